"""The reference's testers (/root/reference/testers/*.m) on top of the device solvers.

``[results, test] = xxxtest(seed, rows, cols, errtol, quiet, options)``: generate the tester's random
problem (``synth.py`` restates each recipe with NumPy's RNG: MATLAB's streams cannot be reproduced),
run the solver on the GPU and evaluate the tester's own pass criterion.  ``test['failed']`` is 0/1
exactly as the reference sets it; the other ``test`` fields carry the quantities the criterion used.
Plots (``showresults``) are not part of the engine.
"""
from __future__ import annotations

import numpy as np

from . import solvers, synth

__all__ = ["lassotest", "ladtest", "huberfittest", "totalvariationtest", "linearsvmtest", "basispursuittest",
           "linearprogramtest", "modeltest", "solvertester"]


def _opts(options, **forced):
    o = dict(options or {})
    o.update(forced)
    return o


def lassotest(seed=0, rows=2 ** 8, cols=2 ** 6, errtol=1e-3, quiet=1, options=None):
    """testers/lassotest.m:31-178: pass when obj(xopt) < obj(testx) (line 143)."""
    p = synth.lasso_problem(seed, rows, cols)
    D, s, lam, testx = p["D"], p["s"], p["lam"], p["testx"]
    obj = lambda x, z: 0.5 * np.sum((D @ x - s) ** 2) + lam * np.sum(np.abs(z))
    results = solvers.lasso(D, s, lam, _opts(options, objevals=1, quiet=quiet))
    xopt = results["xopt"]
    testobj, objopt = obj(testx, testx), obj(xopt, xopt)
    test = dict(D=D, s=s, testx=testx, testobj=testobj, xopt=xopt, admmopt=results["objopt"], objopt=objopt,
                failed=int(not objopt < testobj), objerror=abs((testobj - objopt) / objopt), steps=results["steps"])
    test["lambda"] = lam
    return results, test


def ladtest(seed=0, rows=2 ** 10, cols=2 ** 7, errtol=1e-5, quiet=1, options=None):
    """testers/ladtest.m:37-200: ||xtrue - xopt||_2 < errtol and |objopt - trueobjopt| <= errtol*trueobjopt (149)."""
    p = synth.lad_problem(seed, rows, cols)
    D, s, xtrue = p["D"], p["s"], p["xtrue"]
    results = solvers.lad(D, s, _opts(options, objevals=1, convtest=1, quiet=quiet))
    xopt = results["xopt"]
    trueobj, objopt = np.sum(np.abs(D @ xtrue - s)), np.sum(np.abs(D @ xopt - s))
    xres = np.linalg.norm(xtrue - xopt)
    test = dict(D=D, s=s, truexopt=xtrue, trueobjopt=trueobj, xopt=xopt, admmopt=results["objopt"], objopt=objopt,
                xresidual=xres, xerror=np.sum(np.abs(xtrue - xopt)) / xopt.size,
                failed=int(not (xres < errtol and abs(objopt - trueobj) <= errtol * trueobj)),
                steps=results["steps"], errtol=errtol)
    return results, test


def huberfittest(seed=0, rows=2 ** 11, cols=2 ** 7, errtol=1e-3, quiet=1, options=None):
    """testers/huberfittest.m:43-188: ADMM's objective is below the planted solution's (line 154)."""
    p = synth.huber_problem(seed, rows, cols)
    D, s, testx = p["D"], p["s"], p["testx"]
    hub = lambda r: np.where(np.abs(r) <= 1.0, r * r, 2.0 * np.abs(r) - 1.0)  # CVX huber
    obj = lambda x: 0.5 * np.sum(hub(D @ x - s))
    results = solvers.huberfit(D, s, _opts(options, objevals=1, convtest=1, quiet=quiet))
    xopt = results["xopt"]
    objoptx, objopt = obj(testx), obj(xopt)
    test = dict(D=D, s=s, testx=testx, xopt=xopt, objoptx=objoptx, objopt=objopt, admmopt=results["objopt"],
                xresidual=np.linalg.norm(xopt - testx), failed=int(not objopt < objoptx), steps=results["steps"],
                errtol=errtol)
    return results, test


def totalvariationtest(seed=0, rows=2 ** 7, errtol=0.02, quiet=1, options=None):
    """testers/totalvariationtest.m:35-190: ADMM's objective is below that of the noise-free signal (line 151)."""
    p = synth.tv_problem(seed, rows)
    s, lam, truex = p["s"], p["lam"], p["truex"]
    obj = lambda x: 0.5 * np.sum((x - s) ** 2) + lam * np.sum(np.abs(np.diff(x)))
    results = solvers.totalvariation(s, lam, _opts(options, objevals=1, maxiters=10000, quiet=quiet))
    xopt = results["xopt"]
    trueobj, objopt = obj(truex), obj(xopt)
    test = dict(s=s, truexopt=truex, trueobjopt=trueobj, xopt=xopt, admmopt=results["objopt"], objopt=objopt,
                failed=int(not objopt < trueobj), relerror=abs((trueobj - objopt) / objopt), steps=results["steps"],
                errtol=errtol)
    test["lambda"] = lam
    return results, test


def linearsvmtest(seed=0, mpos=2 ** 7, mneg=2 ** 7, sep=0.2, errtol=0.05, quiet=1, C=0.5, options=None):
    """testers/linearsvmtest.m:47-280: both runs (lossfunction 'hinge', then the string '0-1' -- which is not
    '01', so the hinge prox runs with the 0-1 objective: getProxOps.m:1094, linearsvm.m:231-237) must beat the
    objective of the true separator [1; -1] and recover its slope to errtol (lines 180, 188)."""
    p = synth.svm_problem(seed, mpos, mneg, sep)
    D, ell = p["D"], p["ell"]
    truex = np.array([1.0, -1.0])
    trueobj = 0.5 * np.sum(truex ** 2) + C * np.sum(np.maximum(np.sign(1.0 - ell * (D @ truex)), 0.0))
    base = _opts(options, objevals=1, convtest=1, quiet=quiet, x0=p["x0"], z0=p["z0"], u0=p["u0"])
    out_r, out_t = {}, {}
    for key, loss in (("hingeloss", "hinge"), ("zoloss", "0-1")):
        r = solvers.linearsvm(D, ell, C, dict(base, lossfunction=loss))
        x = r["xopt"]
        if loss == "hinge":
            objopt = 0.5 * np.sum(x ** 2) + C * np.sum(np.maximum(1.0 - ell * (D @ x), 0.0))
        else:
            objopt = 0.5 * np.sum(x ** 2) + C * np.sum(np.maximum(np.sign(1.0 - ell * (D @ x)), 0.0))
        relerr = abs(1.0 - (-x[1] / x[0]))
        out_r[key] = r
        out_t[key] = dict(truexopt=truex, trueobjopt=trueobj, xopt=x, admmopt=r["objopt"], objopt=objopt,
                          failed=int(not (objopt < trueobj and relerr <= errtol)), relerror=relerr,
                          xresidual=np.linalg.norm(x - truex), steps=r["steps"], errtol=errtol)
    return out_r, out_t


def basispursuittest(seed=0, rows=2 ** 6, cols=2 ** 7, errtol=1e-3, quiet=1, options=None):
    """testers/basispursuittest.m:32-175: ||xopt||_1 <= ||testx||_1 and mean relative residual <= errtol (139)."""
    p = synth.basispursuit_problem(seed, rows, cols)
    D, s, testx = p["D"], p["s"], p["testx"]
    results = solvers.basispursuit(D, s, _opts(options, objevals=1, quiet=quiet, maxiters=10000))
    xopt = results["xopt"]
    Dx = D @ xopt
    relerr = float(np.mean(np.abs((Dx - s) / Dx)))
    test = dict(D=D, s=s, testx=testx, xopt=xopt, objopt=np.sum(np.abs(xopt)), testobj=np.sum(np.abs(testx)),
                relerror=relerr,
                failed=int(not (np.sum(np.abs(testx)) >= np.sum(np.abs(xopt)) and relerr <= errtol)),
                steps=results["steps"])
    return results, test


def linearprogramtest(seed=0, rows=2 ** 6, cols=2 ** 7, errtol=1e-3, quiet=1, options=None):
    """testers/linearprogramtest.m:31-170: objective within errtol of the planted one and D*x = s to errtol (130)."""
    p = synth.lp_problem(seed, rows, cols)
    b, D, s, truex = p["b"], p["D"], p["s"], p["truex"]
    results = solvers.linearprogram(b, D, s, _opts(options, objevals=1, quiet=quiet, maxiters=10000))
    xopt = results["xopt"]
    Dx = D @ xopt
    relerr = float(np.mean(np.abs((Dx - s) / Dx)))
    trueobj, objopt = float(b @ truex), float(b @ xopt)
    test = dict(b=b, D=D, s=s, truexopt=truex, xopt=xopt, trueobjopt=trueobj, objopt=objopt, admmopt=results["objopt"],
                relerror=relerr, objerror=abs((trueobj - objopt) / objopt),
                failed=int(not (abs((trueobj - objopt) / objopt) <= errtol and relerr <= errtol)),
                steps=results["steps"])
    return results, test


def modeltest(seed=0, rows=2 ** 7, cols=2 ** 7, errtol=1e-3, quiet=1, options=None):
    """testers/modeltest.m:37-222: closed form x* = (P'P + Q'Q) \\ (P'r + Q's); objective and x within errtol (122, 156)."""
    p = synth.model_problem(seed, rows, cols)
    P, Q, r, s = p["P"], p["Q"], p["r"], p["s"]
    o = _opts(options, objevals=1, quiet=quiet, maxiters=10000, convtest=1, stopcond="both")
    results = solvers.model(P, Q, r, s, o)
    xt = np.linalg.solve(P.T @ P + Q.T @ Q, P.T @ r + Q.T @ s)
    obj = lambda x: 0.5 * np.sum((P @ x - r) ** 2) + 0.5 * np.sum((Q @ x - s) ** 2)
    xopt = results["xopt"]
    test = dict(P=P, Q=Q, r=r, s=s, truex=xt, xopt=xopt, trueobj=obj(xt), objopt=obj(xopt),
                objerror=abs(1.0 - obj(xopt) / obj(xt)), xerror=np.linalg.norm(xt - xopt),
                failed=int(not (abs(1.0 - obj(xopt) / obj(xt)) <= errtol and np.linalg.norm(xt - xopt) <= errtol)),
                steps=results["steps"])
    return results, test


def _sizes(solver, scale, options):
    """Default scalers of testers/solvertester.m (343-361, 396-408, 480-492, 583-595): problem size per scale."""
    scaler = options.get("scaler")
    if callable(scaler):
        m, n = scaler(scale)
        return int(m), int(n)
    ttype = options.get("testtype", "default")
    if solver == "model":
        if ttype == "fat":
            return 2 ** (scale - 1), 2 ** scale
        if ttype == "skinny":
            return 2 ** scale, 2 ** (scale - 1)
        return 2 ** scale, 2 ** scale
    if solver in ("basispursuit", "linearprogram"):
        n = 2 ** scale
        return -(-n // 5), n
    if solver == "totalvariation":
        return 2 ** scale, 1
    # lasso, lad, huberfit: skinny (8x more rows) unless asked otherwise
    if ttype == "fat":
        return max(1, 2 ** (scale - 3)), 2 ** scale
    if ttype == "square":
        return 2 ** scale, 2 ** scale
    return 2 ** scale, max(1, 2 ** (scale - 3))


def solvertester(solver="model", minscale=2, maxscale=8, trials=10, showplots=0, options=None):
    """results = solvertester(solver, minscale, maxscale, trials, showplots, options)
    (testers/solvertester.m:29-275): run the solver's tester `trials` times at every scale from minscale to
    maxscale (problem sizes 2^scale by the reference's default scalers) and collect runtimes and failures.
    ``showplots`` is accepted and ignored (plots are not part of the engine)."""
    options = dict(options or {})
    tests = {"model": modeltest, "basispursuit": basispursuittest, "linearprogram": linearprogramtest,
             "lasso": lassotest, "totalvariation": totalvariationtest, "lad": ladtest, "huberfit": huberfittest}
    if solver not in tests:
        raise ValueError("Given solver is not a supported solver to test!")
    if not (int(minscale) >= 1 and int(maxscale) >= int(minscale) and int(trials) >= 1):
        raise ValueError("minscale, maxscale and trials must be positive integers with maxscale >= minscale!")
    if "errtol" not in options:  # solvertester.m:93-101
        options["errtol"] = 1e-10 if solver in ("basispursuit", "linearprogram") else 1e-3
    seed_rng = np.random.default_rng(options.get("seed"))
    passthrough = {k: v for k, v in options.items() if k not in ("errtol", "testtype", "scaler", "seed")}
    nscale = int(maxscale) - int(minscale) + 1
    runtimes = np.zeros((nscale, int(trials)))
    failed = np.zeros((nscale, int(trials)), dtype=int)
    steps = np.zeros((nscale, int(trials)), dtype=int)
    seeds = np.zeros((nscale, int(trials)), dtype=np.int64)
    for r, scale in enumerate(range(int(minscale), int(maxscale) + 1)):
        m, n = _sizes(solver, scale, options)
        for c in range(int(trials)):
            seed = int(seed_rng.integers(0, 2 ** 31 - 1))
            if solver == "totalvariation":
                res, test = tests[solver](seed, m, options["errtol"], 1, passthrough)
            else:
                res, test = tests[solver](seed, m, n, options["errtol"], 1, passthrough)
            seeds[r, c] = seed
            runtimes[r, c] = res["solverruntime"]
            failed[r, c] = test["failed"]
            steps[r, c] = res["steps"]
    return dict(solver=solver, scales=list(range(int(minscale), int(maxscale) + 1)), runtimes=runtimes, failed=failed,
                steps=steps, seeds=seeds, avetimes=runtimes.mean(axis=1), failure=int(failed.any()), errtol=options["errtol"])
