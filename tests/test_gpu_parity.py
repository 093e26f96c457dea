"""Per-iteration parity of the HIP engine (through the C ABI) against the CPU oracle and the
committed golden fixtures.  The bar set by BASELINE.json's north_star is 1e-6 relative in
fp64 on (x, z, u, residuals); these tests assert TOL = 1e-8 (summation order is the only
difference) and the same iteration count."""
import glob
import os

import numpy as np
import pytest

from oracle import solvers_ref as S

pytestmark = pytest.mark.gpu

TOL = 1e-8
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
HIST = ("xvals", "zvals", "uvals", "vvals", "uhatvals", "pnorm", "dnorm", "perr", "derr", "objevals", "Hnormsq",
        "avals", "dvals", "restarted", "xopt", "zopt", "uopt")


def _close(name, got, ref, tol=TOL, limit=None):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    if limit is not None:
        got, ref = got[..., :limit], ref[..., :limit]
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    nan_g, nan_r = np.isnan(got), np.isnan(ref)
    assert np.array_equal(nan_g, nan_r), f"{name}: NaN pattern differs"
    if got.size == 0 or nan_r.all():
        return
    g, r = got[~nan_r], ref[~nan_r]
    if r.ndim == 1 and got.ndim == 1 and name not in ("xopt", "zopt", "uopt"):
        scale = np.maximum(np.abs(r), 1e-12 + 1e-3 * np.max(np.abs(r)))  # scalar histories: per entry
        err = float(np.max(np.abs(g - r) / scale))
    else:
        err = float(np.max(np.abs(g - r)) / max(1e-300, np.max(np.abs(r))))
    assert err < tol, f"{name}: relative error {err:.3e} >= {tol:g}"


def _compare(got, ref, keys=HIST, tol=TOL, limit=None):
    if "steps" not in ref:  # q4: the convergence test aborted the reference run (admm.m:692-701)
        assert "steps" not in got and "xopt" not in got
        assert got["convtest_failed_at"] == ref["convtest_failed_at"]
    else:
        assert got["steps"] == ref["steps"], (got["steps"], ref["steps"])
    for k in keys:
        if k in ref:
            assert k in got, f"result field {k} missing"
            _close(k, got[k], ref[k], tol, limit)
    for k in ("pnorm", "dnorm", "perr", "derr"):  # q8: absent for accelerated ADMM
        assert (k in got) == (k in ref), k


# ---------------------------------------------------------------------------- oracle parity
@pytest.mark.parametrize("xsolve", ["trsv", "inverse"])
@pytest.mark.parametrize("rows,cols,seed", [(256, 64, 0), (300, 150, 1), (97, 65, 2), (1000, 200, 3),
                                            (2500, 1100, 4)])  # n = 1100: 3 x 9 tiles of the symmetric-half GEMV
def test_lasso_tall(gpu, rows, cols, seed, xsolve):
    p = gpu.synth.lasso_problem(seed, rows, cols)
    o = dict(objevals=1)
    got = gpu.lasso(p["D"], p["s"], p["lam"], dict(o, xsolve=xsolve))
    ref = S.lasso(p["D"], p["s"], p["lam"], o)
    _compare(got, ref)
    assert got["objopt"] == pytest.approx(ref["objopt"], rel=1e-9)
    obj = lambda x: 0.5 * np.sum((p["D"] @ x - p["s"]) ** 2) + p["lam"] * np.sum(np.abs(x))
    assert obj(got["xopt"]) < obj(p["testx"])  # lassotest.m:143
    assert got["solverruntime"] >= got["runtime"] > 0


@pytest.mark.parametrize("opts", [dict(relax=1.5, objevals=1, maxiters=8, domaxiters=1),
                                  dict(relax=0.6, rho=2.0, maxiters=8, domaxiters=1),
                                  dict(relax=1.8, maxiters=6, domaxiters=1, record_history=0),
                                  # relaxation combined with fast / accelerated ADMM (admm.m:515-532 with 563-600)
                                  dict(relax=1.3, fast=1, fasttype="strong", maxiters=8, domaxiters=1),
                                  dict(relax=0.7, fast=1, fasttype="weak", objevals=1, maxiters=8, domaxiters=1)])
@pytest.mark.parametrize("n", [128, 5001])
def test_total_variation_relaxed(gpu, n, opts):
    """a10 with the TV closures: the reference hands Axhat to a z-closure that applies D to it (getProxOps.m:199),
    i.e. z = soft(u + D*(alpha*D*x + (1-alpha)*z_prev)) -- reproduced as is.  That iteration is not a valid ADMM step
    and grows geometrically (1e62 after 1000 iterations at alpha = 1.5), so parity is checked over a few iterations."""
    p = gpu.synth.tv_problem(seed=n + 1, n=n)
    ref_opts = {k: v for k, v in opts.items() if k != "record_history"}
    _compare(gpu.totalvariation(p["s"], p["lam"], dict(opts)), S.totalvariation(p["s"], p["lam"], ref_opts),
             keys=HIST if opts.get("record_history", 1) else ("pnorm", "dnorm", "perr", "derr", "objevals"))


@pytest.mark.parametrize("opts", [dict(maxiters=7, domaxiters=1), dict(maxiters=8, domaxiters=1), dict(),
                                  dict(stopcond="hnorm", maxiters=40), dict(objevals=1, rho=2.0)])
@pytest.mark.parametrize("n", [5000, 70001])
def test_total_variation_without_history(gpu, n, opts):
    """record_history=0: the fused kernel stops storing x; the final x is rebuilt from the surviving forward-sweep
    vector after the loop (odd / even step counts, early stops)."""
    p = gpu.synth.tv_problem(seed=n, n=n)
    got = gpu.totalvariation(p["s"], p["lam"], dict(opts, record_history=0))
    ref = S.totalvariation(p["s"], p["lam"], dict(opts))
    assert got["steps"] == ref["steps"]
    for k in ("xopt", "zopt", "uopt", "pnorm", "dnorm", "perr", "derr"):
        _close(k, got[k], ref[k], TOL, None)
    if "objevals" in opts:
        assert got["objopt"] == pytest.approx(ref["objopt"], rel=1e-9)
    again = gpu.totalvariation(p["s"], p["lam"], dict(opts))  # with history: x stored every iteration
    np.testing.assert_allclose(got["xopt"], again["xopt"], rtol=0, atol=1e-13 * np.max(np.abs(again["xopt"])))


@pytest.mark.parametrize("n,iters", [(5000, 1), (5000, 2), (5000, 3), (5000, 10), (70001, 13)])
def test_total_variation_compact_state_with_arbitrary_start(gpu, n, iters):
    """The direct 1-D kernel carries v = z + u between iterations and rebuilds z = soft(v), u = v - z from it.  A start
    (z0, u0) that does NOT satisfy z0 = soft(z0 + u0) must still be read as given by the first iteration; 1-3 forced
    iterations exercise every branch of the final x / z / u reconstruction (initial buffers, first and later states)."""
    p = gpu.synth.tv_problem(seed=n + iters, n=n)
    rng = np.random.default_rng(iters)
    o = dict(objevals=1, maxiters=iters, domaxiters=1, x0=rng.standard_normal(n), z0=rng.standard_normal(n),
             u0=0.3 * rng.standard_normal(n))
    ref = S.totalvariation(p["s"], p["lam"], dict(o))
    _compare(gpu.totalvariation(p["s"], p["lam"], dict(o)), ref)
    got = gpu.totalvariation(p["s"], p["lam"], dict(o, record_history=0))
    assert got["steps"] == iters
    for k in ("xopt", "zopt", "uopt", "pnorm", "dnorm", "perr", "derr"):
        _close(k, got[k], ref[k], TOL, None)


@pytest.mark.parametrize("xsolve", ["inverse", "trsv"])
@pytest.mark.parametrize("rows,cols,opts", [(256, 64, dict()), (2500, 1600, dict(maxiters=12, domaxiters=1)),
                                            (700, 130, dict(rho=2.5, relax=1.4)), (400, 90, dict(fast=1, fasttype="strong", maxiters=40)),
                                            (400, 90, dict(fast=1, fasttype="weak", maxiters=30)),
                                            (60, 300, dict(maxiters=80)), (90, 700, dict(rho=0.3, relax=1.5, maxiters=60))])  # fat: lasso.m:172
def test_lasso_objective_through_the_gram_matrix(gpu, rows, cols, opts, xsolve):
    """objgram=1 (engine-side option): 1/2*||D*x - s||^2 evaluated as 1/2*x'Gx - x'D's + 1/2*s's with G x = y - rho*x
    from the right-hand side y the x-update solved with (no pass over D or G; plain, relaxed and accelerated ADMM, both
    x-solve forms) -- same iterates, objective equal to the literal form up to eps*||s||^2."""
    p = gpu.synth.lasso_problem(5, rows, cols)
    o = dict(objevals=1, **opts)
    got = gpu.lasso(p["D"], p["s"], p["lam"], dict(o, objgram=1, xsolve=xsolve))
    ref = S.lasso(p["D"], p["s"], p["lam"], dict(o))
    _compare(got, ref, keys=tuple(k for k in HIST if k != "objevals"))
    bound = 1e-13 * float(p["s"] @ p["s"])
    assert np.max(np.abs(got["objevals"] - ref["objevals"])) <= bound
    assert abs(got["objopt"] - ref["objopt"]) <= bound


@pytest.mark.parametrize("xsolve", ["trsv", "inverse"])
def test_lasso_fat(gpu, xsolve):
    p = gpu.synth.lasso_problem(1, 32, 256)
    o = dict(objevals=1)
    _compare(gpu.lasso(p["D"], p["s"], p["lam"], dict(o, xsolve=xsolve)), S.lasso(p["D"], p["s"], p["lam"], o))


@pytest.mark.parametrize("opts", [
    dict(rho=2.5), dict(relax=1.6), dict(relax=0.7, convtest=1), dict(stopcond="hnorm", maxiters=50),
    dict(stopcond="both"), dict(domaxiters=1, maxiters=30), dict(abstol=1e-8, reltol=1e-7, maxiters=400),
    dict(nodualerror=1), dict(stopcond="bogus", maxiters=25), dict(Hreltol=1e-3, stopcond="hnorm"),
])
def test_lasso_options(gpu, opts):
    p = gpu.synth.lasso_problem(4, 200, 50)
    o = dict(objevals=1, **opts)
    _compare(gpu.lasso(p["D"], p["s"], p["lam"], o), S.lasso(p["D"], p["s"], p["lam"], o))


@pytest.mark.parametrize("solver", ["lasso", "lad"])
def test_wvals_of_the_h_norm_runs(gpu, solver):
    """results.wvals = [x; z; rho*u] per iteration (admm.m:678-681), assembled behind the ABI (ADMM_F_WVALS) -- A = 1
    (all three parts of length n) and A = D (x of length n, z and u of length m), rho != 1"""
    if solver == "lasso":
        p = gpu.synth.lasso_problem(6, 150, 40)
        o = dict(stopcond="both", rho=2.5, maxiters=40)
        got, ref = gpu.lasso(p["D"], p["s"], p["lam"], o), S.lasso(p["D"], p["s"], p["lam"], o)
        rows = 3 * 40
    else:
        p = gpu.synth.lad_problem(3, 120, 30)
        o = dict(convtest=1, convtol=1e3, rho=0.5, maxiters=25, domaxiters=1)
        got, ref = gpu.lad(p["D"], p["s"], o), S.lad(p["D"], p["s"], o)
        rows = 30 + 120 + 120
    _compare(got, ref)
    assert got["wvals"].shape == (rows, got["steps"])
    _close("wvals", got["wvals"], ref["wvals"])
    np.testing.assert_array_equal(got["wvals"][:got["xvals"].shape[0]], got["xvals"])


def test_lasso_warm_start(gpu):
    p = gpu.synth.lasso_problem(5, 128, 32)
    rng = np.random.default_rng(0)
    o = dict(x0=rng.standard_normal(32), z0=rng.standard_normal(32), u0=rng.standard_normal(32))
    got, ref = gpu.lasso(p["D"], p["s"], p["lam"], o), S.lasso(p["D"], p["s"], p["lam"], o)
    _compare(got, ref)
    np.testing.assert_array_equal(got["z0"], o["z0"])


@pytest.mark.parametrize("fasttype", ["weak", "strong"])
@pytest.mark.parametrize("xsolve", ["trsv", "inverse"])
def test_lasso_fast(gpu, fasttype, xsolve):
    p = gpu.synth.lasso_problem(0, 256, 64)
    o = dict(objevals=1, fast=1, fasttype=fasttype, maxiters=60, stopcond="both")
    got = gpu.lasso(p["D"], p["s"], p["lam"], dict(o, xsolve=xsolve))
    ref = S.lasso(p["D"], p["s"], p["lam"], o)
    _compare(got, ref)
    if fasttype == "weak":
        assert got["dvaltol"] == 1e-8 and "pnorm" not in got


@pytest.mark.parametrize("solver", ["lad", "huberfit"])
@pytest.mark.parametrize("opts", [dict(convtest=1), dict(relax=1.5), dict(fast=1, fasttype="strong", maxiters=40),
                                  dict(fast=1, fasttype="weak", maxiters=40, stopcond="both"),
                                  dict(nodualerror=1, stopcond="both")])
@pytest.mark.parametrize("xsolve", ["trsv", "inverse"])
def test_lad_huber(gpu, solver, opts, xsolve):
    p = (gpu.synth.lad_problem if solver == "lad" else gpu.synth.huber_problem)(0, 512, 64)
    o = dict(objevals=1, **opts)
    got = getattr(gpu, solver)(p["D"], p["s"], dict(o, xsolve=xsolve))
    ref = getattr(S, solver)(p["D"], p["s"], o)
    _compare(got, ref)


def test_lad_reference_criterion_default_size(gpu):  # ladtest.m:100-102, 149
    p = gpu.synth.lad_problem(0)
    r = gpu.lad(p["D"], p["s"], dict(objevals=1, convtest=1))
    assert np.linalg.norm(p["xtrue"] - r["xopt"]) < 1e-5
    tobj = np.sum(np.abs(p["D"] @ p["xtrue"] - p["s"]))
    assert abs(np.sum(np.abs(p["D"] @ r["xopt"] - p["s"])) - tobj) <= 1e-5 * tobj


def test_huber_reference_criterion_default_size(gpu):  # huberfittest.m:106-107, 154
    p = gpu.synth.huber_problem(0)
    r = gpu.huberfit(p["D"], p["s"], dict(objevals=1, convtest=1))
    obj = lambda x: 0.5 * np.sum(S.huber_cvx(p["D"] @ x - p["s"]))
    assert obj(r["xopt"]) < obj(p["testx"])


@pytest.mark.parametrize("xsolve", ["trsv", "inverse"])
def test_svm_hinge(gpu, xsolve):
    p = gpu.synth.svm_problem(0)
    o = dict(objevals=1, convtest=1, x0=p["x0"], z0=p["z0"], u0=p["u0"])
    got = gpu.linearsvm(p["D"], p["ell"], p["C"], dict(o, xsolve=xsolve))
    ref = S.linearsvm(p["D"], p["ell"], p["C"], o)
    _compare(got, ref, tol=1e-7)  # x = pinv(D)(z-u) vs chol(D'D) solve: kappa(D)^2 * eps apart
    x = got["xopt"]
    assert abs(1 - (-x[1] / x[0])) <= 0.05  # linearsvmtest.m:175-180
    assert np.isnan(got["dnorm"]).all()


def test_svm_01_margin_guarded(gpu):
    """0-1 prox is discontinuous (q24): compare the first iterations, where no component sits
    within 1e-6 of a decision boundary of minz01 (getProxOps.m:1175)."""
    p = gpu.synth.svm_problem(0)
    o = dict(objevals=1, lossfunction="01", x0=p["x0"], z0=p["z0"], u0=p["u0"])
    got = gpu.linearsvm(p["D"], p["ell"], p["C"], o)
    ref = S.linearsvm(p["D"], p["ell"], p["C"], o)
    D, ell, C = p["D"], p["ell"], p["C"]
    k = 0
    u = p["u0"]
    for i in range(min(ref["steps"], 40)):
        sarg = ell * (D @ ref["xvals"][:, i] + u)
        margin = min(np.min(np.abs(sarg - 1.0)), np.min(np.abs(sarg - (1 - np.sqrt(2.0 * C)))))
        if margin < 1e-6:
            break
        u = ref["uvals"][:, i]
        k = i + 1
    assert k >= 10
    for key in ("xvals", "zvals", "uvals", "pnorm", "perr", "objevals"):
        _close(key, got[key], ref[key], 1e-7, limit=k)


def test_svm_mnist_shaped(gpu):
    """Config 3 stand-in: 1500 x 400 sparse-ish pixels, real MNIST label distribution is
    irrelevant to parity; hinge loss, fixed x0/z0/u0, 60 forced iterations."""
    p = gpu.synth.mnist_like_problem(seed=1, m=1500, n=400, digit=3)
    o = dict(objevals=1, x0=p["x0"], z0=p["z0"], u0=p["u0"], domaxiters=1)
    got = gpu.linearsvm(p["D"], p["ell"], p["C"], dict(o, record_history=0))
    ref = S.linearsvm(p["D"], p["ell"], p["C"], o)
    assert got["steps"] == ref["steps"] == 1000
    for k in ("pnorm", "perr", "objevals", "Hnormsq", "xopt"):
        _close(k, got[k], ref[k], 1e-6)
    assert "xvals" not in got


@pytest.mark.parametrize("xsolve", ["auto", "trsv", "inverse", "pinv"])
def test_svm_rank_deficient_matches_pinv(gpu, xsolve):
    """BASELINE config 3's real input is rank deficient: cropped MNIST has pixels that are zero in every sample
    (mnistsvm.m:61-72), which is why the reference computes x = pinv(D)*(z-u) (linearsvm.m:185,
    unwrappedadmm.m:76-78).  5 % all-zero columns + 3 % duplicated columns: chol(D'D) breaks down, the engine
    switches to the pseudo-inverse of D'D by itself and follows the oracle's pinv (MATLAB tolerance rule)."""
    p = gpu.synth.rank_deficient_pixels(seed=1, m=1500, n=400, digit=3)
    assert np.linalg.matrix_rank(p["D"]) == p["rank"] < 400
    o = dict(objevals=1, x0=p["x0"], z0=p["z0"], u0=p["u0"])  # unwrappedadmm.m:90-92: both stop conditions, <= 1000 it.
    got = gpu.linearsvm(p["D"], p["ell"], p["C"], dict(o, xsolve=xsolve))
    ref = S.linearsvm(p["D"], p["ell"], p["C"], o)
    _compare(got, ref, tol=1e-6)
    info = got["engine_info"]
    assert info["pinv_used"] and info["rank"] == p["rank"] and info["xsolve_used"] == "inverse"
    # minimum-norm solution: nothing in the null space of D (zero columns stay exactly at 0)
    dead = np.where(~p["D"].any(axis=0))[0]
    assert dead.size and np.max(np.abs(got["xopt"][dead])) < 1e-12


def test_svm_with_callers_pseudo_inverse(gpu):
    """args.Dplus = pinv(D) from the caller (linearsvm.m:185-186) is applied literally, x = Dplus*(z-u)
    (getProxOps.m:1067): same iterates as the oracle on a rank-deficient D, dual residual included"""
    p = gpu.synth.rank_deficient_pixels(seed=2, m=900, n=200, digit=1)
    Dplus = S.pinv_matlab(p["D"])
    minx, minz, _ = gpu.getproxops("linearsvm", dict(D=p["D"], ell=p["ell"], C=p["C"], lossfunction="hinge",
                                                     Dplus=Dplus))
    m, n = p["D"].shape
    o = dict(A=p["D"], At=p["D"].T, B=-1, c=0, m=m, nA=n, nB=m, x0=p["x0"], z0=p["z0"], u0=p["u0"],
             maxiters=50, domaxiters=1, stopcond="both")
    got = gpu.admm(minx, minz, dict(o))
    from oracle import admm as ref_admm, getproxops as ref_getproxops
    _, rz, _ = ref_getproxops("LinearSVM", dict(D=p["D"], Dt=p["D"].T, ell=p["ell"], C=p["C"], lossfunction="hinge",
                                                Dplus=Dplus))
    ref = ref_admm(lambda x, z, u, rho: Dplus @ (z - u), rz, dict(o))
    _compare(got, ref, tol=1e-8)
    assert got["engine_info"]["xsolve_used"] == "pinv"


def test_svm_rank_deficient_tiny(gpu):
    """duplicated column in the linearsvmtest.m geometry (n = 3, rank 2), run to convergence"""
    p = gpu.synth.svm_problem(0)
    D = np.asfortranarray(np.column_stack([p["D"], p["D"][:, 0]]))
    rng = np.random.default_rng(5)
    o = dict(objevals=1, x0=rng.random(3), z0=p["z0"], u0=p["u0"])
    got = gpu.linearsvm(D, p["ell"], p["C"], o)
    ref = S.linearsvm(D, p["ell"], p["C"], o)
    _compare(got, ref, tol=1e-6)
    assert got["engine_info"]["pinv_used"] and got["engine_info"]["rank"] == 2
    assert abs(got["xopt"][0] - got["xopt"][2]) < 1e-10  # min-norm: equal weight on the two copies


def test_lad_rank_deficient_is_an_error_like_the_reference(gpu):
    """lad.m:134 `chol(D'*D,'lower')` errors on a rank-deficient D; so does the engine (no silent pinv there)"""
    p = gpu.synth.lad_problem(0, 256, 32)
    D = p["D"].copy(order="F")
    D[:, 5] = D[:, 7]
    with pytest.raises(gpu.AdmmError) as ei:
        gpu.lad(D, p["s"], {})
    assert ei.value.code == gpu._lib.E_NUMERIC


@pytest.mark.parametrize("kappa", [1e3, 1e4])
@pytest.mark.parametrize("xsolve", ["auto", "trsv", "inverse"])
@pytest.mark.parametrize("solver", ["lad", "huberfit"])
def test_ill_conditioned_factor_guard(gpu, solver, xsolve, kappa):
    """q20: LAD / Huber factor the UN-shifted D'D, cond = kappa(D)^2.  The explicit inverse loses cond^1.5*eps and
    must not be used there: create() probes both forms and falls back to the blocked triangular solves, whatever
    was requested; every form then matches the oracle (LAPACK triangular solves) at 1e-6."""
    p = gpu.synth.lad_problem_conditioned(0, 640, 300, kappa)
    o = dict(objevals=1, domaxiters=1, maxiters=40)
    got = getattr(gpu, solver)(p["D"], p["s"], dict(o, xsolve=xsolve))
    ref = getattr(S, solver)(p["D"], p["s"], o)
    _compare(got, ref, tol=1e-6)
    info = got["engine_info"]
    assert info["cond_estimate"] > 10.0
    if xsolve != "trsv":
        assert info["probed"]
        # whichever form runs passed the probe (or is the triangular solve)
        assert info["xsolve_used"] == "trsv" or (
            info["probe_err_inverse"] <= max(1e-9, 2 * info["probe_err_trsv"])
            and info["probe_diff"] <= max(1e-9, 4 * info["probe_err_trsv"]))


def test_factor_guard_rejects_a_bad_inverse(gpu):
    """a factor handed in by the caller (args.L, lasso.m:183) whose diagonal spans 12 orders of magnitude: the
    explicit inverse loses digits the triangular solves keep, the probe sees it and the engine runs trsv"""
    rng = np.random.default_rng(11)
    n = 300
    Lf = np.tril(rng.standard_normal((n, n)) * 0.05)
    Lf[np.diag_indices(n)] = np.geomspace(1.0, 1e-6, n)
    D = np.asfortranarray(rng.standard_normal((400, n)))
    s = rng.standard_normal(400)
    eng = gpu.Engine(gpu._lib.PROB_LASSO, D=D, s=s, lam=0.1, rho=1.0, Lfactor=np.asfortranarray(Lf),
                     xsolve=gpu._lib.XSOLVE_INVERSE)
    info = eng.info()
    eng.close()
    assert info["probed"] and info["cond_estimate"] >= 1e11
    assert info["xsolve_used"] in ("trsv", "inverse")
    if info["xsolve_used"] == "inverse":
        assert info["probe_err_inverse"] <= max(1e-9, 2 * info["probe_err_trsv"])


@pytest.mark.parametrize("rows,cols,seed", [(600, 300, 5), (2500, 1100, 4), (4200, 1700, 6)])
@pytest.mark.parametrize("opts", [dict(objevals=1), dict(maxiters=9, domaxiters=1, record_history=0),
                                  dict(fast=1, fasttype="strong", maxiters=12, domaxiters=1),
                                  dict(fast=1, fasttype="weak", objevals=1, maxiters=12, domaxiters=1),
                                  dict(relax=1.6, maxiters=12, domaxiters=1)])
def test_lasso_through_the_one_block_triangular_solves(gpu, rows, cols, seed, opts, monkeypatch):
    """xsolve = trsv with the whole factor as ONE pre-inverted block (two passes over inv(L), symv.hip: tri1_*): the
    forward pass carries the deferred finalize, the element update sums the backward pass's partial rows -- every loop
    variant against the oracle's LAPACK substitution, through early stops and batch ends"""
    monkeypatch.setenv("ADMM_TRSV_FORM", "one")
    p = gpu.synth.lasso_problem(seed, rows, cols)
    got = gpu.lasso(p["D"], p["s"], p["lam"], dict(opts, xsolve="trsv"))
    ref_opts = {k: v for k, v in opts.items() if k != "record_history"}
    ref = S.lasso(p["D"], p["s"], p["lam"], ref_opts)
    keys = HIST if opts.get("record_history", 1) else ("xopt", "zopt", "uopt", "pnorm", "dnorm", "perr", "derr")
    _compare(got, ref, keys=keys)
    info = got["engine_info"]
    assert info["xsolve_used"] == "trsv" and info["trsv_blocks"] == 1
    assert info["probe_err_trsv_one"] < 1e-12 and info["probe_err_trsv"] < 1e-12


@pytest.mark.parametrize("solver", ["lad", "huberfit"])
def test_a_equals_d_problems_through_the_one_block_triangular_solves(gpu, solver, monkeypatch):
    """the same form where x is needed as a vector (D*x follows): backward pass + tri1_reduce_kernel"""
    monkeypatch.setenv("ADMM_TRSV_FORM", "one")
    p = gpu.synth.lad_problem(2, 900, 400)
    o = dict(objevals=1, domaxiters=1, maxiters=30)
    got = getattr(gpu, solver)(p["D"], p["s"], dict(o, xsolve="trsv"))
    ref = getattr(S, solver)(p["D"], p["s"], o)
    _compare(got, ref)
    assert got["engine_info"]["trsv_blocks"] == 1


@pytest.mark.parametrize("kappa", [1e3, 1e4, 1e5])
def test_one_block_form_is_chosen_by_measurement(gpu, kappa):
    """no override: at create the one-block form replaces the blocked substitution only while the probe finds it as
    accurate (4x / 8x) or below 1e-11; whichever runs, the iterates match the oracle"""
    p = gpu.synth.lad_problem_conditioned(1, 2400, 1600, kappa)
    o = dict(objevals=1, domaxiters=1, maxiters=25)
    got = gpu.lad(p["D"], p["s"], dict(o, xsolve="trsv"))
    ref = S.lad(p["D"], p["s"], o)
    _compare(got, ref, tol=1e-6)
    info = got["engine_info"]
    assert info["xsolve_used"] == "trsv" and info["trsv_blocks"] >= 1
    e1, eb = info["probe_err_trsv_one"], info["probe_err_trsv"]
    assert e1 == e1 and eb == eb  # both were measured
    if info["trsv_blocks"] == 1:
        assert e1 <= max(1e-11, 4 * eb)


def test_well_conditioned_auto_keeps_the_inverse(gpu):
    p = gpu.synth.lasso_problem(3, 1000, 300)
    got = gpu.lasso(p["D"], p["s"], p["lam"], dict(xsolve="auto"))
    info = got["engine_info"]
    assert info["xsolve_used"] == "inverse" and info["probed"] and info["probe_err_inverse"] < 1e-12
    assert info["probe_err_trsv"] < 1e-12


@pytest.mark.parametrize("xsolve", ["trsv", "inverse"])
def test_qp_bounded(gpu, xsolve):
    p = gpu.synth.qp_bounded_problem(0, 128)
    o = dict(objevals=1, stopcond="both")
    got = gpu.quadraticprogram(p["P"], p["q"], p["r"], p["lb"], p["ub"], dict(o, xsolve=xsolve))
    ref = S.quadraticprogram_bounded(p["P"], p["q"], p["r"], p["lb"], p["ub"], o)
    _compare(got, ref)
    assert np.all(got["zopt"] >= p["lb"] - 1e-15) and np.all(got["zopt"] <= p["ub"] + 1e-15)


def test_qp_bounded_objective_from_the_right_hand_side(gpu):
    """quadraticprogram.m:242 evaluates 1/2*x'Px + q'x + r with a pass over P; the engine's own factor gives P x = y - rho*x
    from the right-hand side y of the x-update: calibrated against the P*x form in the first batch, then used alone."""
    L = gpu._lib
    p = gpu.synth.qp_bounded_problem(3, 300)
    kw = dict(P=p["P"], q=p["q"], lb=p["lb"], ub=p["ub"], rho=1.0, r=float(p["r"]))
    auto, lit = gpu.Engine(L.PROB_QP_BOUNDED, **kw), gpu.Engine(L.PROB_QP_BOUNDED, obj_gram=-1, **kw)
    try:
        run = dict(maxiters=40, domaxiters=1, objevals=1, check_every=8)
        sa, sl = auto.run(**run), lit.run(**run)
        assert sa.obj_gram_used == 1 and sl.obj_gram_used == 0
        oa, ol = auto.fetch(L.F_OBJEVALS, 40), lit.fetch(L.F_OBJEVALS, 40)
        assert np.array_equal(oa[:8], ol[:8])  # the calibration batch records the P*x values
        scale = np.abs(ol) + abs(float(p["r"])) + 1.0
        assert np.max(np.abs(oa - ol) / scale) < 1e-10
        np.testing.assert_array_equal(auto.fetch(L.F_XOPT, 300), lit.fetch(L.F_XOPT, 300))
    finally:
        auto.close(), lit.close()


def test_basispursuit(gpu):
    p = gpu.synth.basispursuit_problem(0, 32, 96)
    o = dict(objevals=1)
    _compare(gpu.basispursuit(p["D"], p["s"], o), S.basispursuit(p["D"], p["s"], o))


@pytest.mark.parametrize("rows,cols", [(1, 2), (2, 3), (7, 30)])
def test_equality_constrained_solvers_with_a_zero_right_hand_side(gpu, rows, cols):
    """s = 0: the create-time probe of the eliminated maps (engine.hip: probe_affine_map) measures D*x = s against the
    size of D times an unstructured vector -- against |D*x| and |s| alone the violation of an exactly feasible x (noise
    over noise) read as 1 and the problem was refused (found by tests/sweeps/fuzz_solvers.py at 1 x 2)."""
    rng = np.random.default_rng(rows * 100 + cols)
    D = np.asfortranarray(rng.standard_normal((rows, cols)))
    s = np.zeros(rows)
    # random starts: from zero every iterate of these problems IS zero
    o = dict(objevals=1, maxiters=25, z0=rng.standard_normal(cols), u0=rng.standard_normal(cols))

    def same(got, ref):  # the solution is x = 0: late iterates are rounding noise, measured against the starts' size
        assert got["steps"] == ref["steps"]
        for k in ("xvals", "zvals", "uvals", "pnorm", "dnorm", "objevals"):
            np.testing.assert_allclose(got[k], ref[k], rtol=1e-8, atol=1e-12, err_msg=k)

    same(gpu.basispursuit(D, s, dict(o)), S.basispursuit(D, s, dict(o)))
    if rows > 1:  # (a 1 x n constraint is a vector to linearprogram.m's argument checks)
        b = rng.random(cols) + 0.5
        same(gpu.linearprogram(b, D, s, dict(o)), S.linearprogram(b, D, s, dict(o)))


@pytest.mark.parametrize("n,lam,rho,opts", [
    (128, 1.0, 1.0, dict(maxiters=10000)),          # totalvariationtest.m defaults
    (1000, 2.0, 1.0, dict()),
    (5000, 0.5, 3.0, dict(stopcond="both")),        # several tiles + halo overlap
    (9973, 1.0, 0.25, dict(convtest=1)),
    (4096, 1.0, 40.0, dict(maxiters=300)),          # slower-decaying recurrence (halo ~ 270, 20-element tiles)
    (30011, 1.0, 800.0, dict(maxiters=6, domaxiters=1)),  # halo ~ 1200: 48-element tiles, odd n
    (2, 1.0, 1.0, dict(maxiters=5)), (257, 0.0, 1.0, dict(maxiters=20)),
    # short signals with a large rho: the damping length of a carry (5900 positions at rho = 430) exceeds every window
    # size, but a window is never longer than the signal (tv_plan; refused before the solver sweep found it)
    (3, 0.21, 430.0, dict(maxiters=33, domaxiters=1)), (10, 6.0, 4800.0, dict(maxiters=23)),
    (3000, 1.0, 6000.0, dict(maxiters=12, domaxiters=1)),
    (30000, 1.0, 30000.0, dict(maxiters=6, domaxiters=1)),  # halo ~ 7200 of a workgroup's 12288 positions
])
def test_total_variation(gpu, n, lam, rho, opts):
    p = gpu.synth.tv_problem(n % 97, n)
    o = dict(objevals=1, rho=rho, **opts)
    got = gpu.totalvariation(p["s"], lam, o)
    ref = S.totalvariation(p["s"], lam, o)
    _compare(got, ref)
    if n == 128:  # totalvariationtest.m:151
        obj = lambda x: 0.5 * np.sum((x - p["s"]) ** 2) + lam * np.sum(np.abs(np.diff(x)))
        assert obj(got["xopt"]) < obj(p["truex"])


def test_total_variation_second_run_and_errors(gpu):
    p = gpu.synth.tv_problem(3, 3000)
    args = dict(s=p["s"])
    args["lambda"] = 1.0
    minx, minz, _ = gpu.getproxops("TotalVariation", args)
    import scipy.sparse as sp
    D = sp.diags([np.ones(3000), -np.ones(2999)], [0, 1], format="csr")
    base = dict(A=D, B=-1, c=0, m=3000, nB=3000)
    a = gpu.admm(minx, minz, dict(base, maxiters=7, domaxiters=1))   # odd step count: ping-pong parity
    b = gpu.admm(minx, minz, dict(base, maxiters=7, domaxiters=1))
    np.testing.assert_array_equal(a["zopt"], b["zopt"])
    np.testing.assert_array_equal(a["zopt"], a["zvals"][:, -1])
    np.testing.assert_array_equal(a["uopt"], a["uvals"][:, -1])
    c = gpu.admm(minx, minz, dict(base, fast=1, maxiters=7, domaxiters=1))  # fast ADMM: the unfused TV path
    assert c["steps"] == 7 and "avals" in c


@pytest.mark.parametrize("xsolve", ["trsv", "inverse"])
@pytest.mark.parametrize("rows,cols,workers,opts", [
    (256, 64, 4, dict()),                      # 4 x (64 x 64) slices, as in the survey's feasibility run
    (1000, 60, 8, dict(rho=2.0)),              # 8 slices of 125 rows (config-4 shape, scaled down)
    (515, 48, 3, dict(maxiters=15, domaxiters=1)),  # uneven slices 172/172/171
    (300, 70, 2, dict(u0=np.linspace(-1, 1, 70))),
    (256, 64, 4, dict(relax=1.5)),             # the closures ignore what admm relaxes: same iterates as relax = 1
])
def test_consensus_lasso(gpu, rows, cols, workers, opts, xsolve):
    """Config 4 semantics (getProxOps.m:1217-1343) incl. quirks q9-q11: z handed to admm is 0,
    squared special norms, threshold lambda/(rho*N); compared with the N-slice oracle."""
    p = gpu.synth.lasso_problem(1, rows, cols)
    o = dict(objevals=1, parallel="both", **opts)
    got = gpu.lasso(p["D"], p["s"], p["lam"], dict(o, workers=workers, xsolve=xsolve))
    ref = S.lasso(p["D"], p["s"], p["lam"], o, workers=workers)
    _compare(got, ref)
    assert np.all(got["zvals"] == 0.0) and np.all(got["zopt"] == 0.0)  # q9
    zc = ref["_consensus"]["_state"]["z"]
    assert np.max(np.abs(got["zconsensus"] - zc)) < 1e-9 * max(1.0, np.max(np.abs(zc)))
    if not opts:
        obj = lambda x: 0.5 * np.sum((p["D"] @ x - p["s"]) ** 2) + p["lam"] * np.sum(np.abs(x))
        assert obj(got["zconsensus"]) < obj(p["testx"])


def test_matrix_free_x_update_reports_the_updates_that_ended_on_the_cap(gpu):
    """cg_maxit too small for cg_tol: the iterates are inexact, and the run says so (results.cg_capped_updates, the
    second entry of ADMM_F_CG_ITERS) instead of returning them as if nothing had happened; with the default cap no
    update of this well-conditioned problem ends on it."""
    p = gpu.synth.lasso_problem(4, 600, 120)
    o = dict(maxiters=12, domaxiters=1, quiet=1)
    res = gpu.lasso(p["D"], p["s"], p["lam"], dict(o, xsolve="cg", cg_maxit=3))
    assert res["cg_capped_updates"] == 12 and res["cg_iters_total"] == 36
    res = gpu.lasso(p["D"], p["s"], p["lam"], dict(o, xsolve="cg"))
    assert res["cg_capped_updates"] == 0 and res["cg_iters_total"] > 12


@pytest.mark.parametrize("solver,opts", [
    ("lasso", dict()), ("lasso", dict(rho=3.0, relax=1.5)), ("lasso", dict(fast=1, fasttype="strong", maxiters=40)),
    ("lad", dict()), ("huberfit", dict(convtest=1)),
])
def test_matrix_free_cg_matches_factor_path(gpu, solver, opts):
    """xsolve='cg' (no reference counterpart: the reference always factors).  Its own oracle is the
    factor-path oracle: with inner tolerance 1e-13 the per-iteration iterates agree to 1e-7."""
    if solver == "lasso":
        p = gpu.synth.lasso_problem(2, 600, 120)
        o = dict(objevals=1, **opts)
        got = gpu.lasso(p["D"], p["s"], p["lam"], dict(o, xsolve="cg", cg_tol=1e-13))
        ref = S.lasso(p["D"], p["s"], p["lam"], o)
    else:
        p = (gpu.synth.lad_problem if solver == "lad" else gpu.synth.huber_problem)(0, 512, 64)
        o = dict(objevals=1, **opts)
        got = getattr(gpu, solver)(p["D"], p["s"], dict(o, xsolve="cg", cg_tol=1e-13))
        ref = getattr(S, solver)(p["D"], p["s"], o)
    _compare(got, ref, tol=1e-7)
    assert 0 < got["cg_iters_total"] <= 60 * got["steps"]


@pytest.mark.parametrize("rho", [1.0, 2.5])
@pytest.mark.parametrize("rows,workers", [(120, 4), (150, 2), (200, 3)])
def test_consensus_lasso_fat_slices_serial_formula(gpu, rows, workers, rho):
    """q12 (documented deviation): slices with fewer rows than columns.  The reference's branch
    (getProxOps.m:426-430, 1251) shifts the wrong entries of D_k D_k'; the engine applies the serial solver's
    form chol(D_k D_k'/rho + I), x = y/rho - D_k'(U\\(L\\(D_k y)))/rho^2 (lasso.m:172, getProxOps.m:1204) and is
    compared with the oracle branch that does the same (fatformula='serial').  (200, 3): 67/67/66-row slices of a
    64-column matrix are tall -- mixed with nothing here, but 150/2 = 75 > 64 is tall too: only 120/4 = 30 is fat;
    the other two cases pin that the tall path is untouched by the flag."""
    p = gpu.synth.lasso_problem(1, rows, 64)
    o = dict(objevals=1, parallel="both", rho=rho)
    got = gpu.lasso(p["D"], p["s"], p["lam"], dict(o, workers=workers))
    ref = S.lasso(p["D"], p["s"], p["lam"], dict(o, slices=0, fatformula="serial"), workers=workers)
    _compare(got, ref, keys=("xvals", "zvals", "uvals", "pnorm", "dnorm", "perr", "derr", "objevals", "Hnormsq"))
    # and against the mathematically identical tall formula on the same slices: (D_k'D_k + rho I) x_k = y_k
    assert got["steps"] == ref["steps"]


def test_consensus_lasso_mixed_fat_and_tall_slices(gpu):
    """explicit slice sizes (errorcheck.m:263-265): one fat slice (20 rows) next to tall ones"""
    p = gpu.synth.lasso_problem(2, 220, 48)
    sl = [100, 20, 100]
    args = dict(slices=np.array(sl, dtype=float), D=p["D"], s=p["s"], rho=1.0, parallel=1)
    args["lambda"] = p["lam"]
    minx, minz, extra = gpu.getproxops("LASSO", args)
    opts = dict(A=1, At=1, m=48, nA=48, nB=48, B=-1, c=0, stopcond="both", altu=extra["altu"],
                specialnorms=extra["specialnorms"])
    got = gpu.admm(minx, minz, dict(opts))
    from oracle import admm as ref_admm, getproxops as ref_getproxops
    rx, rz, rextra = ref_getproxops("LASSO", dict(args, slices=sl, fatformula="serial"))
    ref = ref_admm(rx, rz, dict(opts, altu=rextra["altu"], specialnorms=rextra["specialnorms"]))
    _compare(got, ref, keys=("xvals", "uvals", "pnorm", "dnorm", "Hnormsq"))


def test_precomputed_factor_is_used(gpu):
    """args.L handed in by the caller (lasso.m:183) must give the same iterates as the on-device factor."""
    import scipy.linalg as sla

    p = gpu.synth.lasso_problem(6, 120, 40)
    D, s, lam = p["D"], p["s"], p["lam"]
    Lf = sla.cholesky(D.T @ D + np.eye(40), lower=True)
    args = dict(D=D, s=s, L=Lf, rho=1.0, m=120, n=40, parallel=0)
    args["lambda"] = lam
    minx, minz, _ = gpu.getproxops("LASSO", args)
    opts = dict(A=1, At=1, m=40, nA=40, nB=40, B=-1, c=0, objevals=1)
    got = gpu.admm(minx, minz, opts)
    _compare(got, S.lasso(D, s, lam, dict(objevals=1)))
    F = minx.problem.engine.fetch(gpu._lib.F_FACTOR, 40 * 40, (40, 40))
    np.testing.assert_allclose(F, Lf, rtol=0, atol=0)


def test_device_factor_matches_lapack(gpu):
    import scipy.linalg as sla

    p = gpu.synth.lasso_problem(7, 600, 200)
    args = dict(D=p["D"], s=p["s"], rho=0.5, parallel=0)
    args["lambda"] = p["lam"]
    minx, _, _ = gpu.getproxops("lasso", args)
    F = minx.problem.engine.fetch(gpu._lib.F_FACTOR, 200 * 200, (200, 200))
    ref = sla.cholesky(p["D"].T @ p["D"] + 0.5 * np.eye(200), lower=True)
    assert np.max(np.abs(F - ref)) / np.max(np.abs(ref)) < 1e-12


def test_convtest_abort_is_reproduced(gpu):
    """q4: the reference returns early without steps/xopt when H-norms grow (admm.m:692-701).
    The discontinuous 0-1 prox trips it (same iteration as the oracle)."""
    p = gpu.synth.svm_problem(0)
    o = dict(objevals=1, convtest=1, lossfunction="01", x0=p["x0"], z0=p["z0"], u0=p["u0"])
    ref = S.linearsvm(p["D"], p["ell"], p["C"], o)
    got = gpu.linearsvm(p["D"], p["ell"], p["C"], o)
    assert "steps" not in ref and "steps" not in got and "xopt" not in got
    assert got["convtest_failed_at"] == ref["convtest_failed_at"]
    _close("Hnormsq", got["Hnormsq"], ref["Hnormsq"], 1e-7)


def test_engine_reuse_and_rho_guard(gpu):
    p = gpu.synth.lasso_problem(8, 100, 30)
    args = dict(D=p["D"], s=p["s"], rho=1.0, parallel=0)
    args["lambda"] = p["lam"]
    minx, minz, _ = gpu.getproxops("lasso", args)
    base = dict(A=1, B=-1, c=0, m=30, nA=30, nB=30)
    a = gpu.admm(minx, minz, dict(base))
    b = gpu.admm(minx, minz, dict(base))  # second run on the same device-resident problem
    np.testing.assert_array_equal(a["xvals"], b["xvals"])  # bitwise reproducible (no float atomics)
    with pytest.raises(gpu.AdmmError):
        gpu.admm(minx, minz, dict(base, rho=3.0))  # cached factor was built for rho = 1
    with pytest.raises(ValueError):
        gpu.admm(minx, minz, dict(base, A=p["D"]))  # wrong constraint for this prox pair


def test_relax_with_svm_is_rejected(gpu):
    p = gpu.synth.svm_problem(0, 16, 16)
    with pytest.raises(gpu.AdmmError):
        gpu.linearsvm(p["D"], p["ell"], p["C"], dict(relax=1.5, x0=p["x0"], z0=p["z0"], u0=p["u0"]))


def test_tv_fused_equals_three_kernel_form_large(gpu, monkeypatch):
    """n large enough for > 1024 tiles: the fused iteration kernel (one launch, per-tile partial sums packed
    by a second kernel) against the forward / backward / prox kernels it replaces."""
    n = 3_000_001  # odd: exercises the unpaired tail
    p = gpu.synth.tv_problem(4, n)
    o = dict(maxiters=12, domaxiters=1, objevals=1, record_history=0)
    a = gpu.totalvariation(p["s"], p["lam"], dict(o))
    monkeypatch.setenv("ADMM_HIP_TV_UNFUSED", "1")
    b = gpu.totalvariation(p["s"], p["lam"], dict(o))
    assert a["steps"] == b["steps"] == 12
    for k in ("pnorm", "dnorm", "perr", "derr", "objevals", "xopt", "zopt", "uopt"):
        _close(k, a[k], b[k], 1e-11)


@pytest.mark.parametrize("n", [112, 113, 127, 1935, 1936, 1937, 3 * 1936 - 1, 3 * 1936, 3 * 1936 + 1, 3 * 1936 + 7,
                               3 * 1936 + 8, 3 * 1936 + 9, 5 * 1936 + 57, 40009])
@pytest.mark.parametrize("rho", [1.0, 7.0])
def test_total_variation_thread_run_kernel_tile_boundaries(gpu, n, rho):
    """tv_direct2_kernel (tv_direct2.h): signal lengths around the tile size (1936 owned positions at rho = 1), around
    the 8-position runs of a thread (the one run the end of the signal cuts), and down to two window margins -- both
    ends by mirror images, one tile or several; with and without history / objective (two instantiations), stopping by
    tolerance (launches enqueued behind the stop must leave the state alone) and after a fixed count."""
    p = gpu.synth.tv_problem(n % 89, n)
    for o in (dict(objevals=1, rho=rho, maxiters=9, domaxiters=1), dict(rho=rho)):
        ref = S.totalvariation(p["s"], p["lam"], dict(o))
        _compare(gpu.totalvariation(p["s"], p["lam"], dict(o)), ref)
        got = gpu.totalvariation(p["s"], p["lam"], dict(o, record_history=0))
        assert got["steps"] == ref["steps"]
        for k in ("xopt", "zopt", "uopt", "pnorm", "dnorm", "perr", "derr"):
            _close(k, got[k], ref[k], TOL, None)


def test_total_variation_thread_run_kernel_equals_first_form(gpu, monkeypatch):
    """the two direct kernels on 3e6 + 1 elements (odd: the cut run; 1550 tiles), 1e-11"""
    n = 3_000_001
    p = gpu.synth.tv_problem(5, n)
    o = dict(maxiters=12, domaxiters=1, objevals=1, record_history=0)
    a = gpu.totalvariation(p["s"], p["lam"], dict(o))
    monkeypatch.setenv("ADMM_HIP_TV_DIRECT1", "1")
    b = gpu.totalvariation(p["s"], p["lam"], dict(o))
    for k in ("pnorm", "dnorm", "perr", "derr", "objevals", "xopt", "zopt", "uopt"):
        _close(k, a[k], b[k], 1e-11)


# ---------------------------------------------------------------------------- golden fixtures
def _opts(npz):
    o = {}
    for k in npz.files:
        if k.startswith("opt_"):
            v = npz[k]
            o[k[4:]] = v.item() if v.ndim == 0 else v
    return o


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "*.npz"))), ids=os.path.basename)
def test_golden_fixture(gpu, path):
    z = np.load(path, allow_pickle=False)
    name = os.path.basename(path)
    o = _opts(z)
    inp = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
    if name.startswith("consensus"):
        got = gpu.lasso(inp["D"], inp["s"], float(inp["lam"]), o)
    elif name.startswith("lp"):
        got = gpu.linearprogram(inp["b"], inp["D"], inp["s"], o)
    elif name.startswith("qpstd"):
        got = gpu.quadraticprogram(inp["P"], inp["q"], float(inp["r"]), inp["D"], inp["s"], o)
    elif name.startswith("lasso"):
        got = gpu.lasso(inp["D"], inp["s"], float(inp["lam"]), o)
    elif name.startswith("lad"):
        got = gpu.lad(inp["D"], inp["s"], o)
    elif name.startswith("huber"):
        got = gpu.huberfit(inp["D"], inp["s"], o)
    elif name.startswith("svm"):
        got = gpu.linearsvm(inp["D"], inp["ell"], float(inp["C"]), o)
    elif name.startswith("qp"):
        got = gpu.quadraticprogram(inp["P"], inp["q"], float(inp["r"]), inp["lb"], inp["ub"], o)
    elif name.startswith("basispursuit"):
        got = gpu.basispursuit(inp["D"], inp["s"], o)
    elif name.startswith("tv"):
        got = gpu.totalvariation(inp["s"], float(inp["lam"]), o)
    elif name.startswith("model"):
        got = gpu.model(inp["P"], inp["Q"], inp["r"], inp["s"], o)
    else:
        pytest.fail(f"no runner for fixture {name}")
    ref = {k[4:]: z[k] for k in z.files if k.startswith("out_")}
    if name.startswith("svm_01"):  # discontinuous prox: the fixture keeps the first 40 iterations only
        for k in ("xvals", "zvals", "uvals", "pnorm", "perr", "objevals"):
            _close(k, got[k], ref[k], 1e-6, limit=20)
        return
    assert got["steps"] == int(z["steps"])
    tol = 1e-7 if name.startswith(("svm", "lp", "qpstd")) else TOL
    for k in HIST:
        if k in ref:
            _close(k, got[k], ref[k], tol)


def test_total_variation_one_launch_iteration(gpu, monkeypatch):
    """the opt-in form of the fused TV iteration whose reduction tree and finalize logic run inside the one kernel
    (last tile of a group sums the group, last group finalizes): same iterates and histories as the oracle"""
    monkeypatch.setenv("ADMM_HIP_TV_ONE_LAUNCH", "1")
    for n in (200, 5000, 300011):
        p = gpu.synth.tv_problem(3, n)
        o = dict(objevals=1, maxiters=40)
        got = gpu.totalvariation(p["s"], p["lam"], dict(o))
        ref = S.totalvariation(p["s"], p["lam"], dict(o))
        _compare(got, ref)


def test_consensus_lasso_partial_row_gather_path(gpu):
    """n >= 1536: every slice's x-solve is the lower-triangle kernel and the exchange kernel assembles x_k from its
    partial rows (no symv_reduce launches); 3 slices, against the 3-slice oracle"""
    p = gpu.synth.lasso_problem(7, 3 * 1700, 1600)
    o = dict(objevals=1, parallel="both", maxiters=25)
    got = gpu.lasso(p["D"], p["s"], p["lam"], dict(o, workers=3, xsolve="inverse"))
    ref = S.lasso(p["D"], p["s"], p["lam"], dict(o, slices=0), workers=3)
    _compare(got, ref, keys=("xvals", "zvals", "uvals", "pnorm", "dnorm", "perr", "derr", "objevals", "Hnormsq"))
    assert got["engine_info"]["xsolve_used"] == "inverse"


@pytest.mark.parametrize("m,n,kind", [(257, 3, "hinge"), (1000, 130, "hinge"), (1000, 130, "01"), (6000, 400, "hinge"),
                                      (900, 513, "hinge"), (1500, 400, "deficient"), (640, 96, "dplus")])
def test_unwrapped_two_launch_iteration_matches_the_generic_path(gpu, monkeypatch, m, n, kind):
    """unwrapped.hip: the linear SVM iteration as two launches (explicit pinv(D), partial rows carried across the kernel
    boundaries, the finalize logic riding along with the next launch) against the generic five-launch A = D iteration
    of the same engine, and against the oracle: early stop, histories, objective, both losses, a rank-deficient D
    (pseudo-inverse from the eigen-solver) and a caller-supplied Dplus."""
    if kind == "deficient":
        p = gpu.synth.rank_deficient_pixels(seed=3, m=m, n=n)
    else:
        p = gpu.synth.mnist_like_problem(seed=2, m=m, n=n, digit=1)
    D, ell, Cv = p["D"], p["ell"], p["C"]
    o = dict(objevals=1, maxiters=120, x0=p["x0"], z0=p["z0"], u0=p["u0"])
    if kind == "01":
        o["lossfunction"] = "01"
        o["maxiters"] = 12  # the 0-1 prox is discontinuous (q24): the first iterations only
    if kind == "dplus":
        o["Dplus"] = np.linalg.pinv(D)
    got = gpu.linearsvm(D, ell, Cv, dict(o))
    assert got["engine_info"]["unwrapped_fused"]
    monkeypatch.setenv("ADMM_HIP_NO_UNWRAPPED_FUSED", "1")
    gen = gpu.linearsvm(D, ell, Cv, dict(o))
    monkeypatch.delenv("ADMM_HIP_NO_UNWRAPPED_FUSED")
    assert not gen["engine_info"]["unwrapped_fused"]
    assert got["steps"] == gen["steps"]
    tol = 1e-7 if kind == "deficient" else 1e-9
    for k in ("xvals", "zvals", "uvals", "pnorm", "perr", "objevals", "xopt", "zopt", "uopt"):
        _close(k, got[k], gen[k], tol)
    assert np.isnan(got["dnorm"]).all()
    if kind in ("hinge", "dplus"):
        ref = S.linearsvm(D, ell, Cv, dict(o))
        _compare(got, ref, tol=1e-7)


@pytest.mark.parametrize("solver,m,n,opts", [
    ("linearsvm", 24000, 400, dict(objevals=1, maxiters=60)),                        # stops early
    ("linearsvm", 24000, 400, dict(objevals=1, maxiters=21, domaxiters=1, record_history=0)),
    ("linearsvm", 17011, 448, dict(lossfunction="01", maxiters=9, domaxiters=1)),    # ragged last block, widest D
    ("linearsvm", 60000, 120, dict(objevals=1, maxiters=30, domaxiters=1)),
    ("lad", 20000, 333, dict(objevals=1, nodualerror=1, maxiters=25, domaxiters=1)),
    ("huberfit", 20000, 333, dict(objevals=1, nodualerror=1, maxiters=25, domaxiters=1)),
])
def test_one_pass_iteration_on_a_tall_narrow_matrix(gpu, monkeypatch, solver, m, n, opts):
    """unwrapped.hip: ad_onepass_kernel -- an A = D iteration without a dual residual reads a tall, narrow D ONCE (D*x,
    the element update and the partial rows of D'*(c + z - u) per 64-row block held in registers) instead of twice:
    against the generic path of the same engine (ADMM_HIP_NO_ONEPASS), against the oracle, and that it is the path
    that ran (no D*x launch of its own)."""
    if solver == "linearsvm":
        p = gpu.synth.mnist_like_problem(seed=2, m=m, n=n, digit=1, labels=gpu.synth.reference_mnist_labels("train"))
        o = dict(opts, x0=p["x0"], z0=p["z0"], u0=p["u0"])
        run = lambda: gpu.linearsvm(p["D"], p["ell"], p["C"], dict(o))
        ref = S.linearsvm(p["D"], p["ell"], p["C"], {k: v for k, v in o.items() if k != "record_history"})
    else:
        p = gpu.synth.lad_problem(4, m, n)
        o = dict(opts)
        run = lambda: getattr(gpu, solver)(p["D"], p["s"], dict(o))
        ref = getattr(S, solver)(p["D"], p["s"], dict(o))
    got = run()
    monkeypatch.setenv("ADMM_HIP_NO_ONEPASS", "1")
    gen = run()
    monkeypatch.delenv("ADMM_HIP_NO_ONEPASS")
    assert got["steps"] == gen["steps"] == ref["steps"]
    keys = ["pnorm", "perr", "xopt", "zopt", "uopt"] + (["objevals"] if opts.get("objevals") else [])
    if opts.get("record_history", 1):
        keys += ["xvals", "zvals", "uvals"]
    for k in keys:
        _close(k, got[k], gen[k], 1e-9)
        if opts.get("lossfunction") != "01":  # (the 0-1 prox is discontinuous, q24: the two device paths agree)
            _close(k, got[k], ref[k], 1e-7)
    assert np.isnan(got["dnorm"]).all()
    # the path: an engine of the same shape, event-timed -- the one-pass iteration launches no D*x kernel
    L = gpu._lib
    if solver == "linearsvm":
        eng = gpu.Engine(L.PROB_LINEARSVM, D=p["D"], ell=p["ell"], Cval=p["C"])
    else:
        eng = gpu.Engine(L.PROB_LAD if solver == "lad" else L.PROB_HUBERFIT, D=p["D"], s=p["s"])
    eng.set_profiling([L.K_GEMV_N, L.K_PROX])
    eng.run(maxiters=5, domaxiters=1, record_history=0, nodualerror=1)
    assert eng.kernel_time(L.K_GEMV_N)[1] == 0 and eng.kernel_time(L.K_PROX)[1] == 5
    eng.close()


@pytest.mark.parametrize("opts", [dict(), dict(objevals=1, maxiters=40), dict(fast=1, fasttype="strong", maxiters=60),
                                  dict(relax=1.6, stopcond="both", maxiters=50), dict(convtest=1, stopcond="hnorm"),
                                  dict(maxiters=13, domaxiters=1), dict(record_history=0, maxiters=70)])
def test_lasso_deferred_finalize_on_the_packed_inverse(gpu, monkeypatch, opts):
    """n = 1600 (>= 1536: tile-packed inverse, lower-triangle kernel): the finalize logic of iteration i rides along
    with the x-solve of iteration i + 1 (symv_lower_fin_kernel) and the stop decision still lands on the iteration the
    reference stops at -- early stops in the middle of a host batch, relaxation, fast ADMM, the convergence test, a
    fixed iteration count that is not a multiple of the batch, no histories.  Compared with the oracle and with the
    one-launch tail (ADMM_HIP_NO_DEFERRED_FINALIZE)."""
    p = gpu.synth.lasso_problem(7, 2000, 1600)
    D, s, lam = p["D"], p["s"], p["lam"]
    got = gpu.lasso(D, s, lam, dict(opts, xsolve="inverse"))
    assert got["engine_info"]["xsolve_used"] == "inverse"
    monkeypatch.setenv("ADMM_HIP_NO_DEFERRED_FINALIZE", "1")
    one = gpu.lasso(D, s, lam, dict(opts, xsolve="inverse"))
    monkeypatch.delenv("ADMM_HIP_NO_DEFERRED_FINALIZE")
    assert got["steps"] == one["steps"]
    for k in ("xopt", "zopt", "uopt", "pnorm", "dnorm", "perr", "derr"):
        assert np.array_equal(got[k], one[k]), k  # the same kernels in the same order: bitwise
    if opts.get("record_history", 1):
        ref = S.lasso(D, s, lam, dict(opts))
        _compare(got, ref, tol=1e-7)
    else:
        ref = S.lasso(D, s, lam, {k: v for k, v in opts.items() if k != "record_history"})
        assert got["steps"] == ref["steps"] and "xvals" not in got
        _close("xopt", got["xopt"], ref["xopt"], 1e-7)


def test_lasso_objective_switches_to_the_gram_form_after_calibration(gpu):
    """obj_gram = 0 (default): the first batch of the first objevals run evaluates both forms of 1/2*||D*x - s||^2, the
    literal values are recorded, and the later batches use the form built on the x-update's right-hand side -- the
    recorded objective matches the literal engine (obj_gram = -1) to 1e-10 throughout, and a second run starts in that
    form (at any size: it costs nothing)."""
    L = gpu._lib
    p = gpu.synth.lasso_problem(11, 42000, 1600)
    kw = dict(D=p["D"], s=p["s"], lam=p["lam"], rho=1.0, xsolve=L.XSOLVE_INVERSE)
    auto, lit = gpu.Engine(L.PROB_LASSO, **kw), gpu.Engine(L.PROB_LASSO, obj_gram=-1, **kw)
    small = gpu.Engine(L.PROB_LASSO, D=p["D"][:3000], s=p["s"][:3000], lam=p["lam"], rho=1.0)
    try:
        run = dict(maxiters=40, domaxiters=1, objevals=1, record_history=0, check_every=8)
        sa, sl = auto.run(**run), lit.run(**run)
        assert sa.obj_gram_used == 1 and sl.obj_gram_used == 0
        oa, ol = auto.fetch(L.F_OBJEVALS, 40), lit.fetch(L.F_OBJEVALS, 40)
        assert np.array_equal(oa[:8], ol[:8])  # the calibration batch records the literal values
        assert np.max(np.abs(oa - ol) / np.abs(ol)) < 1e-10
        assert not np.array_equal(oa[8:], ol[8:])  # ... and the later ones really come from the other formula
        s2 = auto.run(**run)
        assert s2.obj_gram_used == 1
        assert np.max(np.abs(auto.fetch(L.F_OBJEVALS, 40) - ol) / np.abs(ol)) < 1e-10
        assert small.run(**run).obj_gram_used == 1
    finally:
        auto.close(), lit.close(), small.close()


@pytest.mark.parametrize("opts", [dict(objevals=1), dict(record_history=0), dict(record_history=0, maxiters=11, domaxiters=1),
                                  dict(stopcond="both", convtest=1, maxiters=90), dict(rho=3.0, objevals=1, maxiters=37)])
@pytest.mark.parametrize("n", [4099, 50000])
def test_total_variation_deferred_tail_matches_the_two_small_launches(gpu, monkeypatch, n, opts):
    """1-D TV, fused kernel: the tile-partial sums and the finalize logic of iteration i ride along with iteration i + 1's
    launch (one extra workgroup); the iteration after a stop has then run speculatively into the other ping-pong buffers
    (three y buffers in rotation), and z, u, x are still those of the stopping iteration -- bitwise the results of the
    form with two small launches per iteration, and the oracle's."""
    p = gpu.synth.tv_problem(3, n)
    got = gpu.totalvariation(p["s"], p["lam"], dict(opts))
    monkeypatch.setenv("ADMM_HIP_NO_DEFERRED_FINALIZE", "1")
    two = gpu.totalvariation(p["s"], p["lam"], dict(opts))
    monkeypatch.delenv("ADMM_HIP_NO_DEFERRED_FINALIZE")
    assert got["steps"] == two["steps"]
    for k in ("xopt", "zopt", "uopt", "pnorm", "dnorm", "perr", "derr", "xvals", "zvals", "uvals", "objevals"):
        assert (k in got) == (k in two), k
        if k in got and k in ("xopt", "zopt", "uopt", "xvals", "zvals", "uvals"):
            assert np.array_equal(got[k], two[k]), k  # the iterates do not depend on how the tile partials are summed
        elif k in got:
            _close(k, got[k], two[k], 1e-12)  # (the passenger and tv_pack associate the tile sums differently)
    ref = S.totalvariation(p["s"], p["lam"], {k: v for k, v in opts.items() if k != "record_history"})
    assert got["steps"] == ref["steps"]
    _close("xopt", got["xopt"], ref["xopt"], 1e-7)
    _close("zopt", got["zopt"], ref["zopt"], 1e-7)


@pytest.mark.parametrize("n,rho", [(1936, 1.0), (1937, 1.0), (1993, 1.0), (3872, 1.0), (3873, 1.0), (3929, 1.0), (5809, 1.0),
                                   (20000, 10.0), (12001, 30.0), (7000, 0.05)])
def test_total_variation_direct_kernel_tile_boundaries(gpu, n, rho):
    """tv_direct_kernel: lengths around multiples of the owned tile (1936 positions at rho = 1: the last tile is then
    empty-but-one, exactly full, or ends inside the margin of the tile before it -- every such tile must take the exact
    scan path), and slowly decaying kernels (rho = 10, 30: 124 / 216 taps per side)."""
    p = gpu.synth.tv_problem(n % 89, n)
    o = dict(objevals=1, rho=rho, maxiters=25)
    _compare(gpu.totalvariation(p["s"], 1.0, dict(o)), S.totalvariation(p["s"], 1.0, dict(o)), tol=1e-7)


@pytest.mark.parametrize("m,n", [(161, 231), (40, 400), (3, 17)])
def test_svm_wide_matrix_uses_the_pseudo_inverse(gpu, m, n):
    """linearsvm.m:185 x = pinv(D)*(z - u) exists for m < n as well (D'D is then rank deficient by construction): the
    engine's eigen-solver path against the oracle's np.linalg.pinv.  With full row rank D*pinv(D) = I, so D*x = z - u
    exactly and u collapses to rounding noise: u is compared absolutely."""
    q = gpu.synth.mnist_like_problem(seed=5, m=m, n=n, digit=3)
    o = dict(objevals=1, x0=q["x0"], z0=q["z0"], u0=q["u0"])
    got = gpu.linearsvm(q["D"], q["ell"], q["C"], dict(o))
    ref = S.linearsvm(q["D"], q["ell"], q["C"], dict(o))
    assert got["engine_info"]["pinv_used"] and got["engine_info"]["rank"] <= m
    assert got["steps"] == ref["steps"]
    for k in ("xvals", "zvals", "pnorm", "perr", "objevals", "xopt", "zopt"):
        _close(k, got[k], ref[k], 1e-6)
    np.testing.assert_allclose(got["uvals"], ref["uvals"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(got["uopt"], ref["uopt"], rtol=1e-6, atol=1e-9)


def test_lasso_objective_falls_back_to_the_literal_form_when_it_cancels(gpu):
    """lasso.m:227 on a near-interpolating problem (s = D*x0 exactly, lambda tiny): 1/2*||D*x - s||^2 falls five orders
    below 1/2*s's, and the right-hand-side form of the objective -- terms of the size of s's that cancel -- loses those
    digits.  The engine tracks eps*|terms|/|objective| per recorded value and returns to the literal D*x pass once it
    leaves 1e-10, for the rest of the run and for later runs; every recorded objective agrees with the oracle entry by
    entry (no floor), the bar being 1e-6."""
    L = gpu._lib
    p = gpu.synth.lasso_problem(7, 400, 100)
    D = p["D"]
    s = D @ p["testx"]
    lam = 1e-7 * float(np.max(np.abs(D.T @ s)))
    o = dict(objevals=1, maxiters=400)
    ref = S.lasso(D, s, lam, dict(o))
    obj_ref = np.asarray(ref["objevals"])
    assert obj_ref[-1] < 1e-5 * 0.5 * float(s @ s)  # the misfit really is that small
    eng = gpu.Engine(L.PROB_LASSO, D=D, s=s, lam=lam, rho=1.0, xsolve=L.XSOLVE_INVERSE)
    try:
        for run in range(2):
            st = eng.run(objevals=1, maxiters=400)
            assert st.steps == ref["steps"]
            got = eng.fetch(L.F_OBJEVALS, st.steps)
            assert np.max(np.abs(got - obj_ref) / np.abs(obj_ref)) < 1e-6
            info = eng.info()
            assert info["obj_form_literal"] and info["obj_bound_max"] > 1e-10
            if run == 1:
                assert int(st.obj_gram_used) == 0  # the second run never leaves the literal form
    finally:
        eng.close()
    # an ordinary problem (lassotest.m's recipe) keeps the one-pass form: bound ~ 1e-15
    q = gpu.synth.lasso_problem(0, 256, 64)
    res = gpu.lasso(q["D"], q["s"], q["lam"], dict(objevals=1))
    assert not res["engine_info"]["obj_form_literal"] and 0 < res["engine_info"]["obj_bound_max"] < 1e-12


@pytest.mark.parametrize("kind", ["lp", "qp", "bp"])
def test_eliminated_constraint_solves_are_probed(gpu, kind):
    """LP / standard-form QP (getProxOps.m:1363, 1410: a pivoted KKT solve per iteration in the reference) and basis
    pursuit (basispursuit.m:116-120) run on affine maps the engine builds ONCE through explicit inverses of D*inv(M)*D'
    resp. D*D' -- cond(D)^2.  create() probes D*x = s on the finished map: a graded D with cond 1e2 / 1e3 stays at
    oracle parity, cond 1e7 is refused (ADMM_E_NUMERIC) instead of iterating off the constraint."""
    rng = np.random.default_rng(11)
    m, n = 24, 80
    for kappa, ok in ((1e2, True), (1e3, True), (1e7, False)):
        D = np.asfortranarray(gpu.synth.graded_matrix(3, n, m, kappa).T)  # m x n, singular values 1 .. 1/kappa
        truex = np.abs(rng.standard_normal(n))
        s = D @ truex
        o = dict(maxiters=60, domaxiters=1, objevals=1)
        if kind == "lp":
            b = rng.random(n) + 0.5
            run = lambda: gpu.linearprogram(b, D, s, dict(o))
            ref = lambda: S.linearprogram(b, D, s, dict(o))
        elif kind == "qp":
            G = rng.standard_normal((n, n))
            P, q = G @ G.T / n + np.eye(n), rng.standard_normal(n)
            run = lambda: gpu.quadraticprogram(P, q, 0.3, D, s, dict(o))
            ref = lambda: S.quadraticprogram_standard(P, q, 0.3, D, s, dict(o))
        else:
            run = lambda: gpu.basispursuit(D, s, dict(o))
            ref = lambda: S.basispursuit(D, s, dict(o))
        if ok:
            got = run()
            _compare(got, ref(), tol=1e-6)
            assert np.max(np.abs(D @ got["xopt"] - s)) <= 1e-8 * np.max(np.abs(s))
        else:
            with pytest.raises(gpu.AdmmError, match="ill-conditioned"):
                run()
