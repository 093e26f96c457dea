"""One-off randomized parity sweep of the kernels added late in round 2 (direct 1-D TV, Toeplitz row stage of the 2-D
TV solve, two-launch SVM iteration, deferred finalize): random shapes and parameters against the oracle."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import admm_project_amd as ap  # noqa: E402
from oracle import solvers_ref as S  # noqa: E402  (test infrastructure: this script is a checker, not the product)

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


worst = {}


def check(tag, got, ref, keys):
    assert got["steps"] == ref["steps"], (tag, got["steps"], ref["steps"])
    for k in keys:
        if k in ref:
            e = rel(got[k], ref[k])
            if k == "uopt" and "zopt" in ref:  # a u that is rounding noise (wide D: z reproduces D*x) is measured against z
                e *= np.max(np.abs(ref[k])) / max(np.max(np.abs(ref[k])), np.max(np.abs(ref["zopt"])))
            worst[tag] = max(worst.get(tag, 0.0), e)
            assert e < 1e-6, (tag, k, e)


for t in range(24):  # 1-D TV: direct kernel for small halos, scan / three-kernel forms otherwise
    n = int(rng.integers(2, 60000))
    rho = float(10 ** rng.uniform(-1.5, 2.2))
    lam = float(10 ** rng.uniform(-1, 1))
    p = ap.synth.tv_problem(t, n)
    o = dict(rho=rho, maxiters=int(rng.integers(3, 40)), objevals=1, record_history=int(rng.integers(0, 2)))
    got = ap.totalvariation(p["s"], lam, dict(o))
    ref = S.totalvariation(p["s"], lam, {k: v for k, v in o.items() if k != "record_history"})
    check("tv1d", got, ref, ("xopt", "zopt", "uopt", "pnorm", "dnorm", "objevals"))
print("tv1d ok", worst.get("tv1d"), flush=True)

for t in range(12):  # 2-D TV, power-of-two sides: column DCT + Toeplitz row stage (or row DCT for large rho / narrow)
    H, W = (int(2 ** rng.integers(3, 8)) for _ in range(2))
    rho = float(10 ** rng.uniform(-1, 1.3))
    img = rng.standard_normal((H, W)) + 2.0 * (rng.random((H, W)) > 0.7)
    o = dict(rho=rho, maxiters=int(rng.integers(3, 25)), objevals=1)
    got = ap.totalvariation2d(img, 0.7, dict(o))
    ref = S.totalvariation2d(img, 0.7, dict(o))
    check("tv2d", got, ref, ("xopt", "zopt", "uopt", "pnorm", "dnorm", "objevals"))
print("tv2d ok", worst.get("tv2d"), flush=True)

for t in range(16):  # linear SVM: two-launch iteration for small m*n, generic otherwise
    m, n = int(rng.integers(3, 3000)), int(rng.integers(1, 300))
    q = ap.synth.mnist_like_problem(seed=t, m=m, n=n, digit=int(rng.integers(0, 10)))
    o = dict(objevals=1, maxiters=int(rng.integers(2, 60)), x0=q["x0"], z0=q["z0"], u0=q["u0"])
    try:
        got = ap.linearsvm(q["D"], q["ell"], q["C"], dict(o))
    except ap.AdmmError as exc:  # wide or rank-deficient random matrices take the pseudo-inverse path or are refused
        print("svm skip", m, n, str(exc)[:60])
        continue
    ref = S.linearsvm(q["D"], q["ell"], q["C"], dict(o))
    try:
        check("svm", got, ref, ("xopt", "zopt", "uopt", "pnorm", "objevals"))
    except AssertionError:
        print("svm case", m, n, o["maxiters"], "max|u| ref/got", np.max(np.abs(ref["uopt"])), np.max(np.abs(got["uopt"])),
              "max|z|", np.max(np.abs(ref["zopt"])), "max|u - u_ref|", np.max(np.abs(got["uopt"] - ref["uopt"])), flush=True)
        raise
print("svm ok", worst.get("svm"), flush=True)

for t in range(6):  # lasso on the packed inverse: deferred finalize, early stops
    m, n = int(rng.integers(1600, 2600)), int(rng.integers(1536, 1700))
    p = ap.synth.lasso_problem(t, max(m, n + 50), n)
    o = dict(maxiters=int(rng.integers(5, 80)), objevals=int(rng.integers(0, 2)), rho=float(10 ** rng.uniform(-0.5, 0.5)))
    got = ap.lasso(p["D"], p["s"], p["lam"], dict(o, xsolve="inverse"))
    ref = S.lasso(p["D"], p["s"], p["lam"], dict(o))
    check("lasso", got, ref, ("xopt", "zopt", "uopt", "pnorm", "dnorm", "perr", "derr", "objevals"))
print("lasso ok", worst.get("lasso"), flush=True)
print("ALL OK", worst)
