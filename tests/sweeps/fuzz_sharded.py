"""Randomized parity sweep of the row-sharded paths: 2-8 ranks as threads of this process on device 0 (parallel.LocalGroup,
host-staged transport), every rank with the rows slicemaker(0, N, m) gives it (errorcheck.m:249-259) -- ragged shards,
shards with fewer rows than columns, random rho / options -- against the UNSHARDED oracle (lasso, LAD, Huber, linear SVM)
resp. the N-slice oracle (consensus lasso).  A checker (test infrastructure), not the product.
    python tests/sweeps/fuzz_sharded.py [seed] [cases per solver]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import admm_project_amd as ap  # noqa: E402
from admm_project_amd import parallel  # noqa: E402
from oracle import solvers_ref as S  # noqa: E402  (this script is a checker)

TOL = 1e-6
rng = np.random.default_rng(0)


def rel(a, b, floor=0.0):
    a, b = np.asarray(a, float), np.asarray(b, float)
    assert a.shape == b.shape, (a.shape, b.shape)
    if not np.array_equal(np.isnan(a), np.isnan(b)):
        return float("inf")
    keep = ~np.isnan(b)
    if not keep.any():
        return 0.0
    return float(np.max(np.abs(a[keep] - b[keep])) / max(1e-300, floor, np.max(np.abs(b[keep]))))


def options():
    o = dict(maxiters=int(rng.integers(1, 40)), rho=float(10 ** rng.uniform(-0.7, 0.8)))
    if rng.random() < 0.6:
        o["objevals"] = 1
    if rng.random() < 0.25:
        o["stopcond"] = "both"
    if rng.random() < 0.2:
        o["domaxiters"] = 1
    return o


def case(kind):
    N = int(rng.integers(2, 9))
    n = int(rng.integers(1, 70))
    m = int(rng.integers(max(2 * N, n + 2), max(2 * N, n + 2) + 900))
    o = options()
    seed = int(rng.integers(1 << 30))
    replicated = ("xvals", "pnorm", "dnorm", "perr", "derr", "objevals")
    if kind == "lasso":
        p = ap.synth.lasso_problem(seed, m, n)
        ref = S.lasso(p["D"], p["s"], p["lam"], dict(o))
        call = lambda lo, hi, comm: ap.lasso(p["D"][lo:hi], p["s"][lo:hi], p["lam"], dict(o, comm=comm))  # noqa: E731
        replicated += ("zvals", "uvals")
        rows = ()
    elif kind in ("lad", "huber"):
        p = (ap.synth.lad_problem if kind == "lad" else ap.synth.huber_problem)(seed, m, n)
        f, g = (ap.lad, S.lad) if kind == "lad" else (ap.huberfit, S.huberfit)
        if rng.random() < 0.3:
            o["relax"] = float(rng.uniform(1.0, 1.8))
        ref = g(p["D"], p["s"], dict(o))
        call = lambda lo, hi, comm: f(p["D"][lo:hi], p["s"][lo:hi], dict(o, comm=comm))  # noqa: E731
        rows = ("zopt", "uopt")
    elif kind == "svm":
        n = max(n, 2)  # (one feature: an argument error of the reference)
        q = ap.synth.mnist_like_problem(seed=seed, m=m, n=n, digit=int(rng.integers(0, 10)))
        o = dict(maxiters=o["maxiters"], x0=q["x0"], z0=q["z0"], u0=q["u0"], **({"objevals": 1} if "objevals" in o else {}))
        ref = S.linearsvm(q["D"], q["ell"], q["C"], dict(o))
        call = lambda lo, hi, comm: ap.linearsvm(q["D"][lo:hi], q["ell"][lo:hi], q["C"],  # noqa: E731
                                                dict(o, z0=q["z0"][lo:hi], u0=q["u0"][lo:hi], comm=comm))
        rows = ("zopt", "uopt")
        replicated = ("xvals", "pnorm", "objevals")
    else:  # consensus lasso: one slice per rank
        m = max(m, N * (n + 1))
        p = ap.synth.lasso_problem(seed, m, n)
        o = dict(maxiters=o["maxiters"], rho=o["rho"], parallel="both", **({"objevals": 1} if "objevals" in o else {}))
        ref = S.lasso(p["D"], p["s"], p["lam"], dict(o, slices=0), workers=N)
        call = lambda lo, hi, comm: ap.lasso(p["D"][lo:hi], p["s"][lo:hi], p["lam"], dict(o, comm=comm, workers=1))  # noqa: E731
        replicated = ("xvals", "uvals", "pnorm", "dnorm", "perr", "derr", "objevals")
        rows = ()
    desc = f"{kind} {m}x{n} on {N} ranks { {k: v for k, v in o.items() if np.ndim(v) == 0} }"

    grp = parallel.LocalGroup(N, devices=[0] * N, transport="shm")
    try:
        def rank(r, comm):
            lo, hi = parallel.my_rows(m, comm)
            return lo, hi, call(lo, hi, comm)

        outs = grp.on_ranks(rank)
    finally:
        grp.close()
    worst = 0.0
    scale = max(float(np.max(np.abs(ref[k]))) for k in ("xopt", "zopt", "uopt"))
    for lo, hi, g in outs:
        assert g["steps"] == ref["steps"], ("steps", g["steps"], ref["steps"])
        for k in replicated:
            if k in ref:
                floor = scale if k.endswith("vals") and k != "objevals" else (1e-8 * scale if k in ("pnorm", "dnorm") else 0.0)
                e = rel(g[k], ref[k], floor)
                assert e < TOL, (k, e)
                worst = max(worst, e)
        for k in rows:
            e = rel(g[k], ref[k][lo:hi], scale)
            assert e < TOL, (k, e)
            worst = max(worst, e)
    return desc, worst


def main(seed=0, cases=6, kinds=("lasso", "lad", "huber", "svm", "consensus")):
    global rng
    rng = np.random.default_rng(seed)
    worst, failures = {}, []
    for kind in kinds:
        for _ in range(cases):
            try:
                desc, e = case(kind)
                worst[kind] = max(worst.get(kind, 0.0), e)
            except Exception as exc:  # noqa: BLE001
                failures.append((kind, repr(exc)[:400]))
                print(f"  FAIL {kind}: {repr(exc)[:400]}", flush=True)
        print(f"{kind}: worst {worst.get(kind)}", flush=True)
    return worst, failures


if __name__ == "__main__":
    w, f = main(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 6)
    print("worst relative errors:", w, flush=True)
    print("failures:", len(f), flush=True)
    sys.exit(1 if f else 0)
