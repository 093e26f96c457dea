"""Is hipGraph replay worth it?  lasso with the literal TRSV x-solve (2*n/64 launches per iteration)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import admm_project_amd as ap  # noqa: E402

L = ap._lib
for rows, cols in ((8000, 2000), (2000, 512)):
    p = ap.synth.lasso_problem(seed=1, rows=rows, cols=cols)
    for xs in (L.XSOLVE_TRSV, L.XSOLVE_INVERSE):
        e = ap.Engine(L.PROB_LASSO, D=p["D"], s=p["s"], lam=p["lam"], rho=1.0, xsolve=xs)
        e.run(maxiters=20, domaxiters=1, record_history=0)
        t0 = time.perf_counter()
        s = e.run(maxiters=400, domaxiters=1, record_history=0)
        dt = time.perf_counter() - t0
        print(rows, cols, "trsv" if xs == L.XSOLVE_TRSV else "inverse", "it/s %.0f" % (s.steps / dt), flush=True)
