"""Infinity-Cache share of the packed inverse (ADMM_HIP_SYMV_CACHE_MB, read at create) against the headline loop's step time.
A 20000 x 10000 lasso: the same 400 MB inverse as config 2, cheap setup."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import admm_project_amd as ap  # noqa: E402

L = ap._lib
p = ap.synth.lasso_problem(seed=1, rows=20000, cols=10000)
for mb in [int(v) for v in sys.argv[1:]] or [168]:
    os.environ["ADMM_HIP_SYMV_CACHE_MB"] = str(mb)
    e = ap.Engine(L.PROB_LASSO, D=p["D"], s=p["s"], lam=p["lam"], rho=1.0, xsolve=L.XSOLVE_INVERSE)
    e.run(maxiters=50, domaxiters=1, record_history=0)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        s = e.run(maxiters=2000, domaxiters=1, record_history=0)
        best = min(best, (time.perf_counter() - t0) / s.steps)
    print("cache MB", mb, "us/it %.2f" % (1e6 * best), flush=True)
    e.close()
