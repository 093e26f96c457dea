"""The literal two-triangular-solve x-update at n = 10^4 (a 20000 x 10000 lasso: same factor size as config 2, cheap setup)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import admm_project_amd as ap  # noqa: E402

L = ap._lib
p = ap.synth.lasso_problem(seed=1, rows=20000, cols=10000)
e = ap.Engine(L.PROB_LASSO, D=p["D"], s=p["s"], lam=p["lam"], rho=1.0, xsolve=L.XSOLVE_TRSV)
e.run(maxiters=20, domaxiters=1, record_history=0)
t0 = time.perf_counter()
s = e.run(maxiters=400, domaxiters=1, record_history=0)
dt = time.perf_counter() - t0
print("trsv n=10000 it/s %.0f  us/it %.1f" % (s.steps / dt, 1e6 * dt / s.steps), flush=True)
