"""A/B of the symmetric lower-triangle x-solve's panel ring (symv.hip: sy_tile<NBUF>, ADMM_SYMV_NBUF = 2 | 3 | 4):
the headline loop shape (n = 10^4 factor from a 20000 x 10000 lasso: same x-solve, cheap setup) and the 8-slice
consensus loop (8 x 2500 x 10000: same batched x-solve).  One child process per setting (the knob is read once)."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np

    import admm_project_amd as ap

    L = ap._lib
    p = ap.synth.lasso_problem(seed=1, rows=20000, cols=10000)
    e = ap.Engine(L.PROB_LASSO, D=p["D"], s=p["s"], lam=p["lam"], rho=1.0, xsolve=L.XSOLVE_INVERSE)
    e.run(maxiters=20, domaxiters=1, record_history=0)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        s = e.run(maxiters=400, domaxiters=1, record_history=0)
        best = min(best, (time.perf_counter() - t0) / s.steps)
    e.set_profiling([L.K_XSOLVE], stride=4)
    e.run(maxiters=400, domaxiters=1, record_history=0)
    ms, launches = e.kernel_time(L.K_XSOLVE)
    e.close()
    print("NBUF=%s headline-shape loop %.1f us/it, x-solve %.1f us" % (os.environ.get("ADMM_SYMV_NBUF", "2"), best * 1e6,
                                                                     1e3 * ms / max(1, launches)), flush=True)
    m = 20000
    sl = ap.errorcheck.slicemaker(0, 8, m)
    c = ap.Engine(L.PROB_LASSO_CONSENSUS, D=p["D"], s=p["s"], lam=p["lam"], rho=1.0, slices=sl)
    c.run(maxiters=10, domaxiters=1, record_history=0)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        s = c.run(maxiters=100, domaxiters=1, record_history=0)
        best = min(best, (time.perf_counter() - t0) / s.steps)
    c.close()
    print("NBUF=%s consensus 8 x 2500 x 10000 loop %.1f us/it  frac %.3f" % (
        os.environ.get("ADMM_SYMV_NBUF", "2"), best * 1e6, 8 * 4.0 * 10000 * 10001 / best / 8e12), flush=True)
    sys.exit(0)

for nb in (sys.argv[1:] or ["2", "3", "4", "2", "3"]):
    env = dict(os.environ, ADMM_SYMV_NBUF=nb)
    subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=False)
