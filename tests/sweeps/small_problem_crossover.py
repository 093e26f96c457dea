"""Where the device engine stops paying off: per-iteration time of the cached-factor lasso loop on the GPU (fixed cost
per admm_engine_run call + cost per iteration, least squares over several iteration counts) next to the CPU restatement
of the reference's loop (oracle, host BLAS) on the same box, from the testers' default size (lassotest.m: 2^8 x 2^6)
upwards.  A checker-side measurement (imports the oracle), not part of the product.
    python tests/sweeps/small_problem_crossover.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import admm_project_amd as ap  # noqa: E402
from oracle import solvers_ref as S  # noqa: E402

L = ap._lib
print("rows x cols | engine: fixed us per run + us per iteration | oracle (CPU): us per iteration | ratio per iteration")
for m, n in ((256, 64), (1024, 128), (2048, 512), (4096, 1024), (8192, 2048), (16384, 4096)):
    p = ap.synth.lasso_problem(seed=1, rows=m, cols=n)
    e = ap.Engine(L.PROB_LASSO, D=p["D"], s=p["s"], lam=p["lam"], rho=1.0, obj_gram=-1)
    e.run(maxiters=50, domaxiters=1, record_history=0)
    Ks, ts = [1, 5, 20, 80, 320], []
    for K in Ks:
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            e.run(maxiters=K, domaxiters=1, record_history=0)
            best = min(best, time.perf_counter() - t0)
        ts.append(best)
    b, a = np.polyfit(Ks, ts, 1)
    e.close()
    it = 200 if n <= 1024 else 40
    t0 = time.perf_counter()
    r = S.lasso(p["D"], p["s"], p["lam"], dict(maxiters=it, domaxiters=1))
    t_all = time.perf_counter() - t0
    t0 = time.perf_counter()
    S.lasso(p["D"], p["s"], p["lam"], dict(maxiters=1, domaxiters=1))  # (setup + one iteration)
    t_one = time.perf_counter() - t0
    cpu = (t_all - t_one) / (it - 1)
    print(f"{m} x {n} | {a * 1e6:.0f} + {b * 1e6:.1f} | {cpu * 1e6:.1f} | {cpu / b:.2f}x", flush=True)
