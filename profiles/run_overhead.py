"""Fixed cost of one admm_engine_run call on the headline problem: time(K) for several K, least-squares a + b*K."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import admm_project_amd as ap  # noqa: E402

L = ap._lib
m, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100000, 10000)
p = ap.synth.lasso_problem(seed=1, rows=m, cols=n)
e = ap.Engine(L.PROB_LASSO, D=p["D"], s=p["s"], lam=p["lam"], rho=1.0, xsolve=L.XSOLVE_INVERSE, obj_gram=-1)
e.run(maxiters=50, domaxiters=1, record_history=0)
Ks, ts = [1, 2, 5, 10, 20, 40, 80, 160], []
for K in Ks:
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        e.run(maxiters=K, domaxiters=1, record_history=0)
        best = min(best, time.perf_counter() - t0)
    ts.append(best)
    print(K, "%.1f us total, %.2f us/step" % (best * 1e6, best * 1e6 / K), flush=True)
b, a = np.polyfit(Ks, ts, 1)
print("fixed %.1f us + %.2f us per step" % (a * 1e6, b * 1e6))
