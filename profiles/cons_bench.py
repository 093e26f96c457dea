"""Config 4 on one GPU alone: consensus lasso, 8 local slices of 12500 x 10000 (used while tuning consensus.hip)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import admm_project_amd as ap  # noqa: E402

L = ap._lib
m, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100000, 10000)
p = ap.synth.lasso_problem(seed=1, rows=m, cols=n)
sl = ap.errorcheck.slicemaker(0, 8, m)
eng = ap.Engine(L.PROB_LASSO_CONSENSUS, D=p["D"], s=p["s"], lam=p["lam"], rho=1.0, slices=sl)
kw = dict(domaxiters=1, record_history=0, rho=1.0, stopcond="both")
eng.run(maxiters=5, **kw)
for rep in range(3):
    t0 = time.perf_counter()
    s = eng.run(maxiters=200, **kw)
    dt = time.perf_counter() - t0
    print("ms/it %.4f" % (1e3 * dt / s.steps), flush=True)
eng.set_profiling([L.K_XSOLVE, L.K_PROX])
eng.run(maxiters=64, **kw)
eng.set_profiling(False)
for name, k in (("xsolve", L.K_XSOLVE), ("tail kernels", L.K_PROX)):
    ms, cnt = eng.kernel_time(k)
    print(name, "avg ms %.4f over %d" % (ms / max(1, cnt), cnt))
