"""Developer timing: 1-D TV (n = 4096^2) iteration time against the cache policy of tv_fused_kernel
(ADMM_HIP_TV_CACHE: 1 = everything streamed, 2 = s cacheable, 3 = y ping-pong cacheable, 4 = both) and 2-D TV."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import admm_project_amd as ap  # noqa: E402
from admm_project_amd import _lib as L  # noqa: E402

n = 4096 * 4096
p = ap.synth.tv_problem(seed=1, n=n)
for mode in sys.argv[1:] or ["1", "2", "3", "4"]:
    os.environ["ADMM_HIP_TV_CACHE"] = mode
    tv = ap.Engine(L.PROB_TOTALVARIATION, s=p["s"], lam=1.0, nvec=n, device=0)
    tv.run(maxiters=20, domaxiters=1, record_history=0)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        s = tv.run(maxiters=300, domaxiters=1, record_history=0)
        best = min(best, (time.perf_counter() - t0) / s.steps)
    print(f"mode {mode}: {best * 1e3:.4f} ms/iteration  ({7 * 8 * n / best / 1e12:.2f} TB/s algorithmic)", flush=True)
    tv.close()
