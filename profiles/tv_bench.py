"""Config 5 alone: 1-D total variation, n = 4096^2, fixed iteration count (used while tuning tv.hip)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import admm_project_amd as ap  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096 * 4096
p = ap.synth.tv_problem(1, n)
TIMED = int(sys.argv[2]) if len(sys.argv) > 2 else 200
for tag, iters in (("warm", 20), ("timed", TIMED)):
    r = ap.totalvariation(p["s"], p["lam"], dict(maxiters=iters, domaxiters=1, record_history=0, objevals=0))
    print(tag, r["steps"], "it/s %.1f" % (r["steps"] / r["runtime"]), "ms/it %.4f" % (1e3 * r["runtime"] / r["steps"]),
          flush=True)
