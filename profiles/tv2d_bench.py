"""BASELINE config 5 as literally written: anisotropic TV of a 4096 x 4096 image, matrix-free x-update."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import admm_project_amd as ap  # noqa: E402

H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rng = np.random.default_rng(1)
img = np.zeros((H, W))
img[H // 5:H // 2, W // 6:W // 2] = 2.0
img[H // 3:4 * H // 5, W // 3:5 * W // 6] += 1.0
img += rng.standard_normal((H, W))
L = ap._lib
XS = {"auto": L.XSOLVE_AUTO, "cg": L.XSOLVE_CG}[sys.argv[2] if len(sys.argv) > 2 else "auto"]
e = ap.Engine(L.PROB_TV2D, s=np.asfortranarray(img).reshape(-1, order="F"), lam=1.0, shape=(H, W), xsolve=XS)
e.set_profiling(True)
TIMED = int(sys.argv[3]) if len(sys.argv) > 3 else (20 if XS == L.XSOLVE_CG else 200)
for tag, k in (("warm", 3), ("timed", TIMED)):
    t0 = time.perf_counter()
    s = e.run(maxiters=k, domaxiters=1, record_history=0)
    dt = time.perf_counter() - t0
    inner = float(e.fetch(L.F_CG_ITERS, 1)[0])
    print(tag, s.steps, "it/s %.1f" % (s.steps / dt), "ms/it %.3f" % (1e3 * dt / s.steps), "inner/it %.1f" % (inner / s.steps),
          flush=True)
for name, kid in (("xsolve", L.K_XSOLVE), ("prox", L.K_PROX), ("finalize", L.K_FINALIZE)):
    ms, cnt = e.kernel_time(kid)
    print(name, "total ms %.2f" % ms, "launches", cnt, "avg ms %.4f" % (ms / max(1, cnt)))
