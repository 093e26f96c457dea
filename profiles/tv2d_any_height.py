import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import admm_project_amd as ap
L = ap._lib
for (H, W) in ((3000, 2000), (4000, 4096), (1080, 1920)):
    rng = np.random.default_rng(1)
    img = rng.standard_normal((H, W))
    for xs, k in ((L.XSOLVE_AUTO, 100), (L.XSOLVE_CG, 10)):
        e = ap.Engine(L.PROB_TV2D, s=np.asfortranarray(img).reshape(-1, order="F"), lam=1.0, shape=(H, W), xsolve=xs)
        e.run(maxiters=3, domaxiters=1, record_history=0)
        t0 = time.perf_counter()
        s = e.run(maxiters=k, domaxiters=1, record_history=0)
        dt = time.perf_counter() - t0
        print(H, W, "xsolve", "auto" if xs == L.XSOLVE_AUTO else "cg", "it/s %.1f ms/it %.3f inner %.1f" % (s.steps / dt, 1e3 * dt / s.steps, float(e.fetch(L.F_CG_ITERS, 1)[0]) / s.steps), flush=True)
        e.close()
