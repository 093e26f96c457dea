"""2-D TV on an image whose width / rho rule out the Toeplitz row stage and the row transform: the exact tridiagonal row
stage (dct.hip: tv2d_rows_thomas_kernel) against the matrix-free CG x-update it replaces (ADMM_HIP_TV2D_NO_THOMAS=1).
    python profiles/tv2d_rows_thomas.py [H] [W] [rho] [iterations]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import admm_project_amd as ap  # noqa: E402

H = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
rho = float(sys.argv[3]) if len(sys.argv) > 3 else 100.0
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 30
rng = np.random.default_rng(1)
img = rng.standard_normal((H, W)) + 2.0 * (rng.random((H, W)) > 0.7)
L = ap._lib
e = ap.Engine(L.PROB_TV2D, s=np.asfortranarray(img).reshape(-1, order="F"), lam=1.0, shape=(H, W))
e.run(maxiters=2, domaxiters=1, record_history=0, rho=rho)
t0 = time.perf_counter()
s = e.run(maxiters=iters, domaxiters=1, record_history=0, rho=rho)
dt = time.perf_counter() - t0
inner = float(e.fetch(L.F_CG_ITERS, 1)[0])
print(f"{H}x{W} rho {rho:g}: {1e3 * dt / s.steps:.3f} ms per iteration, {inner / s.steps:.1f} CG steps per x-update "
      f"({'matrix-free' if inner else 'spectral: column transform + row stage, no CG'})", flush=True)
