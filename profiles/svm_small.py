"""Config 3 at 6000 x 400 only, 1000 iterations (for kernel traces)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import admm_project_amd as ap  # noqa: E402

L = ap._lib
m = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
q = ap.synth.mnist_like_problem(seed=1, m=m, n=400, digit=0)
svm = ap.Engine(L.PROB_LINEARSVM, D=q["D"], ell=q["ell"], Cval=q["C"], xsolve=L.XSOLVE_INVERSE)
kw = dict(maxiters=1000, domaxiters=1, record_history=0, nodualerror=1, stopcond="both", x0=q["x0"], z0=q["z0"], u0=q["u0"])
svm.run(**dict(kw, maxiters=20))
t0 = time.perf_counter()
s = svm.run(**kw)
print(m, "it/s %.0f" % (s.steps / (time.perf_counter() - t0)), flush=True)
