"""Setup only (upload + Gram + Cholesky + inverse) of the 100000 x 10000 lasso: the target of GEMM profiling."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import admm_project_amd as ap  # noqa: E402

m, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100000, 10000)
p = ap.synth.lasso_problem(seed=1, rows=m, cols=n)
e = ap.Engine(ap._lib.PROB_LASSO, D=p["D"], s=p["s"], lam=p["lam"], rho=1.0, xsolve=ap._lib.XSOLVE_INVERSE)
print("setup_seconds", e.setup_seconds)
