"""Prints what create()'s x-solve probe measures (admm_engine_info) on graded-singular-value matrices and on the
rank-deficient SVM input: probe errors of the explicit inverse and of the blocked triangular solves, the form chosen."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import admm_project_amd as ap

L = ap._lib
for kappa in (1e1, 1e2, 1e3, 1e4, 1e5):
    p = ap.synth.lad_problem_conditioned(0, 640, 300, kappa)
    e = ap.Engine(L.PROB_LAD, D=p["D"], s=p["s"], xsolve=L.XSOLVE_AUTO)
    i = e.info()
    print(f"kappa(D)={kappa:8.0e}  cond_est={i['cond_estimate']:.2e}  err_inverse={i['probe_err_inverse']:.2e}  "
          f"err_trsv={i['probe_err_trsv']:.2e}  used={i['xsolve_used']}")
    e.close()
p = ap.synth.rank_deficient_pixels(seed=1, m=6000, n=400, digit=3)
e = ap.Engine(L.PROB_LINEARSVM, D=p["D"], ell=p["ell"], Cval=p["C"])
print("svm rank-deficient 6000x400:", e.info(), "setup_s", e.setup_seconds)
e.close()
p = ap.synth.mnist_like_problem(seed=1, m=6000, n=400, digit=3)
e = ap.Engine(L.PROB_LINEARSVM, D=p["D"], ell=p["ell"], Cval=p["C"])
print("svm full rank 6000x400:", e.info(), "setup_s", e.setup_seconds)
e.close()
