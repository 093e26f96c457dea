"""The triangular-solve x-update at n = 10^4 in its two forms (a 20000 x 10000 lasso: same factor size as config 2,
cheap setup): blocked substitution (2K + 1 launches) against the one-block form (two passes over inv(L)).
Usage: python profiles/trsv_forms.py [blocked|one|auto ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import admm_project_amd as ap  # noqa: E402

L = ap._lib
p = ap.synth.lasso_problem(seed=1, rows=20000, cols=10000)
for form in (sys.argv[1:] or ["blocked", "one", "auto"]):
    if form == "auto":
        os.environ.pop("ADMM_TRSV_FORM", None)
    else:
        os.environ["ADMM_TRSV_FORM"] = form
    e = ap.Engine(L.PROB_LASSO, D=p["D"], s=p["s"], lam=p["lam"], rho=1.0, xsolve=L.XSOLVE_TRSV)
    info = e.info()
    e.run(maxiters=20, domaxiters=1, record_history=0)
    e.set_profiling([L.K_XSOLVE])
    t0 = time.perf_counter()
    s = e.run(maxiters=400, domaxiters=1, record_history=0)
    dt = time.perf_counter() - t0
    ms, launches = e.kernel_time(L.K_XSOLVE)
    e.set_profiling(False)
    t0 = time.perf_counter()
    s2 = e.run(maxiters=400, domaxiters=1, record_history=0)
    dt2 = time.perf_counter() - t0
    print("form=%s blocks=%d err_one=%.2e err_blocked=%.2e  us/it %.1f (timed x-solve %.1f us over %d groups); "
          "untimed loop us/it %.1f  frac %.3f" % (form, info["trsv_blocks"], info["probe_err_trsv_one"], info["probe_err_trsv"],
                                  1e6 * dt / s.steps, 1e3 * ms / max(1, launches), launches, 1e6 * dt2 / s2.steps,
                                  8.0 * 10000 * 10001 / (dt2 / s2.steps) / 8e12), flush=True)
    e.close()
