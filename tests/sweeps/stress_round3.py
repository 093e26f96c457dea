"""Randomized parity sweep of the paths added in round 3 against the oracle: the one-block triangular solves (lasso and
bounded QP, every loop variant, early stops), the three-launch 2-D TV iteration (cooperative row stage, fused pass into
the forward transform; heights 64..2048, any even width, rho on both sides of the cooperative kernel's truncation
limit), the ABI's binding layer feeding the same engine.  `python tests/sweeps/stress_round3.py [seed]`"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["ADMM_TRSV_FORM"] = "one"
import admm_project_amd as ap  # noqa: E402
from oracle import solvers_ref as S  # noqa: E402  (test infrastructure: this script is a checker, not the product)

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
worst = {}


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


def check(tag, got, ref, keys, tol=1e-7):
    assert got.get("steps") == ref.get("steps"), (tag, got.get("steps"), ref.get("steps"))
    assert got.get("convtest_failed_at") == ref.get("convtest_failed_at"), tag
    for k in keys:
        if k in ref:
            e = rel(got[k], ref[k])
            worst[tag] = max(worst.get(tag, 0.0), e)
            assert e < tol, (tag, k, e)


VARIANTS = [dict(), dict(relax=1.5), dict(fast=1, fasttype="strong"), dict(fast=1, fasttype="weak"),
            dict(stopcond="both"), dict(convtest=1), dict(record_history=0)]
for t in range(28):
    cols = int(rng.integers(256, 2600))
    rows = cols + int(rng.integers(1, 1500))
    p = ap.synth.lasso_problem(t, rows, cols)
    o = dict(VARIANTS[t % len(VARIANTS)], objevals=1, rho=float(10 ** rng.uniform(-0.5, 0.7)))
    if rng.random() < 0.5:
        o.update(maxiters=int(rng.integers(2, 30)), domaxiters=1)
    got = ap.lasso(p["D"], p["s"], p["lam"], dict(o, xsolve="trsv"))
    if "engine_info" in got:  # (absent after a failed convergence test: admm.m:692-701 returns early, q4)
        assert got["engine_info"]["trsv_blocks"] == 1, got["engine_info"]
    ref = S.lasso(p["D"], p["s"], p["lam"], {k: v for k, v in o.items() if k != "record_history"})
    keys = ("xopt", "zopt", "uopt", "pnorm", "dnorm", "perr", "derr", "objevals", "Hnormsq")
    check("lasso/one-block", got, ref, keys + (() if o.get("record_history", 1) == 0 else ("xvals", "uvals")))
print("lasso through the one-block triangular solves ok", worst.get("lasso/one-block"), flush=True)

for t in range(8):
    n = int(rng.integers(256, 1800))
    q = ap.synth.qp_bounded_problem(t, n)
    o = dict(objevals=1, maxiters=int(rng.integers(5, 60)), domaxiters=1)
    got = ap.quadraticprogram(q["P"], q["q"], q["r"], q["lb"], q["ub"], dict(o, xsolve="trsv"))
    ref = S.quadraticprogram_bounded(q["P"], q["q"], q["r"], q["lb"], q["ub"], o)
    check("qp/one-block", got, ref, ("xopt", "zopt", "uopt", "pnorm", "dnorm", "objevals"))
print("bounded QP through the one-block triangular solves ok", worst.get("qp/one-block"), flush=True)

nspec = 0
for t in range(20):
    H = int(2 ** rng.integers(6, 12))
    W = 2 * int(rng.integers(100, 700)) if H <= 512 else 2 * int(rng.integers(100, 260))
    rho = float(10 ** rng.uniform(-0.8, 0.55))  # 0.16 .. 3.5: the cooperative row stage covers rho <= ~2.2
    img = rng.standard_normal((H, W)) + 2.0 * (rng.random((H, W)) > 0.7)
    o = dict(rho=rho, objevals=1)
    if rng.random() < 0.7:
        o.update(maxiters=int(rng.integers(2, 28)), domaxiters=1)
    else:
        o.update(stopcond="both", maxiters=60)
    if rng.random() < 0.3:
        o["record_history"] = 0
    got = ap.totalvariation2d(img, 0.6, dict(o))
    ref = S.totalvariation2d(img, 0.6, {k: v for k, v in o.items() if k != "record_history"})
    spectral = got["cg_iters_total"] == 0  # (CG where the height has no column transform)
    nspec += spectral
    check("tv2d" if spectral else "tv2d/cg", got, ref, ("xopt", "zopt", "uopt", "pnorm", "dnorm", "objevals"),
          tol=1e-8 if spectral else 1e-6)
print("2-D TV ok: %d spectral (three-launch) cases" % nspec, worst.get("tv2d"), worst.get("tv2d/cg"), flush=True)
print("worst relative errors:", worst)
