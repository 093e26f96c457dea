#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X ADMM engine (driver contract: see DESIGN.md).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): lasso.m on synthetic dense D 100000 x 10000 fp64, rho = 1,
lassotest.m:109-122 recipe (seed 1), fixed work via the reference's own switch domaxiters=1
(admm.m:59, 711).  A "step" is one ADMM iteration (admm.m:496-743) with D, the cached factor
and all iterates resident in HBM.  With N > 1 the SAME problem is row-sharded over the ranks
(slicemaker(0, N, m), errorcheck.m:249-259) -> strong scaling.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
_DEV = "cuda"  # device of the torch tensors used for rendezvous collectives ("cpu" in --one-gpu rehearsals)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rows", type=int, default=100000)
    ap.add_argument("--cols", type=int, default=10000)
    ap.add_argument("--xsolve", default="inverse", choices=["trsv", "inverse"])
    ap.add_argument("--transport", default="rccl", choices=["rccl", "shm"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the objevals=1 and A-streaming side measurements")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--one-gpu", action="store_true",
                    help="rehearsal on a single-GPU box: every rank uses HIP device 0, torch.distributed runs "
                         "over gloo and the engine over the shm transport (RCCL refuses two ranks on one device)")
    return ap.parse_args(argv)


def dist_setup(one_gpu=False):
    """One process per GPU (torch.distributed, backend nccl == RCCL).  Returns (rank, world, local, dist|None)."""
    global _DEV
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1:
        return 0, 1, 0, None
    import torch
    import torch.distributed as dist

    if one_gpu:
        _DEV = "cpu"
        dist.init_process_group(backend="gloo")
        return rank, world, 0, dist
    torch.cuda.set_device(local)
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    return rank, world, local, dist


def max_over_ranks(dist, value, device=None):
    if dist is None:
        return float(value)
    import torch

    t = torch.tensor([float(value)], dtype=torch.float64, device=device or _DEV)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sync_all(dist):
    if dist is not None:
        import torch

        dist.barrier()
        if _DEV == "cuda":
            torch.cuda.synchronize()


def timed_run(eng, dist, steps, **kw):
    """barrier + sync, EXACTLY `steps` iterations, sync + barrier; MAX over ranks."""
    sync_all(dist)
    t0 = time.perf_counter()
    s = eng.run(maxiters=steps, domaxiters=1, record_history=0, **kw)  # returns after its stream drained
    sync_all(dist)
    dt = max_over_ranks(dist, time.perf_counter() - t0)
    assert s.steps == steps, (s.steps, steps)
    return dt, s


def make_problem(ap_mod, dist, m, n, lo, hi):
    """lassotest.m:109-122 at full size; each rank keeps only its rows [lo, hi) of D and s."""
    if dist is None:
        return ap_mod.synth.lasso_problem(seed=1, rows=m, cols=n)
    import torch

    p = ap_mod.synth.lasso_problem(seed=1, rows=m, cols=n, row_range=(lo, hi))
    g = torch.from_numpy(p["D"].T @ p["s"]).to(_DEV)  # lambda = 0.1*||D's||_inf needs the global D's
    dist.all_reduce(g)
    p["lam"] = 0.1 * float(g.abs().max().item())
    return p


def cpu_baseline(p, factor, seconds, rho):
    """The oracle's lasso loop (same algorithm and operation order as the reference: two
    triangular solves with the cached factor per iteration) on the host cores.  The factor is the
    one built on the GPU, handed over through args.L exactly as lasso.m:183 hands it to getproxops."""
    from oracle import admm as ref_admm
    from oracle import getproxops as ref_getproxops

    try:
        from threadpoolctl import threadpool_info
        threads = max([d.get("num_threads", 1) for d in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count() or 1
    D, s, lam = p["D"], p["s"], p["lam"]
    m, n = D.shape
    args = dict(D=D, Dts=D.T @ s, L=factor, U=factor.T, m=m, n=n, parallel=0, rho=rho)
    args["lambda"] = lam
    minx, minz, _ = ref_getproxops("LASSO", args)
    base = dict(A=1, At=1, m=n, nA=n, nB=n, B=-1, c=0, domaxiters=1, rho=rho)
    t0 = time.perf_counter()
    ref_admm(minx, minz, dict(base, maxiters=2))
    per = (time.perf_counter() - t0) / 2
    k = int(max(3, min(500, seconds / max(per, 1e-6))))
    r = ref_admm(minx, minz, dict(base, maxiters=k))
    out = dict(value=k / r["runtime"], unit="iterations/s", cores=int(threads), kind="port",
               sample=f"{k} iterations of the same {m}x{n} lasso loop (oracle restatement of admm.m:496-743 + "
                      f"getProxOps.m:1192-1206, SciPy/LAPACK triangular solves, factor taken from the GPU setup); "
                      f"loop only, as results.runtime",
               note="cores = size of the BLAS thread pool the loop was allowed to use; LAPACK's triangular solve "
                    "(BLAS-2) barely threads, compare single_thread")
    # MATLAB applies the factor it stored SPARSE (lasso.m:175-176) with single-threaded triangular solves:
    # the same loop pinned to one BLAS thread is the closer stand-in for the reference's CPU path
    try:
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=1):
            k1 = int(max(2, min(60, 0.4 * seconds / max(per, 1e-6))))
            r1 = ref_admm(minx, minz, dict(base, maxiters=k1))
        out["single_thread"] = dict(value=k1 / r1["runtime"], unit="iterations/s", cores=1,
                                    sample=f"{k1} iterations, BLAS limited to one thread")
    except Exception as exc:  # threadpoolctl missing: report the multi-threaded number only
        out["single_thread"] = dict(error=repr(exc))
    return out


def other_configs(ap, L, device, steps):
    """BASELINE.json configs 3 and 5 on one GPU (side lines; the headline stays config 2)."""
    res = {}
    # config 5: total variation, 1-D signal of length 4096^2 (the reference's TV is 1-D: totalvariation.m:127)
    n = 4096 * 4096
    p = ap.synth.tv_problem(seed=1, n=n)
    tv = ap.Engine(L.PROB_TOTALVARIATION, s=p["s"], lam=1.0, nvec=n, device=device)
    k = max(20, steps // 2)
    tv.run(maxiters=5, domaxiters=1, record_history=0)
    t0 = time.perf_counter()
    s = tv.run(maxiters=k, domaxiters=1, record_history=0)
    dt = time.perf_counter() - t0
    passes = 7  # fused iteration kernel: reads y, z, u, s; writes z, u, y (x only with histories)  
    res["totalvariation_16777216"] = {"iters_per_s": s.steps / dt, "ms_per_step": dt / s.steps * 1e3,
                                      "algorithmic_GB_per_iter": passes * 8.0 * n / 1e9,
                                      "achieved_GBs": passes * 8.0 * n * s.steps / dt / 1e9,
                                      "frac": passes * 8.0 * n * s.steps / dt / 1e9 / HBM_PEAK_GBS}
    tv.close()
    # config 5 as literally written: anisotropic TV of a 4096 x 4096 IMAGE, no cached factor -- an engine-side
    # extension (the reference's solver is 1-D); own oracle, see tests/test_gpu_tv2d.py.  Two x-updates:
    # the direct spectral solve (2-D DCT, dct.hip; the default for power-of-two sides) and warm-started CG.
    hw = 4096
    rng = np.random.default_rng(1)
    img = np.zeros((hw, hw))
    img[hw // 5:hw // 2, hw // 6:hw // 2] = 2.0
    img[hw // 3:4 * hw // 5, hw // 3:5 * hw // 6] += 1.0
    img += rng.standard_normal((hw, hw))
    npix = hw * hw
    flat = np.asfortranarray(img).reshape(-1, order="F")
    for tag, xs, k2 in (("", L.XSOLVE_AUTO, max(50, steps)), ("_cg", L.XSOLVE_CG, max(5, steps // 20))):
        tv2 = ap.Engine(L.PROB_TV2D, s=flat, lam=1.0, shape=(hw, hw), xsolve=xs, device=device)
        tv2.run(maxiters=2, domaxiters=1, record_history=0)
        t0 = time.perf_counter()
        s2 = tv2.run(maxiters=k2, domaxiters=1, record_history=0)
        dt2 = time.perf_counter() - t0
        inner = float(tv2.fetch(L.F_CG_ITERS, 1)[0]) / s2.steps
        if xs == L.XSOLVE_CG:
            # doubles per ADMM iteration: CG start 12N + fused z/u/dual/rhs pass 11N + 10N per inner iteration
            # (fused direction + stencil: reads r, p, writes p, q; update: reads x, p, r, q, writes x, r)
            gb = (23.0 + 10.0 * inner) * 8.0 * npix / 1e9
            extra = {"x_update": "cg", "cg_inner_iters_per_step": inner, "cg_tol": 1e-11}
        else:
            # fused z/u/dual/rhs pass 6N read + 5N written; spectral solve: five in-place passes (column DCT,
            # transpose, row DCT + scale + inverse, transpose, column inverse) = 10N
            gb = 21.0 * 8.0 * npix / 1e9
            extra = {"x_update": "dct"}
        res["totalvariation2d_4096x4096" + tag] = dict(
            {"iters_per_s": s2.steps / dt2, "ms_per_step": dt2 / s2.steps * 1e3, "algorithmic_GB_per_iter": gb,
             "achieved_GBs": gb * s2.steps / dt2, "frac": gb * s2.steps / dt2 / HBM_PEAK_GBS}, **extra)
        tv2.close()
    # config 3: linear SVM, hinge, MNIST-shaped synthetic pixels (image files are absent from the reference)
    for m in (6000, 60000):
        q = ap.synth.mnist_like_problem(seed=1, m=m, n=400, digit=0)
        svm = ap.Engine(L.PROB_LINEARSVM, D=q["D"], ell=q["ell"], Cval=q["C"], xsolve=L.XSOLVE_INVERSE, device=device)
        kw = dict(maxiters=1000, domaxiters=1, record_history=0, nodualerror=1, stopcond="both",
                  x0=q["x0"], z0=q["z0"], u0=q["u0"])  # unwrappedadmm.m:87-92
        svm.run(**dict(kw, maxiters=20))
        t0 = time.perf_counter()
        s = svm.run(**kw)
        dt = time.perf_counter() - t0
        res[f"linearsvm_{m}x400"] = {"iters_per_s": s.steps / dt, "ms_per_step": dt / s.steps * 1e3,
                                     "algorithmic_MB_per_iter": 16.0 * m * 400 / 1e6,
                                     "note": "L2/MALL-resident: latency-bound, HBM fraction not meaningful"}
        svm.close()
    return res


def _leg(name):
    """progress marker on stderr (one line per bench leg: locates a failure under a profiler)"""
    print(f"bench.py: leg {name}", file=sys.stderr, flush=True)


def side_engines(ap, L, a, dist, p, xs, local, comm, lo, hi, n, rho, world, out):
    """The A-streaming legs (lad.m, matrix-free lasso) on the same D, s, and configs 3 / 5 at N = 1."""
    _leg("a_streaming (lad)")
    # A-streaming iteration on the same D, s: lad.m (x = R'\(R\(D'(s+z-u))), z = soft(Dx+u-s)) --
    # exactly one D*x and one D'*[3 rhs] pass per iteration = the "A'(Ax-b)" unit, 16mn bytes,
    # row-sharded with ONE all-reduce per iteration when N > 1 (unwrappedadmm.m:96-141).
    lad = ap.Engine(L.PROB_LAD, D=p["D"], s=p["s"], xsolve=xs, device=local, comm=comm)
    k2 = max(5, a.steps // 4)
    timed_run(lad, dist, 2)
    lad.set_profiling([L.K_GEMV_N, L.K_GEMV_T])
    dt2, _ = timed_run(lad, dist, k2)
    lad.set_profiling(False)
    gn_ms, gn_cnt = lad.kernel_time(L.K_GEMV_N)
    gt_ms, gt_cnt = lad.kernel_time(L.K_GEMV_T)
    pair_ms = gn_ms / max(1, gn_cnt) + gt_ms / max(1, gt_cnt)
    rows_local = hi - lo
    gbs = 16.0 * rows_local * n / (pair_ms * 1e-3) / 1e9 if pair_ms > 0 else 0.0
    out["a_streaming"] = {"workload": "lad.m on the same D,s: D*x + D'*[s+z-u, dz, u] per iteration (16mn B), "
                                      "transpose reduction over the row shards",
                          "iters_per_s": k2 / dt2, "ms_per_step": dt2 / k2 * 1e3,
                          "AtAx_unit_ms": pair_ms, "AtAx_GBs_per_gpu": gbs, "AtAx_frac": gbs / HBM_PEAK_GBS,
                          "gemv_n_avg_ms": gn_ms / max(1, gn_cnt), "gemv_t_avg_ms": gt_ms / max(1, gt_cnt),
                          "setup_seconds": max_over_ranks(dist, lad.setup_seconds)}
    lad.close()

    # objevals=1 with the objective taken from the cached Gram matrix (opt-in args.objgram): 4n^2 B instead of 8mn B
    _leg("objevals1_gram")
    lg = ap.Engine(L.PROB_LASSO, D=p["D"], s=p["s"], lam=p["lam"], rho=rho, xsolve=xs, device=local, comm=comm,
                   obj_gram=1)
    kg = max(20, a.steps)
    timed_run(lg, dist, 5, rho=rho, objevals=1)
    dtg, _ = timed_run(lg, dist, kg, rho=rho, objevals=1)
    out["objevals1_gram"] = {"iters_per_s": kg / dtg, "ms_per_step": dtg / kg * 1e3,
                             "note": "objevals=1 with 1/2*||D*x - s||^2 = 1/2*x'Gx - x'D's + 1/2*s's from the cached "
                                     "G = D'D (one more pass over an n x n lower triangle per iteration); opt-in: "
                                     "absolute rounding error ~1e-16*||s||^2"}
    lg.close()

    # the same loop with the factor applied literally (two triangular solves, the reference's form): a chain of
    # 2*n/64 dependent block steps -- latency-bound, the reason xsolve=inverse exists
    if world == 1:
        _leg("xsolve_trsv")
        lt = ap.Engine(L.PROB_LASSO, D=p["D"], s=p["s"], lam=p["lam"], rho=rho, xsolve=L.XSOLVE_TRSV, device=local)
        kt = max(10, a.steps // 4)
        timed_run(lt, dist, 3, rho=rho)
        dtt, _ = timed_run(lt, dist, kt, rho=rho)
        out["xsolve_trsv"] = {"iters_per_s": kt / dtt, "ms_per_step": dtt / kt * 1e3,
                              "algorithmic_GB_per_iter": 8.0 * n * (n + 1) / 1e9,
                              "frac": 8.0 * n * (n + 1) * kt / dtt / 1e9 / HBM_PEAK_GBS,
                              "note": "x = L'\\(L\\y) as getProxOps.m:1200 writes it; same iterates as the headline"}
        lt.close()

    # matrix-free lasso (xsolve = cg): same iterates as the cached-factor loop (inner tolerance
    # 1e-10), every inner iteration one A'(A p) unit, nothing n x n stored
    _leg("matrix_free (cg)")
    mf = ap.Engine(L.PROB_LASSO, D=p["D"], s=p["s"], lam=p["lam"], rho=rho, xsolve=L.XSOLVE_CG, device=local,
                   comm=comm, cg_tol=1e-10)
    k3 = max(3, a.steps // 40)
    timed_run(mf, dist, 1, rho=rho)
    mf.set_profiling([L.K_GEMV_N, L.K_GEMV_T])
    dt3, _ = timed_run(mf, dist, k3, rho=rho)
    mf.set_profiling(False)
    inner = float(mf.fetch(L.F_CG_ITERS, 1)[0])
    gn_ms, _ = mf.kernel_time(L.K_GEMV_N)
    gt_ms, _ = mf.kernel_time(L.K_GEMV_T)
    # operator applications that did work: one per inner iteration + the initial residual of every solve
    # (launches enqueued past convergence are no-ops and must not be counted as units)
    pair_ms = (gn_ms + gt_ms) / max(1.0, inner + k3)
    gbs = 16.0 * (hi - lo) * n / (pair_ms * 1e-3) / 1e9 if pair_ms > 0 else 0.0
    out["matrix_free"] = {"workload": "lasso, x-update by warm-started CG on (D'D + rho I), tol 1e-10",
                          "iters_per_s": k3 / dt3, "ms_per_step": dt3 / k3 * 1e3,
                          "inner_iters_per_step": inner / k3, "AtAx_unit_ms": pair_ms,
                          "AtAx_GBs_per_gpu": gbs, "AtAx_frac": gbs / HBM_PEAK_GBS,
                          "setup_seconds": max_over_ranks(dist, mf.setup_seconds)}
    mf.close()

    if world > 1:
        # config 4: consensus lasso (getProxOps.m:383-442, 1217-1343), one row slice per GPU: own x_k, u_k and
        # factor chol(D_k'D_k + rho I) per rank, ONE all-reduce of [sum x_k; sum u_k] (2n doubles) per iteration
        cons = ap.Engine(L.PROB_LASSO_CONSENSUS, D=p["D"], s=p["s"], lam=p["lam"], rho=rho, xsolve=xs, device=local,
                         comm=comm, slices=[hi - lo])
        k4 = max(5, a.steps // 4)
        timed_run(cons, dist, 2, rho=rho, stopcond="both")
        dt4, _ = timed_run(cons, dist, k4, rho=rho, stopcond="both")
        out["consensus_lasso"] = {"workload": f"consensus lasso, {world} row slices of {hi - lo} x {n}, one per GPU",
                                  "iters_per_s": k4 / dt4, "ms_per_step": dt4 / k4 * 1e3,
                                  "allreduce_doubles_per_iter": 2 * n + 2,
                                  "setup_seconds": max_over_ranks(dist, cons.setup_seconds)}
        cons.close()
    if world == 1:
        _leg("other_configs (tv, tv2d, svm)")
        out["other_configs"] = other_configs(ap, L, local, a.steps)


def main():
    a = parse()
    rank, world, local, dist = dist_setup(a.one_gpu)
    if a.one_gpu:
        a.transport = "shm"
    if world != a.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run",
                  file=sys.stderr)
        sys.exit(2)
    import admm_project_amd as ap
    from admm_project_amd import parallel

    L = ap._lib
    L.require_device()
    m, n = a.rows, a.cols
    rho = 1.0
    comm = None
    transport = None
    lo, hi = 0, m
    if dist is not None:
        transport = a.transport
        try:
            comm = parallel.init_from_torch(dist, device=local, transport=transport)
        except ap.AdmmError as exc:  # RCCL refused the topology: host-staged transport still measures the GPUs
            if rank == 0:
                print(f"bench.py: RCCL communicator failed ({exc}); falling back to the shm transport", file=sys.stderr)
            transport = "shm"
            comm = parallel.init_from_torch(dist, device=local, transport=transport)
        lo, hi = parallel.my_rows(m, comm)

    t0 = time.perf_counter()
    p = make_problem(ap, dist, m, n, lo, hi)
    t_gen = time.perf_counter() - t0

    xs = {"trsv": L.XSOLVE_TRSV, "inverse": L.XSOLVE_INVERSE}[a.xsolve]
    eng = ap.Engine(L.PROB_LASSO, D=p["D"], s=p["s"], lam=p["lam"], rho=rho, xsolve=xs, device=local, comm=comm)
    setup_s = max_over_ranks(dist, eng.setup_seconds)

    # ---- headline: objevals = 0 ----------------------------------------------------------
    _leg("headline")
    timed_run(eng, dist, max(1, a.warmup), rho=rho)
    # only the dominant kernel class, every 4th iteration: the event pairs sit inside the timed region
    eng.set_profiling([L.K_XSOLVE], stride=4)
    dt, _ = timed_run(eng, dist, a.steps, rho=rho)
    eng.set_profiling(False)
    xs_ms, xs_cnt = eng.kernel_time(L.K_XSOLVE)
    value = a.steps / dt
    # the same K steps without the HIP event records of the roofline leg in the stream
    dt_plain, _ = timed_run(eng, dist, a.steps, rho=rho)
    if a.xsolve == "inverse":
        alg_bytes = 8.0 * n * (n + 1) / 2  # lower triangle of the symmetric inverse, read once
        kname = "symv_lower_kernel (+ symv_reduce): x = inv(D'D+rho I) * y from the lower triangle only"
    else:
        alg_bytes = 8.0 * n * (n + 1)  # two triangular solves
        kname = "trsv_fwd/bwd_step kernels (x = L'\\(L\\y))"
    # HBM bytes per x-solve from the committed PMC passes (profiles/r1_traffic.json; bench.py cannot
    # run rocprofv3 on itself): valid for the default problem size and the symmetric-half kernel only
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "r1_traffic.json")
    if a.xsolve == "inverse" and n == 10000 and os.path.exists(tfile):
        with open(tfile) as fh:
            tj = json.load(fh)
        try:
            traffic = sum(tj[k]["fetch_bytes_per_launch"] + tj[k]["write_bytes_per_launch"]
                          for k in ("admm::symv_lower_kernel<true>", "admm::symv_reduce_kernel"))
        except KeyError:
            traffic = None
    xs_avg_ms = xs_ms / max(1, xs_cnt)
    achieved = alg_bytes / (xs_avg_ms * 1e-3) / 1e9 if xs_cnt else 0.0
    out = {
        "metric": "ADMM iterations/sec (lasso 100k x 10k, fp64)", "value": value, "unit": "iterations/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"lasso.m cached-factor loop, D {m}x{n} fp64, rho=1, lassotest.m recipe seed=1, "
                               f"domaxiters=1, objevals=0, xsolve={a.xsolve}",
                   "rows": m, "cols": n, "rho": rho, "parallelism": f"rows{world}", "collective": transport},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": kname,
                     "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": xs_avg_ms, "launches": xs_cnt},
        "setup_seconds": setup_s, "datagen_seconds": t_gen,
        "iters_per_s_without_event_timing": a.steps / dt_plain,
        "achieved_hbm_GBs_whole_iteration": (alg_bytes + 8.0 * 21 * n) * a.steps / dt / 1e9,
    }

    # ---- side measurements (same resident data) --------------------------------------------
    if not a.no_extras:
        _leg("objevals1")
        k1 = max(5, a.steps // 4)
        timed_run(eng, dist, 2, rho=rho, objevals=1)
        eng.set_profiling([L.K_GEMV_N])
        dt1, _ = timed_run(eng, dist, k1, rho=rho, objevals=1)
        eng.set_profiling(False)
        gn_ms, gn_cnt = eng.kernel_time(L.K_GEMV_N)
        rows_local = hi - lo
        gbs = 8.0 * rows_local * n / (gn_ms / max(1, gn_cnt) * 1e-3) / 1e9 if gn_cnt else 0.0
        out["objevals1"] = {"iters_per_s": k1 / dt1, "ms_per_step": dt1 / k1 * 1e3,
                            "gemv_n_GBs_per_gpu": gbs, "gemv_n_frac": gbs / HBM_PEAK_GBS,
                            "gemv_n_avg_ms": gn_ms / max(1, gn_cnt),
                            "note": "lassotest.m:131 sets objevals=1: one extra D*x pass (8mn B) per iteration, "
                                    "row-sharded when N > 1; timing includes the residual-norm kernel"}

    factor = None
    if not a.no_cpu_baseline and world == 1:
        factor = eng.fetch(L.F_FACTOR, n * n, (n, n))
    eng.close()

    if not a.no_extras:
        try:
            side_engines(ap, L, a, dist, p, xs, local, comm, lo, hi, n, rho, world, out)
        except Exception as exc:  # the side measurements must never cost the headline line
            out["extras_error"] = repr(exc)

    if not a.no_cpu_baseline and world == 1 and rank == 0:
        out["cpu_baseline"] = cpu_baseline(p, factor, a.cpu_seconds, rho)
    if rank == 0:
        print(json.dumps(out))
    if comm is not None:
        comm.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
