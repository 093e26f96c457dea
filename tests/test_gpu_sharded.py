"""GPU: the row-sharded engines, two ranks (processes) sharing the ONE GPU of the test box.

RCCL refuses two ranks on one device, so the ranks talk through the library's host-staged
shared-memory transport (comm.hip, ADMM_COMM_SHM) -- the engine code path (packing, one
all-reduce per iteration, reduced slots in the finalize kernel, global lengths) is the same as
with RCCL.  Each rank must reproduce the UNSHARDED oracle: x and every scalar history to 1e-9,
its rows of z and u likewise.  A single-rank RCCL communicator is also exercised (plumbing)."""
import os
import sys

import numpy as np
import pytest

from oracle import solvers_ref as S

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rank_main(rank, world, uid, case, q):
    sys.path.insert(0, ROOT)
    import admm_project_amd as ap
    from admm_project_amd import parallel

    try:
        comm = parallel.Comm(uid, rank, world, device=0, transport=os.environ.get("ADMM_TEST_TRANSPORT", "shm"))
        out = {}
        tot = comm.allreduce_sum(np.array([1.0 + rank, 10.0]))
        out["allreduce"] = tot
        if case == "lad":
            p = ap.synth.lad_problem(0, 515, 48)
            lo, hi = parallel.my_rows(515, comm)
            for xs in ("trsv", "inverse"):
                r = ap.lad(p["D"][lo:hi], p["s"][lo:hi], dict(objevals=1, comm=comm, xsolve=xs))
                out["lad_" + xs] = {k: r[k] for k in ("steps", "xvals", "pnorm", "dnorm", "perr", "derr", "objevals",
                                                       "zopt", "uopt")}
            r = ap.huberfit(p["D"][lo:hi], p["s"][lo:hi], dict(objevals=1, comm=comm, fast=1, fasttype="weak",
                                                               maxiters=30, stopcond="both"))
            out["huber_weak"] = {k: r[k] for k in ("steps", "xvals", "dvals", "restarted", "Hnormsq", "objevals")}
            out["rows"] = (lo, hi)
        elif case == "svm":
            p = ap.synth.svm_problem(0, 100, 101)
            lo, hi = parallel.my_rows(201, comm)
            o = dict(objevals=1, comm=comm, x0=p["x0"], z0=p["z0"][lo:hi], u0=p["u0"][lo:hi])
            r = ap.linearsvm(p["D"][lo:hi], p["ell"][lo:hi], p["C"], o)
            out["svm"] = {k: r[k] for k in ("steps", "xvals", "pnorm", "perr", "Hnormsq", "objevals", "zopt")}
            out["rows"] = (lo, hi)
        elif case == "consensus":
            p = ap.synth.lasso_problem(1, 256, 64)
            lo, hi = parallel.my_rows(256, comm)  # 128 rows per rank, 2 local slices of 64 -> 4 slices in total
            r = ap.lasso(p["D"][lo:hi], p["s"][lo:hi], p["lam"],
                         dict(objevals=1, parallel="both", comm=comm, workers=2, xsolve="inverse"))
            out["consensus"] = {k: r[k] for k in ("steps", "xvals", "zvals", "uvals", "pnorm", "dnorm", "perr",
                                                  "derr", "objevals", "Hnormsq", "zconsensus")}
        elif case == "lasso":
            p = ap.synth.lasso_problem(2, 301, 64)
            lo, hi = parallel.my_rows(301, comm)
            r = ap.lasso(p["D"][lo:hi], p["s"][lo:hi], p["lam"], dict(objevals=1, comm=comm, xsolve="inverse"))
            out["lasso"] = {k: r[k] for k in ("steps", "xvals", "zvals", "uvals", "pnorm", "dnorm", "objevals")}
        elif case == "big":  # the lower-triangle x-solve itself is split over the ranks
            # (forced: with the host-staged test transport the measured all-reduce latency would veto the split)
            os.environ["ADMM_HIP_XSPLIT"] = "1"
            p = ap.synth.lasso_problem(3, 3600, 1600)
            lo, hi = parallel.my_rows(3600, comm)
            o = dict(objevals=1, comm=comm, xsolve="inverse", maxiters=6, domaxiters=1)
            r = ap.lasso(p["D"][lo:hi], p["s"][lo:hi], p["lam"], o)
            out["lasso"] = {k: r[k] for k in ("steps", "xvals", "zvals", "uvals", "pnorm", "dnorm", "objevals")}
            r = ap.lad(p["D"][lo:hi], p["s"][lo:hi], dict(o, maxiters=4))
            out["lad"] = {k: r[k] for k in ("steps", "xvals", "pnorm", "dnorm", "objevals")}
        q.put((rank, out))
        comm.close()
    except Exception as exc:  # surface the failure in the parent instead of a queue timeout
        import traceback

        q.put((rank, {"error": f"{exc}\n{traceback.format_exc()}"}))


@pytest.fixture(autouse=True, params=["shm", "p2p"])
def transport(request, monkeypatch):
    """every two-process case over the host-staged transport AND over the engine's one-shot peer-to-peer all-reduce
    (ADMM_COMM_P2P: the two processes map each other's device buffers through HIP IPC handles; one kernel per rank
    and collective on the engine's stream, no host synchronisation -- the stream-ordered path)"""
    monkeypatch.setenv("ADMM_TEST_TRANSPORT", request.param)
    return request.param


def _run_two_ranks(case):
    import multiprocessing as mp

    import admm_project_amd as ap
    from admm_project_amd import parallel

    ap._lib.require_device()
    uid = parallel.unique_id()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, uid, case, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in procs:
        rank, out = q.get(timeout=280)
        res[rank] = out
    for p in procs:
        p.join(timeout=60)
    for rank in (0, 1):
        assert "error" not in res[rank], res[rank].get("error")
        np.testing.assert_array_equal(res[rank]["allreduce"], [3.0, 20.0])
    return res


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


def test_sharded_lad_and_huber_match_unsharded_oracle(gpu):
    res = _run_two_ranks("lad")
    p = gpu.synth.lad_problem(0, 515, 48)
    ref = S.lad(p["D"], p["s"], dict(objevals=1))
    refh = S.huberfit(p["D"], p["s"], dict(objevals=1, fast=1, fasttype="weak", maxiters=30, stopcond="both"))
    for rank in (0, 1):
        lo, hi = res[rank]["rows"]
        assert (lo, hi) == ((0, 258), (258, 515))[rank]
        for xs in ("trsv", "inverse"):
            g = res[rank]["lad_" + xs]
            assert g["steps"] == ref["steps"]
            for k in ("xvals", "pnorm", "dnorm", "perr", "derr", "objevals"):
                assert _rel(g[k], ref[k]) < 1e-9, (k, xs)
            assert _rel(g["zopt"], ref["zopt"][lo:hi]) < 1e-9 and _rel(g["uopt"], ref["uopt"][lo:hi]) < 1e-9
        g = res[rank]["huber_weak"]
        assert g["steps"] == refh["steps"]
        np.testing.assert_array_equal(g["restarted"], refh["restarted"])
        for k in ("xvals", "dvals", "Hnormsq", "objevals"):
            assert _rel(g[k], refh[k]) < 1e-8, k
    # both ranks hold bitwise identical replicated x
    np.testing.assert_array_equal(res[0]["lad_trsv"]["xvals"], res[1]["lad_trsv"]["xvals"])


def test_sharded_svm_matches_unsharded_oracle(gpu):
    res = _run_two_ranks("svm")
    p = gpu.synth.svm_problem(0, 100, 101)
    ref = S.linearsvm(p["D"], p["ell"], p["C"], dict(objevals=1, x0=p["x0"], z0=p["z0"], u0=p["u0"]))
    for rank in (0, 1):
        lo, hi = res[rank]["rows"]
        g = res[rank]["svm"]
        assert g["steps"] == ref["steps"]
        for k in ("xvals", "pnorm", "perr", "Hnormsq", "objevals"):
            assert _rel(g[k], ref[k]) < 1e-7, k
        assert _rel(g["zopt"], ref["zopt"][lo:hi]) < 1e-7


def test_sharded_lasso_matches_unsharded_oracle(gpu):
    res = _run_two_ranks("lasso")
    p = gpu.synth.lasso_problem(2, 301, 64)
    ref = S.lasso(p["D"], p["s"], p["lam"], dict(objevals=1))
    for rank in (0, 1):
        g = res[rank]["lasso"]
        assert g["steps"] == ref["steps"]
        for k in ("xvals", "zvals", "uvals", "pnorm", "dnorm", "objevals"):
            assert _rel(g[k], ref[k]) < 1e-9, k


def test_sharded_symmetric_xsolve_matches_unsharded_oracle(gpu):
    """Each rank streams half of the lower-triangle tiles of the inverse, one all-reduce of n doubles assembles x
    (lasso: the only collective of the loop; LAD: in addition to the 3n+16 one).  In production the split is taken
    only when it removes more streaming time than the communicator's measured all-reduce latency costs."""
    res = _run_two_ranks("big")
    p = gpu.synth.lasso_problem(3, 3600, 1600)
    o = dict(objevals=1, maxiters=6, domaxiters=1)
    ref = S.lasso(p["D"], p["s"], p["lam"], o)
    ref_lad = S.lad(p["D"], p["s"], dict(o, maxiters=4))
    for rank in (0, 1):
        g = res[rank]["lasso"]
        assert g["steps"] == ref["steps"]
        for k in ("xvals", "zvals", "uvals", "pnorm", "dnorm", "objevals"):
            assert _rel(g[k], ref[k]) < 1e-8, k
        g = res[rank]["lad"]
        assert g["steps"] == ref_lad["steps"]
        for k in ("xvals", "pnorm", "dnorm", "objevals"):
            assert _rel(g[k], ref_lad[k]) < 1e-7, k
    # both ranks hold bitwise the same replicated x
    np.testing.assert_array_equal(res[0]["lasso"]["xvals"], res[1]["lasso"]["xvals"])


def test_sharded_consensus_lasso_matches_four_slice_oracle(gpu):
    """Config 4 (e2): slices spread over two ranks, one all-reduce of [sum x_k; sum u_k] per iteration."""
    res = _run_two_ranks("consensus")
    p = gpu.synth.lasso_problem(1, 256, 64)
    ref = S.lasso(p["D"], p["s"], p["lam"], dict(objevals=1, parallel="both"), workers=4)
    for rank in (0, 1):
        g = res[rank]["consensus"]
        assert g["steps"] == ref["steps"]
        for k in ("xvals", "zvals", "uvals", "pnorm", "dnorm", "perr", "derr", "objevals", "Hnormsq"):
            assert np.max(np.abs(np.asarray(g[k]) - ref[k])) <= 1e-8 * max(1e-30, np.max(np.abs(ref[k]))), k
        assert _rel(g["zconsensus"], ref["_consensus"]["_state"]["z"]) < 1e-9


def test_single_rank_rccl_communicator(gpu):
    """RCCL plumbing (dlopen, unique id, ncclCommInitRank, all-reduce on a stream) with one rank."""
    from admm_project_amd import parallel

    comm = parallel.Comm(parallel.unique_id(), 0, 1, device=0, transport="rccl")
    np.testing.assert_array_equal(comm.allreduce_sum(np.arange(5.0)), np.arange(5.0))
    p = gpu.synth.lad_problem(1, 300, 40)
    a = gpu.lad(p["D"], p["s"], dict(comm=comm))
    b = gpu.lad(p["D"], p["s"], {})
    np.testing.assert_array_equal(a["xvals"], b["xvals"])
    comm.close()
