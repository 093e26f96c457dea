"""CPU, world_size 2, gloo: the multi-rank host path.

* the row partition and the unique-id hand-off used by the sharded engines
  (admm_project_amd.parallel) over a real torch.distributed group;
* the transpose-reduction scheme itself: each rank runs the oracle's sharded loop on ITS rows
  with gloo all-reduces and must reproduce the unsharded oracle solvers (lad / huber / SVM)
  iterate by iterate -- this is the algebra the engine's one-all-reduce-per-iteration implements;
* bench.py's barrier + max-over-ranks timing helper.
No GPU compute happens here (the HIP engine has no CPU fallback)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import admm_project_amd as ap
        from admm_project_amd import parallel
        from oracle import parallel_ref
        from oracle import solvers_ref as S

        out = {}
        # --- unique id hand-off (rank 0 creates, broadcast_object_list distributes)
        box = [parallel.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        out["uid"] = box[0]

        def allreduce(a):
            t = torch.from_numpy(np.array(a, dtype=np.float64, copy=True))
            dist.all_reduce(t)
            return t.numpy()

        # --- sharded oracle == unsharded oracle
        checks = {}
        p = ap.synth.lad_problem(0, 515, 48)  # 515 rows: uneven split 258 + 257
        lo, hi = parallel.my_rows(515, rank, world)
        out["rows"] = (lo, hi)
        r = parallel_ref.sharded_unwrapped("lad", p["D"][lo:hi], p["s"][lo:hi], allreduce)
        full = S.lad(p["D"], p["s"], {})
        checks["lad"] = (r["steps"] == full["steps"],
                         float(np.max(np.abs(r["xvals"] - full["xvals"]))),
                         float(np.max(np.abs(r["pnorm"] - full["pnorm"]) / full["pnorm"])),
                         float(np.max(np.abs(r["dnorm"] - full["dnorm"]) / np.maximum(full["dnorm"], 1e-12))),
                         float(np.max(np.abs(r["zopt"] - full["zopt"][lo:hi]))))
        p = ap.synth.huber_problem(1, 300, 40)
        lo2, hi2 = parallel.my_rows(300, rank, world)
        r = parallel_ref.sharded_unwrapped("huber", p["D"][lo2:hi2], p["s"][lo2:hi2], allreduce, rho=1.0)
        full = S.huberfit(p["D"], p["s"], {})
        checks["huber"] = (r["steps"] == full["steps"], float(np.max(np.abs(r["xvals"] - full["xvals"]))))
        p = ap.synth.svm_problem(0, 64, 64)
        lo3, hi3 = parallel.my_rows(128, rank, world)
        r = parallel_ref.sharded_unwrapped("svm-hinge", p["D"][lo3:hi3], p["ell"][lo3:hi3], allreduce, C=p["C"],
                                           nodualerror=True, stopcond="both", z0=p["z0"][lo3:hi3],
                                           u0=p["u0"][lo3:hi3])
        full = S.linearsvm(p["D"], p["ell"], p["C"], dict(x0=p["x0"], z0=p["z0"], u0=p["u0"]))
        checks["svm"] = (r["steps"] == full["steps"], float(np.max(np.abs(r["xvals"] - full["xvals"]))),
                         float(np.max(np.abs(r["Hnormsq"] - full["Hnormsq"]) / np.maximum(full["Hnormsq"], 1e-12))))
        # --- consensus lasso with ONE packed exchange per iteration == the 4-slice oracle (2 local slices per rank)
        p = ap.synth.lasso_problem(1, 256, 64)
        lo4, hi4 = parallel.my_rows(256, rank, world)
        r = parallel_ref.sharded_consensus_lasso(p["D"][lo4:hi4], p["s"][lo4:hi4], p["lam"], [64, 64], allreduce)
        full = S.lasso(p["D"], p["s"], p["lam"], dict(parallel="both", slices=0), workers=4)
        rel = lambda a, b: float(np.max(np.abs(np.asarray(a) - np.asarray(b)) / np.maximum(np.abs(np.asarray(b)), 1e-300)))
        checks["consensus"] = (r["steps"] == full["steps"], float(np.max(np.abs(r["xvals"] - full["xvals"]))),
                               float(np.max(np.abs(r["uvals"] - full["uvals"]))), rel(r["pnorm"], full["pnorm"]),
                               rel(r["dnorm"], full["dnorm"]), rel(r["perr"], full["perr"]), rel(r["derr"], full["derr"]))
        out["checks"] = checks

        # --- gather of row-sharded slices
        out["gathered"] = parallel.gather_rows(dist, np.arange(lo, hi, dtype=np.float64), 515)

        # --- bench timing helper: max over ranks
        sys.argv = ["bench.py"]
        import bench

        out["tmax"] = bench.max_over_ranks(dist, 1.0 + rank, device="cpu")
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in procs:
        rank, out = q.get(timeout=240)
        res[rank] = out
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0]["uid"] == res[1]["uid"] and len(res[0]["uid"]) == 128
    assert res[0]["rows"] == (0, 258) and res[1]["rows"] == (258, 515)
    for rank in (0, 1):
        c = res[rank]["checks"]
        # x to 1e-9 absolute; the residual norms (differences of nearly equal vectors near
        # convergence) to the 1e-6 relative bar of the north star
        assert c["lad"][0] and c["lad"][1] < 1e-9 and c["lad"][2] < 1e-6 and c["lad"][3] < 1e-6 and c["lad"][4] < 1e-9
        assert c["huber"][0] and c["huber"][1] < 1e-9
        assert c["svm"][0] and c["svm"][1] < 1e-7 and c["svm"][2] < 1e-6
        # the packed exchange: iterates to 1e-10; lassonorms' first value is a difference of two shrinking sums
        # (q - N*||xave - c||^2): about one digit lost, far inside the 1e-6 bar
        cc = c["consensus"]
        assert cc[0] and cc[1] < 1e-10 and cc[2] < 1e-10 and cc[3] < 1e-8 and cc[4] < 1e-8 and cc[5] < 1e-10 and cc[6] < 1e-10
        np.testing.assert_array_equal(res[rank]["gathered"], np.arange(515.0))
        assert res[rank]["tmax"] == 2.0
