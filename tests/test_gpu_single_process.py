"""ONE host process driving several ranks (parallel.LocalGroup = admm_comm_init_all; Engine.create_all / run_all =
admm_engine_create_all / _run_all): the deployment of a MATLAB session with the MEX gateway (the reference opens
its pool from one session, admm.m:347-356) and the way a one-GPU box rehearses an 8-GPU node -- eight ranks, each
with the rows slicemaker(0, 8, m) gives it (errorcheck.m:249-259), all on device 0 over the host-staged transport.
Every rank must reproduce the unsharded / 8-slice oracle."""
import numpy as np
import pytest

from oracle import solvers_ref as S

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


@pytest.fixture()
def group8(gpu):
    from admm_project_amd import parallel
    g = parallel.LocalGroup(8, devices=[0] * 8, transport="shm")
    yield g
    g.close()


def test_world8_lad_matches_unsharded_oracle(gpu, group8):
    from admm_project_amd import parallel
    m, n = 1003, 40  # 1003 = 8*125 + 3: the first three ranks get one row more (errorcheck.m:255-259)
    p = gpu.synth.lad_problem(0, m, n)
    ref = S.lad(p["D"], p["s"], dict(objevals=1))

    def rank(r, comm):
        lo, hi = parallel.my_rows(m, comm)
        res = gpu.lad(p["D"][lo:hi], p["s"][lo:hi], dict(objevals=1, comm=comm))
        return lo, hi, res

    outs = group8.on_ranks(rank)
    sizes = [hi - lo for lo, hi, _ in outs]
    assert sizes == [126, 126, 126, 125, 125, 125, 125, 125] and outs[-1][1] == m
    for lo, hi, g in outs:
        assert g["steps"] == ref["steps"]
        for k in ("xvals", "pnorm", "dnorm", "perr", "derr", "objevals"):
            assert _rel(g[k], ref[k]) < 1e-9, k
        assert _rel(g["zopt"], ref["zopt"][lo:hi]) < 1e-9 and _rel(g["uopt"], ref["uopt"][lo:hi]) < 1e-9


@pytest.mark.parametrize("objgram", [0, 1])
def test_world8_row_sharded_lasso_objective_forms(gpu, group8, objgram):
    """Row-sharded lasso (transpose reduction): x, z, u replicated.  objgram = 1: the objective's data term comes out of
    the x-update's right-hand side on every rank (no pass over the row shards, no collective; 1/2*s's summed over the
    shards once at create); objgram = 0: sharded engines keep the literal form (one scalar all-reduce per iteration)."""
    from admm_project_amd import parallel
    m, n = 1003, 60
    p = gpu.synth.lasso_problem(2, m, n)
    ref = S.lasso(p["D"], p["s"], p["lam"], dict(objevals=1))

    def rank(r, comm):
        lo, hi = parallel.my_rows(m, comm)
        return gpu.lasso(p["D"][lo:hi], p["s"][lo:hi], p["lam"], dict(objevals=1, comm=comm, objgram=objgram))

    for g in group8.on_ranks(rank):
        assert g["steps"] == ref["steps"]
        for k in ("xvals", "zvals", "uvals", "pnorm", "dnorm", "perr", "derr", "xopt"):
            assert _rel(g[k], ref[k]) < 1e-8, k
        assert np.max(np.abs(g["objevals"] - ref["objevals"])) <= 1e-12 * float(p["s"] @ p["s"])


def test_world8_consensus_lasso_matches_8_slice_oracle(gpu, group8):
    """config 4's layout: one row slice per rank, ONE all-reduce of [sum x_k; sum u_k; q] per iteration"""
    from admm_project_amd import parallel
    m, n = 8 * 96 + 5, 64
    p = gpu.synth.lasso_problem(3, m, n)
    ref = S.lasso(p["D"], p["s"], p["lam"], dict(objevals=1, parallel="both", slices=0), workers=8)

    def rank(r, comm):
        lo, hi = parallel.my_rows(m, comm)
        return gpu.lasso(p["D"][lo:hi], p["s"][lo:hi], p["lam"],
                         dict(objevals=1, parallel="both", comm=comm, workers=1, xsolve="inverse"))

    for g in group8.on_ranks(rank):
        assert g["steps"] == ref["steps"]
        for k in ("xvals", "zvals", "uvals", "pnorm", "dnorm", "perr", "derr", "objevals", "Hnormsq"):
            assert _rel(g[k], ref[k]) < 1e-8, k
        assert np.all(g["zopt"] == 0.0)  # q9


def test_native_create_all_run_all(gpu):
    """the C entry points a single-process binding calls: admm_comm_init_all, admm_engine_create_all, _run_all"""
    from admm_project_amd import parallel
    L = gpu._lib
    m, n, R = 640, 48, 4
    p = gpu.synth.lad_problem(1, m, n)
    ref = S.lad(p["D"], p["s"], dict(objevals=1))
    grp = parallel.LocalGroup(R, devices=[0] * R, transport="shm")
    try:
        kws = []
        for c in grp.comms:
            lo, hi = parallel.my_rows(m, c)
            kws.append(dict(D=p["D"][lo:hi], s=p["s"][lo:hi], comm=c))
        engines = gpu.Engine.create_all(L.PROB_LAD, kws)
        sums = gpu.Engine.run_all(engines, objevals=1)
        for e, sm in zip(engines, sums):
            assert sm.steps == ref["steps"]
            x = e.fetch(L.F_XOPT, n)
            assert _rel(x, ref["xopt"]) < 1e-9
            pn = e.fetch(L.F_PNORM, sm.steps)
            assert _rel(pn, ref["pnorm"]) < 1e-9
            e.close()
        # a failing rank reports through the caller's last error, with its rank (no communicator: nothing blocks)
        bad = [dict(D=p["D"][:100], s=p["s"][:100]), dict(D=p["D"][:100])]  # rank 1 lacks the signal vector
        with pytest.raises(gpu.AdmmError) as ei:
            gpu.Engine.create_all(L.PROB_LAD, bad)
        assert "rank 1" in str(ei.value)
    finally:
        grp.close()


def test_a_rank_failing_outside_a_collective_releases_its_peers(gpu):
    """admm_engine_create_all with ONE rank whose description is unusable (no matrix): that rank fails before its first
    collective while the other seven have entered the all-reduce of D_g'D_g.  The failing rank's thread abandons the
    group (comm_abort: the shm header's flag; ncclCommAbort on RCCL), so the call returns the first failure in seconds
    instead of sitting in the barrier until its 120 s timeout (for ever, inside RCCL).  A fresh group works."""
    import time

    from admm_project_amd import parallel
    L = gpu._lib
    m, n = 1003, 40
    p = gpu.synth.lad_problem(0, m, n)
    g = parallel.LocalGroup(8, devices=[0] * 8, transport="shm")
    try:
        kw = []
        for r in range(8):
            lo, hi = parallel.my_rows(m, g.comms[r])
            kw.append(dict(D=p["D"][lo:hi], s=p["s"][lo:hi], comm=g.comms[r]))
        kw[3] = dict(s=p["s"][:5], comm=g.comms[3], nvec=n)  # no D: admm_engine_create refuses it on the spot
        t0 = time.perf_counter()
        with pytest.raises(gpu.AdmmError) as ei:
            gpu.Engine.create_all(L.PROB_LAD, kw)
        assert time.perf_counter() - t0 < 30.0
        assert "rank 3" in str(ei.value)
    finally:
        g.close()
    g2 = parallel.LocalGroup(2, devices=[0, 0], transport="shm")
    try:
        ref = S.lad(p["D"], p["s"], dict(maxiters=5, domaxiters=1))

        def rank(r, comm):
            lo, hi = parallel.my_rows(m, comm)
            return gpu.lad(p["D"][lo:hi], p["s"][lo:hi], dict(maxiters=5, domaxiters=1, comm=comm))

        for res in g2.on_ranks(rank):
            assert _rel(res["xvals"], ref["xvals"]) < 1e-9
    finally:
        g2.close()


@pytest.mark.parametrize("nranks", [2])
def test_one_process_ranks_over_the_one_shot_p2p_all_reduce(gpu, nranks):
    """ADMM_COMM_P2P between ranks that are threads of ONE process (the MEX-gateway deployment): the peers' buffers are
    plain device pointers here, the kernels of the ranks run on their engines' streams at the same time, nothing
    synchronises with the host inside an iteration.  (On this box all ranks share device 0, and a rank's kernel waits
    for the kernels of the others: each needs a hardware queue of its own.  The runtime deals a process's streams onto a
    pool of queues per stream priority, depending on every stream made before -- two ranks on one queue trip the polling
    limit (seen once the suite grew: ADMM_E_COMM, no hang) --, so same-device ranks of one process take different
    priorities (comm.hip: comm_stream_create): up to three ranks are separated for certain.  One GPU per rank has no
    such limit.)"""
    from admm_project_amd import parallel
    m, n = 1003, 40
    p = gpu.synth.lad_problem(0, m, n)
    ref = S.lad(p["D"], p["s"], dict(objevals=1))
    g = parallel.LocalGroup(nranks, devices=[0] * nranks, transport="p2p")
    try:
        tot = g.on_ranks(lambda r, comm: comm.allreduce_sum(np.array([1.0 + r, 10.0, -2.0 * r])))
        for t in tot:
            np.testing.assert_array_equal(t, [sum(1.0 + r for r in range(nranks)), 10.0 * nranks,
                                              -2.0 * sum(range(nranks))])

        def rank(r, comm):
            lo, hi = parallel.my_rows(m, comm)
            return lo, hi, gpu.lad(p["D"][lo:hi], p["s"][lo:hi], dict(objevals=1, comm=comm))

        outs = g.on_ranks(rank)
        for lo, hi, res in outs:
            assert res["steps"] == ref["steps"]
            for k in ("xvals", "pnorm", "dnorm", "perr", "derr", "objevals"):
                assert _rel(res[k], ref[k]) < 1e-9, k
            assert _rel(res["zopt"], ref["zopt"][lo:hi]) < 1e-9
        # every rank holds bitwise the same replicated x (rank-ordered sums)
        for _, _, res in outs[1:]:
            assert np.array_equal(res["xvals"], outs[0][2]["xvals"])
    finally:
        g.close()
