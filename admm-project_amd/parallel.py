"""Row-sharded (multi-GPU) runs: one process per GPU, rows of D split with the reference's
``slicemaker(0, workers, len)`` rule (errorcheck.m:249-259), sums of D_g' products and of the
residual partials exchanged by ONE all-reduce per iteration inside the engine (RCCL over xGMI).

This replaces the reference's ``parfor`` transpose reduction (unwrappedadmm.m:96-141) and is
used by passing ``options['comm'] = Comm`` and the LOCAL rows to the ordinary solvers:

    comm = parallel.init_from_torch(dist, device=local_rank)          # after init_process_group
    lo, hi = parallel.my_rows(m, comm)
    results = admm_project_amd.lad(D[lo:hi], s[lo:hi], dict(comm=comm))

x (and everything derived from it) is replicated and identical on all ranks; z, u and their
histories are this rank's row slice.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from .errorcheck import rank_rows

__all__ = ["Comm", "LocalGroup", "unique_id", "init_from_torch", "my_rows", "gather_rows"]


def unique_id():
    """128-byte communicator id (RCCL's ncclUniqueId when a GPU is present, random bytes otherwise)."""
    buf = C.create_string_buffer(L.COMM_ID_BYTES)
    L.check(L.load().admm_comm_unique_id(buf))
    return bytes(buf.raw)


class Comm:
    """Owner of one ``admm_comm`` handle."""

    def __init__(self, uid, rank, nranks, device=0, transport="rccl"):
        if len(uid) != L.COMM_ID_BYTES:
            raise ValueError("unique id must be 128 bytes")
        tr = {"rccl": L.COMM_RCCL, "shm": L.COMM_SHM, "p2p": L.COMM_P2P}[transport]
        h = C.c_void_p()
        L.check(L.load().admm_comm_init(uid, int(rank), int(nranks), int(device), tr, C.byref(h)))
        self.handle = h
        self.rank, self.nranks, self.device, self.transport = int(rank), int(nranks), int(device), transport

    def allreduce_sum(self, host_array):
        """Sum a host fp64 array over the ranks (convenience / tests; the loop reduces on the device)."""
        a = np.ascontiguousarray(host_array, dtype=np.float64)
        L.check(L.load().admm_comm_allreduce_sum(self.handle, L.as_dp(a), a.size))
        return a

    def allreduce_latency_us(self, count, reps=50):
        """average microseconds of one all-reduce of ``count`` doubles, enqueued back to back (collective call)"""
        us = C.c_double(0.0)
        L.check(L.load().admm_comm_measure_latency(self.handle, int(count), int(reps), C.byref(us)))
        return us.value

    def close(self):
        if getattr(self, "handle", None):
            L.load().admm_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LocalGroup:
    """``nranks`` communicators of one group inside THIS process (admm_comm_init_all): the single-process deployment,
    one host driving several GPUs as a MATLAB session with the MEX gateway does (the reference opens its pool from
    one session: admm.m:347-356).  ``devices[r]`` is rank r's HIP device; with the ``shm`` transport the same device
    may serve several ranks (how a one-GPU box rehearses 8 ranks).  Use with ``Engine.create_all`` / ``Engine.run_all``
    or run the ordinary per-rank solvers on threads with ``on_ranks``."""

    def __init__(self, nranks, devices=None, transport="shm"):
        devices = list(devices) if devices is not None else [0] * int(nranks)
        if len(devices) != int(nranks):
            raise ValueError("one device per rank")
        tr = {"rccl": L.COMM_RCCL, "shm": L.COMM_SHM, "p2p": L.COMM_P2P}[transport]
        dev = (C.c_int * int(nranks))(*devices)
        hs = (C.c_void_p * int(nranks))()
        L.check(L.load().admm_comm_init_all(int(nranks), dev, tr, hs))
        self.comms = []
        for r in range(int(nranks)):
            c = Comm.__new__(Comm)
            c.handle = C.c_void_p(hs[r])
            c.rank, c.nranks, c.device, c.transport = r, int(nranks), int(devices[r]), transport
            self.comms.append(c)
        self.nranks = int(nranks)

    def on_ranks(self, fn):
        """fn(rank, comm) on one Python thread per rank (the ctypes calls release the GIL and block inside the
        collectives, as the native *_all entry points do); returns the results in rank order."""
        import threading

        out, err = [None] * self.nranks, [None] * self.nranks

        def body(r):
            try:
                out[r] = fn(r, self.comms[r])
            except BaseException as exc:  # noqa: BLE001 -- reported to the caller below
                err[r] = exc

        th = [threading.Thread(target=body, args=(r,)) for r in range(self.nranks)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for e in err:
            if e is not None:
                raise e
        return out

    def close(self):
        for c in self.comms:
            c.close()
        self.comms = []


def init_from_torch(dist, device=0, transport="rccl"):
    """Create the engine communicator from an initialised ``torch.distributed`` group: rank 0
    makes the unique id, ``broadcast_object_list`` hands it to the others (works with gloo and nccl)."""
    rank, world = dist.get_rank(), dist.get_world_size()
    box = [unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return Comm(box[0], rank, world, device=device, transport=transport)


def my_rows(length, comm_or_rank, nranks=None):
    """[lo, hi) rows of this rank: slicemaker(0, nranks, length) (errorcheck.m:249-259)."""
    if nranks is None:
        return rank_rows(length, comm_or_rank.rank, comm_or_rank.nranks)
    return rank_rows(length, int(comm_or_rank), int(nranks))


def gather_rows(dist, local, length):
    """All-gather row-sharded result slices (first axis) back into the full array, in rank order."""
    world = dist.get_world_size()
    parts = [None] * world
    dist.all_gather_object(parts, np.asarray(local))
    out = np.concatenate(parts, axis=0)
    assert out.shape[0] == length
    return out
