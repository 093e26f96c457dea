"""Host-side validation helpers mirroring /root/reference/errorcheck.m.

Only ``slicemaker`` (errorcheck.m:216-267) shapes the hot path: it defines the row
partition used by consensus lasso and by the transpose-reduction (row-sharded) engines.
"""
from __future__ import annotations

import numpy as np


def slicemaker(slices, workers, length):
    """Balance a slice specification over ``workers`` (errorcheck.m:216-267).

    * scalar 0  -> ``floor(len/workers)`` rows each, the first ``mod(len, workers)`` slices
      get one extra row (errorcheck.m:249-259);
    * scalar k>0 -> blocks of k rows, last block the remainder.  Deviation q13 (documented):
      the reference overwrites the last full block with ``mod(len,k) = 0`` when k divides
      len; we return the evenly divided blocks instead;
    * vector    -> must sum to ``length`` (errorcheck.m:263-265).
    """
    sl = np.atleast_1d(np.asarray(slices))
    if sl.dtype.kind not in "iuf":
        raise TypeError("Argument slices is not a numeric vector or integer!")
    sl = np.floor(np.real(sl)).astype(np.int64)
    length = int(length)
    workers = int(workers)
    if sl.size == 1 and sl[0] > 0:
        size = int(sl[0])
        out = [size] * (length // size)
        if length % size:
            out.append(length % size)
        return out
    if sl.size == 1 and sl[0] == 0:
        if workers <= 0:
            raise ValueError("There are no workers on this machine, cannot perform parallel ADMM!")
        rem = length % workers
        size = length // workers
        return [size + 1] * rem + [size] * (workers - rem)
    if int(sl.sum()) != length:
        raise ValueError("The number of parallel slices does not match length of x!")
    return [int(k) for k in sl]


def slice_ranges(slices):
    """0-based [lo, hi) row ranges of consecutive slices (getProxOps.m:402-412)."""
    starts = np.concatenate([[0], np.cumsum(np.asarray(slices, dtype=np.int64))])
    return [(int(starts[i]), int(starts[i + 1])) for i in range(len(slices))]


def rank_rows(length, rank, nranks):
    """Row range of ``rank`` when ``length`` rows are sharded over ``nranks`` devices with
    slicemaker(0, nranks, length)."""
    return slice_ranges(slicemaker(0, nranks, length))[rank]


def is_nonnegative_real(v, name):
    if not np.isscalar(v) or not np.isreal(v) or v < 0:
        raise ValueError(f"Argument {name} is not a nonnegative real number!")
    return float(v)


def is_positive_real(v, name):
    if not np.isscalar(v) or not np.isreal(v) or v <= 0:
        raise ValueError(f"Argument {name} is not a positive real number!")
    return float(v)
