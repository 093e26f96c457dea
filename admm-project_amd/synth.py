"""Synthetic problem generators: the reference testers' recipes, re-stated with NumPy RNG.

MATLAB's ``randn/sprandn/randsample`` streams cannot be reproduced without
MATLAB, so inputs follow the same distributions and sizes with
``numpy.random.default_rng(seed)``.  All matrices are returned column-major
(Fortran order), fp64, matching the MATLAB layout the engine consumes.
Host-side only; no oracle or GPU dependency.
"""
from __future__ import annotations

import os
import struct
from concurrent.futures import ThreadPoolExecutor

import numpy as np


def host_cores():
    """CPU cores this process may really use: the scheduler affinity, capped by the cgroup's CPU quota when there is
    one (a BLAS pool sized from os.cpu_count() on a box with a smaller share oversubscribes itself)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:  # cgroup v2: "<quota> <period>" or "max <period>"
            quota, period = fh.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                quota, period = int(fq.read()), int(fp.read())
            if quota > 0 and period > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return max(1, n)


def _randn_cols(rng_seed, rows, cols, threads=None):
    """randn(rows, cols) in Fortran order, generated column-block-wise in parallel."""
    out = np.empty((rows, cols), dtype=np.float64, order="F")
    if rows * cols < (1 << 22):
        rng = np.random.default_rng(rng_seed)
        for j0 in range(0, cols, 256):
            j1 = min(cols, j0 + 256)
            out[:, j0:j1] = rng.standard_normal((j1 - j0, rows)).T
        return out
    threads = threads or int(os.environ.get("ADMM_SYNTH_THREADS", "0")) or min(32, os.cpu_count() or 1)
    blk = 64  # fixed column-block size: the data must not depend on the thread count
    blocks = [(j0, min(cols, j0 + blk)) for j0 in range(0, cols, blk)]
    seeds = np.random.SeedSequence(rng_seed).spawn(len(blocks))

    def fill(args):
        (j0, j1), ss = args
        rng = np.random.default_rng(ss)
        # one column at a time keeps the per-thread scratch small
        for j in range(j0, j1):
            rng.standard_normal(rows, out=out[:, j])

    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(fill, zip(blocks, seeds)))
    return out


def _unit_cols_rows(rng_seed, rows, cols, lo, hi, threads=None):
    """Rows [lo, hi) of the column-normalised randn(rows, cols) that ``_randn_cols`` (large path)
    + normalisation produce, without ever holding the full matrix (row-sharded ranks)."""
    out = np.empty((hi - lo, cols), dtype=np.float64, order="F")
    threads = threads or int(os.environ.get("ADMM_SYNTH_THREADS", "0")) or min(32, os.cpu_count() or 1)
    blk = 64  # fixed column-block size: the data must not depend on the thread count
    blocks = [(j0, min(cols, j0 + blk)) for j0 in range(0, cols, blk)]
    seeds = np.random.SeedSequence(rng_seed).spawn(len(blocks))

    def fill(args):
        (j0, j1), ss = args
        rng = np.random.default_rng(ss)
        col = np.empty(rows)
        for j in range(j0, j1):
            rng.standard_normal(rows, out=col)
            out[:, j] = col[lo:hi] / np.sqrt(col @ col)

    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(fill, zip(blocks, seeds)))
    return out


def lasso_problem(seed=0, rows=2 ** 8, cols=2 ** 6, threads=None, row_range=None):
    """testers/lassotest.m:109-122: D=randn with unit-norm columns, 60 %-dense planted
    testx, s = D*testx + sqrt(0.001)*randn, lambda = 0.1*||D's||_inf.

    ``row_range=(lo, hi)``: return only those rows of D and s (one rank's shard of the same
    problem); ``lam`` is then computed from the LOCAL D's and must be re-derived by the caller
    from the all-reduced D's."""
    rng = np.random.default_rng(seed)
    testx = np.where(rng.random(cols) < 0.6, rng.standard_normal(cols), 0.0)
    if row_range is not None:
        lo, hi = row_range
        if rows * cols < (1 << 22):
            full = lasso_problem(seed, rows, cols, threads)
            D = np.asfortranarray(full["D"][lo:hi])
            return dict(D=D, s=full["s"][lo:hi].copy(), lam=full["lam"], testx=full["testx"])
        D = _unit_cols_rows(seed + 7919, rows, cols, lo, hi, threads)
        noise = np.sqrt(0.001) * rng.standard_normal(rows)
        s = D @ testx + noise[lo:hi]
        lam = 0.1 * float(np.max(np.abs(D.T @ s)))
        return dict(D=D, s=s, lam=lam, testx=testx)
    D = _randn_cols(seed + 7919, rows, cols, threads)
    # column normalisation, blockwise to bound temporaries at large sizes
    for j0 in range(0, cols, 512):
        j1 = min(cols, j0 + 512)
        nrm = np.sqrt(np.einsum("ij,ij->j", D[:, j0:j1], D[:, j0:j1]))
        D[:, j0:j1] /= nrm
    s = D @ testx + np.sqrt(0.001) * rng.standard_normal(rows)
    lam = 0.1 * float(np.max(np.abs(D.T @ s)))
    return dict(D=D, s=s, lam=lam, testx=testx)


def lad_problem(seed=0, rows=2 ** 10, cols=2 ** 7):
    """testers/ladtest.m:116-123: s = D*xtrue with 2 % of rows hit by 100*randn outliers."""
    rng = np.random.default_rng(seed)
    D = np.asfortranarray(rng.standard_normal((cols, rows)).T)
    xtrue = 10.0 * rng.standard_normal(cols)
    s = D @ xtrue
    k = int(np.ceil(rows / 50))
    idx = rng.choice(rows, size=k, replace=False)
    s[idx] += 100.0 * rng.standard_normal(k)
    return dict(D=D, s=s, xtrue=xtrue)


def huber_problem(seed=0, rows=2 ** 11, cols=2 ** 7):
    """testers/huberfittest.m:122-128."""
    rng = np.random.default_rng(seed)
    testx = rng.standard_normal(cols)
    D = np.asfortranarray(rng.standard_normal((cols, rows)).T)
    D /= np.sqrt(np.sum(D * D, axis=0))
    s = D @ testx + np.sqrt(0.01) * rng.standard_normal(rows)
    sparse_noise = np.where(rng.random(rows) < 200.0 / rows, rng.random(rows), 0.0)
    s = s + 10.0 * sparse_noise
    return dict(D=D, s=s, testx=testx)


def tv_problem(seed=0, n=2 ** 7):
    """testers/totalvariationtest.m:109-127: three random plateaus + N(0,1) noise."""
    rng = np.random.default_rng(seed)
    truex = np.ones(n)
    for _ in range(3):
        rs = int(rng.integers(1, n + 1))
        ri = int(rng.integers(1, 11))
        lo = int(np.ceil(rs / 2))
        truex[lo - 1:rs] *= ri
    s = truex + rng.standard_normal(n)
    return dict(s=s, truex=truex, lam=1.0)


def svm_problem(seed=0, mpos=2 ** 7, mneg=2 ** 7, sep=0.2):
    """testers/linearsvmtest.m:133-144: two noisy bands around the line x1 = x2."""
    rng = np.random.default_rng(seed)
    tpos = np.linspace(0.0, 2.0, mpos)
    tneg = np.linspace(0.0, 2.0, mneg)
    pos = np.stack([tpos + rng.random(mpos) - sep * rng.random(mpos),
                    tpos - rng.random(mpos) + sep * rng.random(mpos)], axis=1)
    neg = np.stack([tneg - rng.random(mneg) + sep * rng.random(mneg),
                    tneg + rng.random(mneg) - sep * rng.random(mneg)], axis=1)
    D = np.asfortranarray(np.concatenate([pos, neg], axis=0))
    ell = np.ones(mpos + mneg)
    ell[mpos:] = -1.0
    m, n = D.shape
    return dict(D=D, ell=ell, C=0.5, x0=rng.random(n), z0=rng.random(m), u0=rng.random(m))


def model_problem(seed=0, rows=2 ** 7, cols=2 ** 7):
    """testers/modeltest.m:115-118."""
    rng = np.random.default_rng(seed)
    P = np.asfortranarray(rng.standard_normal((rows, cols)))
    Q = np.asfortranarray(rng.standard_normal((rows, cols)))
    return dict(P=P, Q=Q, r=rng.standard_normal(rows), s=rng.standard_normal(rows))


def qp_bounded_problem(seed=0, n=2 ** 7):
    """Bounded QP in the shape quadraticprogram.m's 'bounded' branch accepts
    (SPD P, q, box lb<=ub); quadraticprogramtest.m's own pin is unusable (q25)."""
    rng = np.random.default_rng(seed)
    G = rng.standard_normal((n, n))
    P = np.asfortranarray(G.T @ G / n + 0.1 * np.eye(n))
    q = rng.standard_normal(n)
    lb = -rng.random(n)
    ub = rng.random(n)
    return dict(P=P, q=q, r=0.0, lb=lb, ub=ub)


def lp_problem(seed=0, rows=2 ** 6, cols=2 ** 7):
    """testers/linearprogramtest.m:107-111: b = rand+0.5, truexopt = |randn|, D = |randn|, s = D*truexopt."""
    rng = np.random.default_rng(seed)
    b = rng.random(cols) + 0.5
    truex = np.abs(rng.standard_normal(cols))
    D = np.asfortranarray(np.abs(rng.standard_normal((rows, cols))))
    return dict(b=b, D=D, s=D @ truex, truex=truex)


def qp_standard_problem(seed=0, rows=2 ** 6, cols=2 ** 7):
    """testers/quadraticprogramtest.m:120-131: P = V*diag(1+rand)*V' from the eigenvectors of a random
    symmetric matrix, q = randn, r = randn, D = |randn|, s = D*|randn|."""
    rng = np.random.default_rng(seed)
    G = rng.random((cols, cols))
    _, V = np.linalg.eigh(G + G.T)
    P = np.asfortranarray(V @ np.diag(1.0 + rng.random(cols)) @ V.T)
    P = 0.5 * (P + P.T)
    q = rng.standard_normal(cols)
    r = float(rng.standard_normal())
    truex = np.abs(rng.standard_normal(cols))
    D = np.asfortranarray(np.abs(rng.standard_normal((rows, cols))))
    return dict(P=P, q=q, r=r, D=D, s=D @ truex, truex=truex)


def basispursuit_problem(seed=0, rows=2 ** 6, cols=2 ** 7):
    """testers/basispursuittest.m:106-109: fat D, s = D*testx.  The reference asks for
    ``sprandn(cols, 1, 0.1*cols)``, i.e. a density of 0.1*cols >= 1 for cols >= 10, which
    saturates at a fully dense N(0,1) vector; ``density`` is restated as min(1, 0.1*cols)."""
    rng = np.random.default_rng(seed)
    D = np.asfortranarray(rng.standard_normal((rows, cols)))
    density = min(1.0, 0.1 * cols)
    testx = np.where(rng.random(cols) < density, rng.standard_normal(cols), 0.0)
    return dict(D=D, s=D @ testx, testx=testx)


def read_idx1_labels(path, count=None):
    """examples/mnistsvm.m:215-229: big-endian idx1 label file (magic 2049)."""
    with open(path, "rb") as f:
        magic, n = struct.unpack(">ii", f.read(8))
        if magic != 2049:
            raise ValueError("Invalid label file header")
        if count is not None and n < count:
            raise ValueError("Trying to read too many digits")
        data = np.frombuffer(f.read(n if count is None else count), dtype=np.uint8)
    return data.astype(np.float64)


def read_idx3_images(path, count=None, border=4):
    """examples/mnistsvm.m:188-256: big-endian idx3 images (magic 2051), `border`-pixel
    trim (28x28 -> 20x20), /255, row-major flatten (mnistsvm.m:61-72) -> count x 400."""
    with open(path, "rb") as f:
        magic, n, h, w = struct.unpack(">iiii", f.read(16))
        if magic != 2051:
            raise ValueError("Invalid image file header")
        k = n if count is None else count
        if n < k:
            raise ValueError("Trying to read too many digits")
        raw = np.frombuffer(f.read(k * h * w), dtype=np.uint8).reshape(k, h, w)
    img = raw[:, border:h - border, border:w - border].astype(np.float64) / 255.0
    return np.asfortranarray(img.reshape(k, -1))


def reference_mnist_labels(which="train"):
    """The label files the reference tree holds (examples/MNIST/{train,t10k}-labels.idx1-ubyte), committed byte for
    byte under tests/golden/mnist/; None where the repository's tests are not installed next to the package."""
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "mnist",
                        f"{which}-labels.idx1-ubyte")
    return read_idx1_labels(path) if os.path.exists(path) else None


def mnist_like_problem(seed=1, m=6000, n=400, digit=0, labels=None):
    """Config-3 stand-in (the MNIST image files are absent from the reference tree):
    entries 0 w.p. 0.5 else U(0,1) as cropped /255 pixels; labels digit-vs-rest -> +-1
    (mnistsvm.m:136-142) from the real label file when given, else synthetic digits."""
    rng = np.random.default_rng(seed)
    D = np.asfortranarray(np.where(rng.random((n, m)) < 0.5, 0.0, rng.random((n, m))).T)
    if labels is None:
        labels = rng.integers(0, 10, size=m).astype(np.float64)
    lab = np.asarray(labels[:m], dtype=np.float64)
    ell = np.where(lab == digit, 1.0, -1.0)
    return dict(D=D, ell=ell, C=0.5, x0=rng.random(n), z0=rng.random(m), u0=rng.random(m))


def graded_matrix(seed, rows, cols, kappa):
    """rows x cols matrix with singular values graded geometrically from 1 down to 1/kappa (random singular
    vectors): the conditioning stress case for the un-shifted chol(D'D) of lad.m:134 / huberfit.m:166 (q20)."""
    rng = np.random.default_rng(seed)
    U, _ = np.linalg.qr(rng.standard_normal((rows, cols)))
    V, _ = np.linalg.qr(rng.standard_normal((cols, cols)))
    sv = np.geomspace(1.0, 1.0 / kappa, cols)
    return np.asfortranarray((U * sv) @ V.T)


def lad_problem_conditioned(seed=0, rows=512, cols=64, kappa=1e3):
    """ladtest.m's recipe (outliers on 2 % of the rows) on a matrix with cond(D) = kappa."""
    rng = np.random.default_rng(seed + 1000)
    D = graded_matrix(seed, rows, cols, kappa)
    xtrue = rng.standard_normal(cols)
    s = D @ xtrue
    k = int(np.ceil(rows / 50))
    idx = rng.choice(rows, size=k, replace=False)
    s[idx] += 10.0 * rng.standard_normal(k)
    return dict(D=D, s=s, xtrue=xtrue)


def rank_deficient_pixels(seed=1, m=1500, n=400, digit=0, dead=0.05, dup=0.03):
    """mnist_like_problem whose matrix is rank deficient the way cropped MNIST is (mnistsvm.m:61-72): `dead` of the
    pixel columns are zero in every sample, `dup` of them duplicate another column."""
    p = mnist_like_problem(seed=seed, m=m, n=n, digit=digit)
    rng = np.random.default_rng(seed + 77)
    D = p["D"].copy(order="F")
    cols = rng.permutation(n)
    nd, nu = int(round(dead * n)), int(round(dup * n))
    D[:, cols[:nd]] = 0.0
    for j in cols[nd:nd + nu]:
        src = cols[nd + nu + int(rng.integers(0, n - nd - nu))]
        D[:, j] = D[:, src]
    p["D"] = np.asfortranarray(D)
    p["rank"] = n - nd - nu
    return p
