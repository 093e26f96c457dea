"""Oracle (test infrastructure): problem solvers, restating /root/reference/solvers/*.m.

Each function performs the reference solver's one-time setup (Gram matrix,
Cholesky, pseudo-inverse, difference operator), builds the same ``options``
constraint fields and calls the oracle's ``getproxops`` + ``admm``.  MATLAB's
parallel pool size (``gcp().NumWorkers``) is passed explicitly as ``workers``.
Parity pin status: see ``oracle/__init__.py``.
"""
from __future__ import annotations

import time

import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp

from .admm_ref import admm
from .proxops_ref import getproxops, soft_threshold


def slicemaker(slices, workers, length):
    """errorcheck.m:216-267.  Returns a list of slice sizes."""
    sl = np.atleast_1d(np.floor(np.real(np.asarray(slices, dtype=np.float64)))).astype(np.int64)
    if sl.size == 1 and sl[0] > 0:
        # 238-243 (bug q13: when `length` divides evenly, the last full slice is
        # overwritten by mod(len, size) = 0 -- restated literally).
        size = int(sl[0])
        nfull = length // size
        out = [size] * nfull
        last = -(-length // size)  # ceil
        while len(out) < last:
            out.append(0)
        out[last - 1] = length % size
        return out
    if sl.size == 1 and sl[0] == 0:
        if length % workers != 0:  # 249-255
            rem = length % workers
            size = length // workers
            return [size + 1] * rem + [size] * (workers - rem)
        return [length // workers] * workers  # 258
    if int(sl.sum()) != length:
        raise ValueError("The number of parallel slices does not match length of x!")
    return [int(k) for k in sl]


def pinv_matlab(A):
    """MATLAB's built-in ``pinv(A)`` as called at linearsvm.m:185 / unwrappedadmm.m:76: singular values
    ``<= tol = max(size(A)) * eps(norm(A))`` are treated as zero (documented default of the built-in; NumPy's own
    default ``rcond = 1e-15`` keeps directions MATLAB drops)."""
    A = np.asarray(A, dtype=np.float64)
    smax = float(np.linalg.norm(A, 2)) if A.size else 0.0
    if smax == 0.0:
        return np.zeros(A.shape[::-1])
    tol = max(A.shape) * np.spacing(smax)
    return np.linalg.pinv(A, rcond=tol / smax)


def huber_cvx(x):
    """CVX ``huber`` (huberfit.m:180): x^2 if |x|<=1 else 2|x|-1."""
    ax = np.abs(x)
    return np.where(ax <= 1.0, x * x, 2.0 * ax - 1.0)


def lasso(D, s, lam, options=None, workers=1):
    """solvers/lasso.m:77-245."""
    options = dict(options or {})
    t0 = time.perf_counter()
    rho = float(options.get("rho", 1.0))
    parallel = options.get("parallel", "none") in ("both", "zming", "xminf")  # 144-156
    if parallel:
        options["parallel"] = "none"
        options["stopcond"] = "both"
    m, n = D.shape
    if not parallel:  # 159-192
        Dts = D.T @ s
        if m >= n:
            L = sla.cholesky(D.T @ D + rho * np.eye(n), lower=True)
        else:
            L = sla.cholesky((D @ D.T) / rho + np.eye(m), lower=True)
        args = dict(D=D, Dts=Dts, L=L, U=L.T, m=m, n=n)
        args["lambda"] = lam
        args["parallel"] = 0
        args["rho"] = rho
        minx, minz, extra = getproxops("LASSO", args)
    else:  # 193-224
        slices = options.get("slices", 0)
        slices = np.atleast_1d(slices)[0]  # lasso.m:197 takes only slices(1)
        slices = slicemaker(slices, workers, m)
        args = dict(slices=slices, D=D, s=s, rho=rho, parallel=1)
        args["lambda"] = lam
        if "fatformula" in options:  # q12: "serial" = the corrected fat-slice formula (see proxops_ref)
            args["fatformula"] = options["fatformula"]
        minx, minz, extra = getproxops("LASSO", args)
        options["altu"] = extra["altu"]
        options["specialnorms"] = extra["specialnorms"]
    options["obj"] = lambda x, z: 0.5 * float(np.sum((D @ x - s) ** 2)) + lam * float(np.sum(np.abs(z)))  # 227
    options.update(A=1, At=1, m=n, nA=n, nB=n, B=-1, c=0, parallel="none")  # 232-239
    results = admm(minx, minz, options)
    results["solverruntime"] = time.perf_counter() - t0
    if parallel:
        results["_consensus"] = extra  # oracle-side: true consensus z lives in extra["_state"]["z"]
    return results


def lad(D, s, options=None):
    """solvers/lad.m:51-154."""
    options = dict(options or {})
    t0 = time.perf_counter()
    m, n = D.shape
    args = dict(D=D, s=s)
    if options.get("relax", 1) != 1:  # 124-126
        args["userelax"] = 1
    args["R"] = sla.cholesky(D.T @ D, lower=True)  # 134 (un-shifted, q20)
    minx, minz, _ = getproxops("lad", args)
    options.update(A=D, B=-1, c=s, m=m, nA=n, nB=m)  # 140-145
    options["obj"] = lambda x, z: float(np.sum(np.abs(z)))  # 148
    results = admm(minx, minz, options)
    results["solverruntime"] = time.perf_counter() - t0
    return results


def huberfit(D, s, options=None):
    """solvers/huberfit.m:83-186."""
    options = dict(options or {})
    t0 = time.perf_counter()
    m, n = D.shape
    args = dict(D=D, s=s)
    if options.get("relax", 1) != 1:
        args["userelax"] = 1
    args["R"] = sla.cholesky(D.T @ D, lower=True)  # 166
    minx, minz, _ = getproxops("huberfit", args)
    options.update(A=D, B=-1, c=s, m=m, nA=n, nB=m)
    options["obj"] = lambda x, z: 0.5 * float(np.sum(huber_cvx(z)))  # 180
    results = admm(minx, minz, options)
    results["solverruntime"] = time.perf_counter() - t0
    return results


def tv_operator(n):
    """totalvariation.m:127  D = spdiags([ones -ones], 0:1, n, n)."""
    return sp.diags([np.ones(n), -np.ones(n - 1)], [0, 1], shape=(n, n), format="csr")


def totalvariation(s, lam, options=None):
    """solvers/totalvariation.m:62-167."""
    options = dict(options or {})
    t0 = time.perf_counter()
    s = np.asarray(s, dtype=np.float64).reshape(-1)
    n = s.size
    D = tv_operator(n)
    Dt = D.T.tocsr()
    DtD = (Dt @ D).tocsc()
    objective = lambda x, z: 0.5 * float(np.sum((x - s) ** 2)) + lam * float(np.sum(np.abs(x[1:] - x[:-1])))  # 134-135
    args = dict(D=D, Dt=Dt, DtD=DtD, s=s, banded=options.pop("banded", 0))  # banded: oracle-side, see proxops_ref
    args["lambda"] = lam
    xmin, zmin, _ = getproxops("TotalVariation", args)
    options.update(A=D, At=Dt, B=-1, nB=n, c=0, m=n)  # 151-157
    options["obj"] = objective
    results = admm(xmin, zmin, options)
    results["solverruntime"] = time.perf_counter() - t0
    return results


def tv2d_operator(H, W):
    """D = [Dv; Dh] (2N x N, N = H*W, x = X(:) column-major): (Dv x)[i,j] = x[i,j] - x[i+1,j] for i < H-1,
    (Dh x)[i,j] = x[i,j] - x[i,j+1] for j < W-1; the rows of the last image row / column are zero.
    No reference counterpart (totalvariation.m is 1-D): this is the oracle of the engine's own extension."""
    N = H * W
    idx = np.arange(N).reshape((H, W), order="F")
    rows, cols, vals = [], [], []
    v = idx[:-1, :].reshape(-1)
    rows += [v, v]
    cols += [v, idx[1:, :].reshape(-1)]
    vals += [np.ones(v.size), -np.ones(v.size)]
    h = idx[:, :-1].reshape(-1)
    rows += [N + h, N + h]
    cols += [h, idx[:, 1:].reshape(-1)]
    vals += [np.ones(h.size), -np.ones(h.size)]
    return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(2 * N, N))


def totalvariation2d(S, lam, options=None):
    """Anisotropic 2-D TV denoising by the same ADMM splitting as totalvariation.m (D*x - z = 0), with the
    x-update solved directly (sparse LU) -- the device solves it by CG to 1e-11."""
    import scipy.sparse.linalg as spla

    options = dict(options or {})
    t0 = time.perf_counter()
    img = np.asarray(S, dtype=np.float64)
    H, W = img.shape
    N = H * W
    s = img.reshape(-1, order="F")
    D = tv2d_operator(H, W)
    Dt = D.T.tocsr()
    rho = float(options.get("rho", 1.0))
    lu = spla.splu((sp.identity(N, format="csc") + rho * (Dt @ D)).tocsc())
    xmin = lambda _x, z, u, r_: lu.solve(s + r_ * (Dt @ (z - u)))
    zmin = lambda x, _z, u, r_: soft_threshold(u + D @ x, lam / r_)
    options.update(A=D, At=Dt, B=-1, nA=N, nB=2 * N, c=0, m=2 * N)
    options["obj"] = lambda x, z: 0.5 * float(np.sum((x - s) ** 2)) + lam * float(np.sum(np.abs(D @ x)))
    results = admm(xmin, zmin, options)
    results["xopt"] = results["xopt"].reshape((H, W), order="F")
    results["solverruntime"] = time.perf_counter() - t0
    return results


def unwrappedadmm(zming, D, options=None, workers=1, keep_init=True):
    """solvers/unwrappedadmm.m:1-143.

    ``keep_init`` (documented deviation q14): explicit ``x0/z0/u0`` win over the
    reference's unconditional ``rand`` so that runs are reproducible.
    """
    options = dict(options or {})
    m, n = D.shape
    par = options.get("parallel", "none") in ("xminf", "zming", "both")
    if par:  # 45-74
        options["parallel"] = "zming" if options["parallel"] == "both" else "none"
        slices = slicemaker(options.get("slices", 0), workers, m)
        options["slices"] = slices
        starts = np.concatenate([[0], np.cumsum(slices)])
        st = {}

        def preprocess():  # 96-123
            W = np.zeros((n, n))
            for i in range(len(slices)):
                Di = D[starts[i]:starts[i + 1], :]
                W = W + Di.T @ Di
            st["W"] = W

        def proxf(_x, z, u, _rho):  # 125-141
            d = 0
            for i in range(len(slices)):
                lo, hi = starts[i], starts[i + 1]
                d = d + D[lo:hi, :].T @ (z[lo:hi] - u[lo:hi])
            return np.linalg.solve(st["W"], d)

        xminf = proxf
        options["preprocess"] = preprocess
    else:  # 76-78
        Dplus = pinv_matlab(D)
        xminf = lambda _x, z, u, _rho: Dplus @ (z - u)
    options.update(A=D, At=D.T, B=-1, nB=m, c=0, m=m)
    rng = np.random.default_rng(0)
    for key, size in (("x0", n), ("z0", m), ("u0", m)):  # 87-89
        if not (keep_init and key in options):
            options[key] = rng.random(size)
    options["maxiters"] = 1000  # 90
    options["stopcond"] = "both"
    options["nodualerror"] = 1
    if options.get("parallel") == "zming":
        # admm.m:397-407, 447-467: zming(x,z,u,rho,k) per slice, concatenated.
        sl = options["slices"]
        zi = zming
        zming = lambda x, z, u, rho: np.concatenate([zi(x, z, u, rho, k) for k in range(len(sl))])
        options["parallel"] = "none"
    return admm(xminf, zming, options)


def linearsvm(D, ell, C, options=None, workers=1):
    """solvers/linearsvm.m:92-246."""
    options = dict(options or {})
    t0 = time.perf_counter()
    loss = options.get("lossfunction", "hinge")  # 154-158
    par = options.get("parallel", "none") in ("both", "zming", "xminf")
    args = dict(D=D, Dt=D.T, ell=ell, C=C, lossfunction=loss)
    if par:
        options["parallel"] = "both"  # 174
        args["slices"] = slicemaker(options.get("slices", 0), workers, D.shape[0])
        options["slices"] = args["slices"]
    else:
        args["Dplus"] = pinv_matlab(D)  # 185
    _, minz, _ = getproxops("LinearSVM", args)
    if loss == "hinge":  # 231-237
        options["obj"] = lambda x, z: 0.5 * float(x @ x) + C * float(np.sum(np.maximum(1 - ell * (D @ x), 0)))
    else:
        options["obj"] = lambda x, z: 0.5 * float(x @ x) + C * float(np.sum(np.maximum(np.sign(1 - ell * (D @ x)), 0)))
    results = unwrappedadmm(minz, D, options, workers=workers)
    results["solverruntime"] = time.perf_counter() - t0
    return results


def basispursuit(D, s, options=None):
    """solvers/basispursuit.m:52-145."""
    options = dict(options or {})
    t0 = time.perf_counter()
    n = D.shape[1]
    DDt = D @ D.T
    P = np.eye(n) - D.T @ np.linalg.solve(DDt, D)  # 116-119
    q = D.T @ np.linalg.solve(DDt, s)  # 120
    minx, minz, _ = getproxops("BasisPursuit", dict(P=P, q=q))
    options.update(A=1, B=-1, c=0, m=n, nA=n, nB=n)
    options["obj"] = lambda x, z: float(np.sum(np.abs(x)))  # 140
    results = admm(minx, minz, options)
    results["solverruntime"] = time.perf_counter() - t0
    return results


def model(P, Q, r, s, options=None):
    """solvers/model.m:47-143."""
    options = dict(options or {})
    t0 = time.perf_counter()
    n = P.shape[1]
    args = dict(PtP=P.T @ P, Ptr=P.T @ r, QtQ=Q.T @ Q, Qts=Q.T @ s, n=n)
    minx, minz, _ = getproxops("Model", args)
    options.update(A=1, B=-1, c=0, m=n, nA=n, nB=n)
    options["obj"] = lambda x, z: 0.5 * float(np.sum((P @ x - r) ** 2)) + 0.5 * float(np.sum((Q @ z - s) ** 2))
    results = admm(minx, minz, options)
    results["solverruntime"] = time.perf_counter() - t0
    return results


def quadraticprogram_bounded(P, q, r, lb, ub, options=None):
    """solvers/quadraticprogram.m:99-246, 'bounded' constraint branch (210-219)."""
    options = dict(options or {})
    t0 = time.perf_counter()
    n = P.shape[0]
    rho = float(options.get("rho", 1.0))
    args = dict(P=P, q=q, lb=lb, ub=ub, rho=rho, n=n, constraint="bounded")
    minx, minz, _ = getproxops("quadraticprogram", args)
    options.update(A=1, B=-1, c=0, m=n, nA=n, nB=n)
    options["obj"] = lambda x, z: 0.5 * float(x @ (P @ x)) + float(q @ x) + r  # 242
    results = admm(minx, minz, options)
    results["solverruntime"] = time.perf_counter() - t0
    return results


def quadraticprogram_standard(P, q, r, D, s, options=None):
    """solvers/quadraticprogram.m, 'standard' constraint branch (193-209)."""
    options = dict(options or {})
    t0 = time.perf_counter()
    n = P.shape[0]
    rho = float(options.get("rho", 1.0))
    args = dict(P=P, q=q, D=D, s=s, rho=rho, n=n, constraint="standard")
    minx, minz, _ = getproxops("quadraticprogram", args)
    options.update(A=1, B=-1, c=0, m=n, nA=n, nB=n)
    options["obj"] = lambda x, z: 0.5 * float(x @ (P @ x)) + float(q @ x) + r
    results = admm(minx, minz, options)
    results["solverruntime"] = time.perf_counter() - t0
    return results


def linearprogram(b, D, s, options=None):
    """solvers/linearprogram.m:81-185."""
    options = dict(options or {})
    t0 = time.perf_counter()
    n = D.shape[1]
    minx, minz, _ = getproxops("LinearProgram", dict(D=D, b=b, s=s, n=n))
    options.update(A=1, B=-1, c=0, m=n, nA=n, nB=n)
    options["obj"] = lambda x, z: float(b @ x)  # 180
    results = admm(minx, minz, options)
    results["solverruntime"] = time.perf_counter() - t0
    return results
