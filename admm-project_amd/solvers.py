"""Problem solvers with the reference's signatures (/root/reference/solvers/*.m).

Each function is the host half of the corresponding MATLAB solver: validate arguments,
hand the data to ``getproxops`` (which uploads it and builds the cached factor on the
GPU), set the constraint fields of ``options`` exactly as the reference does, call
``admm`` and add ``solverruntime``.  The objective handles the reference installs in
``options.obj`` are realised by device kernels; ``options['obj']`` is therefore a marker.
"""
from __future__ import annotations

import time

import numpy as np

from .api import admm, getproxops
from .errorcheck import is_nonnegative_real, is_positive_real, slicemaker

__all__ = ["lasso", "lad", "huberfit", "linearsvm", "unwrappedadmm", "quadraticprogram", "basispursuit",
           "totalvariation", "totalvariation2d", "model", "linearprogram"]

_ENGINE_OBJ = "<engine-native objective>"


def _matrix(D, name):
    D = np.asarray(D, dtype=np.float64)
    if D.ndim != 2:
        raise ValueError(f"Argument {name} is not a matrix!")
    return D


def _colvec(s, name):
    s = np.asarray(s, dtype=np.float64)
    if s.ndim == 2 and 1 in s.shape:
        s = s.reshape(-1)
    if s.ndim != 1:
        raise ValueError(f"Argument {name} is not a vector!")
    return s


def _engine_args(options, args):
    for key in ("xsolve", "device", "comm", "cg_tol", "cg_maxit", "objgram"):
        if key in options:
            args[key] = options[key]
    return args


def lasso(D, s, lam, options=None):
    """results = lasso(D, s, lambda, options)   (solvers/lasso.m:77-245)

    minimise 1/2*||D*x - s||_2^2 + lambda*||x||_1.  Setup on the device: Dts = D'*s,
    L = chol(D'*D + rho*I) (tall) or chol(D*D'/rho + I) (fat) (lasso.m:160-176).
    """
    if not isinstance(options, dict):
        raise TypeError("Given options is not a struct! At least pass empty struct!")
    options = dict(options)
    t0 = time.perf_counter()
    lam = is_nonnegative_real(lam, "lambda")
    D = _matrix(D, "D")
    s = _colvec(s, "s")
    rho = is_positive_real(options["rho"], "options.rho") if "rho" in options else 1.0
    m, n = D.shape
    if s.size != m:
        raise ValueError("The number of rows in argument D do not match size of s!")
    if options.get("parallel", "none") in ("both", "zming", "xminf"):  # lasso.m:144-156, 193-224
        # consensus lasso: row slices each with their own x_k, u_k and cached factor.  `workers`
        # stands for gcp().NumWorkers (lasso.m:203-207); with options['comm'] D, s are this rank's rows
        # and the slices given here are this rank's local slices.
        options["parallel"] = "none"
        options["stopcond"] = "both"
        slices = np.atleast_1d(options.get("slices", 0))[0]  # lasso.m:197 takes only slices(1)
        workers = int(options.get("workers", 1 if "comm" in options else 8))
        args = _engine_args(options, dict(D=D, s=s, rho=rho, parallel=1, slices=slicemaker(slices, workers, m)))
        args["lambda"] = lam
        minx, minz, extra = getproxops("LASSO", args)
        options["altu"] = extra["altu"]  # lasso.m:222-223
        options["specialnorms"] = extra["specialnorms"]
    else:
        args = _engine_args(options, dict(D=D, s=s, m=m, n=n, parallel=0, rho=rho))
        args["lambda"] = lam
        minx, minz, _ = getproxops("LASSO", args)
    options["obj"] = _ENGINE_OBJ  # lasso.m:227  0.5*sum((D*x - s).^2) + lambda*norm(z,1)
    options.update(A=1, At=1, m=n, nA=n, nB=n, B=-1, c=0, parallel="none")  # lasso.m:232-239
    results = admm(minx, minz, options)
    results["solverruntime"] = time.perf_counter() - t0
    return results


def _lad_like(kind, D, s, options):
    if not isinstance(options, dict):
        raise TypeError("Argument options is not a struct!")
    options = dict(options)
    t0 = time.perf_counter()
    D = _matrix(D, "D")
    s = _colvec(s, "s")
    m, n = D.shape
    if s.size != m:
        raise ValueError("The number of rows in argument D do not match size of s!")
    args = _engine_args(options, dict(D=D, s=s))
    if options.get("relax", 1) != 1:  # lad.m:124-126, huberfit.m:156-158
        args["userelax"] = 1
    minx, minz, _ = getproxops(kind, args)
    options.update(A=D, B=-1, c=s, m=m, nA=n, nB=m)  # lad.m:140-145
    options["obj"] = _ENGINE_OBJ  # lad.m:148 norm(z,1) | huberfit.m:180 1/2*sum(huber(z))
    results = admm(minx, minz, options)
    results["solverruntime"] = time.perf_counter() - t0
    return results


def lad(D, s, options=None):
    """results = lad(D, s, options)   (solvers/lad.m:51-154): minimise ||D*x - s||_1."""
    return _lad_like("lad", D, s, options if options is not None else {})


def huberfit(D, s, options=None):
    """results = huberfit(D, s, options)   (solvers/huberfit.m:83-186): 1/2*sum(huber(D*x - s))."""
    return _lad_like("huberfit", D, s, options if options is not None else {})


def _single_column_quirk(D):
    """admm.m:151 takes nA from a matrix A only when it has more than one column, and unwrappedadmm.m:81-86 /
    linearsvm.m:221-227 pass no options.nA: with a single-column D the check of admm.m:188-190 (rows of A' against
    nA = 0) fails -- the reference cannot run these solvers on one feature, and neither does the mirror."""
    if D.shape[1] == 1 and D.shape[0] != 1:
        raise ValueError("Number of rows in At (A transpose) do not match number of columns in A, in constraint Ax + Bz = c")
    if D.shape == (1, 1):
        raise ValueError("Given scalar as matrix A with no number of columns nA specified in options struct; cannot infer nA "
                         "- please specify nA in options!")


def unwrappedadmm(zming, D, options=None):
    """results = unwrappedadmm(zming, D, options)   (solvers/unwrappedadmm.m:1-143)

    Transpose-reduction ADMM: minimise g(z) s.t. D*x = z with x = D^+ (z - u)
    (unwrappedadmm.m:76-78).  Forces maxiters=1000, stopcond='both', nodualerror=1
    (unwrappedadmm.m:90-92).  Deviation q14 (documented): explicit options.x0/z0/u0 win
    over the reference's unconditional ``rand`` so that runs are reproducible.
    """
    options = dict(options or {})
    D = _matrix(D, "D")
    m, n = D.shape
    _single_column_quirk(D)
    if options.get("parallel", "none") in ("xminf", "zming", "both"):
        # unwrappedadmm.m:45-74: the x-update becomes the transpose reduction (W = sum D_i'D_i, d = sum D_i'(z_i-u_i),
        # x = W\d: the engine's cached factor of D'D does exactly that) and admm slices the z-prox ('zming').
        options["slices"] = slicemaker(options.get("slices", 0), int(options.get("workers", 1)), m)
        options["parallel"] = "zming" if options["parallel"] == "both" else "none"
    from .api import ProxOp
    if isinstance(zming, ProxOp):
        prob = zming.problem
    elif callable(zming):
        # the reference's own use of this solver: the CALLER's z-prox handle (unwrappedadmm.m:1).  The x-update
        # x = D^+ (z - u) is the linear-SVM engine's (same closure: getProxOps.m:1062-1068 == unwrappedadmm.m:78);
        # the handle runs on device tensors between the two halves of the fused kernel.
        args = _engine_args(options, dict(D=D, Dt=None, ell=np.ones(m), C=1.0, lossfunction="hinge"))
        prob = getproxops("LinearSVM", args)[0].problem
    else:
        raise TypeError("Given zming is not a function handle!")
    xminf = ProxOp(prob, "x")
    options.update(A=D, At=D.T, B=-1, nB=m, c=0, m=m)
    rng = np.random.default_rng()
    for key, size in (("x0", n), ("z0", m), ("u0", m)):  # unwrappedadmm.m:87-89
        if key not in options:
            options[key] = rng.random(size)
    options["maxiters"] = 1000
    options["stopcond"] = "both"
    options["nodualerror"] = 1
    return admm(xminf, zming, options)


def linearsvm(D, ell, C, options=None):
    """results = linearsvm(D, ell, C, options)   (solvers/linearsvm.m:92-246)."""
    if not isinstance(options, dict):
        raise TypeError("Given options is not a struct! At least pass empty struct!")
    options = dict(options)
    t0 = time.perf_counter()
    if not np.isscalar(C) or np.real(C) < 0:  # linearsvm.m:270-274
        raise ValueError("Given regularization parameter C is not a nonnegative number!")
    C = float(np.real(C))
    ell = _colvec(ell, "ell")
    D = _matrix(D, "D")
    if D.shape[0] != ell.size:
        raise ValueError("Product ell*D is not possible; sizes incompatible!")
    _single_column_quirk(D)
    loss = options.get("lossfunction", "hinge")  # linearsvm.m:154-158
    args = _engine_args(options, dict(D=D, Dt=None, ell=ell, C=C, lossfunction=loss))
    if options.get("parallel", "none") in ("both", "zming", "xminf"):  # linearsvm.m:170-205
        options["parallel"] = "both"
        args["slices"] = options["slices"] = slicemaker(options.get("slices", 0), int(options.get("workers", 1)),
                                                         D.shape[0])
    _, minz, _ = getproxops("LinearSVM", args)
    options["obj"] = _ENGINE_OBJ  # linearsvm.m:231-237
    results = unwrappedadmm(minz, D, options)
    results["solverruntime"] = time.perf_counter() - t0
    return results


def quadraticprogram(P, q, r, cons1, cons2, options=None):
    """results = quadraticprogram(P, q, r, cons1, cons2, options)  (solvers/quadraticprogram.m:99-246)

    'bounded' form: cons1 = lb, cons2 = ub vectors.  'standard' form: one matrix D and one vector s
    (either order, quadraticprogram.m:321-329) for D*x = s, x >= 0.  ``options['altproxg']`` (a handle on
    device tensors) replaces the z-prox as in quadraticprogram.m:221-227.
    """
    if not isinstance(options, dict):
        raise TypeError("Argument options is not a struct!")
    options = dict(options)
    t0 = time.perf_counter()
    P = _matrix(P, "P")
    if P.shape[0] != P.shape[1]:
        raise ValueError("Argument P is not a square matrix!")
    q = _colvec(q, "q")
    if q.size != P.shape[0]:
        raise ValueError("The dimensions of square matrix P and vector q do not match!")
    c1, c2 = np.asarray(cons1, dtype=np.float64), np.asarray(cons2, dtype=np.float64)
    vec1, vec2 = (c1.ndim <= 1 or 1 in c1.shape), (c2.ndim <= 1 or 1 in c2.shape)
    if not vec1 and not vec2:
        raise ValueError("It appears that both constraint inputs are matrices! If trying to use standard "
                         "constraint form, only one can be a matrix.")  # quadraticprogram.m:361-363
    if not (vec1 and vec2):  # 'standard': D*x = s, x >= 0   (quadraticprogram.m:319-345)
        Dm, sv = (c2, c1) if vec1 else (c1, c2)
        sv = sv.reshape(-1)
        if Dm.shape[1] != P.shape[0]:
            raise ValueError("Number of columns in constraint matrix in standard form do not match lengths of "
                             "P and q!")  # quadraticprogram.m:348-349
        if Dm.shape[0] != sv.size:
            raise ValueError("Number of rows in constraint matrix in standard form does not match length of "
                             "constraint vector! (D and s in standard form)")  # quadraticprogram.m:351-353
        n = P.shape[0]
        rho = float(options.get("rho", 1.0))
        args = _engine_args(options, dict(P=P, q=q, D=Dm, s=sv, rho=rho, n=n, constraint="standard", r=float(r)))
        minx, minz, _ = getproxops("quadraticprogram", args)
        if callable(options.get("altproxg")):
            minz = options["altproxg"]
        options.update(A=1, B=-1, c=0, m=n, nA=n, nB=n)
        options["obj"] = _ENGINE_OBJ  # quadraticprogram.m:242
        results = admm(minx, minz, options)
        results["solverruntime"] = time.perf_counter() - t0
        return results
    lb, ub = c1.reshape(-1), c2.reshape(-1)
    if lb.size != ub.size:
        raise ValueError("Lengths of lower and upper bound constraints on solution x do not match!")
    if lb.size != P.shape[0]:
        raise ValueError("Bound vectors do not match predicted length of solution x!")
    if np.array_equal(np.maximum(lb, ub), lb):  # quadraticprogram.m:319-322 swap
        lb, ub = ub, lb
    elif not np.array_equal(np.maximum(lb, ub), ub):
        raise ValueError("Given constraint variables do not specify an upper and lower bound on solution x!")
    n = P.shape[0]
    rho = float(options.get("rho", 1.0))
    args = _engine_args(options, dict(P=P, q=q, lb=lb, ub=ub, rho=rho, n=n, constraint="bounded", r=float(r)))
    minx, minz, _ = getproxops("quadraticprogram", args)
    if callable(options.get("altproxg")):  # quadraticprogram.m:221-227
        minz = options["altproxg"]
    options.update(A=1, B=-1, c=0, m=n, nA=n, nB=n)
    options["obj"] = _ENGINE_OBJ  # quadraticprogram.m:242
    results = admm(minx, minz, options)
    results["solverruntime"] = time.perf_counter() - t0
    return results


def linearprogram(b, D, s, options=None):
    """results = linearprogram(b, D, s, options)   (solvers/linearprogram.m:81-185)

    minimise b'*x subject to D*x = s, x >= 0.  The reference solves the (n+m) x (n+m) KKT system in
    every x-update (getProxOps.m:1363); here it is reduced once on the host to x = K*y + k0 and the
    per-iteration n x n GEMV, the pos() prox, the u-update and the residuals run on the device.
    """
    if options is None:
        options = {}
    if not isinstance(options, dict):
        raise TypeError("Given options is not a struct! At least pass empty struct!")
    options = dict(options)
    t0 = time.perf_counter()
    D = _matrix(D, "D")
    b = _colvec(b, "b")
    s = _colvec(s, "s")
    m, n = D.shape
    if b.size != n:
        raise ValueError("Number of columns in D do not match length of vector b!")  # linearprogram.m:241-244
    if s.size != m:
        raise ValueError("Number of rows in D does not match length of vector s!")
    rho = is_positive_real(options["rho"], "options.rho") if "rho" in options else 1.0
    args = _engine_args(options, dict(D=D, Dt=D.T, b=b, s=s, n=n, rho=rho))
    minx, minz, _ = getproxops("LinearProgram", args)
    if callable(options.get("altproxg")):  # linearprogram.m:158-164
        minz = options["altproxg"]
    options.update(A=1, B=-1, c=0, m=n, nA=n, nB=n)  # linearprogram.m:172-177
    options["obj"] = _ENGINE_OBJ  # linearprogram.m:178  b'*x
    results = admm(minx, minz, options)
    results["solverruntime"] = time.perf_counter() - t0
    return results


def basispursuit(D, s, options=None):
    """results = basispursuit(D, s, options)   (solvers/basispursuit.m:52-145).

    The projector P = I - D'(DD')^-1 D and q = D'(DD')^-1 s (basispursuit.m:116-120) are formed once by the
    engine on the device (MFMA GEMMs + Cholesky + explicit m x m inverse); the per-iteration n x n GEMV and the
    shrinkage run on the device.
    """
    if not isinstance(options, dict):
        raise TypeError("Given options is not a struct! At least pass empty struct!")
    options = dict(options)
    t0 = time.perf_counter()
    D = _matrix(D, "D")
    s = _colvec(s, "s")
    mD, nD = D.shape
    if mD == nD and mD == s.size:
        raise ValueError("Square matrix problem Dx = s; don't need Basis Pursuit to solve this!")
    if mD > nD and mD == s.size:
        raise ValueError("Overdetermined system Dx = s; use Unwrapped ADMM solver instead.")
    if mD != s.size:
        raise ValueError("The number of rows in matrix D must match the number of rows in signal vector s!")
    n = nD
    minx, minz, _ = getproxops("BasisPursuit", _engine_args(options, dict(D=D, s=s)))
    options.update(A=1, B=-1, c=0, m=n, nA=n, nB=n)
    options["obj"] = _ENGINE_OBJ  # basispursuit.m:140 norm(x,1)
    results = admm(minx, minz, options)
    results["solverruntime"] = time.perf_counter() - t0
    return results


def totalvariation(s, lam, options=None):
    """results = totalvariation(s, lambda, options)   (solvers/totalvariation.m:62-167)

    minimise 1/2*||x - s||_2^2 + lambda*sum|x_{i+1} - x_i| (1-D signal).  The difference operator
    D = spdiags([1 -1], 0:1, n, n) (totalvariation.m:127) is never formed on the device: its
    stencils are fused into the kernels and (I + rho*D'D) is solved by parallel recurrences.
    """
    import scipy.sparse as sp

    if not isinstance(options, dict):
        raise TypeError("Given options is not a struct! At least pass empty struct!")
    options = dict(options)
    t0 = time.perf_counter()
    if not np.isscalar(lam) or np.real(lam) < 0:  # totalvariation.m:190-194
        raise ValueError("Given lambda parameter is not a nonnegative number!")
    lam = float(np.real(lam))
    s = _colvec(s, "s")
    n = s.size
    if n == 1:  # admm.m:145-148: totalvariation.m:151 hands admm a 1 x 1 matrix D and no options.nA
        raise ValueError("Given scalar as matrix A with no number of columns nA specified in options struct; cannot infer nA "
                         "- please specify nA in options!")
    args = _engine_args(options, dict(s=s))
    args["lambda"] = lam
    xmin, zmin, _ = getproxops("TotalVariation", args)
    D = sp.diags([np.ones(n), -np.ones(n - 1)], [0, 1], shape=(n, n), format="csr") if n <= (1 << 22) else \
        _ShapeOnly((n, n))
    options.update(A=D, At=None, B=-1, mB=n, nB=n, c=0, m=n)  # totalvariation.m:151-157
    options["obj"] = _ENGINE_OBJ  # totalvariation.m:134-135
    results = admm(xmin, zmin, options)
    results["solverruntime"] = time.perf_counter() - t0
    return results


def totalvariation2d(S, lam, options=None):
    """results = totalvariation2d(S, lambda, options) -- engine-side extension (the reference's
    totalvariation.m is 1-D): anisotropic TV denoising of an image,

        minimise 1/2*||X - S||_F^2 + lambda*(sum|X[i+1,j] - X[i,j]| + sum|X[i,j+1] - X[i,j]|),

    as ADMM on D*x - z = 0 with D = [Dv; Dh] (2N x N forward differences, x = X(:) column-major).  The
    x-update (I + rho*D'D) x = s + rho*D'(z - u) is solved spectrally where the height has a column transform (8 to
    4096 rows, or a power of two up to 8192): column DCT, a row stage (Toeplitz kernel, row DCT or exact tridiagonal
    solve, whichever applies to this width and rho), inverse column DCT -- and matrix-free by warm-started CG otherwise
    or with ``options['xsolve'] = 'cg'`` (``options['cg_tol']``, default 1e-11 relative; ``results['cg_capped_updates']``
    counts x-updates that ended on the step cap); nothing is factored.  ``xopt`` is returned as an image.
    """
    if options is None:
        options = {}
    if not isinstance(options, dict):
        raise TypeError("Given options is not a struct! At least pass empty struct!")
    options = dict(options)
    t0 = time.perf_counter()
    if not np.isscalar(lam) or np.real(lam) < 0:
        raise ValueError("Given lambda parameter is not a nonnegative number!")
    img = np.asarray(S, dtype=np.float64)
    if img.ndim != 2:
        raise ValueError("Argument S is not an image (2-D array)!")
    H, W = img.shape
    N = H * W
    args = _engine_args(options, dict(s=img))
    args["lambda"] = float(np.real(lam))
    xmin, zmin, _ = getproxops("TotalVariation2D", args)
    options.update(A=_ShapeOnly((2 * N, N)), At=None, B=-1, nA=N, nB=2 * N, c=0, m=2 * N)
    options["obj"] = _ENGINE_OBJ
    results = admm(xmin, zmin, options)
    results["xopt"] = results["xopt"].reshape((H, W), order="F")
    results["solverruntime"] = time.perf_counter() - t0
    return results


class _ShapeOnly:
    """Stand-in for options.A when materialising the sparse operator would only cost memory."""

    def __init__(self, shape):
        self.shape = shape


def model(P, Q, r, s, options=None):
    """results = model(P, Q, r, s, options)   (solvers/model.m:47-143)

    minimise 1/2*||P*x - r||_2^2 + 1/2*||Q*z - s||_2^2 subject to x - z = 0.  The host forms the
    Gram data exactly as model.m:111-121 does and passes it to ``getproxops('model', args)``; the
    matrices themselves travel along (engine-side extension) so that the device can evaluate the
    objective of model.m:133-134 when ``objevals`` is set.
    """
    if options is None:
        options = {}
    if not isinstance(options, dict):
        raise TypeError("Given options is not a struct! At least pass empty struct!")
    options = dict(options)
    t0 = time.perf_counter()
    P = _matrix(P, "P")
    Q = _matrix(Q, "Q")
    r = _colvec(r, "r")
    s = _colvec(s, "s")
    if P.shape[0] != Q.shape[0]:  # model.m:204-211
        raise ValueError("Number of rows in P do not match number of rows in Q!")
    if P.shape[1] != Q.shape[1]:
        raise ValueError("Number of columns in P do not match number of columns in Q!")
    if P.shape[0] != r.size:
        raise ValueError("Number of rows in P does not match length of vector r!")
    if Q.shape[0] != s.size:
        raise ValueError("Number of rows in Q does not match length of vector s!")
    n = P.shape[1]
    rho = is_positive_real(options["rho"], "options.rho") if "rho" in options else 1.0
    args = dict(PtP=P.T @ P, Ptr=P.T @ r, QtQ=Q.T @ Q, Qts=Q.T @ s, n=n, rho=rho, P=P, Q=Q, r=r, s=s)
    minx, minz, _ = getproxops("Model", _engine_args(options, args))
    options.update(A=1, B=-1, c=0, m=n, nA=n, nB=n)  # model.m:124-130
    options["obj"] = _ENGINE_OBJ
    results = admm(minx, minz, options)
    results["solverruntime"] = time.perf_counter() - t0
    return results
