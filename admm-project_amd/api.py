"""The reference's plug-in surface on top of the HIP engine.

    [minx, minz, extra] = getproxops(problem, args)      (getProxOps.m:13)
    results             = admm(xminf, zming, options)    (admm.m:24)

``getproxops`` returns *descriptor* prox operators: objects that name an engine-native
x-/z-update and share one device-resident problem (data + cached factor).  When ``admm``
receives a matching pair it runs the whole loop (admm.m:496-743) on the device through
the C ABI.  ``options`` and ``results`` are dicts with the reference's field names.
Caller-supplied prox handles (the reference's plain function handles, as in
examples/convergencechecking.m, or the zming of unwrappedadmm) are supported next to a library
operator (A and B then belong to that problem: B = -1) or as a pair of their own with any A / At / B
the reference accepts (scalars, matrices, function handles; admm.m:113-245): they are called with
zero-copy CUDA tensors of the engine's state on the engine's HIP stream and must return
CUDA tensors; the rest of the iteration stays in the fused kernels.  There is no CPU
fallback: a handle that returns a host array is rejected loudly.
"""
from __future__ import annotations

import math

import numpy as np

from . import _lib as L
from .engine import Engine

__all__ = ["admm", "getproxops", "ProxOp"]

_PROBLEMS = ("model", "basispursuit", "totalvariation", "linearsvm", "lasso", "linearprogram",
             "quadraticprogram", "covarianceselection", "lad", "huberfit",
             "totalvariation2d")  # engine-side extension, no reference counterpart


class _Problem:
    """Shared state behind a (minx, minz) pair: what getproxops' closure captured."""

    def __init__(self, kind, engine, expect, extra=None, rebuild=None, rho=1.0):
        self.kind = kind
        self.engine = engine
        self.expect = expect  # constraint the solver must pass to admm: dict(A=..., c=..., nA, nB)
        self.extra = extra or {}
        # closures of the reference that re-derive their cached data when admm passes a different rho
        # (rhoprev logic: getProxOps.m:968-975, 1005-1008, 1441-1453; KKT systems 1363/1410 use the current rho):
        # rebuild(rho) returns a fresh engine for that rho
        self.rebuild = rebuild
        self.rho = float(rho)


class ProxOp:
    """Descriptor of an engine-native proximal operator (role 'x' or 'z')."""

    def __init__(self, problem, role):
        self.problem = problem
        self.role = role

    def __call__(self, *a, **k):
        raise NotImplementedError(
            "engine-native proximal operators run inside admm(); calling one on host arrays "
            "would need a CPU path, which this package does not have")

    def __repr__(self):
        return f"<ProxOp {self.problem.kind}:{self.role}>"


class EngineHook:
    """Marker for ``extra.altu`` / ``extra.specialnorms`` of consensus lasso (getProxOps.m:441-442):
    the solver stores them in ``options`` (lasso.m:222-223) and ``admm`` recognises that the
    engine implements them (altuLASSO 1312-1326, lassonorms 1335-1343) on the device."""

    def __init__(self, problem, name):
        self.problem = problem
        self.name = name

    def __call__(self, *a, **k):
        raise NotImplementedError("engine-native hook; it runs inside admm() on the device")


def _get(args, name):
    if name not in args:
        raise KeyError(f"args.{name} is required for this problem (getProxOps.m)")
    return args[name]


def getproxops(problem, args):
    """Prox-operator factory (getProxOps.m:13-917).  ``args`` uses the reference's field names.

    Engine-side extensions: the cached factor (``args.L``/``args.U``/``args.R``) and
    ``args.Dts`` may be omitted -- they are then built on the device (MFMA Gram + blocked
    Cholesky) instead of being passed in from host MATLAB code; ``args.xsolve`` selects how
    the factor is applied ('trsv' | 'inverse'); ``args.device`` the GPU ordinal.
    """
    if not isinstance(problem, str):
        raise TypeError("Given problem argument is not a string specifying for which problem "
                        "proximal operators are needed!")
    if not isinstance(args, dict):
        raise TypeError("Given struct args is not a struct containing arguments needed for "
                        "proximal operators for the given problem!")
    kind = problem.lower()
    if kind not in _PROBLEMS:
        raise ValueError("Invalid input for problem - given string is not a solver!")
    xs = {"auto": L.XSOLVE_AUTO, "trsv": L.XSOLVE_TRSV, "inverse": L.XSOLVE_INVERSE,
          "cg": L.XSOLVE_CG, "pinv": L.XSOLVE_PINV}[str(args.get("xsolve", "auto")).lower()]
    dev = int(args.get("device", 0))
    comm = args.get("comm")  # parallel.Comm: D/s/ell are then this rank's rows (see parallel.py)
    if comm is not None:
        dev = comm.device
    extra = {}
    cg = {k: args[k] for k in ("cg_tol", "cg_maxit") if k in args}  # xsolve='cg' (matrix-free) knobs

    if kind == "lasso":
        if args.get("parallel", 0):  # getProxOps.m:383-442: consensus over row slices
            D, s, lam = _get(args, "D"), _get(args, "s"), _get(args, "lambda")
            slices = [int(k) for k in np.atleast_1d(_get(args, "slices"))]
            rho = float(args.get("rho", 1.0))
            n = D.shape[1]
            eng = Engine(L.PROB_LASSO_CONSENSUS, D=D, s=s, lam=lam, rho=rho, xsolve=xs, device=dev, comm=comm,
                         slices=slices)
            prob = _Problem("lasso-consensus", eng, dict(A=1, c=0.0, nA=n, nB=n))
            extra = {"altu": EngineHook(prob, "altu"), "specialnorms": EngineHook(prob, "specialnorms")}
            return ProxOp(prob, "x"), ProxOp(prob, "z"), extra
        D = _get(args, "D")
        lam = _get(args, "lambda")
        rho = float(args.get("rho", 1.0))
        m, n = D.shape
        s = args.get("s")
        if s is None:
            raise KeyError("args.s is required (the engine forms Dts = D'*s itself)")
        Lf = args.get("L")
        if Lf is not None and hasattr(Lf, "toarray"):
            Lf = Lf.toarray()  # lasso.m:175 stores the factor sparse
        # args.objgram (engine-side extension): objective through the cached Gram matrix, see admm_engine.h obj_gram
        eng = Engine(L.PROB_LASSO, D=D, s=s, lam=lam, rho=rho, Lfactor=Lf, xsolve=xs, device=dev, comm=comm,
                     obj_gram=int(args.get("objgram", 0)), **cg)
        prob = _Problem("lasso", eng, dict(A=1, c=0.0, nA=n, nB=n))
    elif kind in ("lad", "huberfit"):
        D, s = _get(args, "D"), _get(args, "s")
        m, n = D.shape
        code = L.PROB_LAD if kind == "lad" else L.PROB_HUBERFIT
        eng = Engine(code, D=D, s=s, Lfactor=args.get("R"), userelax=int(bool(args.get("userelax", 0))),
                     xsolve=xs, device=dev, comm=comm, **cg)
        prob = _Problem(kind, eng, dict(A="D", c="s", nA=n, nB=m))
    elif kind == "linearsvm":
        D, ell, Cval = _get(args, "D"), _get(args, "ell"), _get(args, "C")
        loss = args.get("lossfunction", "hinge")
        m, n = D.shape
        # args.Dplus (linearsvm.m:185-186), when the caller supplies it, is applied literally: x = Dplus*(z-u)
        eng = Engine(L.PROB_LINEARSVM, D=D, ell=ell, Cval=Cval,
                     loss={"hinge": L.LOSS_HINGE, "01": L.LOSS_01}.get(loss, L.LOSS_HINGE_OBJ01), xsolve=xs,
                     device=dev, comm=comm, Dplus=args.get("Dplus"), **cg)
        prob = _Problem("linearsvm", eng, dict(A="D", c=0.0, nA=n, nB=m))
    elif kind == "linearprogram" or (kind == "quadraticprogram" and _get(args, "constraint") == "standard"):
        # getProxOps.m:1363 / 1410 solve [M D'; D 0] \ [rho*(z-u) - q; s] every iteration (M = rho*I for the
        # LP, P + rho*I for the QP).  The engine eliminates that KKT system ONCE, on the device (create():
        # build_kkt_map), to x = K*y + k0; the loop runs one n x n GEMV per x-update.
        D, s = _get(args, "D"), np.asarray(_get(args, "s"), dtype=np.float64).reshape(-1)
        D = np.asarray(D, dtype=np.float64)
        n = D.shape[1]
        rho = float(args.get("rho", 1.0))
        lp = kind == "linearprogram"
        q = np.asarray(_get(args, "b" if lp else "q"), dtype=np.float64).reshape(-1)
        P = None if lp else np.asarray(_get(args, "P"), dtype=np.float64)
        r0 = float(args.get("r", 0.0))

        def build(rho_):
            if lp:
                return Engine(L.PROB_LINEARPROGRAM, D=D, s=s, q=q, rho=rho_, device=dev)
            return Engine(L.PROB_QP_STANDARD, D=D, s=s, P=P, q=q, rho=rho_, r=r0, device=dev)

        prob = _Problem(kind, build(rho), dict(A=1, c=0.0, nA=n, nB=n), rebuild=build, rho=rho)
    elif kind == "quadraticprogram":
        P, q = _get(args, "P"), _get(args, "q")
        n = P.shape[0]
        lb, ub, r0 = _get(args, "lb"), _get(args, "ub"), float(args.get("r", 0.0))
        build = lambda rho_: Engine(L.PROB_QP_BOUNDED, P=P, q=q, lb=lb, ub=ub, rho=rho_, r=r0, xsolve=xs, device=dev)
        rho = float(_get(args, "rho"))
        prob = _Problem("quadraticprogram", build(rho), dict(A=1, c=0.0, nA=n, nB=n), rebuild=build, rho=rho)
    elif kind == "totalvariation":
        sig = np.asarray(_get(args, "s"), dtype=np.float64).reshape(-1)
        n = sig.size
        eng = Engine(L.PROB_TOTALVARIATION, s=sig, lam=float(_get(args, "lambda")), nvec=n, device=dev)
        prob = _Problem("totalvariation", eng, dict(A="D", c=0.0, nA=n, nB=n))
    elif kind == "totalvariation2d":
        img = np.asarray(_get(args, "s"), dtype=np.float64)
        if img.ndim != 2:
            raise ValueError("Argument s is not an image (2-D array)!")
        H, W = img.shape
        eng = Engine(L.PROB_TV2D, s=np.asfortranarray(img).reshape(-1, order="F"), lam=float(_get(args, "lambda")),
                     shape=(H, W), xsolve=xs, device=dev, **cg)
        prob = _Problem("totalvariation2d", eng, dict(A="D", c=0.0, nA=H * W, nB=2 * H * W))
    elif kind == "basispursuit":
        if "P" in args:  # getProxOps.m:137-138: the projector and offset computed by the caller (basispursuit.m:116-120)
            P, q = _get(args, "P"), _get(args, "q")
            n = P.shape[0]
            eng = Engine(L.PROB_BASISPURSUIT, P=P, q=q, device=dev)
        else:            # engine-side: the same two from D and s, on the device (create(): build_bp_projector)
            D = np.asarray(_get(args, "D"), dtype=np.float64)
            n = D.shape[1]
            eng = Engine(L.PROB_BASISPURSUIT, D=D, s=_get(args, "s"), device=dev)
        prob = _Problem("basispursuit", eng, dict(A=1, c=0.0, nA=n, nB=n))
    elif kind == "model":
        # getProxOps.m:83-89: the Gram data; engine-side extension: args.P/Q/r/s (the matrices
        # themselves) let the device evaluate model.m:133-134's objective when objevals is set
        PtP, Ptr = _get(args, "PtP"), _get(args, "Ptr")
        QtQ, Qts = _get(args, "QtQ"), _get(args, "Qts")
        n = int(args.get("n", np.asarray(PtP).shape[0]))
        objdata = {}
        if all(k in args for k in ("P", "Q", "r", "s")):
            objdata = dict(D=args["P"], s=args["r"], D2=args["Q"], s2=args["s"])
        build = lambda rho_: Engine(L.PROB_MODEL, P=PtP, q=Ptr, Q=QtQ, qz=Qts, nvec=n, rho=rho_, xsolve=xs, device=dev,
                                    **objdata)
        rho = float(args.get("rho", 1.0))
        prob = _Problem("model", build(rho), dict(A=1, c=0.0, nA=n, nB=n), rebuild=build, rho=rho)
    else:
        raise NotImplementedError(f"problem '{kind}' is outside the engine's hot-path scope (SURVEY.md section 8)")
    return ProxOp(prob, "x"), ProxOp(prob, "z"), extra


# ---------------------------------------------------------------------------------------
def _setopt(options, name, default):
    """admm.m:780-971.  q2: ``Hnormtol`` is read from ``Hreltol`` in the reference; accept both."""
    if name == "Hnormtol":
        if "Hreltol" in options:
            return options["Hreltol"]
        return options.get("Hnormtol", default)
    return options.get(name, default)


def _check_constraint(options, prob):
    """The solver passes A, B, c, m, nA, nB (e.g. lasso.m:232-238, lad.m:140-145); they must
    describe the same constraint the engine-native pair implements (B = -I always)."""
    if "A" not in options:
        raise ValueError("Must specify a matrix A in constraint Ax + Bz = c!")
    if "B" not in options:
        raise ValueError("Must specify a matrix B in constraint Ax + Bz = c!")
    B = options["B"]
    if prob.kind == "generic":
        return  # both handles are the caller's: A, B, c and the sizes were taken from options (_generic_problem)
    if not (np.isscalar(B) and float(B) == -1.0):
        raise ValueError("engine-native problems use B = -1 (z enters the constraint as -z)")
    exp = prob.expect
    A = options["A"]
    if exp["A"] == 1:
        if not (np.isscalar(A) and float(A) == 1.0):
            raise ValueError(f"{prob.kind}: constraint matrix A must be the scalar 1")
    else:
        if np.isscalar(A) or tuple(A.shape) != (exp["nB"], exp["nA"]):
            raise ValueError(f"{prob.kind}: constraint matrix A must be the data matrix D")
    c = options.get("c", None)
    if c is None:
        if int(options.get("m", 0)) <= 0:
            raise ValueError("Must specify a vector c in constraint Ax + Bz = c!")
    elif np.isscalar(c):
        if int(options.get("m", 0)) == 0:
            raise ValueError("Given vector c is scalar and no length m has been provided")
        if exp["c"] == "s" and float(c) != 0.0:
            raise ValueError("scalar non-zero c is not supported")
    for key in ("nA", "nB"):
        if key in options and int(options[key]) not in (0, exp[key]):
            raise ValueError(f"options.{key} does not match the problem size")


_CALLBACK_KINDS = ("model", "lasso", "quadraticprogram", "linearprogram", "basispursuit",  # A = 1 problems
                   "lad", "huberfit", "linearsvm")  # A = D problems


def _generic_problem(options):
    """Both prox operators are the caller's handles: the state, the u-update, residuals, histories and stop logic of
    admm.m:496-743 still run on the device.  A: the scalar 1 (examples/convergencechecking.m:110-115), another scalar,
    a matrix or a function handle with At (admm.m:113-195); B: the shorthand -1, another scalar, an m x nB matrix or a
    function handle with options.nB (admm.m:198-245)."""
    A, B = options.get("A"), options.get("B")
    if A is None:
        raise ValueError("Must specify a matrix A in constraint Ax + Bz = c!")
    if B is None:
        raise ValueError("Must specify a matrix B in constraint Ax + Bz = c!")
    if hasattr(A, "toarray"):
        A = A.toarray()  # sparse operators (totalvariation.m:127) are streamed as dense columns
    if hasattr(B, "toarray"):
        B = B.toarray()
    dev = int(options.get("device", 0))
    if callable(B) and int(options.get("nB", 0)) <= 0:  # admm.m:206-212
        raise ValueError("Matrix B is a function handle, but no number of columns nB specified for it; cannot infer nB "
                         "- please specify it in options struct!")
    if not (callable(B) or np.isscalar(B) or np.ndim(B) == 2):  # admm.m:217-222
        raise ValueError("Given B in constraint Ax + Bz = c is neither a numeric matrix nor function handle of single "
                         "vector!")

    def cvector(m):
        c = options.get("c", 0.0)
        if np.isscalar(c):
            if float(c) != 0.0:
                raise NotImplementedError("scalar non-zero c is not supported")
            return np.zeros(m)
        cvec = np.asarray(c, dtype=np.float64).reshape(-1)
        if cvec.size != m:
            raise ValueError("Given vector c does not match the problem size")
        return cvec

    if np.isscalar(A) and float(A) != 1.0:  # A = a*I as the pair of handles v -> a*v (admm.m:117-120)
        a = float(A)
        c = options.get("c", 0.0)
        mm = int(options.get("m", 0)) or (0 if np.isscalar(c) else np.size(c)) or int(options.get("nA", 0))
        A = lambda v: a * v  # noqa: E731
        options = dict(options, A=A, At=A, nA=int(options.get("nA", 0)) or mm, m=mm)
    if not np.isscalar(A) and not callable(A) and np.ndim(A) == 2:
        # a constraint matrix (admm.m:117-120: A(v) = A*v, At(v) = A'*v): the A = D engine of lad.m with no
        # factor at all -- both prox operators are the caller's; D*x, D'*(.), the residuals run on the device
        Am = np.asfortranarray(np.asarray(A, dtype=np.float64))
        m, n = Am.shape
        At = options.get("At")
        if At is not None and not np.isscalar(At) and not callable(At) and np.shape(At) != (n, m):
            raise ValueError("options.At is not the transpose of options.A")
        eng = Engine(L.PROB_LAD, D=Am, s=cvector(m), xsolve=L.XSOLVE_CALLBACK, device=dev)
        prob = _Problem("generic", eng, dict(A="D", c="s", nA=n, nB=m, m=m))
    elif callable(A):
        # options.A / options.At as function handles (admm.m:117-158): no matrix exists; A(x) and At(v) are device
        # callbacks like the prox operators, everything else of the loop stays in the fused kernels
        At = options.get("At")
        if not callable(At):
            raise ValueError("options.A is a function handle: options.At must be one too (admm.m:139-158)")
        nA = int(options.get("nA", 0))
        c = options.get("c", 0.0)
        m = int(options.get("m", 0)) or (0 if np.isscalar(c) else np.size(c))
        if not m and np.isscalar(B):
            m = int(options.get("nB", 0))
        if nA <= 0 or m <= 0:
            raise ValueError("function-handle operators need the sizes options.nA and options.m (or nB)")
        eng = Engine(L.PROB_LAD, s=cvector(m), shape=(m, nA), xsolve=L.XSOLVE_CALLBACK, device=dev)
        eng.set_operators(A, At)
        prob = _Problem("generic", eng, dict(A="D", c="s", nA=nA, nB=m, m=m))
    elif np.isscalar(A):
        c = options.get("c", 0.0)
        n = int(options.get("nA", 0) or options.get("m", 0) or (options.get("nB", 0) if np.isscalar(B) else 0))
        cvec = None
        if not np.isscalar(c):
            cvec = np.asarray(c, dtype=np.float64).reshape(-1)
            n = n or cvec.size
            if cvec.size != n:
                raise ValueError("Given vector c does not match the problem size")
        elif float(c) != 0.0:
            raise NotImplementedError("scalar non-zero c is not supported")
        if n <= 0:
            raise ValueError("Given vector c is scalar and no length m has been provided")
        eng = Engine(L.PROB_MODEL, nvec=n, c=cvec, rho=float(_setopt(options, "rho", 1.0)), device=dev)
        prob = _Problem("generic", eng, dict(A=1, c=0.0, nA=n, nB=n, m=n))
    else:
        raise ValueError("Given A in constraint Ax + Bz = c is neither a numeric matrix nor function handle of single "
                         "vector!")  # admm.m:131-134
    m = prob.expect["m"]
    try:
        if callable(B):  # admm.m:206-216
            nB = int(options["nB"])
            eng.set_constraint_b(B, nB)
            prob.expect.update(nB=nB, B="general")
        elif np.isscalar(B):
            if float(B) != -1.0:
                eng.set_constraint_b(float(B))
                prob.expect["B"] = "general"
        elif np.ndim(B) == 2:  # admm.m:202-204, 226-229
            if np.shape(B)[0] != m:
                raise ValueError("Number of rows in matrix B do not match length of column vector c in constraint "
                                 "Ax + Bz = c")
            eng.set_constraint_b(B)
            prob.expect.update(nB=int(np.shape(B)[1]), B="general")
    except BaseException:
        eng.close()
        raise
    return prob


_ADAPTIVE_KINDS = ("lasso", "lad", "huberfit", "totalvariation", "linearsvm", "quadraticprogram", "basispursuit",
                   "linearprogram", "generic")  # B = -1: ||B dz|| of the H-norm is ||dz||


def _admm_adaptive(xminf, zming, options, prob):
    """options.adaptive (experimental, admm.m:724-741): rho changes after every iteration i > 2, so the loop is
    stepped from the host -- one device iteration per step, warm-started from the previous (x, z, u), with the
    closures' "rho ~= rhoprev" re-factorisation (prob.rebuild) in between.  The H-norm keeps the rho it was
    created with (MATLAB anonymous functions capture by value, admm.m:305-309) while w = [x; z; rho*u] uses the
    current one (admm.m:678)."""
    if _setopt(options, "fast", 0):
        raise NotImplementedError("options.adaptive together with fast ADMM is not supported")
    if prob.kind not in _ADAPTIVE_KINDS or prob.expect.get("B") == "general":
        raise NotImplementedError(f"options.adaptive is not supported for the '{prob.kind}' operators"
                                  + (" with a general B" if prob.expect.get("B") == "general" else ""))
    rho = float(_setopt(options, "rho", 1.0))
    rho_H = rho
    N = _setopt(options, "maxiters", 1000)
    N = int(math.ceil(float(np.real(N)))) if N > 0 else 1000
    domaxiters = _setopt(options, "domaxiters", 0)
    nodualerror = _setopt(options, "nodualerror", 0)
    objevals = bool(_setopt(options, "objevals", 0))
    convtol = _setopt(options, "convtol", 1e-10)
    Hnormtol = _setopt(options, "Hnormtol", 1e-6)
    stopcond = _setopt(options, "stopcond", "standard")
    nA, nB = prob.expect["nA"], prob.expect["nB"]
    state = {k: (np.zeros(n) if options.get(k) is None else np.array(options[k], dtype=np.float64).reshape(-1))
             for k, n in (("x0", nA), ("z0", nB), ("u0", nB))}
    results = dict(state)
    results["Hnormtol"] = Hnormtol
    hist = {k: [] for k in ("xvals", "zvals", "uvals", "wvals", "pnorm", "dnorm", "perr", "derr", "objevals",
                            "Hnormsq")}
    x, z, u = state["x0"], state["z0"], state["u0"]
    w = np.concatenate([x, z, rho * u])
    runtime, step, failed, i = 0.0, None, 0, 0

    def pack():
        for k in ("xvals", "zvals", "uvals", "wvals"):
            results[k] = np.asfortranarray(np.stack(hist[k], axis=1))
        for k in ("pnorm", "dnorm", "perr", "derr", "Hnormsq") + (("objevals",) if objevals else ()):
            results[k] = np.array(hist[k])

    for i in range(1, N + 1):
        # closures without a "rho ~= rhoprev" branch keep their factor (xminLASSO, getProxOps.m:1192-1206)
        o = dict(options, adaptive=0, convtest=0, stopcond="standard", maxiters=1, domaxiters=1, quiet=1, rho=rho, preprocess=None,
                 x0=x, z0=z, u0=u, record_history=0, stale_factor_ok=int(prob.rebuild is None))
        step = admm(xminf, zming, o)
        runtime += step["runtime"]
        x, z, u = step["xopt"], step["zopt"], step["uopt"]
        for k, v in (("xvals", x), ("zvals", z), ("uvals", u)):
            hist[k].append(v)
        for k in ("pnorm", "dnorm", "perr", "derr"):
            hist[k].append(float(step[k][0]))
        if objevals:
            hist["objevals"].append(float(step["objevals"][0]))
        wprev, w = w, np.concatenate([x, z, rho * u])  # admm.m:677-682
        hist["wvals"].append(w)
        dw = wprev - w
        hist["Hnormsq"].append(rho_H * float(dw[nA:nA + nB] @ dw[nA:nA + nB]) + rho_H * float(dw[nA + nB:] @ dw[nA + nB:]))
        H1 = H2 = None
        if i >= 2:
            H2, H1 = hist["Hnormsq"][-1], hist["Hnormsq"][-2]
            if H1 > np.finfo(float).eps and H2 > H1 and not ((H2 - H1) <= H1 * convtol):  # q4: early return
                failed = i
                break
        if stopcond in ("standard", "both") and not domaxiters and hist["pnorm"][-1] < hist["perr"][-1] and (
                nodualerror or hist["dnorm"][-1] < hist["derr"][-1]):
            break
        if stopcond in ("hnorm", "both") and not domaxiters and i > 2 and hist["Hnormsq"][-1] <= Hnormtol:
            break
        if i > 2:  # admm.m:724-741 (wdiff is the scalar H1 - H2)
            growthtol = 5
            wdiff = np.float64(H1 - H2)  # MATLAB arithmetic: 0/0 is NaN, not an exception
            rhoprev = rho
            rho = rho * (wdiff * rhoprev) / (wdiff * wdiff)
            rhodiff = abs(rho - rhoprev)
            if rhodiff >= rhoprev * growthtol:
                rho = rho / growthtol
            elif rhodiff <= rhoprev / growthtol:
                rho = rho * growthtol
    pack()
    if failed:
        print(f"Iteration {failed}: H norms not converging to given relative tolerance: "
              f"{(H2 - H1) / (H1 + np.finfo(float).eps):g} is not less or equal to tol. {convtol:g}")
        print("ADMM seems to not be converging! Please check that your proximal operators are correct!")
        results["convtest_failed_at"] = failed
        return results
    results["steps"] = i
    results["xopt"], results["zopt"], results["uopt"] = x, z, u
    if objevals:
        results["objopt"] = float(step["objopt"])
    results["runtime"] = runtime
    results["rho_final"] = rho  # extension: the step size the next iteration would have used
    results["options"] = options
    return results


def admm(xminf, zming, options):
    """Run ADMM on the device (admm.m:24).  See the module docstring."""
    if not isinstance(options, dict):
        raise TypeError("Given options is not a struct! At least pass empty struct!")
    x_lib, z_lib = isinstance(xminf, ProxOp), isinstance(zming, ProxOp)
    for f, name in ((xminf, "xminf"), (zming, "zming")):
        if not (isinstance(f, ProxOp) or callable(f)):
            raise TypeError(f"Given {name} is not a function handle!")  # admm.m:33-45
    if x_lib and z_lib:
        if xminf.problem is not zming.problem or xminf.role != "x" or zming.role != "z":
            raise ValueError("xminf/zming must be the (minx, minz) pair returned by one getproxops call")
        prob = xminf.problem
    elif x_lib or z_lib:
        # one library operator + one caller-supplied handle (examples/convergencechecking.m:125-136)
        op = xminf if x_lib else zming
        if op.role != ("x" if x_lib else "z"):
            raise ValueError("a library proximal operator was passed in the wrong slot")
        prob = op.problem
        if prob.kind not in _CALLBACK_KINDS:
            raise NotImplementedError(f"caller-supplied prox handles cannot be mixed with the '{prob.kind}' operators "
                                      "(supported: " + ", ".join(_CALLBACK_KINDS) + ")")
    else:
        prob = _generic_problem(options)  # both handles are the caller's: admm.m:24 as is
    rho_run = float(_setopt(options, "rho", 1.0))
    if prob.rebuild is not None and rho_run != prob.rho and rho_run > 0:
        prob.engine.close()  # the closure's "rho ~= rhoprev" branch: new cached factor / KKT reduction
        prob.engine = prob.rebuild(rho_run)
        prob.rho = rho_run
    eng = prob.engine
    _check_constraint(options, prob)
    user_obj = options.get("obj") if callable(options.get("obj")) else None
    if user_obj is not None and prob.kind not in _CALLBACK_KINDS + ("generic",):
        user_obj = None  # the solver-supplied objective of this problem is engine-native (objevals switch)
    callbacks = dict(xmin=None if x_lib else xminf, zmin=None if z_lib else zming, obj=user_obj)
    use_callbacks = any(v is not None for v in callbacks.values())
    # options.altu / options.specialnorms (admm.m:553-559, 612-616): getproxops' own hooks (consensus lasso) are
    # engine-native; a caller's handle becomes a device callback (Engine.set_hooks)
    hooks = {}
    for hook in ("altu", "specialnorms"):
        h = options.get(hook)
        if h is None:
            continue
        if isinstance(h, EngineHook):
            if not (h.problem is prob and h.name == hook):
                raise ValueError(f"options.{hook} belongs to another getproxops call")
        elif callable(h):
            hooks[hook] = h
        elif hook == "altu":  # admm.m:538 tests isfield only: anything else would fail at the call (553-559)
            raise TypeError("options.altu is not a function handle")
        # (a non-handle options.specialnorms is ignored: admm.m:612-613 tests isa(..., 'function_handle'))
    if hooks and prob.kind == "lasso-consensus":
        raise NotImplementedError("consensus lasso runs with the hooks of its own getproxops call (lasso.m:222-223)")
    pre = options.get("preprocess")  # admm.m:473-476: called once, before the loop
    if pre is not None and not isinstance(pre, EngineHook):
        if not callable(pre):
            raise TypeError("options.preprocess is not a function handle")
        pre()
    if prob.kind == "lasso-consensus" and not ("altu" in options and "specialnorms" in options):
        raise ValueError("consensus lasso needs options.altu and options.specialnorms from getproxops' extra "
                         "(lasso.m:222-223)")
    if _setopt(options, "adaptive", 0) and _setopt(options, "convtest", 0):  # admm.m:724: only with convtest
        return _admm_adaptive(xminf, zming, options, prob)
    par = _setopt(options, "parallel", "none")
    if par in ("xminf", "zming", "both"):
        # admm.m:343-468 (parproxf / parproxg): the prox is evaluated slice by slice -- handle(x, z, u, rho, k)
        # returns piece k -- and the pieces are concatenated.
        from .errorcheck import slicemaker
        workers = int(options.get("workers", 1))
        if workers <= 0:
            raise ValueError("There are no workers on this machine, cannot perform parallel ADMM!")  # admm.m:350
        slices = options.get("slices", 0)
        is_cell = isinstance(slices, (list, tuple)) and len(slices) == 2 and all(
            isinstance(v, (list, tuple, np.ndarray)) for v in slices)
        if par == "both" and not is_cell:  # admm.m:357-361
            raise ValueError("For parallelizing both proximal ops, please:\n\tSpecify slices for f as a vector in "
                             "options.slices.\n\tSpecify slices for g as a vector in options.slices.")
        if par != "both" and is_cell:  # admm.m:362-364
            raise ValueError("Trying to parallelize both proximal operators, but options.slices is not a 2 element "
                             "cell!")
        sl = {"xminf": slices[0] if par == "both" else slices, "zming": slices[1] if par == "both" else slices}

        def sliced(fn, lengths, what):
            # the caller's handle for ONE slice (k is 0-based here; MATLAB's is 1-based); the device loop sees an
            # ordinary prox handle.  The reference's parfor is a plain loop on the engine's stream: the GPU is the
            # parallel resource.
            import torch

            def whole(x, z, u, rho_):
                pieces = []
                for k, ln in enumerate(lengths):
                    piece = fn(x, z, u, rho_, k)
                    if not isinstance(piece, torch.Tensor) or piece.numel() != ln:
                        raise ValueError(f"{what}: slice {k} must be a tensor of {ln} elements")
                    pieces.append(piece.reshape(-1))
                return torch.cat(pieces)

            return whole

        for name, is_lib, ln in (("xminf", x_lib, prob.expect["nA"]), ("zming", z_lib, prob.expect["nB"])):
            if par not in (name, "both"):
                continue
            lengths = slicemaker(sl[name], workers, ln)  # errorcheck.m:216-267
            if not is_lib:
                callbacks[name[:4]] = sliced(callbacks[name[:4]], [int(v) for v in lengths], name)
            elif not (prob.kind == "linearsvm" and name == "zming"):
                # For the separable z-prox of the linear SVM, slicing is the computation the fused kernel already
                # does over all rows at once (and the transpose reduction W = sum D_i'D_i of unwrappedadmm.m:96-141
                # is the engine's cached factor of D'D): the option only has to be validated.
                raise NotImplementedError("options.parallel on a library operator is engine-native for the linear "
                                          "SVM z-update only; shard rows with parallel.Comm instead")

    quiet = _setopt(options, "quiet", 1)
    rho = float(_setopt(options, "rho", 1.0))
    N = _setopt(options, "maxiters", 1000)
    N = int(math.ceil(float(np.real(N)))) if N > 0 else 1000  # admm.m:334-339
    fast = _setopt(options, "fast", 0)
    fasttype = _setopt(options, "fasttype", "weak")
    alg = L.FAST_OFF
    if fast:
        alg = L.FAST_WEAK if fasttype == "weak" else L.FAST_STRONG
    objevals = bool(_setopt(options, "objevals", 0))
    convtest = bool(_setopt(options, "convtest", 0))
    stopcond = _setopt(options, "stopcond", "standard")
    if stopcond not in ("standard", "hnorm", "both"):
        stopcond = "none"  # the reference's strcmp chain silently matches nothing
    record_history = bool(options.get("record_history", 1))
    nA, nB = prob.expect["nA"], prob.expect["nB"]
    mC = prob.expect.get("m", nB)  # length of u and c; differs from nB only under a general B (admm.m:252-254)

    x0 = options.get("x0")
    z0 = options.get("z0")
    u0 = options.get("u0")
    for name, v0, ln in (("x0", x0, nA), ("z0", z0, nB), ("u0", u0, mC)):
        if v0 is not None and np.asarray(v0).size != ln:
            raise ValueError(f"options.{name} has the wrong length")

    if use_callbacks:
        eng.set_callbacks(**callbacks)
    if hooks:
        eng.set_hooks(**hooks)
    try:
        summ = eng.run(rho=rho, maxiters=N, domaxiters=_setopt(options, "domaxiters", 0),
                       relax=_setopt(options, "relax", 1), fast=alg, objevals=objevals, convtest=convtest,
                       convtol=_setopt(options, "convtol", 1e-10),
                       stopcond=stopcond,
                       nodualerror=_setopt(options, "nodualerror", 0), abstol=_setopt(options, "abstol", 1e-5),
                       reltol=_setopt(options, "reltol", 1e-3), Hnormtol=_setopt(options, "Hnormtol", 1e-6),
                       restart=_setopt(options, "restart", 0.999), dvaltol=_setopt(options, "dvaltol", 1e-8),
                       record_history=record_history, check_every=int(options.get("check_every", 0)),
                       x0=x0, z0=z0, u0=u0, stale_factor_ok=options.get("stale_factor_ok", 0))
    finally:
        if use_callbacks:
            eng.set_callbacks()  # the library operators are the engine's default again
        if hooks:
            eng.set_hooks()
    steps = int(summ.steps)
    results = {}
    results["x0"] = np.zeros(nA) if x0 is None else np.array(x0, dtype=np.float64).reshape(-1)
    results["z0"] = np.zeros(nB) if z0 is None else np.array(z0, dtype=np.float64).reshape(-1)
    results["u0"] = np.zeros(mC) if u0 is None else np.array(u0, dtype=np.float64).reshape(-1)
    use_h = convtest or stopcond in ("hnorm", "both")
    if alg == L.FAST_WEAK:
        results["dvaltol"] = _setopt(options, "dvaltol", 1e-8)
    if use_h:
        results["Hnormtol"] = _setopt(options, "Hnormtol", 1e-6)

    if record_history:
        results["xvals"] = eng.fetch(L.F_XVALS, nA * steps, (nA, steps))
        results["zvals"] = eng.fetch(L.F_ZVALS, nB * steps, (nB, steps))
        results["uvals"] = eng.fetch(L.F_UVALS, mC * steps, (mC, steps))
        if alg != L.FAST_OFF:
            results["vvals"] = eng.fetch(L.F_VVALS, nB * steps, (nB, steps))
            results["uhatvals"] = eng.fetch(L.F_UHATVALS, mC * steps, (mC, steps))
    if alg != L.FAST_WEAK:  # q8: accelerated mode records no norms/tolerances (admm.m:619-640)
        for key, fld in (("pnorm", L.F_PNORM), ("dnorm", L.F_DNORM), ("perr", L.F_PERR), ("derr", L.F_DERR)):
            results[key] = eng.fetch(fld, steps)
    if alg != L.FAST_OFF:
        results["avals"] = eng.fetch(L.F_AVALS, steps)
        if alg == L.FAST_WEAK:
            results["dvals"] = eng.fetch(L.F_DVALS, steps)
            results["restarted"] = eng.fetch(L.F_RESTARTED, steps)
    if objevals:
        results["objevals"] = eng.fetch(L.F_OBJEVALS, steps)
    if use_h:
        results["Hnormsq"] = eng.fetch(L.F_HNORMSQ, steps)
        if record_history:  # admm.m:678-681  w = [x; z; rho*u], assembled behind the ABI (ADMM_F_WVALS)
            nw = nA + nB + mC
            results["wvals"] = eng.fetch(L.F_WVALS, nw * steps, (nw, steps))

    if not quiet and alg != L.FAST_WEAK:  # admm.m:318-330, 661-673
        hdr = ["Iteration", "Primal Residual Norm", "Primal Error", "Dual Residual Norm", "Dual Error"]
        if objevals:
            hdr.append("Objective Value")
        print("\t".join(f"{h:>20s}" if k else f"{h:>7s}" for k, h in enumerate(hdr)))
        for i in range(steps):
            row = (f"{i + 1:3d}\t{results['pnorm'][i]:10.4f}\t{results['perr'][i]:10.4f}\t"
                   f"{results['dnorm'][i]:10.4f}\t{results['derr'][i]:10.4f}")
            if objevals:
                row += f"\t{results['objevals'][i]:10.2f}"
            print(row)

    if summ.convtest_failed_at > 0:
        # q4 (admm.m:692-701): the reference prints a diagnostic and returns before steps/xopt/... exist
        i = int(summ.convtest_failed_at)
        H2, H1 = results["Hnormsq"][i - 1], results["Hnormsq"][i - 2]
        print(f"Iteration {i}: H norms not converging to given relative tolerance: "
              f"{(H2 - H1) / (H1 + np.finfo(float).eps):g} is not less or equal to tol. "
              f"{_setopt(options, 'convtol', 1e-10):g}")
        print("ADMM seems to not be converging! Please check that your proximal operators are correct!")
        results["convtest_failed_at"] = i
        return results

    results["steps"] = steps
    results["engine_info"] = eng.info()  # engine extension: which x-solve form runs, probe errors, rank (not a reference field)
    try:  # matrix-free x-update: total inner CG iterations (engine extension, not a reference field)
        cg = eng.fetch(L.F_CG_ITERS, 2)
        results["cg_iters_total"] = int(cg[0])
        results["cg_capped_updates"] = int(cg[1])  # x-updates that ended on cg_maxit above cg_tol (inexact iterates)
        if cg[1] > 0 and not _setopt(options, "quiet", 0):
            print(f"admm: {int(cg[1])} of {steps} x-updates ended on the CG iteration cap above cg_tol; raise cg_maxit")
    except L.AdmmError:
        pass
    if prob.kind == "lasso-consensus":
        # q9: the z admm holds is identically zero; the closure's consensus z is exposed as an extra field
        results["zconsensus"] = eng.fetch(L.F_ZCONSENSUS, nA)
    results["xopt"] = eng.fetch(L.F_XOPT, nA)
    results["zopt"] = eng.fetch(L.F_ZOPT, nB)
    results["uopt"] = eng.fetch(L.F_UOPT, mC)
    if objevals:
        results["objopt"] = float(summ.objopt)
    results["runtime"] = float(summ.runtime_s)
    if not quiet:
        print(f"Elapsed time is {results['runtime']:g} seconds.", end="")
        print(f"Number of steps to convergence: {steps:d}", end="")
    results["options"] = options
    return results
