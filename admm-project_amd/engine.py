"""Engine: Python owner of one ``admm_engine`` handle (C ABI, include/admm_engine.h)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L

_STOPCOND = {"standard": L.STOP_STANDARD, "hnorm": L.STOP_HNORM, "both": L.STOP_BOTH, "none": L.STOP_NONE}


def _f64(a, order="F"):
    return np.require(np.asarray(a, dtype=np.float64), dtype=np.float64, requirements=[order, "A"])


class Engine:
    """Device-resident problem (data + cached factor) and its iteration loop."""

    def __init__(self, problem, *, D=None, s=None, ell=None, P=None, q=None, lb=None, ub=None, Lfactor=None,
                 lam=0.0, Cval=0.0, r=0.0, rho=1.0, loss=L.LOSS_HINGE, userelax=0, xsolve=L.XSOLVE_AUTO,
                 device=0, slices=None, comm=None, nvec=None, cg_tol=None, cg_maxit=None,
                 Q=None, qz=None, D2=None, s2=None, c=None, K=None, k0=None, shape=None, obj_gram=0,
                 Dplus=None, Dts=None, _defer=None):
        lib = L.load()
        L.require_device()
        d = L.ProblemDesc()
        lib.admm_problem_desc_default(C.byref(d))
        d.problem = problem
        keep = []  # arrays must outlive the create call

        def vec(a):
            a = _f64(a).reshape(-1)
            keep.append(a)
            return L.as_dp(a)

        if D is not None:
            Dm = _f64(D)
            if Dm.ndim != 2:
                raise ValueError("D must be a matrix")
            keep.append(Dm)
            d.m, d.n = Dm.shape
            d.D = L.as_dp(Dm)
            d.ldD = Dm.shape[0]
        if P is not None:
            Pm = _f64(P)
            keep.append(Pm)
            d.n = Pm.shape[0]
            if D is None:
                d.m = Pm.shape[0]
            d.P = L.as_dp(Pm)
        if nvec is not None:  # problems without a data matrix (total variation, generic admm): vector length
            if D is None:
                d.m = int(nvec)
            d.n = int(nvec)
        if shape is not None:  # 2-D total variation: the image is shape[0] x shape[1], column-major in s
            d.m, d.n = int(shape[0]), int(shape[1])  # (also the m x n of an operator-form engine without a matrix)
        if Q is not None:  # model problem: QtQ, Qts and the optional objective data (getProxOps.m:83-89)
            Qm = _f64(Q)
            keep.append(Qm)
            d.Q = L.as_dp(Qm)
        if qz is not None:
            d.qz = vec(qz)
        if D2 is not None:
            D2m = _f64(D2)
            keep.append(D2m)
            d.D2 = L.as_dp(D2m)
            d.m2, d.ldD2 = D2m.shape[0], D2m.shape[0]
        if s2 is not None:
            d.s2 = vec(s2)
        if c is not None:
            d.c = vec(c)
        if K is not None:  # LP / standard-form QP: reduced KKT map x = K*y + k0
            Km = _f64(K)
            keep.append(Km)
            d.K = L.as_dp(Km)
            d.n = Km.shape[0]
            if D is None:
                d.m = Km.shape[0]
            d.k0 = vec(k0)
        if s is not None:
            d.s = vec(s)
        if ell is not None:
            d.ell = vec(ell)
        if q is not None:
            d.q = vec(q)
        if lb is not None:
            d.lb = vec(lb)
        if ub is not None:
            d.ub = vec(ub)
        if Dplus is not None:  # args.Dplus = pinv(D) handed in by the caller (linearsvm.m:185-186)
            Dp = _f64(Dplus)
            if D is None or Dp.shape != (d.n, d.m):
                raise ValueError("Dplus must be n x m for D m x n")
            keep.append(Dp)
            d.Dplus = L.as_dp(Dp)
        if Dts is not None:  # args.Dts (lasso.m:160, 182)
            d.Dts = vec(Dts)
        if Lfactor is not None:
            Lm = _f64(Lfactor)
            keep.append(Lm)
            d.L = L.as_dp(Lm)
        d.lambda_ = float(lam)
        d.C = float(Cval)
        d.r = float(r)
        d.rho = float(rho)
        d.loss = int(loss)
        d.userelax = int(userelax)
        d.xsolve = int(xsolve)
        d.mem = L.MEM_HOST
        d.device = int(device)
        if cg_tol is not None:
            d.cg_tol = float(cg_tol)
        if cg_maxit is not None:
            d.cg_maxit = int(cg_maxit)
        d.obj_gram = int(obj_gram)  # 0 automatic, 1 Gram form, -1 literal form (admm_engine.h)
        if slices is not None:
            sl = np.ascontiguousarray(np.asarray(slices, dtype=np.int64))
            keep.append(sl)
            d.nslices = sl.size
            d.slices = sl.ctypes.data_as(C.POINTER(C.c_int64))
        if comm is not None:
            d.comm = comm.handle
        self._comm = comm
        self._lib = lib
        if _defer is not None:  # Engine.create_all: the native call creates every rank's engine at once
            _defer.append((self, d, keep))
            return
        h = C.c_void_p()
        L.check(lib.admm_engine_create(C.byref(d), C.byref(h)))
        self._attach(h, d)
        del keep

    def _attach(self, h, d):
        problem = int(d.problem)
        self._h = h
        self.problem = problem
        self.m, self.n = int(d.m), int(d.n)
        # lengths of x and of z, u (admm.m: nA, nB): the A = D problems constrain D*x - z = c
        self.nA = self.n
        self.nB = self.m if problem in (L.PROB_LAD, L.PROB_HUBERFIT, L.PROB_LINEARSVM) else self.n
        if problem == L.PROB_TV2D:
            self.nA, self.nB = self.m * self.n, 2 * self.m * self.n
        self.mC = self.nB  # length of u, c and A*x; equals nB unless a general B was set (set_constraint_b)
        self.device = int(d.device)
        self._cb_keep = None
        self._cb_error = None
        self._relax = 1.0

    # ------------------------------------------------------------------ one process, several ranks
    @classmethod
    def create_all(cls, problem, per_rank_kwargs):
        """One engine per rank of a ``parallel.LocalGroup`` in THIS process (admm_engine_create_all): each dict holds
        that rank's rows and its ``comm``.  The native call runs the blocking per-rank creates on threads of its own."""
        pending = []
        engines = [cls(problem, _defer=pending, **kw) for kw in per_rank_kwargs]
        n = len(engines)
        descs = (L.ProblemDesc * n)(*[d for _, d, _ in pending])
        handles = (C.c_void_p * n)()
        L.check(L.load().admm_engine_create_all(n, descs, handles))
        for (eng, d, _), h in zip(pending, handles):
            eng._attach(C.c_void_p(h), d)
        return engines

    @staticmethod
    def run_all(engines, **kw):
        """The same run on every rank's engine (admm_engine_run_all); x0 / z0 / u0 may be per-rank lists."""
        n = len(engines)
        opts = (L.Options * n)()
        keep = []
        for r, eng in enumerate(engines):
            kr = {k: (v[r] if k in ("x0", "z0", "u0") and isinstance(v, (list, tuple)) else v) for k, v in kw.items()}
            opts[r] = eng._options(keep, **kr)
        handles = (C.c_void_p * n)(*[e._h for e in engines])
        sums = (L.RunSummary * n)()
        L.check(L.load().admm_engine_run_all(n, handles, opts, 1, sums))
        for r, eng in enumerate(engines):
            eng.last = sums[r]
        return list(sums)

    # ------------------------------------------------------------------ caller-supplied prox operators
    def set_callbacks(self, xmin=None, zmin=None, obj=None):
        """Replace the x-/z-update (and the objective hook) by Python callables working on DEVICE
        tensors: ``xmin(x, z, u, rho)``, ``zmin(xh, z, u, rho)`` return a float64 CUDA tensor,
        ``obj(x, z)`` a scalar (tensor or float).  The arguments are zero-copy torch views of the
        engine's state; the calls run on the engine's HIP stream (admm_engine.h: admm_prox_callback).
        ``None`` keeps / restores the engine-native operator."""
        if xmin is None and zmin is None and obj is None:
            L.check(self._lib.admm_engine_set_callbacks(self._h, C.cast(None, L.PROX_CALLBACK), None,
                                                        C.cast(None, L.PROX_CALLBACK), None,
                                                        C.cast(None, L.OBJ_CALLBACK), None))
            self._cb_keep = None
            return
        import torch  # device memory / stream plumbing only

        dev = torch.device("cuda", self.device)
        nA = self.nA

        class _View:  # __cuda_array_interface__ carrier: torch.as_tensor makes a zero-copy tensor of it
            def __init__(self, ptr, count):
                self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False),
                                                 "version": 2}

        def view(ptr, count):
            return torch.as_tensor(_View(ptr, count), device=dev)

        def as_result(val, count, what):
            if not (isinstance(val, torch.Tensor) and val.is_cuda):
                raise TypeError(f"{what} must return a CUDA tensor: host arrays would need a CPU path, "
                                "which this package does not have")
            if val.numel() != count:
                raise ValueError(f"{what} returned {val.numel()} elements, expected {count}")
            return val.to(torch.float64).reshape(-1)

        def wrap_prox(fn, what):
            if fn is None:
                return C.cast(None, L.PROX_CALLBACK)

            def cb(_user, x, z, u, rho, out, nout, stream):
                try:
                    with torch.cuda.stream(torch.cuda.ExternalStream(int(stream), device=dev)):
                        # xminf gets x; zming gets x, or the relaxed Axhat (nB elements) when relax != 1 (admm.m:521-530)
                        first = nA if (what == "xminf" or self._relax == 1.0) else self.mC
                        res = fn(view(x, first), view(z, self.nB), view(u, self.mC), float(rho))
                        view(out, nout).copy_(as_result(res, nout, what))
                    return 0
                except BaseException as exc:  # noqa: BLE001 - must not propagate through the C frame
                    self._cb_error = exc
                    return 1

            return L.PROX_CALLBACK(cb)

        def wrap_obj(fn):
            if fn is None:
                return C.cast(None, L.OBJ_CALLBACK)

            def cb(_user, x, nA, z, nB, out, stream):
                try:
                    with torch.cuda.stream(torch.cuda.ExternalStream(int(stream), device=dev)):
                        val = fn(view(x, nA), view(z, nB))
                        if not isinstance(val, torch.Tensor):
                            val = torch.tensor(float(val), dtype=torch.float64, device=dev)
                        view(out, 1).copy_(val.to(device=dev, dtype=torch.float64).reshape(1))
                    return 0
                except BaseException as exc:  # noqa: BLE001
                    self._cb_error = exc
                    return 1

            return L.OBJ_CALLBACK(cb)

        keep = (wrap_prox(xmin, "xminf"), wrap_prox(zmin, "zming"), wrap_obj(obj))
        L.check(self._lib.admm_engine_set_callbacks(self._h, keep[0], None, keep[1], None, keep[2], None))
        self._cb_keep = keep  # the C side holds raw pointers to these thunks

    def set_hooks(self, altu=None, specialnorms=None):
        """options.altu / options.specialnorms as the CALLER's handles (admm.m:553-559, 612-616), callables on DEVICE
        tensors like the prox callbacks: ``altu(u, Ax, Bz, c) -> new u`` (m elements; Ax is the relaxed Axhat when
        relax != 1, c a zero vector when the constraint has none), ``specialnorms(x, z, u, rho) -> (pnorm, dnorm)``
        (a 2-element tensor or two scalars).  ``None, None`` restores the engine's own u-update and norms."""
        if altu is None and specialnorms is None:
            L.check(self._lib.admm_engine_set_hooks(self._h, C.cast(None, L.ALTU_CALLBACK), None,
                                                    C.cast(None, L.NORMS_CALLBACK), None))
            self._hook_keep = None
            return
        import torch  # device memory / stream plumbing only

        dev = torch.device("cuda", self.device)

        class _View:
            def __init__(self, ptr, count):
                self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False),
                                                 "version": 2}

        def view(ptr, count):
            return torch.as_tensor(_View(ptr, count), device=dev)

        def wrap_altu(fn):
            if fn is None:
                return C.cast(None, L.ALTU_CALLBACK)

            def cb(_user, u, ax, bz, c, m, out, stream):
                try:
                    with torch.cuda.stream(torch.cuda.ExternalStream(int(stream), device=dev)):
                        res = fn(view(u, m), view(ax, m), view(bz, m), view(c, m))
                        if not (isinstance(res, torch.Tensor) and res.is_cuda):
                            raise TypeError("options.altu must return a CUDA tensor: host arrays would need a CPU path, "
                                            "which this package does not have")
                        if res.numel() != m:
                            raise ValueError(f"options.altu returned {res.numel()} elements, expected {m}")
                        view(out, m).copy_(res.to(torch.float64).reshape(-1))
                    return 0
                except BaseException as exc:  # noqa: BLE001 - must not propagate through the C frame
                    self._cb_error = exc
                    return 1

            return L.ALTU_CALLBACK(cb)

        def wrap_norms(fn):
            if fn is None:
                return C.cast(None, L.NORMS_CALLBACK)

            def cb(_user, x, nA, z, nB, u, m, rho, out2, stream):
                try:
                    with torch.cuda.stream(torch.cuda.ExternalStream(int(stream), device=dev)):
                        res = fn(view(x, nA), view(z, nB), view(u, m), float(rho))
                        if not isinstance(res, torch.Tensor):
                            res = torch.stack([r if isinstance(r, torch.Tensor) else
                                               torch.tensor(float(r), dtype=torch.float64, device=dev) for r in res])
                        if res.numel() < 2:
                            raise ValueError("options.specialnorms must return two values (admm.m:613-616)")
                        view(out2, 2).copy_(res.to(device=dev, dtype=torch.float64).reshape(-1)[:2])
                    return 0
                except BaseException as exc:  # noqa: BLE001
                    self._cb_error = exc
                    return 1

            return L.NORMS_CALLBACK(cb)

        keep = (wrap_altu(altu), wrap_norms(specialnorms))
        L.check(self._lib.admm_engine_set_hooks(self._h, keep[0], None, keep[1], None))
        self._hook_keep = keep

    def set_operators(self, A, At):
        """options.A / options.At as function handles (admm.m:117-158) for an engine created without a matrix:
        callables on DEVICE tensors, ``A(x) -> nB elements``, ``At(v) -> nA elements`` (zero-copy torch views, the
        engine's HIP stream), exactly like the prox callbacks."""
        import torch  # device memory / stream plumbing only

        dev = torch.device("cuda", self.device)

        class _View:
            def __init__(self, ptr, count):
                self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False),
                                                 "version": 2}

        def wrap(fn, what):
            def cb(_user, vin, nin, out, nout, stream):
                try:
                    with torch.cuda.stream(torch.cuda.ExternalStream(int(stream), device=dev)):
                        res = fn(torch.as_tensor(_View(vin, nin), device=dev))
                        if not (isinstance(res, torch.Tensor) and res.is_cuda):
                            raise TypeError(f"options.{what} must return a CUDA tensor: host arrays would need a CPU "
                                            "path, which this package does not have")
                        if res.numel() != nout:
                            raise ValueError(f"options.{what} returned {res.numel()} elements, expected {nout}")
                        torch.as_tensor(_View(out, nout), device=dev).copy_(res.to(torch.float64).reshape(-1))
                    return 0
                except BaseException as exc:  # noqa: BLE001 - must not propagate through the C frame
                    self._cb_error = exc
                    return 1

            return L.OPERATOR_CALLBACK(cb)

        keep = (wrap(A, "A"), wrap(At, "At"))
        L.check(self._lib.admm_engine_set_operators(self._h, keep[0], None, keep[1], None))
        self._op_keep = keep

    def set_constraint_b(self, B, nB=0):
        """options.B other than -1 (admm.m:198-245) for an engine whose two prox operators are the caller's: a scalar,
        an m x nB matrix (NumPy, host) or a callable on DEVICE tensors ``B(z) -> m elements`` (then ``nB`` is
        required).  Afterwards z has nB elements; u, c and A*x keep m."""
        none = C.cast(None, L.OPERATOR_CALLBACK)
        if callable(B):
            import torch  # device memory / stream plumbing only

            dev = torch.device("cuda", self.device)

            class _View:
                def __init__(self, ptr, count):
                    self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8",
                                                     "data": (int(ptr), False), "version": 2}

            def cb(_user, vin, nin, out, nout, stream):
                try:
                    with torch.cuda.stream(torch.cuda.ExternalStream(int(stream), device=dev)):
                        res = B(torch.as_tensor(_View(vin, nin), device=dev))
                        if not (isinstance(res, torch.Tensor) and res.is_cuda):
                            raise TypeError("options.B must return a CUDA tensor: host arrays would need a CPU path, "
                                            "which this package does not have")
                        if res.numel() != nout:
                            raise ValueError(f"options.B returned {res.numel()} elements, expected {nout}")
                        torch.as_tensor(_View(out, nout), device=dev).copy_(res.to(torch.float64).reshape(-1))
                    return 0
                except BaseException as exc:  # noqa: BLE001 - must not propagate through the C frame
                    self._cb_error = exc
                    return 1

            self._b_keep = L.OPERATOR_CALLBACK(cb)
            L.check(self._lib.admm_engine_set_constraint_b(self._h, None, 0, int(nB), L.MEM_HOST, 0.0, self._b_keep,
                                                           None))
            self.nB = int(nB)
        elif np.isscalar(B):
            L.check(self._lib.admm_engine_set_constraint_b(self._h, None, 0, 0, L.MEM_HOST, float(B), none, None))
        else:
            Bm = np.asfortranarray(np.asarray(B, dtype=np.float64))
            L.check(self._lib.admm_engine_set_constraint_b(self._h, L.as_dp(Bm), Bm.shape[0], Bm.shape[1], L.MEM_HOST,
                                                           0.0, none, None))
            self.nB = int(Bm.shape[1])

    # ------------------------------------------------------------------ lifecycle
    def close(self):
        if getattr(self, "_h", None):
            self._lib.admm_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def setup_seconds(self):
        v = C.c_double(0)
        L.check(self._lib.admm_engine_setup_seconds(self._h, C.byref(v)))
        return v.value

    def info(self):
        """What create() decided about the x-update factor (admm_engine_info): the form in use, the probe errors
        of both forms, the condition estimate, and rank / sweeps when the pseudo-inverse path ran."""
        i = L.EngineInfo()
        i.struct_size = C.sizeof(L.EngineInfo)
        L.check(self._lib.admm_engine_info(self._h, C.byref(i)))
        names = {L.XSOLVE_AUTO: "auto", L.XSOLVE_TRSV: "trsv", L.XSOLVE_INVERSE: "inverse", L.XSOLVE_CG: "cg",
                 L.XSOLVE_CALLBACK: "callback", L.XSOLVE_PINV: "pinv"}
        return dict(xsolve_requested=names.get(i.xsolve_requested, i.xsolve_requested),
                    xsolve_used=names.get(i.xsolve_used, i.xsolve_used), pinv_used=bool(i.pinv_used),
                    probed=bool(i.probed), unwrapped_fused=bool(i.unwrapped_fused), trsv_blocks=i.trsv_blocks, jacobi_sweeps=i.jacobi_sweeps,
                    factor_n=i.factor_n, rank=i.rank, cond_estimate=i.cond_estimate,
                    probe_err_inverse=i.probe_err_inverse, probe_err_trsv=i.probe_err_trsv, probe_err_trsv_one=i.probe_err_trsv_one, probe_diff=i.probe_diff,
                    xsolve_cacheable_bytes=i.xsolve_cacheable_bytes, xsolve_stream_bytes=i.xsolve_stream_bytes,
                    obj_bound_max=i.obj_bound_max, obj_form_literal=bool(i.obj_form_literal))

    def set_profiling(self, on, stride=1):
        """True/False, or an iterable of kernel classes (L.K_XSOLVE, ...) to time with HIP events; stride > 1 times
        only every stride-th launch of a class (sampling)."""
        L.check(self._lib.admm_engine_set_profiling_stride(self._h, int(stride)))
        if on is True:
            mask = -1
        elif not on:
            mask = 0
        else:
            mask = 0
            for k in on:
                mask |= 1 << int(k)
        L.check(self._lib.admm_engine_set_profiling(self._h, mask))

    def kernel_time(self, which):
        ms, cnt = C.c_double(0), C.c_int64(0)
        L.check(self._lib.admm_engine_kernel_time(self._h, which, C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    # ------------------------------------------------------------------ run / fetch
    def run(self, *, rho=1.0, maxiters=1000, domaxiters=0, relax=1.0, fast=L.FAST_OFF, objevals=0, convtest=0,
            convtol=1e-10, stopcond="standard", nodualerror=0, abstol=1e-5, reltol=1e-3, Hnormtol=1e-6,
            restart=0.999, dvaltol=1e-8, record_history=1, check_every=0, x0=None, z0=None, u0=None,
            stale_factor_ok=0):
        keep = []
        o = self._options(keep, rho=rho, maxiters=maxiters, domaxiters=domaxiters, relax=relax, fast=fast,
                          objevals=objevals, convtest=convtest, convtol=convtol, stopcond=stopcond,
                          nodualerror=nodualerror, abstol=abstol, reltol=reltol, Hnormtol=Hnormtol, restart=restart,
                          dvaltol=dvaltol, record_history=record_history, check_every=check_every, x0=x0, z0=z0, u0=u0,
                          stale_factor_ok=stale_factor_ok)
        s = L.RunSummary()
        self._cb_error = None
        rc = self._lib.admm_engine_run(self._h, C.byref(o), C.byref(s))
        if rc != L.OK and self._cb_error is not None:  # a Python prox callback raised: surface ITS exception
            exc, self._cb_error = self._cb_error, None
            raise exc
        L.check(rc)
        self.last = s
        return s

    def _options(self, keep, *, rho=1.0, maxiters=1000, domaxiters=0, relax=1.0, fast=L.FAST_OFF, objevals=0,
                 convtest=0, convtol=1e-10, stopcond="standard", nodualerror=0, abstol=1e-5, reltol=1e-3,
                 Hnormtol=1e-6, restart=0.999, dvaltol=1e-8, record_history=1, check_every=0, x0=None, z0=None,
                 u0=None, stale_factor_ok=0):
        o = L.Options()
        self._lib.admm_options_default(C.byref(o))
        o.rho, o.relax, o.abstol, o.reltol = float(rho), float(relax), float(abstol), float(reltol)
        o.Hnormtol, o.convtol, o.restart, o.dvaltol = float(Hnormtol), float(convtol), float(restart), float(dvaltol)
        o.maxiters = int(maxiters)
        o.domaxiters = int(bool(domaxiters))
        o.fast = int(fast)
        o.objevals = int(bool(objevals))
        o.convtest = int(bool(convtest))
        o.stopcond = _STOPCOND[stopcond]
        o.nodualerror = int(bool(nodualerror))
        o.record_history = int(bool(record_history))
        o.check_every = int(check_every)
        o.stale_factor_ok = int(bool(stale_factor_ok))
        for name, val in (("x0", x0), ("z0", z0), ("u0", u0)):
            if val is not None:
                a = _f64(val).reshape(-1)
                keep.append(a)
                setattr(o, name, L.as_dp(a))
        self._relax = float(relax)
        return o

    def fetch(self, field, count, shape=None):
        out = np.empty(int(count), dtype=np.float64)
        written = C.c_size_t(0)
        L.check(self._lib.admm_engine_fetch(self._h, field, L.as_dp(out), out.size, C.byref(written)))
        out = out[:written.value]
        if shape is not None:
            out = out.reshape(shape, order="F")
        return out
