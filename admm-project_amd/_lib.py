"""ctypes binding of libadmm_hip.so (include/admm_engine.h).

The product path has no CPU fallback: if the shared library is missing or no HIP device
is visible, every compute entry point raises ``AdmmError`` loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libadmm_hip.so")

ABI_VERSION = 5

# error codes
OK, E_INVALID, E_UNSUPPORTED, E_DEVICE, E_NUMERIC, E_COMM, E_CAPACITY = 0, -1, -2, -3, -4, -5, -6

# problem kinds
PROB_LASSO, PROB_LASSO_CONSENSUS, PROB_LAD, PROB_HUBERFIT = 1, 2, 3, 4
PROB_LINEARSVM, PROB_TOTALVARIATION, PROB_QP_BOUNDED, PROB_BASISPURSUIT = 5, 6, 7, 8
PROB_MODEL, PROB_LINEARPROGRAM, PROB_QP_STANDARD, PROB_TV2D = 9, 10, 11, 12
LOSS_HINGE, LOSS_01, LOSS_HINGE_OBJ01 = 0, 1, 2
XSOLVE_AUTO, XSOLVE_TRSV, XSOLVE_INVERSE, XSOLVE_CG, XSOLVE_CALLBACK, XSOLVE_PINV = 0, 1, 2, 3, 4, 5
MEM_HOST, MEM_DEVICE = 0, 1
STOP_STANDARD, STOP_HNORM, STOP_BOTH, STOP_NONE = 0, 1, 2, 3
FAST_OFF, FAST_STRONG, FAST_WEAK = 0, 1, 2

# fetch fields
(F_XOPT, F_ZOPT, F_UOPT, F_XVALS, F_ZVALS, F_UVALS, F_PNORM, F_DNORM, F_PERR, F_DERR, F_OBJEVALS, F_HNORMSQ,
 F_AVALS, F_DVALS, F_RESTARTED, F_VVALS, F_UHATVALS, F_ZCONSENSUS, F_FACTOR, F_CG_ITERS, F_CONS_X,
 F_CONS_U) = range(1, 23)

K_XSOLVE, K_GEMV_N, K_GEMV_T, K_PROX, K_FINALIZE, K_COUNT = 0, 1, 2, 3, 4, 5
COMM_ID_BYTES = 128
COMM_RCCL, COMM_SHM, COMM_P2P = 0, 1, 2

_dp = C.POINTER(C.c_double)


class ProblemDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("problem", C.c_int32),
        ("m", C.c_int64), ("n", C.c_int64),
        ("D", _dp), ("ldD", C.c_int64),
        ("s", _dp), ("ell", _dp), ("P", _dp), ("q", _dp), ("lb", _dp), ("ub", _dp), ("L", _dp),
        ("lambda_", C.c_double), ("C", C.c_double), ("r", C.c_double), ("rho", C.c_double),
        ("loss", C.c_int32), ("userelax", C.c_int32), ("xsolve", C.c_int32), ("mem", C.c_int32),
        ("device", C.c_int32), ("nslices", C.c_int32),
        ("slices", C.POINTER(C.c_int64)),
        ("comm", C.c_void_p),
        ("cg_tol", C.c_double), ("cg_maxit", C.c_int32), ("obj_gram", C.c_int32),
        ("Q", _dp), ("qz", _dp), ("D2", _dp), ("m2", C.c_int64), ("ldD2", C.c_int64), ("s2", _dp), ("c", _dp),
        ("K", _dp), ("k0", _dp), ("Dplus", _dp), ("Dts", _dp),
    ]


# caller-supplied prox operators / objective (device pointers as integers; see admm_engine.h)
PROX_CALLBACK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p,
                            C.c_int64, C.c_void_p)
OPERATOR_CALLBACK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p)
OBJ_CALLBACK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p)
ALTU_CALLBACK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                            C.c_void_p)
NORMS_CALLBACK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                             C.c_double, C.c_void_p, C.c_void_p)


class Options(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("maxiters", C.c_int32),
        ("rho", C.c_double), ("relax", C.c_double), ("abstol", C.c_double), ("reltol", C.c_double),
        ("Hnormtol", C.c_double), ("convtol", C.c_double), ("restart", C.c_double), ("dvaltol", C.c_double),
        ("domaxiters", C.c_int32), ("fast", C.c_int32), ("objevals", C.c_int32), ("convtest", C.c_int32),
        ("stopcond", C.c_int32), ("nodualerror", C.c_int32), ("record_history", C.c_int32),
        ("check_every", C.c_int32),
        ("x0", _dp), ("z0", _dp), ("u0", _dp),
        ("stale_factor_ok", C.c_int32), ("reserved1", C.c_int32),
    ]


class RunSummary(C.Structure):
    _fields_ = [
        ("steps", C.c_int32), ("stopped_early", C.c_int32), ("convtest_failed_at", C.c_int32),
        ("obj_gram_used", C.c_int32), ("runtime_s", C.c_double), ("objopt", C.c_double),
    ]


class EngineInfo(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("xsolve_requested", C.c_int32), ("xsolve_used", C.c_int32),
        ("pinv_used", C.c_int32), ("probed", C.c_int32), ("trsv_blocks", C.c_int32), ("jacobi_sweeps", C.c_int32),
        ("unwrapped_fused", C.c_int32), ("factor_n", C.c_int64), ("rank", C.c_int64),
        ("cond_estimate", C.c_double), ("probe_err_inverse", C.c_double), ("probe_err_trsv", C.c_double),
        ("probe_diff", C.c_double), ("xsolve_cacheable_bytes", C.c_int64), ("xsolve_stream_bytes", C.c_int64),
        ("obj_bound_max", C.c_double), ("obj_form_literal", C.c_int32), ("reserved0", C.c_int32),
        ("probe_err_trsv_one", C.c_double),
    ]


class Field(C.Structure):  # admm_field: one field of a host struct as the binding layer reads it
    _fields_ = [("name", C.c_char_p), ("kind", C.c_int32), ("reserved", C.c_int32), ("data", C.POINTER(C.c_double)),
                ("rows", C.c_int64), ("cols", C.c_int64), ("text", C.c_char_p), ("ir", C.POINTER(C.c_uint64)),
                ("jc", C.POINTER(C.c_uint64))]


class BindingInfo(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("problem", C.c_int32), ("nA", C.c_int64), ("nB", C.c_int64),
                ("nU", C.c_int64), ("a_handle", C.c_int32), ("b_kind", C.c_int32), ("b_scalar", C.c_double),
                ("b_matrix", C.POINTER(C.c_double)), ("b_ld", C.c_int64)]


class ResultField(C.Structure):
    _fields_ = [("name", C.c_char_p), ("kind", C.c_int32), ("source", C.c_int32), ("rows", C.c_int64),
                ("cols", C.c_int64), ("scalar", C.c_double)]


FIELD_NUMERIC, FIELD_TEXT, FIELD_HANDLE, FIELD_SPARSE, FIELD_OTHER = 0, 1, 2, 3, 4
RES_FETCH, RES_SCALAR, RES_START = 0, 1, 2
F_WVALS = 23


class AdmmError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"[admm_hip {code}] {message}")
        self.code = code


# every symbol include/admm_engine.h declares, with its signature
_SIGNATURES = {
    "admm_abi_version": (C.c_int, []),
    "admm_last_error": (C.c_char_p, []),
    "admm_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "admm_device_info": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "admm_options_default": (None, [C.POINTER(Options)]),
    "admm_problem_desc_default": (None, [C.POINTER(ProblemDesc)]),
    "admm_engine_create": (C.c_int, [C.POINTER(ProblemDesc), C.POINTER(C.c_void_p)]),
    "admm_engine_set_callbacks": (C.c_int, [C.c_void_p, PROX_CALLBACK, C.c_void_p, PROX_CALLBACK, C.c_void_p,
                                            OBJ_CALLBACK, C.c_void_p]),
    "admm_engine_set_operators": (C.c_int, [C.c_void_p, OPERATOR_CALLBACK, C.c_void_p, OPERATOR_CALLBACK, C.c_void_p]),
    "admm_engine_set_constraint_b": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_int64, C.c_int64, C.c_int32,
                                               C.c_double, OPERATOR_CALLBACK, C.c_void_p]),
    "admm_engine_run": (C.c_int, [C.c_void_p, C.POINTER(Options), C.POINTER(RunSummary)]),
    "admm_engine_fetch": (C.c_int, [C.c_void_p, C.c_int, _dp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "admm_engine_set_hooks": (C.c_int, [C.c_void_p, ALTU_CALLBACK, C.c_void_p, NORMS_CALLBACK, C.c_void_p]),
    "admm_engine_info": (C.c_int, [C.c_void_p, C.POINTER(EngineInfo)]),
    "admm_engine_setup_seconds": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "admm_engine_kernel_time": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "admm_engine_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "admm_engine_set_profiling_stride": (C.c_int, [C.c_void_p, C.c_int]),
    "admm_engine_destroy": (None, [C.c_void_p]),
    "admm_binding_create": (C.c_int, [C.c_char_p, C.POINTER(Field), C.c_int32, C.POINTER(Field), C.c_int32,
                                      C.POINTER(C.c_void_p)]),
    "admm_binding_destroy": (None, [C.c_void_p]),
    "admm_binding_desc": (C.POINTER(ProblemDesc), [C.c_void_p]),
    "admm_binding_get_info": (C.c_int, [C.c_void_p, C.POINTER(BindingInfo)]),
    "admm_binding_apply": (C.c_int, [C.c_void_p, C.c_void_p]),
    "admm_binding_options": (C.c_int, [C.c_void_p, C.POINTER(Field), C.c_int32, C.POINTER(Field), C.c_int32,
                                       C.POINTER(Options)]),
    "admm_binding_results": (C.c_int, [C.c_void_p, C.POINTER(Options), C.POINTER(RunSummary), C.POINTER(ResultField),
                                       C.c_int32, C.POINTER(C.c_int32)]),
    "admm_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "admm_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "admm_op_gemv_n": (C.c_int, [_dp, C.c_int64, C.c_int64, C.c_int64, _dp, _dp]),
    "admm_op_gemv_t": (C.c_int, [_dp, C.c_int64, C.c_int64, C.c_int64, _dp, C.c_int64, C.c_int32, _dp, C.c_int64]),
    "admm_op_gram": (C.c_int, [_dp, C.c_int64, C.c_int64, C.c_int64, C.c_double, _dp]),
    "admm_op_cholesky": (C.c_int, [_dp, C.c_int64, C.c_int64]),
    "admm_op_trsv_pair": (C.c_int, [_dp, C.c_int64, C.c_int64, _dp, _dp]),
    "admm_op_soft_threshold": (C.c_int, [_dp, C.c_int64, C.c_double, _dp]),
    "admm_comm_unique_id": (C.c_int, [C.c_char_p]),
    "admm_comm_init": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "admm_comm_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "admm_comm_allreduce_sum": (C.c_int, [C.c_void_p, _dp, C.c_size_t]),
    "admm_comm_measure_latency": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_double)]),
    "admm_comm_destroy": (None, [C.c_void_p]),
    "admm_comm_init_all": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]),
    "admm_engine_create_all": (C.c_int, [C.c_int, C.POINTER(ProblemDesc), C.POINTER(C.c_void_p)]),
    "admm_engine_run_all": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.POINTER(Options), C.c_int,
                                      C.POINTER(RunSummary)]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def load():
    """Load libadmm_hip.so (once).  Raises AdmmError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AdmmError(E_DEVICE, f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                                  f"g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
    if os.environ.get("ADMM_HIP_NO_TORCH") != "1":
        # torch ships its own HIP runtime (torch/lib/libamdhip64.so).  Whichever runtime is loaded first owns
        # the GPU: if ours (/opt/rocm) came first, torch would later report "No HIP GPUs are available" and
        # RCCL sharing / device-tensor prox callbacks could not work.  Importing torch first makes
        # libadmm_hip.so bind to the runtime already in the process (same soname).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.admm_abi_version() != ABI_VERSION:
        raise AdmmError(E_INVALID, "libadmm_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc):
    if rc != OK:
        msg = load().admm_last_error()
        raise AdmmError(rc, msg.decode("utf-8", "replace") if msg else "unknown error")


def device_count():
    n = C.c_int(0)
    rc = load().admm_device_count(C.byref(n))
    return n.value if rc == OK else 0


def require_device():
    if device_count() <= 0:
        raise AdmmError(E_DEVICE, "no HIP device visible: the ADMM engine has no CPU fallback")


def as_dp(a):
    return a.ctypes.data_as(_dp)
