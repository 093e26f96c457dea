"""The binding layer of the C ABI (include/admm_engine.h: admm_binding_*) from Python: host structs (dicts) flattened
into ``admm_field`` arrays, exactly as the MEX gateway flattens MATLAB structs (csrc/admm_mex.cpp: Fields).  What the
fields MEAN -- getproxops' args per problem (getProxOps.m:52-917), admm's option defaults (admm.m:780-971), the layout of
results (admm.m:257-767) -- is decided behind the ABI (csrc/binding.hip), on the host: nothing here needs a GPU."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


class Handle:
    """Stands for a function handle in a host struct (the binding layer only asks whether one is there)."""


class Sparse:
    """A real CSC matrix (MATLAB's sparse storage: lasso.m:175 hands the factor over like this)."""

    def __init__(self, dense):
        a = np.asarray(dense, dtype=np.float64)
        self.shape = a.shape
        cols = [np.nonzero(a[:, j])[0] for j in range(a.shape[1])]
        self.jc = np.concatenate([[0], np.cumsum([len(c) for c in cols])]).astype(np.uint64)
        self.ir = (np.concatenate(cols) if cols else np.zeros(0)).astype(np.uint64)
        self.data = np.concatenate([a[c, j] for j, c in enumerate(cols)]).astype(np.float64) if cols else np.zeros(0)


class Fields:
    """dict -> (admm_field array, count); keeps every buffer the array points into alive."""

    def __init__(self, struct):
        self._keep = []
        items = list((struct or {}).items())
        self.array = (L.Field * max(1, len(items)))()
        self.count = len(items)
        for f, (name, value) in zip(self.array, items):
            f.name = name.encode()
            f.kind = L.FIELD_OTHER
            if isinstance(value, Handle) or callable(value):
                f.kind = L.FIELD_HANDLE
            elif isinstance(value, str):
                f.kind = L.FIELD_TEXT
                f.text = value.encode()
            elif isinstance(value, Sparse):
                f.kind = L.FIELD_SPARSE
                f.rows, f.cols = value.shape
                f.data = value.data.ctypes.data_as(C.POINTER(C.c_double))
                f.ir = value.ir.ctypes.data_as(C.POINTER(C.c_uint64))
                f.jc = value.jc.ctypes.data_as(C.POINTER(C.c_uint64))
                self._keep.append(value)
            else:
                a = np.asfortranarray(np.atleast_1d(np.asarray(value, dtype=np.float64)))
                if a.ndim == 1:
                    a = a.reshape(-1, 1)
                f.kind = L.FIELD_NUMERIC
                f.rows, f.cols = a.shape
                f.data = a.ctypes.data_as(C.POINTER(C.c_double))
                self._keep.append(a)


class Binding:
    """``admm_binding``: a problem description built from getproxops' argument struct."""

    def __init__(self, problem, args, handles=None):
        self._lib = L.load()
        self._args, self._handles = Fields(args), Fields(handles)  # (the description borrows their arrays)
        self._h = C.c_void_p()
        L.check(self._lib.admm_binding_create(problem.encode(), self._args.array, self._args.count, self._handles.array,
                                              self._handles.count, C.byref(self._h)))

    def close(self):
        if self._h:
            self._lib.admm_binding_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        self.close()

    @property
    def desc(self):
        return self._lib.admm_binding_desc(self._h).contents

    def info(self):
        i = L.BindingInfo()
        i.struct_size = C.sizeof(L.BindingInfo)
        L.check(self._lib.admm_binding_get_info(self._h, C.byref(i)))
        return dict(problem=i.problem, nA=i.nA, nB=i.nB, nU=i.nU, a_handle=bool(i.a_handle), b_kind=i.b_kind,
                    b_scalar=i.b_scalar, b_ld=i.b_ld)

    def options(self, options=None, handles=None):
        fo, fh = Fields(options), Fields(handles)
        o = L.Options()
        L.check(self._lib.admm_binding_options(self._h, fo.array, fo.count, fh.array, fh.count, C.byref(o)))
        o._keep = (fo, fh)  # x0 / z0 / u0 point into the option fields
        return o

    def results(self, opts, steps, convtest_failed_at=0, objopt=float("nan"), runtime=0.0):
        s = L.RunSummary()
        s.steps, s.convtest_failed_at, s.objopt, s.runtime_s = steps, convtest_failed_at, objopt, runtime
        out = (L.ResultField * 64)()
        n = C.c_int32()
        L.check(self._lib.admm_binding_results(self._h, C.byref(opts), C.byref(s), out, 64, C.byref(n)))
        return [dict(name=out[i].name.decode(), kind=out[i].kind, source=out[i].source, rows=out[i].rows,
                     cols=out[i].cols, scalar=out[i].scalar) for i in range(n.value)]
