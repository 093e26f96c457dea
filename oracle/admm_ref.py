"""Oracle (test infrastructure): generic ADMM loop, restating /root/reference/admm.m.

``admm(xminf, zming, options)`` follows admm.m:24-769 statement by statement:
option defaults (admm.m:51-76, setopt 780-971), operator normalisation
(79-245), Fast/Accelerated state (262-298), H-norm closure (302-313), the
iteration loop (496-743) and result packing (746-767).  ``options`` and
``results`` are plain dicts using the reference's field names.  Vectors are 1-D
float64 arrays.  Parity pin status: see ``oracle/__init__.py``.

Not restated: the PCT ``parfor`` in-prox slicing wrapper (admm.m:343-468) -- it
is a pure map+concatenate, restated in ``oracle/parallel_ref.py`` on top of this
loop; N-D (matrix) variables (used only by covarianceselection, out of scope).
"""
from __future__ import annotations

import math
import time

import numpy as np

_EPS = np.finfo(np.float64).eps


def _setopt(options, name, default):
    """admm.m:780-971 -- value if present, otherwise the default.

    Quirk q2 (admm.m:927-928): the reference reads ``Hnormtol`` from a field
    called ``Hreltol``; we accept either spelling.
    """
    if name == "Hnormtol":
        # literal reference: the value is taken from Hreltol when Hnormtol is present (and
        # errors when Hreltol is missing).  Stance D: either spelling sets the tolerance.
        if "Hreltol" in options:
            return options["Hreltol"]
        if "Hnormtol" in options:
            return options["Hnormtol"]
        return default
    return options[name] if name in options else default


def _as_operator(M):
    """admm.m:117-120, 165-167, 202-204: a matrix/scalar becomes ``v -> M*v``."""
    if callable(M):
        return M, None
    if np.isscalar(M):
        val = float(M)
        return (lambda v: val * v), (1, 1)
    if hasattr(M, "shape") and len(M.shape) == 2:  # ndarray or scipy.sparse
        return (lambda v: M @ v), tuple(M.shape)
    raise ValueError("operator is neither numeric matrix nor function handle")


def _fro(v):
    v = np.asarray(v, dtype=np.float64)
    return float(np.sqrt(np.sum(v * v)))


def admm(xminf, zming, options):
    if not isinstance(options, dict):
        raise TypeError("Given options is not a struct! At least pass empty struct!")
    options = dict(options)
    results = {}

    adaptive = _setopt(options, "adaptive", 0)
    rho = float(_setopt(options, "rho", 1.0))
    N = _setopt(options, "maxiters", 1000)
    domaxiters = _setopt(options, "domaxiters", 0)
    relax = float(_setopt(options, "relax", 1))
    fast = _setopt(options, "fast", 0)
    fasttype = _setopt(options, "fasttype", "weak")
    obj = _setopt(options, "obj", 0)
    objevals = _setopt(options, "objevals", 0)
    convtest = _setopt(options, "convtest", 0)
    convtol = _setopt(options, "convtol", 1e-10)
    stopcond = _setopt(options, "stopcond", "standard")
    nodualerror = _setopt(options, "nodualerror", 0)
    ABSTOL = _setopt(options, "abstol", 1e-5)
    RELTOL = _setopt(options, "reltol", 1e-3)
    HNORMTOL = _setopt(options, "Hnormtol", 1e-6)
    m = int(_setopt(options, "m", 0))
    nA = int(_setopt(options, "nA", 0))
    nB = int(_setopt(options, "nB", 0))

    # --- c (admm.m:79-110)
    if "c" in options:
        c = options["c"]
        if np.isscalar(c):
            if m == 0:
                raise ValueError("scalar c and no length m provided")
            c = float(c)
        else:
            c = np.asarray(c, dtype=np.float64).reshape(-1)
            if c.size != 1:
                m = c.size
            elif m == 0:
                raise ValueError("scalar c and no length m provided")
    else:
        if m > 0:
            c = np.zeros(m)
        else:
            raise ValueError("Must specify a vector c in constraint Ax + Bz = c!")

    # --- A (admm.m:113-158)
    if "A" not in options:
        raise ValueError("Must specify a matrix A in constraint Ax + Bz = c!")
    Araw = options["A"]
    A, shpA = _as_operator(Araw)
    if shpA is None:
        if nA == 0:
            raise ValueError("A is a function handle but nA not specified")
        mA, nAtemp = np.atleast_1d(A(np.zeros(nA))).shape[0], 1
    else:
        mA, nAtemp = shpA
        # admm.m:119 overwrites options.At with A' whenever A is a matrix.
        options["At"] = Araw if np.isscalar(Araw) else Araw.T
    if mA != m and mA != 1:
        raise ValueError("rows of A do not match length of c")
    if nA == 0 and nAtemp == 1 and mA == 1:
        raise ValueError("scalar A with no nA specified")
    elif nAtemp != 1 and nA != nAtemp:
        nA = nAtemp

    # --- At (admm.m:161-195)
    if "At" not in options:
        raise ValueError("Must specify a matrix A in constraint Ax + Bz = c!")
    At, shpAt = _as_operator(options["At"])
    if shpAt is not None:
        nAt, mAt = shpAt
        if mAt != mA:
            raise ValueError("columns of At do not match rows of A")
        if nAt != nA and not (nAt == 1 and mAt == 1):
            raise ValueError("rows of At do not match columns of A")

    # --- B (admm.m:198-245)
    if "B" not in options:
        raise ValueError("Must specify a matrix B in constraint Ax + Bz = c!")
    B, shpB = _as_operator(options["B"])
    if shpB is None:
        if nB == 0:
            raise ValueError("B is a function handle but nB not specified")
        mB, nBtemp = np.atleast_1d(B(np.zeros(nB))).shape[0], 1
    else:
        mB, nBtemp = shpB
    if mB != m and mB != 1:
        raise ValueError("rows of B do not match length of c")
    if nB == 0 and nBtemp == 1 and mB == 1:
        raise ValueError("scalar B with no nB specified")
    elif nBtemp != 1 and nB != nBtemp:
        nB = nBtemp

    canEvalObj = bool(objevals) and callable(obj)

    # --- initial iterates (admm.m:252-259)
    x = np.array(_setopt(options, "x0", np.zeros(nA)), dtype=np.float64).reshape(-1)
    z = np.array(_setopt(options, "z0", np.zeros(nB)), dtype=np.float64).reshape(-1)
    u = np.array(_setopt(options, "u0", np.zeros(m)), dtype=np.float64).reshape(-1)
    results["x0"], results["z0"], results["u0"] = x.copy(), z.copy(), u.copy()

    # --- Fast / Accelerated ADMM state (admm.m:262-298)
    alg = 0
    if fast:
        v = z.copy()
        uhat = u.copy()
        acurr = 1.0
        aprev = 1.0
        if fasttype == "weak":
            d = math.inf
            dprev = math.inf
            nrst = _setopt(options, "restart", 0.999)
            if nrst <= 0 or nrst >= 1:
                nrst = 0.999
            DVALTOL = _setopt(options, "dvaltol", 1e-8)
            results["dvaltol"] = DVALTOL
            alg = 2
        else:
            alg = 1

    # --- H-norm (admm.m:302-313).  The anonymous function captures rho at
    # creation time (MATLAB semantics), hence rho_H.
    use_h = bool(convtest) or stopcond in ("hnorm", "both")
    if use_h:
        rho_H = rho

        def H_norm_sq(wdiff):
            return rho_H * _fro(B(wdiff[nA:nA + nB])) ** 2 + rho_H * _fro(wdiff[nA + nB:nA + nB + m]) ** 2

        w = np.concatenate([x, z, rho * u])
        results["Hnormtol"] = HNORMTOL

    start = time.perf_counter()

    # admm.m:334-339
    if N > 0:
        N = int(math.ceil(float(np.real(N))))
    else:
        N = 1000

    if callable(options.get("preprocess", None)):
        options["preprocess"]()

    hist = {k: [] for k in ("xvals", "zvals", "uvals", "pnorm", "dnorm", "perr", "derr",
                            "objevals", "Hnormsq", "wvals", "vvals", "uhatvals", "avals",
                            "dvals", "restarted")}
    H1 = H2 = None
    i = 0
    for i in range(1, N + 1):
        zprev = z
        # x-update (admm.m:501-511)
        if alg == 0:
            x = np.asarray(xminf(x, z, u, rho), dtype=np.float64).reshape(-1)
        else:
            aprev = acurr
            uprev = u
            x = np.asarray(xminf(x, v, uhat, rho), dtype=np.float64).reshape(-1)
            if alg == 2:
                dprev = d

        # z-update, optionally over-relaxed (admm.m:515-532)
        if relax != 1:
            Axhat = relax * A(x) - (1 - relax) * (B(zprev) - c)
            if alg == 0:
                z = zming(Axhat, z, u, rho)
            else:
                z = zming(Axhat, z, uhat, rho)
        else:
            if alg == 0:
                z = zming(x, z, u, rho)
            else:
                z = zming(x, z, uhat, rho)
        z = np.asarray(z, dtype=np.float64).reshape(-1)

        Ax = A(x)
        Bz = B(z)

        # u-update (admm.m:538-560)
        if "altu" not in options:
            if relax != 1:
                u = (u if alg == 0 else uhat) + (Axhat + Bz - c)
            else:
                u = (u if alg == 0 else uhat) + (Ax + Bz - c)
        else:
            if relax != 1:
                u = options["altu"](u, Axhat, Bz, c)
            else:
                u = options["altu"](u, Ax, Bz, c)
        u = np.asarray(u, dtype=np.float64).reshape(-1)

        # Fast / Accelerated extrapolation (admm.m:563-600)
        if alg in (1, 2):
            if alg == 1:
                acurr = 0.5 * (1 + math.sqrt(1 + 4 * aprev ** 2))
                v = z + (aprev - 1) / acurr * (z - zprev)
                uhat = u + (aprev - 1) / acurr * (u - uprev)
            else:
                d = 1 / rho * _fro(u - uhat) ** 2 + rho * _fro(B(z - v)) ** 2
                if d < nrst * dprev:
                    acurr = 0.5 * (1 + math.sqrt(1 + 4 * aprev ** 2))
                    v = z + (aprev - 1) / acurr * (z - zprev)
                    uhat = u + (aprev - 1) / acurr * (u - uprev)
                    hist["restarted"].append(0)
                else:
                    acurr = 1.0
                    v = zprev
                    uhat = uprev
                    d = dprev / nrst
                    hist["restarted"].append(1)
                hist["dvals"].append(d)
            hist["vvals"].append(np.array(v, copy=True))
            hist["uhatvals"].append(np.array(uhat, copy=True))
            hist["avals"].append(acurr)

        if canEvalObj:
            hist["objevals"].append(float(obj(x, z)))

        hist["xvals"].append(x.copy())
        hist["zvals"].append(z.copy())
        hist["uvals"].append(u.copy())

        # residual norms (admm.m:612-637)
        if callable(options.get("specialnorms", None)):
            vv = options["specialnorms"](x, z, u, rho)
            hist["pnorm"].append(float(vv[0]))
            hist["dnorm"].append(float(vv[1]))
        elif alg == 0:
            hist["pnorm"].append(_fro(Ax + Bz - c))
            hist["dnorm"].append(_fro(rho * At(B(z - zprev))) if not nodualerror else math.nan)
        elif alg == 1:
            hist["pnorm"].append(_fro(Ax + Bz - c))
            hist["dnorm"].append(rho * _fro(At(B(z - v))) if not nodualerror else math.nan)

        # tolerances (admm.m:640-658)
        if alg in (0, 1):
            M1 = np.size(Ax)
            M2 = np.size(Bz)
            hist["perr"].append(math.sqrt(M1) * ABSTOL + RELTOL * max(max(_fro(Ax), _fro(Bz)), _fro(c)))
            if not nodualerror:
                hist["derr"].append(math.sqrt(M2) * ABSTOL + RELTOL * _fro(rho * At(u)))
            else:
                hist["derr"].append(math.nan)

        # H-norm / convergence test (admm.m:676-703)
        if use_h:
            wprev = w
            w = np.concatenate([x, z, rho * u])
            hist["wvals"].append(w.copy())
            hist["Hnormsq"].append(H_norm_sq(wprev - w))
            if convtest and i >= 2:
                H2 = hist["Hnormsq"][i - 1]
                H1 = hist["Hnormsq"][i - 2]
                if alg == 0 and H1 > _EPS and H2 > H1 and not ((H2 - H1) <= H1 * convtol):
                    # quirk q4: early return, steps/xopt/... never set.
                    _pack_histories(results, hist)
                    results["convtest_failed_at"] = i  # oracle-side annotation only
                    return results

        # stopping (admm.m:706-722)
        if alg == 2 and i >= 2 and abs(d - dprev) <= DVALTOL * dprev:
            break
        elif alg in (0, 1):
            if stopcond in ("standard", "both") and (
                    (not domaxiters) and hist["pnorm"][-1] < hist["perr"][-1]
                    and (nodualerror or hist["dnorm"][-1] < hist["derr"][-1])):
                break
        if stopcond in ("hnorm", "both") and (not domaxiters) and i > 2 and hist["Hnormsq"][-1] <= HNORMTOL:
            break

        # experimental adaptive rho (admm.m:724-741)
        if adaptive and convtest and i > 2:
            growthtol = 5
            wdiff = np.float64(H1 - H2)  # MATLAB arithmetic: 0/0 is NaN, not an exception
            rhoprev = rho
            rho = rho * (wdiff * rhoprev) / (wdiff * wdiff)
            rhodiff = abs(rho - rhoprev)
            if rhodiff >= rhoprev * growthtol:
                rho = rho / growthtol
            elif rhodiff <= rhoprev / growthtol:
                rho = rho * growthtol

    _pack_histories(results, hist)
    results["steps"] = i
    results["xopt"] = x
    results["zopt"] = z
    results["uopt"] = u
    if objevals and callable(obj):
        results["objopt"] = float(obj(x, z))
    results["runtime"] = time.perf_counter() - start
    results["options"] = options
    return results


def _pack_histories(results, hist):
    """Histories as reference-shaped arrays: vectors are (len x steps) column-major."""
    for k in ("xvals", "zvals", "uvals", "wvals", "vvals", "uhatvals"):
        if hist[k]:
            results[k] = np.asfortranarray(np.stack(hist[k], axis=1))
    for k in ("pnorm", "dnorm", "perr", "derr", "objevals", "Hnormsq", "avals", "dvals", "restarted"):
        if hist[k]:
            results[k] = np.asarray(hist[k], dtype=np.float64)
