"""Oracle (test infrastructure): prox-operator factory, restating getProxOps.m.

``getproxops(problem, args)`` returns ``(minx, minz, extra)`` closures with the
reference signature ``f(x, z, u, rho)``; each nested function cites the
getProxOps.m lines it follows.  Dense fp64 NumPy/SciPy; triangular solves go
through LAPACK (scipy.linalg.solve_triangular) where the reference uses
MATLAB's ``\\`` on a (sparse-stored) triangular factor.
Parity pin status: see ``oracle/__init__.py``.
"""
from __future__ import annotations

import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp
import scipy.sparse.linalg as spla


def soft_threshold(v, t):
    """getProxOps.m:933-938  sign(v).*subplus(abs(v) - t)."""
    return np.sign(v) * np.maximum(np.abs(v) - t, 0.0)


def minz01(s, t):
    """getProxOps.m:1158-1180  y=1; y(s>=1 | s<1-sqrt(2/t)) = s."""
    y = np.ones_like(s)
    inds = (s >= 1) | (s < (1 - np.sqrt(2.0 / t)))
    y[inds] = s[inds]
    return y


def _lower_solve(L, b):
    return sla.solve_triangular(L, b, lower=True, check_finite=False)


def _upper_solve(U, b):
    return sla.solve_triangular(U, b, lower=False, check_finite=False)


def getproxops(problem, args):
    if not isinstance(problem, str):
        raise TypeError("problem must be a string")
    if not isinstance(args, dict):
        raise TypeError("args must be a struct (dict)")
    problem = problem.lower()
    extra = {}

    if problem == "model":  # getProxOps.m:55-95, 952-1012
        PtP, Ptr, QtQ, Qts, n = args["PtP"], args["Ptr"], args["QtQ"], args["Qts"], args["n"]

        def xmin(_x, z, u, rho):
            # 968-974: rhoprev stays 0 (quirk q22) so the diagonal is refreshed each call.
            return np.linalg.solve(PtP + rho * np.eye(n), Ptr + rho * (z - u))

        def zmin(x, _z, u, rho):
            return np.linalg.solve(QtQ + rho * np.eye(n), Qts + rho * (x + u))

        return xmin, zmin, extra

    if problem == "basispursuit":  # 98-142, 1027-1032
        P, q = args["P"], args["q"]
        return (lambda _x, z, u, _rho: P @ (z - u) + q,
                lambda x, _z, u, rho: soft_threshold(u + x, 1.0 / rho), extra)

    if problem == "totalvariation":  # 145-199, 1044-1048
        D, Dt, DtD, s, lam = args["D"], args["Dt"], args["DtD"], args["s"], args["lambda"]
        Id = sp.identity(DtD.shape[0], format="csc")

        def xmin(_x, z, u, rho):  # 1044-1048: (I + rho*DtD) \ (s + rho*Dt*(z - u)), re-assembled on every call
            rhs = s + rho * (Dt @ (z - u))
            if args.get("banded", 0):
                # oracle-side switch for n ~ 1e7: MATLAB's backslash on this SPD tridiagonal matrix IS a banded
                # Cholesky solve; SuperLU (spsolve) computes the same x but needs minutes and tens of GB there.
                M = (Id + rho * DtD).tocsc()
                ab = np.zeros((2, M.shape[0]))
                ab[0] = M.diagonal(0)
                ab[1, :-1] = M.diagonal(-1)
                return sla.solveh_banded(ab, rhs, lower=True, check_finite=False)
            return spla.spsolve((Id + rho * DtD).tocsc(), rhs)

        return xmin, (lambda x, _z, u, rho: soft_threshold(u + D @ x, lam / rho)), extra

    if problem == "linearsvm":  # 202-310, 1062-1143
        D, ell, C, loss = args["D"], args["ell"], args["C"], args["lossfunction"]

        def zprox(Dx, uu, ll, rho):
            Dxplusu = Dx + uu
            v = ll * Dxplusu
            if loss != "01":  # 1094: anything that is not '01' runs the hinge prox (quirk q18)
                return Dxplusu + ll * np.maximum(np.minimum(1 - v, C / rho), 0.0)
            with np.errstate(divide="ignore"):  # MATLAB arithmetic: rho/0 is Inf (C = 0: the threshold 1 - sqrt(2/t) is 1)
                return ll * minz01(v, np.float64(rho) / np.float64(C))

        if "slices" in args:  # 284-303, 1120-1143
            slices = [int(k) for k in args["slices"]]
            starts = np.concatenate([[0], np.cumsum(slices)])

            def zmin_par(x, _z, u, rho, i):
                lo, hi = starts[i], starts[i + 1]
                return zprox(D[lo:hi, :] @ x, u[lo:hi], ell[lo:hi], rho)

            return 0, zmin_par, extra
        Dplus = args["Dplus"]
        return (lambda _x, z, u, _rho: Dplus @ (z - u),
                lambda x, _z, u, rho: zprox(D @ x, u, ell, rho), extra)

    if problem == "lasso":  # 313-456, 1192-1343
        if args.get("parallel", 0):
            return _consensus_lasso(args)
        D, Dts, lam, L, U, m, n = (args["D"], args["Dts"], args["lambda"], args["L"], args["U"],
                                   args["m"], args["n"])

        def xmin(_x, z, u, rho):  # 1192-1206
            y = rho * (z - u) + Dts
            if m >= n:
                return _upper_solve(U, _lower_solve(L, y))
            return y / rho - (D.T @ _upper_solve(U, _lower_solve(L, D @ y))) / rho ** 2

        return xmin, (lambda x, _z, u, rho: soft_threshold(u + x, lam / rho)), extra

    if problem == "linearprogram":  # 459-542, 1357-1382
        D, b, s, n = args["D"], args["b"], args["s"], args["n"]
        mm = D.shape[0]

        def xmin(_x, z, u, rho):
            K = np.block([[rho * np.eye(n), D.T], [D, np.zeros((mm, mm))]])
            return np.linalg.solve(K, np.concatenate([rho * (z - u) - b, s]))[:n]

        return xmin, (lambda x, _z, u, _rho: np.maximum(x + u, 0.0)), extra

    if problem == "quadraticprogram":  # 545-666, 1397-1474
        P, q, n = args["P"], args["q"], args["n"]
        if args["constraint"] == "bounded":
            lb, ub = args["lb"], args["ub"]
            st = {"rhoprev": args["rho"], "R": sla.cholesky(P + args["rho"] * np.eye(n), lower=False)}

            def xmin(_x, z, u, rho):  # 1441-1456
                if rho != st["rhoprev"]:
                    st["R"] = sla.cholesky(P + rho * np.eye(n), lower=False)
                    st["rhoprev"] = rho
                R = st["R"]
                return _upper_solve(R, _lower_solve(R.T, rho * (z - u) - q))

            zmin = lambda x, _z, u, _rho: np.minimum(ub, np.maximum(lb, x + u))  # 1470-1474
        else:
            D, s = args["D"], args["s"]
            mm = D.shape[0]

            def xmin(_x, z, u, rho):  # 1397-1412
                K = np.block([[P + rho * np.eye(n), D.T], [D, np.zeros((mm, mm))]])
                return np.linalg.solve(K, np.concatenate([rho * (z - u) - q, s]))[:n]

            zmin = lambda x, _z, u, _rho: np.maximum(x + u, 0.0)
        if "altproxg" in args:
            zmin = args["altproxg"]
        return xmin, zmin, extra

    if problem in ("lad", "huberfit"):  # 753-912, 1511-1539
        R, D, s = args["R"], args["D"], args["s"]
        Rt = R.T

        def xmin(_x, z, u, _rho):  # 1511-1515
            return _upper_solve(Rt, _lower_solve(R, D.T @ (s + z - u)))

        userelax = bool(args.get("userelax", 0))
        if problem == "lad":
            if userelax:
                zmin = lambda x, _z, u, rho: soft_threshold(x + u - s, 1.0 / rho)  # 808
            else:
                zmin = lambda x, _z, u, rho: soft_threshold(D @ x + u - s, 1.0 / rho)  # 810
        else:
            def huber_prox(Ax, u, rho):  # 1529-1539
                v = Ax + u - s
                return 1.0 / (1.0 + rho) * (rho * v + soft_threshold(v, 1.0 + 1.0 / rho))

            if userelax:
                zmin = lambda Dxhat, _z, u, rho: huber_prox(Dxhat, u, rho)  # 907-908
            else:
                zmin = lambda x, _z, u, rho: huber_prox(D @ x, u, rho)  # 910-911
        return xmin, zmin, extra

    raise ValueError("Invalid input for problem - given string is not a solver!")


def _consensus_lasso(args):
    """getProxOps.m:383-442 (setup) and 1217-1343 (closures), quirks q9-q12 included."""
    D, s, lam = args["D"], args["s"], args["lambda"]
    slices = [int(k) for k in args["slices"]]
    m, n = D.shape
    N = len(slices)
    starts = np.concatenate([[0], np.cumsum(slices)])
    st = {"z": np.zeros(n), "xave": np.zeros(n), "xaveprev": np.zeros(n), "rhoprev": args["rho"]}
    Di = [D[starts[i]:starts[i + 1], :] for i in range(N)]
    Dtsi = [Di[i].T @ s[starts[i]:starts[i + 1]] for i in range(N)]
    xi = [np.zeros(n) for _ in range(N)]
    ui = [np.zeros(n) for _ in range(N)]
    DtDi, Li = [None] * N, [None] * N

    # q12 (documented deviation): the reference's fat-slice branch adds rho along stride n+1 of an mi x mi matrix
    # (not its diagonal) and then applies the /rho^2 formula that belongs to chol(D*D'/rho + I) (lasso.m:172,
    # getProxOps.m:1204).  args["fatformula"] = "serial" selects that serial, correct form for fat slices -- the
    # Woodbury identity of (Di'Di + rho I)^-1 -- which is what the engine implements; the default restates the
    # reference literally.
    serial_fat = args.get("fatformula", "literal") == "serial"

    def refactor(k, rho):
        mi = Di[k].shape[0]
        if serial_fat and mi < n:
            Li[k] = sla.cholesky(DtDi[k] / rho + np.eye(mi), lower=True)  # lasso.m:172
            return
        P = np.array(DtDi[k], order="F", copy=True)
        # 430 / 1232: Pi(1:n+1:end) = DtDi(1:n+1:end) + rho -- stride n+1 in
        # column-major linear indexing even when the matrix is mi x mi (q12).
        flat = P.reshape(-1, order="F")
        src = DtDi[k].reshape(-1, order="F")
        flat[0::n + 1] = src[0::n + 1] + rho
        P = flat.reshape(P.shape, order="F")
        Li[k] = sla.cholesky(P, lower=True)

    pre = args.get("_DtDi")  # oracle-side: slice Gram matrices a test has already formed on the host (same products)
    for k in range(N):  # 419-436
        mi = Di[k].shape[0]
        if pre is not None:
            DtDi[k] = pre[k]
        else:
            DtDi[k] = Di[k].T @ Di[k] if mi >= n else Di[k] @ Di[k].T
        refactor(k, st["rhoprev"])

    def xmin(_x, _z, _u, rho):  # 1217-1260
        newrho = rho != st["rhoprev"]
        for k in range(N):
            if newrho:
                refactor(k, rho)
                st["rhoprev"] = rho
            yi = rho * (st["z"] - ui[k]) + Dtsi[k]
            mi = Di[k].shape[0]
            if mi >= n:
                xi[k] = _upper_solve(Li[k].T, _lower_solve(Li[k], yi))
            else:
                xi[k] = yi / rho - (Di[k].T @ _upper_solve(Li[k].T, _lower_solve(Li[k], Di[k] @ yi))) / rho ** 2
        acc = np.zeros(n)
        for k in range(N):
            acc = acc + xi[k]
        return acc / N

    def zmin(_x, _z, _u, rho):  # 1272-1299
        uave = np.zeros(n)
        minz = uave  # 1276: the value handed back to admm is all zeros (q9)
        st["xaveprev"] = st["xave"]
        xave = np.zeros(n)
        for j in range(N):
            uave = uave + ui[j]
            xave = xave + xi[j]
        uave = uave / N
        xave = xave / N
        st["xave"] = xave
        v = uave + xave
        st["z"] = np.sign(v) * np.maximum(np.abs(v) - lam / (rho * N), 0.0)  # q11
        for j in range(N):
            ui[j] = ui[j] + (xi[j] - st["z"])
        return minz

    def altu(_u, _Ax, _Bz, _c):  # 1312-1326
        acc = np.zeros(n)
        for j in range(N):
            acc = acc + ui[j]
        return acc / N

    def norms(_x, _z, _u, _rho):  # 1335-1343: squared, uses rhoprev (q10)
        v0 = 0.0
        for j in range(N):
            v0 += float(np.sum((xi[j] - st["xave"]) ** 2))
        v1 = N * st["rhoprev"] ** 2 * float(np.sum((st["xave"] - st["xaveprev"]) ** 2))
        return [v0, v1]

    extra = {"altu": altu, "specialnorms": norms,
             # oracle-side peek at the closure state (not in the reference API):
             "_state": st, "_xi": xi, "_ui": ui}
    return xmin, zmin, extra
