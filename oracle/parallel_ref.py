"""Oracle (test infrastructure): the transpose-reduction scheme on row shards.

Restates unwrappedadmm.m:96-141 (W = sum_i D_i'D_i once; per iteration d = sum_i D_i'(z_i - u_i),
x = W \\ d replicated, z-prox row-local) generalised to the constraint vector c = s of lad.m /
huberfit.m, as ONE rank's share of the work: every cross-shard sum goes through the supplied
``allreduce`` callable (gloo ``dist.all_reduce`` in the CPU tests).  Residual norms follow
admm.m:621-658 with the squared partial sums reduced over the shards.  Plain ADMM (alg 0),
no relaxation.  Parity pin status: see ``oracle/__init__.py``.
"""
from __future__ import annotations

import math

import numpy as np
import scipy.linalg as sla

from .proxops_ref import minz01, soft_threshold


def sharded_unwrapped(kind, D, rowvec, allreduce, *, C=0.5, rho=1.0, maxiters=1000, abstol=1e-5, reltol=1e-3,
                      nodualerror=False, stopcond="standard", Hnormtol=1e-6, z0=None, u0=None, domaxiters=False):
    """kind in {'lad','huber','svm-hinge','svm-01'}; D = this rank's rows; rowvec = s (lad/huber) or ell (svm)."""
    m, n = D.shape
    svm = kind.startswith("svm")
    c = np.zeros(m) if svm else np.asarray(rowvec, dtype=np.float64)
    ell = np.asarray(rowvec, dtype=np.float64) if svm else None
    W = allreduce(D.T @ D)  # unwrappedadmm.m:114-122
    R = sla.cholesky(W, lower=True)
    mtot = int(round(float(allreduce(np.array([float(m)]))[0])))
    cnorm = math.sqrt(float(allreduce(np.array([float(c @ c)]))[0]))
    z = np.zeros(m) if z0 is None else np.array(z0, dtype=np.float64)
    u = np.zeros(m) if u0 is None else np.array(u0, dtype=np.float64)
    out = {k: [] for k in ("xvals", "pnorm", "dnorm", "perr", "derr", "Hnormsq")}
    steps = 0
    for i in range(1, maxiters + 1):
        zprev, uprev = z, u
        d = allreduce(D.T @ ((c + z) - u))  # unwrappedadmm.m:127-137
        x = sla.solve_triangular(R.T, sla.solve_triangular(R, d, lower=True), lower=False)
        Ax = D @ x
        v = (Ax + u) - c
        if kind == "lad":
            z = soft_threshold(v, 1.0 / rho)
        elif kind == "huber":
            z = 1.0 / (1.0 + rho) * (rho * v + soft_threshold(v, 1.0 + 1.0 / rho))
        elif kind == "svm-hinge":
            z = v + ell * np.maximum(np.minimum(1 - ell * v, C / rho), 0.0)
        else:
            z = ell * minz01(ell * v, rho / C)
        u = u + ((Ax - z) - c)
        sums = allreduce(np.array([np.sum(((Ax - z) - c) ** 2), np.sum(Ax ** 2), np.sum(z ** 2),
                                   np.sum((z - zprev) ** 2), np.sum((u - uprev) ** 2)]))
        pn = math.sqrt(sums[0])
        pe = math.sqrt(mtot) * abstol + reltol * max(max(math.sqrt(sums[1]), math.sqrt(sums[2])), cnorm)
        if nodualerror:
            dn = de = math.nan
        else:
            g = allreduce(np.stack([D.T @ (z - zprev), D.T @ u]))
            dn = rho * math.sqrt(float(g[0] @ g[0]))
            de = math.sqrt(mtot) * abstol + reltol * (rho * math.sqrt(float(g[1] @ g[1])))
        hn = rho * sums[3] + rho * (rho * rho) * sums[4]
        out["xvals"].append(x)
        out["pnorm"].append(pn)
        out["dnorm"].append(dn)
        out["perr"].append(pe)
        out["derr"].append(de)
        out["Hnormsq"].append(hn)
        steps = i
        if not domaxiters:
            if stopcond in ("standard", "both") and pn < pe and (nodualerror or dn < de):
                break
            if stopcond in ("hnorm", "both") and i > 2 and hn <= Hnormtol:
                break
    res = {k: np.asarray(v) for k, v in out.items()}
    res["xvals"] = np.asfortranarray(np.stack(out["xvals"], axis=1))
    res.update(steps=steps, xopt=x, zopt=z, uopt=u)
    return res


def sharded_consensus_lasso(D, s, lam, slices, allreduce, *, rho=1.0, maxiters=1000, abstol=1e-5, reltol=1e-3,
                            Hnormtol=1e-6, u0=None, domaxiters=False):
    """Consensus lasso (getProxOps.m:383-442, 1217-1343; lasso.m:196-224, stopcond 'both') as ONE rank's share: D, s
    are this rank's rows, ``slices`` their split into local slices.  Exactly ONE vector exchange per iteration:
    ``[sum_k x_k; sum_k u_k; q]`` with ``q = sum_k ||x_k - c||^2`` about the previous mean c (known to every rank
    before the exchange); lassonorms' first value (getProxOps.m:1338-1340) is then ``q - N*||xave - c||^2`` -- the
    identity the engine's packed collective rests on (engine_run_consensus.hip).  Quirks q9-q11 as in the reference."""
    n = D.shape[1]
    starts = np.concatenate([[0], np.cumsum(slices)]).astype(int)
    Di = [D[starts[k]:starts[k + 1]] for k in range(len(slices))]
    Dtsi = [Di[k].T @ s[starts[k]:starts[k + 1]] for k in range(len(slices))]
    Li = [sla.cholesky(Di[k].T @ Di[k] + rho * np.eye(n), lower=True) for k in range(len(slices))]
    N = int(round(float(allreduce(np.array([float(len(slices))]))[0])))
    xi = [np.zeros(n) for _ in slices]
    ui = [np.zeros(n) for _ in slices]
    z = np.zeros(n)
    xave = np.zeros(n)
    ubar = np.zeros(n) if u0 is None else np.array(u0, dtype=np.float64)  # admm's own u (only the first H-norm sees it)
    out = {k: [] for k in ("xvals", "uvals", "pnorm", "dnorm", "perr", "derr", "Hnormsq")}
    steps = 0
    for i in range(1, maxiters + 1):
        for k in range(len(slices)):  # 1228-1253
            y = rho * (z - ui[k]) + Dtsi[k]
            xi[k] = sla.solve_triangular(Li[k].T, sla.solve_triangular(Li[k], y, lower=True), lower=False)
        c = xave
        q = sum(float(np.sum((xk - c) ** 2)) for xk in xi)
        pack = allreduce(np.concatenate([sum(xi), sum(ui), [q]]))  # the one collective
        xprev, xave = xave, pack[:n] / N
        uave = pack[n:2 * n] / N
        v = uave + xave
        z = np.sign(v) * np.maximum(np.abs(v) - lam / (rho * N), 0.0)  # q11
        for k in range(len(slices)):
            ui[k] = ui[k] + (xi[k] - z)
        ubar_old, ubar = ubar, (uave + xave) - z  # altu: mean of the updated u_k
        dx2 = float(np.sum((xave - xprev) ** 2))
        pn = max(pack[2 * n] - N * dx2, 0.0)  # lassonorms: squared sums (q10)
        dn = N * rho ** 2 * dx2
        pe = math.sqrt(n) * abstol + reltol * max(math.sqrt(float(xave @ xave)), 0.0)  # ||Ax||, z = 0 (q9), c = 0
        de = math.sqrt(n) * abstol + reltol * (rho * math.sqrt(float(ubar @ ubar)))
        hn = rho * 0.0 + rho * (rho * rho) * float(np.sum((ubar - ubar_old) ** 2))  # z is identically zero (q9)
        for key, val in (("xvals", xave), ("uvals", ubar), ("pnorm", pn), ("dnorm", dn), ("perr", pe), ("derr", de),
                         ("Hnormsq", hn)):
            out[key].append(val)
        steps = i
        if not domaxiters:
            if pn < pe and dn < de:
                break
            if i > 2 and hn <= Hnormtol:
                break
    res = {k: np.asarray(v) for k, v in out.items()}
    res["xvals"] = np.asfortranarray(np.stack(out["xvals"], axis=1))
    res["uvals"] = np.asfortranarray(np.stack(out["uvals"], axis=1))
    res.update(steps=steps, xopt=xave, zconsensus=z)
    return res
