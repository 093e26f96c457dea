"""BASELINE.json's full sizes on the GPU, checked through size-independent identities of the ADMM updates
(the oracle cannot run these sizes in seconds): every iterate must satisfy the defining equation of its update
-- the normal equations of the x-update, the closed form of the z-prox, the u-update, the residual norms and the
objective -- evaluated on the host with BLAS / O(n) stencils.  One warm-started iteration from a state reached
after K iterations gives (x', z', u') together with the (z, u) they were computed from."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-9


def _soft(v, t):
    return np.sign(v) * np.maximum(np.abs(v) - t, 0.0)


def _rel(a, b):
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


@pytest.fixture(scope="module")
def big(gpu):
    """Config 2: lassotest.m's recipe at 100000 x 10000 (8 GB)."""
    return gpu.synth.lasso_problem(seed=1, rows=100000, cols=10000)


def _warm_step(solver, base, state):
    """One more iteration from (x, z, u): returns the results of that single iteration."""
    o = dict(base, maxiters=1, domaxiters=1, x0=state["xopt"], z0=state["zopt"], u0=state["uopt"])
    return solver(o)


def test_lasso_100k_x_10k_identities(gpu, big):
    D, s, lam, rho = big["D"], big["s"], big["lam"], 1.0
    run = lambda o: gpu.lasso(D, s, lam, o)
    base = dict(rho=rho, objevals=1, xsolve="inverse")
    st = run(dict(base, maxiters=25, domaxiters=1, record_history=0))
    r = _warm_step(run, base, st)
    x, z, u = r["xopt"], r["zopt"], r["uopt"]
    z0, u0 = st["zopt"], st["uopt"]
    # x-update (getProxOps.m:1195-1200): (D'D + rho I) x = rho (z - u) + D's
    lhs = D.T @ (D @ x) + rho * x
    rhs = rho * (z0 - u0) + D.T @ s
    assert _rel(lhs, rhs) < RTOL
    # z-prox (getProxOps.m:455) and u-update (admm.m:548)
    assert _rel(z, _soft(x + u0, lam / rho)) < 1e-12
    assert _rel(u, u0 + x - z) < 1e-12
    # residual norms (admm.m:621-624) and the objective (lasso.m:227)
    assert r["pnorm"][0] == pytest.approx(np.linalg.norm(x - z), rel=1e-9)
    assert r["dnorm"][0] == pytest.approx(rho * np.linalg.norm(z - z0), rel=1e-9)
    assert r["objopt"] == pytest.approx(0.5 * np.sum((D @ x - s) ** 2) + lam * np.sum(np.abs(z)), rel=1e-10)
    # the planted signal is beaten (lassotest.m:143)
    obj = lambda v: 0.5 * np.sum((D @ v - s) ** 2) + lam * np.sum(np.abs(v))
    assert obj(x) < obj(big["testx"])


def test_lad_100k_x_10k_identities(gpu, big):
    """The A-streaming iteration at full size (lad.m on the same D, s)."""
    D, s, rho = big["D"], big["s"], 1.0
    run = lambda o: gpu.lad(D, s, o)
    base = dict(rho=rho, objevals=1, xsolve="inverse")
    st = run(dict(base, maxiters=6, domaxiters=1, record_history=0))
    r = _warm_step(run, base, st)
    x, z, u = r["xopt"], r["zopt"], r["uopt"]
    z0, u0 = st["zopt"], st["uopt"]
    # x-update (getProxOps.m:1511-1515): D'D x = D'(s + z - u)
    assert _rel(D.T @ (D @ x), D.T @ (s + z0 - u0)) < RTOL
    Dx = D @ x
    assert _rel(z, _soft(Dx + u0 - s, 1.0 / rho)) < 1e-10      # getProxOps.m:810
    assert _rel(u, u0 + Dx - z - s) < 1e-10                     # admm.m:548 with c = s
    assert r["pnorm"][0] == pytest.approx(np.linalg.norm(Dx - z - s), rel=1e-8)
    assert r["dnorm"][0] == pytest.approx(rho * np.linalg.norm(D.T @ (z - z0)), rel=1e-8)
    assert r["objopt"] == pytest.approx(np.sum(np.abs(z)), rel=1e-10)  # lad.m:148


def test_svm_60000_x_400_identities(gpu):
    """Config 3 at the full MNIST-shaped size (synthetic pixels; the image files are absent from the reference)."""
    q = gpu.synth.mnist_like_problem(seed=1, m=60000, n=400, digit=0,
                                     labels=gpu.synth.reference_mnist_labels("train"))  # the reference's label file
    D, ell, C, rho = q["D"], q["ell"], q["C"], 1.0
    run = lambda o: gpu.linearsvm(D, ell, C, o)
    st = run(dict(x0=q["x0"], z0=q["z0"], u0=q["u0"], record_history=0, maxiters=50, domaxiters=1))
    # unwrappedadmm.m:90-92 forces maxiters = 1000; a single warm iteration needs the engine directly
    r = gpu.admm(*gpu.getproxops("LinearSVM", dict(D=D, ell=ell, C=C, lossfunction="hinge"))[:2],
                 dict(A=D, At=D.T, B=-1, nB=60000, c=0, m=60000, maxiters=1, domaxiters=1, nodualerror=1,
                      stopcond="both", x0=st["xopt"], z0=st["zopt"], u0=st["uopt"]))
    x, z, u = r["xopt"], r["zopt"], r["uopt"]
    z0, u0 = st["zopt"], st["uopt"]
    assert _rel(D.T @ (D @ x), D.T @ (z0 - u0)) < 1e-8          # x = D^+ (z - u)   unwrappedadmm.m:78
    Dx = D @ x
    v = Dx + u0
    assert _rel(z, v + ell * np.maximum(np.minimum(1.0 - ell * v, C / rho), 0.0)) < 1e-10  # getProxOps.m:1096
    assert _rel(u, u0 + Dx - z) < 1e-10


def test_tv_4096sq_identities(gpu):
    """Config 5 (1-D signal of length 4096^2, as the reference's solver defines it): tridiagonal x-update."""
    n, rho = 4096 * 4096, 1.0
    p = gpu.synth.tv_problem(seed=1, n=n)
    s, lam = p["s"], p["lam"]
    run = lambda o: gpu.totalvariation(s, lam, o)
    base = dict(rho=rho, objevals=1)
    st = run(dict(base, maxiters=11, domaxiters=1, record_history=0))
    r = _warm_step(run, dict(base, record_history=0), st)
    x, z, u = r["xopt"], r["zopt"], r["uopt"]
    z0, u0 = st["zopt"], st["uopt"]
    Dv = lambda v: np.concatenate([v[:-1] - v[1:], v[-1:]])                      # D = spdiags([1 -1], 0:1, n, n)
    Dtv = lambda w: np.concatenate([w[:1], w[1:] - w[:-1]])                      # D'
    assert _rel(x + rho * Dtv(Dv(x)), s + rho * Dtv(z0 - u0)) < RTOL             # getProxOps.m:1047
    assert _rel(z, _soft(u0 + Dv(x), lam / rho)) < 1e-11                         # getProxOps.m:199
    assert _rel(u, u0 + Dv(x) - z) < 1e-11
    assert r["objopt"] == pytest.approx(0.5 * np.sum((x - s) ** 2) + lam * np.sum(np.abs(np.diff(x))), rel=1e-10)


@pytest.mark.parametrize("xsolve", ["auto", "cg"])
def test_tv2d_4096x4096_identities(gpu, xsolve):
    """Config 5 as literally written (image, matrix-free x-update): the spectral solve / the CG solve meet the
    x-update's defining equation."""
    H = W = 4096
    rng = np.random.default_rng(1)
    img = np.zeros((H, W))
    img[H // 5:H // 2, W // 6:W // 2] = 2.0
    img += rng.standard_normal((H, W))
    lam, rho = 1.0, 1.0
    run = lambda o: gpu.totalvariation2d(img, lam, dict(o, xsolve=xsolve))
    st = run(dict(maxiters=3, domaxiters=1, record_history=0))
    r = run(dict(maxiters=1, domaxiters=1, record_history=0, objevals=1, x0=st["xopt"].reshape(-1, order="F"),
                 z0=st["zopt"], u0=st["uopt"]))
    X = r["xopt"]
    N = H * W
    zv0, zh0 = st["zopt"][:N].reshape((H, W), order="F"), st["zopt"][N:].reshape((H, W), order="F")
    uv0, uh0 = st["uopt"][:N].reshape((H, W), order="F"), st["uopt"][N:].reshape((H, W), order="F")

    def D(Xm):
        dv = np.zeros_like(Xm)
        dh = np.zeros_like(Xm)
        dv[:-1, :] = Xm[:-1, :] - Xm[1:, :]
        dh[:, :-1] = Xm[:, :-1] - Xm[:, 1:]
        return dv, dh

    def Dt(wv, wh):
        out = np.zeros_like(wv)
        out[:-1, :] += wv[:-1, :]
        out[1:, :] -= wv[:-1, :]
        out[:, :-1] += wh[:, :-1]
        out[:, 1:] -= wh[:, :-1]
        return out

    dv, dh = D(X)
    lhs = X + rho * Dt(dv, dh)
    rhs = img + rho * Dt(zv0 - uv0, zh0 - uh0)
    assert np.linalg.norm(lhs - rhs) <= 1e-9 * np.linalg.norm(rhs)               # CG tolerance 1e-11, relative
    zv = r["zopt"][:N].reshape((H, W), order="F")
    assert _rel(zv, _soft(uv0 + dv, lam / rho)) < 1e-11
    assert r["objopt"] == pytest.approx(0.5 * np.sum((X - img) ** 2) + lam * (np.abs(dv).sum() + np.abs(dh).sum()),
                                        rel=1e-10)


def test_consensus_lasso_config4_full_size_as_8_local_slices(gpu, big):
    """BASELINE config 4 at full size -- D 100000 x 10000 split by slicemaker(0, 8, 100000) into 8 slices of
    12500 x 10000 (lasso.m:196-208) -- as 8 LOCAL slices of one engine on one GPU.  Checked through the identities
    of getProxOps.m:1217-1343 with the per-slice closure state (x_k, u_k) fetched from the device: two runs of i and
    i + 1 iterations (the loop is deterministic) give (u_k, z) before and (x_k, u_k, z) after iteration i + 1."""
    L = gpu._lib
    D, s, lam, rho = big["D"], big["s"], big["lam"], 1.0
    m, n = D.shape
    K = 8
    sl = gpu.errorcheck.slicemaker(0, K, m)
    assert list(sl) == [12500] * 8
    eng = gpu.Engine(L.PROB_LASSO_CONSENSUS, D=D, s=s, lam=lam, rho=rho, slices=sl, xsolve=L.XSOLVE_AUTO)
    try:
        assert eng.info()["xsolve_used"] == "inverse"
        i = 4
        kw = dict(rho=rho, domaxiters=1, stopcond="both", objevals=1)
        eng.run(maxiters=i, **kw)
        U0 = eng.fetch(L.F_CONS_U, n * K, (n, K))
        z0 = eng.fetch(L.F_ZCONSENSUS, n)
        xbar0 = eng.fetch(L.F_XOPT, n)
        eng.run(maxiters=i + 1, **kw)
        X1 = eng.fetch(L.F_CONS_X, n * K, (n, K))
        U1 = eng.fetch(L.F_CONS_U, n * K, (n, K))
        z1 = eng.fetch(L.F_ZCONSENSUS, n)
        xbar1 = eng.fetch(L.F_XOPT, n)
        ubar1 = eng.fetch(L.F_UOPT, n)
        pn = eng.fetch(L.F_PNORM, i + 1)
        dn = eng.fetch(L.F_DNORM, i + 1)
        obj = eng.fetch(L.F_OBJEVALS, i + 1)
        assert np.all(eng.fetch(L.F_ZOPT, n) == 0.0)  # q9: the z handed back to admm
    finally:
        eng.close()
    r0 = 0
    for k in range(K):  # x_k = (D_k'D_k + rho I) \\ (rho (z - u_k) + D_k's_k)   getProxOps.m:1240-1247
        Dk, sk = D[r0:r0 + sl[k]], s[r0:r0 + sl[k]]
        lhs = Dk.T @ (Dk @ X1[:, k]) + rho * X1[:, k]
        rhs = rho * (z0 - U0[:, k]) + Dk.T @ sk
        assert _rel(lhs, rhs) < RTOL, k
        r0 += sl[k]
    assert _rel(xbar1, X1.mean(axis=1)) < 1e-12                                  # getProxOps.m:1259
    assert _rel(z1, _soft(U0.mean(axis=1) + xbar1, lam / (rho * K))) < 1e-11     # 1286-1292, q11
    assert _rel(U1, U0 + X1 - z1[:, None]) < 1e-12                               # 1296-1298
    assert _rel(ubar1, U1.mean(axis=1)) < 1e-12                                  # altu, 1312-1326
    assert pn[i] == pytest.approx(np.sum((X1 - xbar1[:, None]) ** 2), rel=1e-8)  # lassonorms, squared (q10)
    assert dn[i] == pytest.approx(K * rho ** 2 * np.sum((xbar1 - xbar0) ** 2), rel=1e-8)
    assert obj[i] == pytest.approx(0.5 * np.sum((D @ xbar1 - s) ** 2), rel=1e-10)  # lasso.m:227 with z = 0 (q9)
