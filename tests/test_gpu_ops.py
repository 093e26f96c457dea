"""GPU parity of the stand-alone HIP operators (through the C ABI) against NumPy/SciPy fp64.
Tolerance: 1e-11 relative to the natural scale of each result (summation order differs)."""
import ctypes as C

import numpy as np
import pytest
import scipy.linalg as sla

pytestmark = pytest.mark.gpu


def _dp(ap, a):
    return ap._lib.as_dp(a)


def _rel(a, b):
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


@pytest.mark.parametrize("m,n", [(1, 1), (7, 3), (64, 64), (257, 33), (1000, 129), (4099, 70), (513, 1100)])
def test_gemv_n(gpu, m, n):
    rng = np.random.default_rng(m * 1000 + n)
    D = np.asfortranarray(rng.standard_normal((m, n)))
    x = rng.standard_normal(n)
    y = np.zeros(m)
    gpu._lib.check(gpu._lib.load().admm_op_gemv_n(_dp(gpu, D), m, n, m, _dp(gpu, x), _dp(gpu, y)))
    assert _rel(y, D @ x) < 1e-12


@pytest.mark.parametrize("m,n,nrhs", [(1, 1, 1), (7, 3, 2), (64, 64, 3), (257, 33, 3), (5000, 37, 1),
                                      (4099, 70, 3), (513, 1100, 2), (9001, 5, 3)])
def test_gemv_t(gpu, m, n, nrhs):
    rng = np.random.default_rng(m * 1000 + n + nrhs)
    D = np.asfortranarray(rng.standard_normal((m, n)))
    V = np.asfortranarray(rng.standard_normal((m, nrhs)))
    G = np.zeros((n, nrhs), order="F")
    gpu._lib.check(gpu._lib.load().admm_op_gemv_t(_dp(gpu, D), m, n, m, _dp(gpu, V), m, nrhs, _dp(gpu, G), n))
    assert _rel(G, D.T @ V) < 1e-12


def test_gemv_respects_leading_dimension(gpu):
    rng = np.random.default_rng(0)
    big = np.asfortranarray(rng.standard_normal((50, 9)))
    D = big[:37, :]  # ld = 50 > m = 37
    x = rng.standard_normal(9)
    y = np.zeros(37)
    gpu._lib.check(gpu._lib.load().admm_op_gemv_n(_dp(gpu, big), 37, 9, 50, _dp(gpu, x), _dp(gpu, y)))
    assert _rel(y, D @ x) < 1e-12


@pytest.mark.parametrize("m,n", [(5, 3), (200, 64), (300, 130), (1000, 257), (129, 129)])
def test_gram(gpu, m, n):
    rng = np.random.default_rng(m + n)
    D = np.asfortranarray(rng.standard_normal((m, n)))
    W = np.zeros((n, n), order="F")
    gpu._lib.check(gpu._lib.load().admm_op_gram(_dp(gpu, D), m, n, m, 0.75, _dp(gpu, W)))
    ref = D.T @ D + 0.75 * np.eye(n)
    assert _rel(W, ref) < 1e-12
    np.testing.assert_array_equal(W, W.T)  # mirrored, not recomputed


@pytest.mark.parametrize("n", [1, 5, 64, 65, 127, 200, 513])
def test_cholesky(gpu, n):
    rng = np.random.default_rng(n)
    G = rng.standard_normal((n + 10, n))
    A = np.asfortranarray(G.T @ G + np.eye(n))
    ref = sla.cholesky(A, lower=True)
    gpu._lib.check(gpu._lib.load().admm_op_cholesky(_dp(gpu, A), n, n))
    assert _rel(A, ref) < 1e-11
    assert np.all(np.triu(A, 1) == 0.0)


def test_cholesky_rejects_indefinite(gpu):
    A = np.asfortranarray(np.diag([1.0, 2.0, -1.0, 3.0]))
    rc = gpu._lib.load().admm_op_cholesky(_dp(gpu, A), 4, 4)
    assert rc == gpu._lib.E_NUMERIC
    assert b"positive definite" in gpu._lib.load().admm_last_error()


@pytest.mark.parametrize("n", [1, 3, 63, 64, 65, 127, 128, 129, 130, 400, 1000, 1280, 1281, 1409, 2049, 2560, 2700, 3333, 4100,
                               6500, 10000])
def test_trsv_pair(gpu, n):
    """blocked substitution: one coarse block up to n = 2048, several beyond (ragged last block included; from two blocks
    on the pair runs as ONE launch, trsv.hip: tri_persist_kernel -- 5 blocks at n = 10000);
    the strictly-upper part of the factor buffer holds garbage, as after an in-place Cholesky of a full matrix"""
    rng = np.random.default_rng(n)
    G = rng.standard_normal((n + 20, n)) / np.sqrt(n + 20)
    Lf = np.asfortranarray(sla.cholesky(G.T @ G + np.eye(n), lower=True))
    Lf = np.asfortranarray(Lf + np.triu(rng.standard_normal((n, n)), 1))
    y = rng.standard_normal(n)
    x = np.zeros(n)
    gpu._lib.check(gpu._lib.load().admm_op_trsv_pair(_dp(gpu, Lf), n, n, _dp(gpu, y), _dp(gpu, x)))
    Lc = np.tril(Lf)
    ref = sla.solve_triangular(Lc.T, sla.solve_triangular(Lc, y, lower=True), lower=False)
    assert _rel(x, ref) < 1e-11


def test_soft_threshold(gpu):
    v = np.array([-2.0, -1.0, -0.5, 0.0, 0.5, 1.0, 2.0, 1e-300, -1e300])
    out = np.zeros_like(v)
    gpu._lib.check(gpu._lib.load().admm_op_soft_threshold(_dp(gpu, v), v.size, 1.0, _dp(gpu, out)))
    np.testing.assert_array_equal(out, np.sign(v) * np.maximum(np.abs(v) - 1.0, 0.0))


def test_bad_arguments_are_errors_not_crashes(gpu):
    lib = gpu._lib.load()
    assert lib.admm_op_gemv_n(None, 4, 4, 4, None, None) == gpu._lib.E_INVALID
    d = gpu._lib.ProblemDesc()
    lib.admm_problem_desc_default(C.byref(d))
    d.problem = 999
    h = C.c_void_p()
    assert lib.admm_engine_create(C.byref(d), C.byref(h)) == gpu._lib.E_INVALID
    d.problem = gpu._lib.PROB_LASSO  # no data pointers
    assert lib.admm_engine_create(C.byref(d), C.byref(h)) == gpu._lib.E_INVALID
    d.struct_size = 8
    assert lib.admm_engine_create(C.byref(d), C.byref(h)) == gpu._lib.E_INVALID


@pytest.mark.parametrize("n", [2049, 2700, 4100, 10000])
def test_trsv_pair_one_launch_form(gpu, n, monkeypatch):
    """the same pair as ONE persistent launch (trsv.hip: tri_persist_kernel; opt-in, slower than the stepwise launches):
    tickets, write-through hand-offs between workgroups, row-tile counters -- same sums in the same order"""
    monkeypatch.setenv("ADMM_TRSV_ONE_LAUNCH", "1")
    rng = np.random.default_rng(n)
    G = rng.standard_normal((n + 20, n)) / np.sqrt(n + 20)
    Lf = np.asfortranarray(sla.cholesky(G.T @ G + np.eye(n), lower=True))
    y = rng.standard_normal(n)
    x1, x2 = np.zeros(n), np.zeros(n)
    gpu._lib.check(gpu._lib.load().admm_op_trsv_pair(_dp(gpu, Lf), n, n, _dp(gpu, y), _dp(gpu, x1)))
    monkeypatch.delenv("ADMM_TRSV_ONE_LAUNCH")
    gpu._lib.check(gpu._lib.load().admm_op_trsv_pair(_dp(gpu, Lf), n, n, _dp(gpu, y), _dp(gpu, x2)))
    ref = sla.solve_triangular(Lf.T, sla.solve_triangular(Lf, y, lower=True), lower=False)
    assert _rel(x1, ref) < 1e-11 and _rel(x2, ref) < 1e-11
    # the same tiles and panels; a tile's columns are split over 4 waves here and over 1..16 in the stepwise launches
    # (chosen per step from the grid size), so the sums associate differently: equal to rounding, not bitwise
    assert _rel(x1, x2) < 1e-13


@pytest.mark.parametrize("n", [256, 300, 1000, 2049, 3333, 10000])
def test_trsv_pair_one_block_form(gpu, n, monkeypatch):
    """the triangular solves with the whole factor as ONE pre-inverted block (symv.hip: tri1_*): w = X y by the N-part
    pass + fold, x = X' w by the T-part pass + reduce; against LAPACK's substitution, against the blocked form, and
    twice in a row (fixed-order sums: bitwise the same)"""
    rng = np.random.default_rng(n + 7)
    G = rng.standard_normal((n + 20, n)) / np.sqrt(n + 20)
    Lf = np.asfortranarray(sla.cholesky(G.T @ G + np.eye(n), lower=True))
    Lf = np.asfortranarray(Lf + np.triu(rng.standard_normal((n, n)), 1))  # garbage above the diagonal is ignored
    y = rng.standard_normal(n)
    xb, x1, x2 = np.zeros(n), np.zeros(n), np.zeros(n)
    lib = gpu._lib.load()
    monkeypatch.setenv("ADMM_TRSV_FORM", "blocked")
    gpu._lib.check(lib.admm_op_trsv_pair(_dp(gpu, Lf), n, n, _dp(gpu, y), _dp(gpu, xb)))
    monkeypatch.setenv("ADMM_TRSV_FORM", "one")
    gpu._lib.check(lib.admm_op_trsv_pair(_dp(gpu, Lf), n, n, _dp(gpu, y), _dp(gpu, x1)))
    gpu._lib.check(lib.admm_op_trsv_pair(_dp(gpu, Lf), n, n, _dp(gpu, y), _dp(gpu, x2)))
    Lc = np.tril(Lf)
    ref = sla.solve_triangular(Lc.T, sla.solve_triangular(Lc, y, lower=True), lower=False)
    assert _rel(x1, ref) < 1e-11 and _rel(xb, ref) < 1e-11
    assert np.array_equal(x1, x2)
