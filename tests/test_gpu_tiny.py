"""Degenerate sizes: one column, one row, 1 x 1 -- every kernel's guards against its smallest inputs."""
import numpy as np
import pytest

from oracle import solvers_ref as S
from tests.test_gpu_parity import HIST, _compare

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows,cols", [(1, 1), (3, 1), (2, 2), (5, 3), (1, 4), (2, 7)])
@pytest.mark.parametrize("xsolve", ["trsv", "inverse"])
def test_lasso_tiny(gpu, rows, cols, xsolve):
    rng = np.random.default_rng(rows * 10 + cols)
    D = np.asfortranarray(rng.standard_normal((rows, cols)))
    s = rng.standard_normal(rows)
    lam = 0.1 * float(np.max(np.abs(D.T @ s)))
    o = dict(objevals=1, maxiters=40)
    _compare(gpu.lasso(D, s, lam, dict(o, xsolve=xsolve)), S.lasso(D, s, lam, o), tol=1e-7)


@pytest.mark.parametrize("rows,cols", [(2, 1), (5, 1), (4, 2), (9, 3)])
def test_lad_and_huber_tiny(gpu, rows, cols):
    rng = np.random.default_rng(rows * 10 + cols)
    D = np.asfortranarray(rng.standard_normal((rows, cols)))
    s = rng.standard_normal(rows)
    o = dict(objevals=1, maxiters=40)
    _compare(gpu.lad(D, s, dict(o)), S.lad(D, s, dict(o)), tol=1e-7)
    got, ref = gpu.huberfit(D, s, dict(o)), S.huberfit(D, s, dict(o))
    # with so few rows z can stop moving altogether: dnorm is then rounding noise (1e-16) on both sides
    keys = tuple(k for k in HIST if k != "dnorm")
    _compare(got, ref, keys=keys, tol=1e-7)
    assert np.max(np.abs(got["dnorm"] - ref["dnorm"])) < 1e-12 * max(1.0, np.max(np.abs(ref["pnorm"])))


@pytest.mark.parametrize("n", [2, 3, 5])
def test_totalvariation_tiny(gpu, n):
    rng = np.random.default_rng(n)
    s = rng.standard_normal(n)
    o = dict(objevals=1, maxiters=40)
    _compare(gpu.totalvariation(s, 0.3, dict(o)), S.totalvariation(s, 0.3, dict(o)), tol=1e-7)


def test_model_and_qp_tiny(gpu):
    rng = np.random.default_rng(0)
    P, Q = rng.standard_normal((3, 1)), rng.standard_normal((3, 1))
    r, s = rng.standard_normal(3), rng.standard_normal(3)
    o = dict(objevals=1, maxiters=30)
    _compare(gpu.model(P, Q, r, s, dict(o)), S.model(P, Q, r, s, dict(o)), tol=1e-7)
    Pm = np.array([[2.0]])
    got = gpu.quadraticprogram(Pm, np.array([1.0]), 0.5, np.array([-1.0]), np.array([1.0]), dict(o))
    ref = S.quadraticprogram_bounded(Pm, np.array([1.0]), 0.5, np.array([-1.0]), np.array([1.0]), dict(o))
    _compare(got, ref, tol=1e-7)
