"""The opt-in hipGraph replay of iteration batches (ADMM_HIP_GRAPH=1) gives the same run as eager launches."""
import numpy as np
import pytest

from oracle import solvers_ref as S
from tests.test_gpu_parity import _compare

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("xsolve", ["trsv", "inverse"])
def test_graph_replay_matches_oracle(gpu, monkeypatch, xsolve):
    monkeypatch.setenv("ADMM_HIP_GRAPH", "1")
    p = gpu.synth.lasso_problem(2, 300, 150)
    o = dict(objevals=0, maxiters=40)
    _compare(gpu.lasso(p["D"], p["s"], p["lam"], dict(o, xsolve=xsolve)), S.lasso(p["D"], p["s"], p["lam"], o))


def test_graph_replay_stops_early_like_eager(gpu, monkeypatch):
    p = gpu.synth.svm_problem(0, 64, 64)
    o = dict(x0=p["x0"], z0=p["z0"], u0=p["u0"])
    eager = gpu.linearsvm(p["D"], p["ell"], p["C"], dict(o))
    monkeypatch.setenv("ADMM_HIP_GRAPH", "1")
    graph = gpu.linearsvm(p["D"], p["ell"], p["C"], dict(o))
    assert graph["steps"] == eager["steps"]
    np.testing.assert_array_equal(graph["xopt"], eager["xopt"])  # same kernels, same order: bitwise
