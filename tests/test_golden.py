"""CPU: the oracle reproduces every committed golden fixture (tests/golden/*.npz, produced by
tests/golden/make_golden.py).  Guards the checker itself against drift."""
import glob
import os

import numpy as np
import pytest

from oracle import solvers_ref as S

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FILES = sorted(glob.glob(os.path.join(GOLDEN, "*.npz")))


def _opts(npz):
    o = {}
    for k in npz.files:
        if k.startswith("opt_"):
            v = npz[k]
            o[k[4:]] = v.item() if v.ndim == 0 else v
    return o


def test_fixture_inventory():
    names = {os.path.basename(f) for f in FILES}
    for want in ("lasso_tall_256x64.npz", "lasso_fat_32x256.npz", "lad_512x64.npz", "lad_512x64_relax.npz",
                 "huber_512x64.npz", "svm_hinge_256x2.npz", "svm_01_256x2.npz", "qp_bounded_128.npz",
                 "lasso_fast_weak.npz", "lasso_fast_strong.npz", "lasso_tall_relax.npz", "basispursuit_32x96.npz",
                 "model_plain_200.npz", "model_fast_weak_200.npz", "model_fast_strong_200.npz",
                 "consensus_lasso_4x64.npz", "lp_32x96.npz", "qpstd_24x80.npz"):
        assert want in names


@pytest.mark.parametrize("path", FILES, ids=os.path.basename)
def test_oracle_reproduces_fixture(path):
    z = np.load(path, allow_pickle=False)
    name = os.path.basename(path)
    o = _opts(z)
    inp = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
    if name.startswith("consensus"):
        workers = int(o.pop("workers"))
        r = S.lasso(inp["D"], inp["s"], float(inp["lam"]), o, workers=workers)
    elif name.startswith("lp"):
        r = S.linearprogram(inp["b"], inp["D"], inp["s"], o)
    elif name.startswith("qpstd"):
        r = S.quadraticprogram_standard(inp["P"], inp["q"], float(inp["r"]), inp["D"], inp["s"], o)
    elif name.startswith("lasso"):
        r = S.lasso(inp["D"], inp["s"], float(inp["lam"]), o)
    elif name.startswith("lad"):
        r = S.lad(inp["D"], inp["s"], o)
    elif name.startswith("huber"):
        r = S.huberfit(inp["D"], inp["s"], o)
    elif name.startswith("svm"):
        r = S.linearsvm(inp["D"], inp["ell"], float(inp["C"]), o)
    elif name.startswith("qp"):
        r = S.quadraticprogram_bounded(inp["P"], inp["q"], float(inp["r"]), inp["lb"], inp["ub"], o)
    elif name.startswith("tv"):
        r = S.totalvariation(inp["s"], float(inp["lam"]), o)
    elif name.startswith("model"):
        r = S.model(inp["P"], inp["Q"], inp["r"], inp["s"], o)
    else:
        r = S.basispursuit(inp["D"], inp["s"], o)
    assert r["steps"] == int(z["steps"])
    for k in z.files:
        if not k.startswith("out_") or k == "out_objopt":
            continue
        ref = z[k]
        got = np.asarray(r[k[4:]])[..., :ref.shape[-1]] if ref.ndim else np.asarray(r[k[4:]])
        np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-12, equal_nan=True)
