"""2-D anisotropic total variation (engine-side extension; BASELINE config 5 as literally written): the device
path (stencil kernels + matrix-free CG x-update) against the oracle's sparse-direct restatement.  No reference
counterpart exists (totalvariation.m is 1-D), so this parity is pinned by the oracle alone."""
import numpy as np
import pytest

from oracle import solvers_ref as S
from tests.test_gpu_parity import _close

pytestmark = pytest.mark.gpu


def _image(seed, H, W):
    rng = np.random.default_rng(seed)
    img = np.zeros((H, W))
    img[H // 5:H // 2, W // 6:W // 2] = 2.0
    img[H // 3:4 * H // 5, W // 3:5 * W // 6] += 1.0
    return img + 0.3 * rng.standard_normal((H, W))


@pytest.mark.parametrize("H,W,opts", [
    (24, 17, dict(objevals=1)), (33, 40, dict(objevals=1, rho=2.0)), (1, 50, dict(objevals=1)),
    (40, 1, dict()), (64, 64, dict(objevals=1, convtest=1, stopcond="both", maxiters=60)),
    (20, 30, dict(domaxiters=1, maxiters=25, nodualerror=1)),
])
def test_tv2d_matches_oracle(gpu, H, W, opts):
    img = _image(H * 100 + W, H, W)
    got = gpu.totalvariation2d(img, 0.5, dict(opts))
    ref = S.totalvariation2d(img, 0.5, dict(opts))
    assert got["steps"] == ref["steps"]
    for k in ("xvals", "zvals", "uvals", "pnorm", "dnorm", "perr", "derr", "objevals", "Hnormsq", "xopt", "zopt", "uopt"):
        if k in ref:
            _close(k, got[k], ref[k], 1e-7)
    assert got["xopt"].shape == (H, W)
    spectral = 8 <= H <= 4096  # (the column transform exists; the row stage takes any width)
    assert (got["cg_iters_total"] == 0) if spectral else (got["cg_iters_total"] >= got["steps"])


@pytest.mark.parametrize("H,W,rho", [(170, 110, 2381.0), (61, 75, 4958.0), (24, 17, 1.0), (33, 40, 300.0), (4095, 7, 50.0),
                                     (64, 1, 3.0), (256, 255, 900.0), (1000, 100, 1.0)])
def test_tv2d_exact_row_stage_for_any_width_and_rho(gpu, H, W, rho):
    """Widths below four times the Toeplitz stage's tap count, or a rho whose kernel decays over thousands of columns,
    on widths that are no power of two: the row systems are solved as they stand (dct.hip: tv2d_rows_thomas_kernel),
    no CG -- against the oracle's sparse-direct solve at the spectral path's tolerance."""
    rng = np.random.default_rng(H + W)
    img = rng.standard_normal((H, W)) + 2.0 * (rng.random((H, W)) > 0.7)
    o = dict(objevals=1, maxiters=8, domaxiters=1, rho=rho)
    got, ref = gpu.totalvariation2d(img, 0.6, dict(o)), S.totalvariation2d(img, 0.6, dict(o))
    assert got["steps"] == ref["steps"] and got["cg_iters_total"] == 0
    for k in ("xvals", "zvals", "uvals", "pnorm", "dnorm", "objevals", "xopt"):
        _close(k, got[k], ref[k], 1e-8)
    cg = gpu.totalvariation2d(img, 0.6, dict(o, maxiters=2, xsolve="cg"))  # (and the matrix-free path on the same input)
    assert cg["cg_iters_total"] > 0 and cg["cg_capped_updates"] == 0
    _close("xvals", cg["xvals"], ref["xvals"][:, :2], 1e-7)


@pytest.mark.parametrize("H,W,rho", [(6, 400, 2381.0), (7, 333, 4958.0)])
def test_tv2d_matrix_free_x_update_with_a_large_rho(gpu, H, W, rho):
    """No column transform for fewer than 8 rows: CG on I + rho*D'D, condition number up to 1 + 8 rho.  The DEFAULT
    iteration cap follows rho
    (engine_run_tv.hip: cg_solve_tv2d); with the fixed 500 the x-update stopped short and the iterates of 170 x 110 and
    258 x 265 images -- matrix-free then -- were off by 2e-6 resp. 2e-4 from the first iteration on (found by
    tests/sweeps/fuzz_solvers.py with FUZZ_RHO_WIDE=1 FUZZ_SIZE=8)."""
    rng = np.random.default_rng(H + W)
    img = rng.standard_normal((H, W)) + 2.0 * (rng.random((H, W)) > 0.7)
    o = dict(objevals=1, maxiters=6, domaxiters=1, rho=rho)
    got, ref = gpu.totalvariation2d(img, 0.6, dict(o)), S.totalvariation2d(img, 0.6, dict(o))
    assert got["steps"] == ref["steps"] and got["cg_iters_total"] > 500  # (past the fixed cap in the first x-update)
    for k in ("xvals", "zvals", "uvals", "pnorm", "dnorm", "objevals", "xopt"):
        _close(k, got[k], ref[k], 1e-7)


@pytest.mark.parametrize("H,W,rho", [(8, 8, 1.0), (16, 64, 1.0), (64, 32, 2.5), (128, 256, 0.7), (32, 8, 1.0),
                                     (512, 8, 1.0), (8, 1024, 1.3), (2048, 16, 1.0), (16, 4096, 0.5),
                                     (8192, 8, 1.0), (8, 8192, 2.0)])  # every radix plan; 8192: 128 KB of LDS
def test_tv2d_spectral_x_update(gpu, H, W, rho):
    """Power-of-two sides: (I + rho*D'D) is inverted through the 2-D DCT-II (dct.hip) -- against the oracle's
    sparse-direct solve, and against the engine's own CG path."""
    img = _image(H * 1000 + W, H, W)
    o = dict(objevals=1, rho=rho, maxiters=30, domaxiters=1)
    got = gpu.totalvariation2d(img, 0.4, dict(o))
    ref = S.totalvariation2d(img, 0.4, dict(o))
    cg = gpu.totalvariation2d(img, 0.4, dict(o, xsolve="cg"))
    assert got["cg_iters_total"] == 0 and cg["cg_iters_total"] >= 30
    for k in ("xvals", "zvals", "uvals", "pnorm", "dnorm", "perr", "derr", "objevals", "xopt", "zopt", "uopt"):
        _close(k, got[k], ref[k], 1e-9)
        _close(k, cg[k], ref[k], 1e-7)


def test_tv2d_denoises(gpu):
    """The pass criterion of totalvariationtest.m:151 carried over: ADMM's objective beats the clean image's."""
    H, W = 96, 80
    rng = np.random.default_rng(3)
    clean = np.zeros((H, W))
    clean[20:60, 10:50] = 3.0
    clean[40:90, 30:70] += 2.0
    img = clean + rng.standard_normal((H, W))
    lam = 1.0
    tv = lambda X: np.sum(np.abs(np.diff(X, axis=0))) + np.sum(np.abs(np.diff(X, axis=1)))
    obj = lambda X: 0.5 * np.sum((X - img) ** 2) + lam * tv(X)
    r = gpu.totalvariation2d(img, lam, dict(objevals=1, maxiters=2000, record_history=0))
    assert obj(r["xopt"]) < obj(clean)
    assert r["objopt"] == pytest.approx(obj(r["xopt"]), rel=1e-9)


@pytest.mark.parametrize("H,W", [(5, 17), (24, 17), (32, 64)])  # CG x-update / exact row stage / Toeplitz row stage
@pytest.mark.parametrize("opts", [dict(fast=1, fasttype="strong", objevals=1, maxiters=60),
                                  dict(fast=1, fasttype="weak", objevals=1, maxiters=25),
                                  dict(fast=1, fasttype="strong", stopcond="both", rho=2.0, maxiters=40)])
def test_tv2d_fast_admm(gpu, H, W, opts):
    """fast / accelerated ADMM (admm.m:267-298, 563-600) on the image solver: the x-update takes (v, uhat), the generic
    fused kernel does z, u, v, uhat, the D' stencils come from dz and u.  Accelerated ADMM is compared while its
    restart value is far from rounding noise (its decisions are knife-edge below that)."""
    img = _image(H * 100 + W + 1, H, W)
    got = gpu.totalvariation2d(img, 0.5, dict(opts))
    ref = S.totalvariation2d(img, 0.5, dict(opts))
    assert got["steps"] == ref["steps"]
    keys = ("xvals", "zvals", "uvals", "vvals", "uhatvals", "avals", "dvals", "restarted", "pnorm", "dnorm", "perr", "derr",
            "objevals", "xopt", "zopt", "uopt")
    for k in keys:
        assert (k in got) == (k in ref), k
        if k in ref:
            _close(k, got[k], ref[k], 1e-6)


@pytest.mark.parametrize("H,W,rho,spectral", [(32, 200, 1.0, True), (64, 334, 0.5, True), (16, 1000, 2.0, True),
                                              (32, 100, 1.0, True), (32, 300, 10.0, True), (64, 333, 0.5, True)])
def test_tv2d_any_width_when_the_height_is_a_power_of_two(gpu, H, W, rho, spectral):
    """the row stage of the spectral solve is the Toeplitz kernel of the row operator (no row transform), so only the
    height has to be a power of two -- while the kernel's truncation (42 terms per side at rho = 1) is well inside the
    width (an odd width's last column is both halves of its column pair); otherwise the exact tridiagonal row stage runs"""
    img = _image(H + W, H, W)
    o = dict(objevals=1, rho=rho, maxiters=30)
    got = gpu.totalvariation2d(img, 0.5, dict(o))
    ref = S.totalvariation2d(img, 0.5, dict(o))
    assert got["steps"] == ref["steps"]
    for k in ("xvals", "zvals", "uvals", "pnorm", "dnorm", "perr", "derr", "objevals", "xopt", "zopt", "uopt"):
        _close(k, got[k], ref[k], 1e-7)
    assert (got["cg_iters_total"] == 0) == spectral


@pytest.mark.parametrize("H,W,rho", [(24, 64, 1.0), (100, 128, 0.6), (17, 256, 1.0), (1000, 240, 1.5), (3000, 64, 1.0),
                                     (4095, 250, 2.0), (2049, 256, 1.0), (640, 480, 1.0), (9, 256, 1.0)])
def test_tv2d_any_height_through_the_chirp_transform(gpu, H, W, rho):
    """a height that is not a power of two: the column DCT as a circular convolution of length 2^p >= 2H - 1 (Bluestein,
    dct.hip: dct_cols_*_chirp_kernel) -- the spectral x-update for any image up to 4096 rows, against the oracle's
    sparse-direct solve; odd heights (columns at 8-byte alignment) included.  (The row stage keeps its own conditions:
    the Toeplitz kernel needs 4 x its truncation <= W, the row transform a power-of-two width and an even height.)"""
    img = _image(7 * H + W, H, W)
    o = dict(objevals=1, rho=rho, maxiters=12, domaxiters=1)
    got = gpu.totalvariation2d(img, 0.4, dict(o))
    ref = S.totalvariation2d(img, 0.4, dict(o))
    assert got["steps"] == ref["steps"] and got["cg_iters_total"] == 0
    for k in ("xvals", "zvals", "uvals", "pnorm", "dnorm", "perr", "derr", "objevals", "xopt", "zopt", "uopt"):
        _close(k, got[k], ref[k], 1e-9)
    stop = gpu.totalvariation2d(img, 0.4, dict(objevals=1, rho=rho, stopcond="both", maxiters=80))
    rstop = S.totalvariation2d(img, 0.4, dict(objevals=1, rho=rho, stopcond="both", maxiters=80))
    assert stop["steps"] == rstop["steps"]
    _close("xopt", stop["xopt"], rstop["xopt"], 1e-9)


@pytest.mark.parametrize("H,W,opts", [(256, 255, dict(objevals=1, maxiters=14, domaxiters=1)),           # glued kernel
                                      (64, 513, dict(objevals=1, stopcond="both", maxiters=80)),
                                      (100, 333, dict(objevals=1, maxiters=10, domaxiters=1)),            # chirp transform
                                      (2048, 201, dict(maxiters=6, domaxiters=1, record_history=0))])
def test_tv2d_odd_widths(gpu, H, W, opts):
    """the column kernels transform two real columns as one complex sequence; an odd width's last column is both halves
    of its pair (same values written twice) -- spectral x-update, against the oracle"""
    img = _image(H + 11 * W, H, W)
    got = gpu.totalvariation2d(img, 0.5, dict(opts))
    ref = S.totalvariation2d(img, 0.5, {k: v for k, v in opts.items() if k != "record_history"})
    assert got["steps"] == ref["steps"] and got["cg_iters_total"] == 0
    keys = ("xopt", "zopt", "uopt", "pnorm", "dnorm", "perr", "derr") + (("objevals",) if opts.get("objevals") else ())
    for k in keys + (("xvals", "zvals", "uvals") if opts.get("record_history", 1) else ()):
        _close(k, got[k], ref[k], 1e-9)


@pytest.mark.parametrize("H,W,iters", [(64, 64, 1), (64, 64, 2), (64, 64, 7), (40, 33, 2), (128, 96, 12)])
def test_tv2d_compact_state_with_arbitrary_start(gpu, H, W, iters):
    """The fused pass carries v = z + u; z0, u0 that do not satisfy z0 = soft(z0 + u0) are read as given by the first
    iteration, and the iterates handed back are expanded from the last state (spectral and CG x-updates)."""
    img = _image(H + 7 * W + iters, H, W)
    rng = np.random.default_rng(iters)
    o = dict(objevals=1, maxiters=iters, domaxiters=1, x0=rng.standard_normal(H * W), z0=rng.standard_normal(2 * H * W),
             u0=0.3 * rng.standard_normal(2 * H * W))
    ref = S.totalvariation2d(img, 0.5, dict(o))
    for extra in (dict(), dict(record_history=0)):
        got = gpu.totalvariation2d(img, 0.5, dict(o, **extra))
        assert got["steps"] == ref["steps"] == iters
        keys = ("xopt", "zopt", "uopt", "pnorm", "dnorm", "perr", "derr") + (() if extra else ("xvals", "zvals", "uvals"))
        for k in keys:
            _close(k, got[k], ref[k], 1e-7)


@pytest.mark.parametrize("H,W,opts", [(64, 200, dict(objevals=1, maxiters=30, domaxiters=1)),
                                      (1024, 640, dict(objevals=1, maxiters=19, domaxiters=1)),
                                      (256, 256, dict(objevals=1, stopcond="both", maxiters=200)),       # stops early
                                      (128, 512, dict(maxiters=21, domaxiters=1, record_history=0)),
                                      (512, 384, dict(objevals=1, rho=2.0, maxiters=12, domaxiters=1)),
                                      (8192, 192, dict(objevals=1, maxiters=5, domaxiters=1))])          # 128 KB of LDS
def test_tv2d_fused_pass_into_the_forward_transform(gpu, H, W, opts, monkeypatch):
    """default spectral form: the fused pass hands its right-hand side to the forward column DCT inside one kernel
    (dct.hip: tv2d_fused_dct_kernel; three launches per iteration, the row stage carries the deferred finalize) --
    the same arithmetic per pixel as the four-launch form (ADMM_HIP_TV2D_NO_GLUE=1): the same iterates bit for bit"""
    img = _image(3 * H + W, H, W)
    got = gpu.totalvariation2d(img, 0.45, dict(opts))
    monkeypatch.setenv("ADMM_HIP_TV2D_NO_GLUE", "1")
    old = gpu.totalvariation2d(img, 0.45, dict(opts))
    assert got["steps"] == old["steps"] and got["cg_iters_total"] == 0
    keys = ["xopt", "zopt", "uopt", "pnorm", "dnorm", "perr", "derr"]
    if opts.get("record_history", 1):
        keys += ["xvals", "zvals", "uvals"]
    if opts.get("objevals"):
        keys += ["objevals"]
    for k in keys:
        if k in ("xopt", "zopt", "uopt", "xvals", "zvals", "uvals"):
            assert np.array_equal(np.asarray(got[k]), np.asarray(old[k])), k
        else:  # sums over the image: the two kernels cut it into different blocks
            np.testing.assert_allclose(np.asarray(got[k]), np.asarray(old[k]), rtol=1e-12, atol=0, err_msg=k)
    if H * W <= 1 << 18:
        ref = S.totalvariation2d(img, 0.45, {k: v for k, v in opts.items() if k != "record_history"})
        assert got["steps"] == ref["steps"]
        for k in ("xopt", "zopt", "uopt", "pnorm", "dnorm"):
            _close(k, got[k], ref[k], 1e-8)


def test_tv2d_relaxation_is_a_dimension_error(gpu):
    with pytest.raises(Exception, match="dimension error"):
        gpu.totalvariation2d(_image(3, 16, 16), 0.5, dict(relax=1.5))


def test_tv2d_argument_errors(gpu):
    with pytest.raises(ValueError, match="not an image"):
        gpu.totalvariation2d(np.zeros(10), 1.0, {})
    with pytest.raises(ValueError, match="nonnegative"):
        gpu.totalvariation2d(np.zeros((4, 4)), -1.0, {})
