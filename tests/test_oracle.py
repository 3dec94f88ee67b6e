"""CPU: the two independent restatements of the reference algorithm agree bit for bit,
the hoisted forms equal the literal per-element forms, and the frozen vectors hold."""
import os

import numpy as np
import pytest

import workloads as Wk

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def eq(a, b):
    return np.array_equal(a, b, equal_nan=True)


CASES = {
    "cfg1": lambda: (*Wk.cfg1_inputs(), 8, 32, 1.0),
    "mid": lambda: (*Wk.bench_inputs(R=32, C=5, seed=11), 8, 64, 0.25),
    "ph11": lambda: (*Wk.bench_inputs(R=8, C=3, H=90, W=120, img=480, seed=5), 11, 77, 0.25),
    "batch3": lambda: (*Wk.bench_inputs(R=24, C=2, H=64, W=64, img=256, seed=6, batch=3), 8, 40, 0.25),
}


@pytest.mark.parametrize("name", list(CASES))
def test_c_equals_numpy_equals_literal(oracle, name):
    f, r, ph, pw, s = CASES[name]()
    out, geom = oracle.forward_c(f, r, ph, pw, s, return_geom=True)
    lit, ix, iy = oracle.forward_literal_c(f, r, ph, pw, s)
    npo, jx, jy = oracle.forward_np(f, r, ph, pw, s)
    assert eq(out, lit), "hoisted C forward != literal per-element forward"
    assert eq(out, npo), "C forward != numpy forward"
    assert eq(geom[..., 0], jx) and eq(geom[..., 1], jy)
    # the reference stores the same centre for every channel (kernel.cu:144-145)
    for c in range(f.shape[1]):
        assert eq(ix[:, c], jx) and eq(iy[:, c], jy)
    assert eq(oracle.forward_c(f, r, ph, pw, s, threads=4), out), "OpenMP changes bits"


def test_edge_and_degenerate_rois(oracle):
    rng = np.random.default_rng(1)
    f = rng.standard_normal((1, 3, 160, 160), dtype=np.float32)
    for rois in (Wk.edge_rois(), Wk.degenerate_rois()):
        out = oracle.forward_c(f, rois, 8, 64, 0.25)
        lit, ix, iy = oracle.forward_literal_c(f, rois, 8, 64, 0.25)
        npo, _, _ = oracle.forward_np(f, rois, 8, 64, 0.25)
        assert eq(out, lit) and eq(out, npo)
    # h == 0, w > 0 -> every bin active, all sample the map centre
    d = oracle.forward_c(f, Wk.degenerate_rois()[:1], 8, 64, 0.25)[0]
    centre = 0.25 * (f[0, :, 79, 79] + f[0, :, 79, 80] + f[0, :, 80, 80] + f[0, :, 80, 79])
    assert np.allclose(d, centre[:, None, None], atol=1e-6)
    # h < 0 -> mask false everywhere
    assert not oracle.forward_c(f, Wk.degenerate_rois()[3:4], 8, 64, 0.25).any()


def test_sample_points_are_half_integers(oracle):
    f, r = Wk.bench_inputs(R=64, C=1, seed=2)
    _, geom = oracle.forward_c(f, r, 8, 64, 0.25, return_geom=True)
    assert eq(geom * 2, np.round(geom * 2)), "SURVEY.md fact 1: centres are multiples of 0.5"
    # only one side of each clamp exists (kernel.cu:97-100): a box wholly left of the map
    # has rightMost < 0, so centres may be negative -- such taps simply fail the bounds


def test_row0_col0_never_read(oracle):
    """kernel.cu:116-126: taps need y > 0 and x > 0."""
    f, r = Wk.bench_inputs(R=64, C=2, seed=4)
    a = oracle.forward_c(f, r, 8, 64, 0.25)
    g = f.copy()
    g[:, :, 0, :] = 1e6
    g[:, :, :, 0] = -1e6
    assert eq(a, oracle.forward_c(g, r, 8, 64, 0.25))


def test_linearity_in_features(oracle):
    f, r = Wk.bench_inputs(R=16, C=2, seed=8)
    a = oracle.forward_c(f, r, 8, 64, 0.25)
    assert eq(oracle.forward_c(f * np.float32(4.0), r, 8, 64, 0.25), a * np.float32(4.0))


@pytest.mark.parametrize("name", ["cfg1", "mid", "batch3"])
def test_backward_forms_agree(oracle, name):
    f, r, ph, pw, s = CASES[name]()
    out, ix, iy = oracle.forward_literal_c(f, r, ph, pw, s)
    gout = (2 * out).astype(np.float32)
    lit = oracle.backward_literal_c(gout, r, ix, iy, f.shape, s)   # fp32, index order
    dbl = oracle.backward_c(gout, r, f.shape, s)                   # double accumulation
    npb = oracle.backward_np(gout, r, f.shape, s)
    assert eq(dbl, npb)
    tol = 1e-5 * max(1.0, float(np.abs(dbl).max()))
    assert np.abs(lit - dbl).max() <= tol


def test_backward_border_asymmetry(oracle):
    """kernel.cu:267-274 excludes the last row/column as well: backward is not the
    adjoint of forward at the border."""
    H = W = 32
    f = np.ones((1, 1, H, W), np.float32)
    rois = np.asarray([[0, 16, 16, 64, 64, 0]], np.float32)  # covers the whole map
    gout = np.ones((1, 1, 8, 8), np.float32)
    g = oracle.backward_c(gout, rois, f.shape, 1.0)
    assert not g[0, 0, 0, :].any() and not g[0, 0, :, 0].any()
    assert not g[0, 0, H - 1, :].any() and not g[0, 0, :, W - 1].any()


@pytest.mark.parametrize("name", ["oracle_cfg1", "oracle_mid", "oracle_edge"])
def test_frozen_vectors(oracle, name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    ph, pw = (int(v) for v in z["pooled"])
    out, geom = oracle.forward_c(z["features"], z["rois"], ph, pw, float(z["scale"]), return_geom=True)
    assert eq(out, z["out"]) and eq(geom, z["geom"])
    gin = oracle.backward_c((2.0 * np.nan_to_num(out)).astype(np.float32), z["rois"],
                            z["features"].shape, float(z["scale"]))
    assert eq(gin, z["grad_in"])
    npo, _, _ = oracle.forward_np(z["features"], z["rois"], ph, pw, float(z["scale"]))
    assert eq(npo, z["out"])


def test_touched_pixels_matches_survey(oracle):
    """~98.7 % of the 160x160 map is touched at the bench distribution."""
    _, r = Wk.bench_inputs(R=512, C=1)
    n = oracle.touched_pixels(r, 1, 160, 160, 8, 64, 0.25)
    assert 0.95 * 25600 < n <= 159 * 159


def test_threaded_backward_is_bit_identical_to_the_statement_order_form(oracle):
    """bench.py's CPU baseline for the backward: channel-partitioned OpenMP, same accumulation order per
    feature element -> the same bits for every thread count."""
    import workloads as Wk
    f, r = Wk.bench_inputs(R=24, C=13, H=40, W=56, img=224, seed=3, batch=2)
    rng = np.random.default_rng(0)
    g = rng.standard_normal((24, 13, 8, 32), dtype=np.float32)
    want = oracle.backward_c(g, r, f.shape, 0.25)
    for t in (1, 3, 8):
        assert np.array_equal(oracle.backward_c(g, r, f.shape, 0.25, threads=t), want), t
