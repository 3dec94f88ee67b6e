"""Seeded synthetic inputs shared by the tests."""
import numpy as np


def bench_inputs(R=512, C=256, H=160, W=160, img=640, seed=0, all_active=False,
                 axis_aligned=False, batch=1):
    """BASELINE.json configs[1..3]: features ~ N(0,1); ROIs in img x img pixels,
    cx,cy ~ U[0,img), h ~ U[16,64), w = h*U[4,8), angle ~ U[-90,90)."""
    rng = np.random.default_rng(seed)
    feats = rng.standard_normal((batch, C, H, W), dtype=np.float32)
    cx = rng.uniform(0, img, R)
    cy = rng.uniform(0, img, R)
    h = rng.uniform(16, 64, R)
    ratio = np.full(R, 8.0) if all_active else rng.uniform(4, 8, R)
    w = h * ratio
    ang = np.zeros(R) if axis_aligned else rng.uniform(-90, 90, R)
    bidx = rng.integers(0, batch, R) if batch > 1 else np.zeros(R)
    rois = np.stack([bidx, cx, cy, h, w, ang], 1).astype(np.float32)
    return feats, rois


def cfg1_inputs(seed=0):
    """BASELINE.json configs[0]: 1x3x64x128 map, 4 rotated ROIs, pooled 8x32, scale 1."""
    rng = np.random.default_rng(seed)
    feats = rng.standard_normal((1, 3, 64, 128), dtype=np.float32)
    R = 4
    rois = np.stack([np.zeros(R), rng.uniform(0, 128, R), rng.uniform(0, 64, R),
                     rng.uniform(6, 20, R), rng.uniform(20, 80, R), rng.uniform(-90, 90, R)],
                    1).astype(np.float32)
    return feats, rois


def edge_rois(img_w=640, img_h=640):
    """ROIs that stress the reference's corner cases: outside the map on every side,
    straddling each border, 0/+-45/+-90/180 degrees, w/h < 1 and > 8, integer and
    half-integer centres (round() ties), batch index with a fractional part."""
    rows = []
    for ang in (0.0, 45.0, -45.0, 90.0, -90.0, 180.0, 30.0, -135.0):
        rows.append([0, img_w / 2, img_h / 2, 32, 200, ang])
    for cx, cy in ((-40, 100), (img_w + 40, 100), (100, -40), (100, img_h + 40),
                   (2, 2), (img_w - 2, img_h - 2), (0, 0), (img_w, img_h),
                   (-500, -500), (5000, 5000)):
        rows.append([0, cx, cy, 24, 150, 17.0])
    rows += [
        [0, 320, 320, 64, 32, 10],      # w/h < 1
        [0, 320, 320, 8, 200, -20],     # w/h = 25 >> pooled width ratio
        [0, 320, 320, 16, 128, 0],      # exactly w/h = 8
        [0, 320.5, 320.5, 16, 128, 0],  # half-integer centre
        [0, 100, 200, 40, 160, 90],     # ties at 90 degrees
        [0, 128, 256, 32, 256, 0],      # integer grid, scale .25 -> bin edges on integers
        [0.9, 300, 300, 20, 100, 5],    # batch index truncates to 0
        [0, 320, 320, 1, 4, 3],         # tiny
        [0, 320, 320, 600, 2400, 33],   # much larger than the map
    ]
    return np.asarray(rows, np.float32)


def degenerate_rois():
    """h == 0, w == 0, negative h, NaN / inf fields (SURVEY.md section 7 'Degenerate ROIs')."""
    nan, inf = float("nan"), float("inf")
    return np.asarray([
        [0, 320, 320, 0, 100, 10],
        [0, 320, 320, 20, 0, 10],
        [0, 320, 320, 0, 0, 10],
        [0, 320, 320, -20, 100, 10],
        [0, 320, 320, 20, -100, 10],
        [0, 320, 320, 20, 100, nan],
        [0, nan, 320, 20, 100, 10],
        [0, 320, inf, 20, 100, 10],
        [0, 320, 320, inf, 100, 10],
        [0, 320, 320, 20, inf, 10],
        [0, 1e30, 320, 20, 100, 10],
        [0, 320, 320, 1e-30, 1e30, 10],
    ], np.float32)


def tie_rois(img=640):
    """Rounding-tie stress (SURVEY.md 8c fixture 3): with h = 32, w = 256, pooled 8x64 and scale 0.25 the
    affine has unit steps (Sx = Sy = 1), so at 0 / 90 / 180 / -90 degrees and centres on the integer or
    half-integer grid of the map every bin corner is an exact integer or half-integer: round() sits
    on ties in every bin, and one ulp in the affine (e.g. from FMA contraction) flips the sample."""
    rows = []
    for ang in (0.0, 90.0, 180.0, -90.0, 45.0, -45.0, 30.0):
        for cx, cy in ((320, 320), (322, 318), (321, 321), (320.5, 320.5), (323, 320), (200, 440), (50, 50), (600, 30)):
            rows.append([0, cx, cy, 32, 256, ang])
    for h, w in ((16, 128), (64, 512), (24, 192), (40, 200)):
        rows.append([0, 320, 320, h, w, 0.0])
        rows.append([0, 320, 320, h, w, 90.0])
    return np.asarray(rows, np.float32)


def check_backward(got, want, what="", require_abs=True, rel=1e-4, abs_bar=1e-4):
    """The backward's bar.  BASELINE configs[2] says "autograd.Function parity <= 1e-4": asserted as a MAX-ABS bound
    wherever the data allow it (`require_abs`: cfg1 ... cfg3, the training shapes, smoke -- gradients of magnitude
    tens, sums of at most a few dozen fp32 terms per pixel, measured error 1e-6 ... 2e-5).  fp32 sums do not
    associate (the reference's own atomicAdds are unordered, kernel.cu:267-274), so the heavy-overlap cases -- lists
    of thousands of pairs on one pixel, gradients of magnitude thousands -- keep the bound RELATIVE to max |grad|
    only, and say so.  Prints both errors; returns (abs_err, rel_err)."""
    import numpy as _np
    g, w = _np.asarray(got, dtype=_np.float64), _np.asarray(want, dtype=_np.float64)
    err = float(_np.abs(g - w).max()) if g.size else 0.0
    scale = max(1.0, float(_np.abs(w).max())) if w.size else 1.0
    print(f"[backward parity] {what}: max-abs {err:.3e}, relative to max|grad| = {scale:.3g}: {err / scale:.3e}")
    assert err <= rel * scale, (what, err, scale)
    if require_abs:
        assert err <= abs_bar, (what, err, "absolute bar", abs_bar, "scale", scale)
    return err, err / scale
