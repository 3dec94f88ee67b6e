"""CPU: pin the oracle to the only outputs of the real CUDA op the reference holds.

rroi_align/test2.py:22-104 ran the op on data/timg.jpeg for three hand-annotated
quads (gt2, gt4, gt5), pooled 44 x ceil(44*max(w/h)), scale 1.0, and wrote the
crops to data/res{0,1,2}.jpg and the gradient of pooled.pow(2).sum() to
data/grad.jpg.  Its ROI height carries an unrecorded `random.randint(-2, 2)`
jitter (test2.py:56); exactly one jitter per ROI reproduces the stored crops to
JPEG noise.  What this pins: column order
[idx,cx,cy,h,w,angle], the angle sign, the in_rroi right-side zero padding, the
round/clamp/average sampling rule, NCHW indexing, and the backward's geometry.
"""
import math
import os

import numpy as np
import pytest

PIL = pytest.importorskip("PIL.Image")

DATA = os.path.join(os.path.dirname(__file__), "golden", "ref_data")

# rroi_align/test2.py:42-46 (quads are data, restated here as numbers)
GT2 = [[206, 111], [199, 95], [349, 60], [355, 80]]
GT4 = [[312, 127], [304, 105], [367, 88], [374, 114]]
GT5 = [[133, 168], [118, 112], [175, 100], [185, 154]]
JITTER = (2, 1, 2)


def bgr(name):
    return np.asarray(PIL.open(os.path.join(DATA, name)).convert("RGB"))[:, :, ::-1].astype(np.float32)


def build_rois(jitter):
    """test2.py:50-61."""
    rois = []
    for gt, j in zip((GT2, GT4, GT5), jitter):
        gt = np.asarray(gt)
        center = (gt[0] + gt[1] + gt[2] + gt[3]) / 4
        dw, dh = gt[2] - gt[1], gt[1] - gt[0]
        w = math.sqrt(dw[0] ** 2 + dw[1] ** 2)
        h = math.sqrt(dh[0] ** 2 + dh[1] ** 2) + j
        a = (math.atan2(gt[2][1] - gt[1][1], gt[2][0] - gt[1][0]) +
             math.atan2(gt[3][1] - gt[0][1], gt[3][0] - gt[0][0])) / 2
        rois.append([0, center[0], center[1], h, w, -a / 3.1415926535 * 180])
    return np.asarray(rois, np.float32)


@pytest.fixture(scope="module")
def setup():
    img = bgr("timg.jpeg")
    feats = np.ascontiguousarray(img.transpose(2, 0, 1)[None])
    rois = build_rois(JITTER)
    pooled_w = math.ceil(44 * float((rois[:, 4] / rois[:, 3]).max()))  # test2.py:66-69
    return feats, rois, pooled_w


def test_pooled_width_matches_stored_crops(setup):
    _, _, pooled_w = setup
    assert bgr("res0.jpg").shape == (44, pooled_w, 3) == (44, 349, 3)


def test_forward_matches_cuda_crops(oracle, setup):
    feats, rois, pooled_w = setup
    out = oracle.forward_c(feats, rois, 44, pooled_w, 1.0)
    errs = []
    for i in range(3):
        crop = out[i].transpose(1, 2, 0).astype(np.uint8).astype(np.float32)  # test2.py:81-86
        errs.append(float(np.abs(crop - bgr("res%d.jpg" % i)).mean()))
    # JPEG quantisation noise only (1.51 / 0.53 / 0.46)
    assert errs[0] < 2.0 and errs[1] < 1.0 and errs[2] < 1.0, errs


def test_wrong_jitter_or_sign_is_rejected(oracle, setup):
    """The KAT discriminates: any other height jitter, or a flipped angle, misses by far."""
    feats, _, pooled_w = setup
    ref = [bgr("res%d.jpg" % i) for i in range(3)]

    def err(rois, i):
        o = oracle.forward_c(feats, rois[i:i + 1], 44, pooled_w, 1.0)[0]
        return float(np.abs(o.transpose(1, 2, 0).astype(np.uint8).astype(np.float32) - ref[i]).mean())

    good = build_rois(JITTER)
    for i in range(3):
        best = err(good, i)
        for j in (-2, -1, 0, 1, 2):
            if j == JITTER[i]:
                continue
            jit = list(JITTER)
            jit[i] = j
            assert err(build_rois(jit), i) > 1.2 * best + 0.5
        flipped = good.copy()
        flipped[:, 5] *= -1
        assert err(flipped, i) > 5 * best


def test_backward_support_matches_cuda_grad(oracle, setup):
    """grad.jpg went through a wrapping uint8 cast (test2.py:96), so only its support is a
    usable pin: IoU of the non-zero sets (0.985)."""
    feats, rois, pooled_w = setup
    out = oracle.forward_c(feats, rois, 44, pooled_w, 1.0)
    gin = oracle.backward_c((2 * out).astype(np.float32), rois, feats.shape, 1.0)
    mine = np.abs(gin[0]).sum(0) > 0
    ref = bgr("grad.jpg").sum(2) > 24  # above JPEG ringing
    iou = (mine & ref).sum() / (mine | ref).sum()
    assert iou > 0.95, iou
