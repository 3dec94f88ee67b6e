"""Callers' ROI construction: CPU checks of the restatement,
GPU parity of the device kernel, and the batched launch against per-box launches."""
import math

import numpy as np
import pytest
import torch

from oracle import roi_build_oracle as RB


def random_quads(n, seed=0, integer=False, size=(1280, 704)):
    """Rotated rectangles as 4 corner points, order 0..3 = bl, tl, tr, br of the text box as in
    rroi_align/test2.py:42-46 (edge 1->2 is the long 'width' edge, 0->1 the 'height' edge)."""
    rng = np.random.default_rng(seed)
    q = np.zeros((n, 4, 2))
    for i in range(n):
        cx, cy = rng.uniform(100, size[0] - 100), rng.uniform(100, size[1] - 100)
        h, w = rng.uniform(12, 60), rng.uniform(40, 400)
        a = rng.uniform(-80, 80) / 180 * math.pi
        ux, uy = math.cos(a), math.sin(a)        # along the text
        vx, vy = -math.sin(a), math.cos(a)       # down
        q[i, 0] = (cx - ux * w / 2 + vx * h / 2, cy - uy * w / 2 + vy * h / 2)
        q[i, 1] = (cx - ux * w / 2 - vx * h / 2, cy - uy * w / 2 - vy * h / 2)
        q[i, 2] = (cx + ux * w / 2 - vx * h / 2, cy + uy * w / 2 - vy * h / 2)
        q[i, 3] = (cx + ux * w / 2 + vx * h / 2, cy + uy * w / 2 + vy * h / 2)
    if integer:
        q = np.round(q)
    return q.reshape(n, 8).astype(np.float32)


def test_restatement_on_the_reference_demo_quad():
    """rroi_align/test2.py:43 quad gt2 -> the numbers its own code (test2.py:50-61) produces."""
    gt = np.asarray([[206, 111], [199, 95], [349, 60], [355, 80]], np.float32)
    rois, gw = RB.rois_from_quads(gt.reshape(1, 8), mode=1)
    center = gt.sum(0) / 4
    w = math.sqrt(150 ** 2 + 35 ** 2)
    h = math.sqrt(7 ** 2 + 16 ** 2)
    ang = -(math.atan2(-35, 150) + math.atan2(-31, 149)) / 2 / 3.1415926535 * 180
    want = np.asarray([0, center[0], center[1], h, w, ang], np.float64).astype(np.float32)
    assert np.array_equal(rois[0], want)
    # inference flavour: truncated centre, single-edge angle, width rule of ocr_utils.py:147-150
    r0, g0 = RB.rois_from_quads(gt.reshape(1, 8), mode=0)
    assert r0[0, 1] == 277.0 and r0[0, 2] == 86.0
    assert np.float32(-math.atan2(-35, 150) / 3.1415926535 * 180) == r0[0, 5]
    assert g0[0] == max(2, (int(w * (11 / h)) + 11) // 32) * 32 == 96


GOLD = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "golden", "roi_build.npz")


def test_restatement_equals_the_references_own_statements():
    """VERDICT r01 #6b: `tests/golden/roi_build.npz` holds what the reference's OWN lines
    (tools/ocr_utils.py:133-150, src/ocr_process.py:197-206 and :259-263, executed in place by
    `make_roi_golden.py`) compute for seeded quads: int() truncation of the centre, max(1, h), the
    // 32 rule, the double-precision ground-truth branch with its height jitter, the training
    width rule.  The restatement must reproduce every number, bit for bit."""
    z = np.load(GOLD)
    for tag in ("int", "real"):
        q = z["quads_" + tag]
        rois, gw = RB.rois_from_quads(q, mode=0)
        assert np.array_equal(rois, z["m0_rois_" + tag]) and np.array_equal(gw, z["m0_gw_" + tag])
        for j in (0, -2, 2):
            rois1, _ = RB.rois_from_quads(q[:64], mode=1, jitter=j)
            assert np.array_equal(rois1, z["m1_rois_%s_j%d" % (tag, j)])
            assert RB.train_pooled_width(rois1) == int(z["m1_pw_%s_j%d" % (tag, j)])


def test_pooled_width_rules():
    q = random_quads(64, seed=3)
    rois, gw = RB.rois_from_quads(q, mode=0)
    assert (gw % 32 == 0).all() and (gw >= 64).all()
    # the rule rounds DOWN to a multiple of 32: it may cut a crop (gw < roi_pooled_width = 11*w/h,
    # kernel.cu:68) but never by more than 32 - 11 columns
    rpw = 11 * rois[:, 4] / np.maximum(rois[:, 3], 1)
    assert (gw > rpw - 21).all() and ((gw == 64) | (gw <= rpw + 11)).all()
    assert RB.train_pooled_width(rois) == math.ceil(11 * float((rois[:, 4] / rois[:, 3]).max()))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1])
def test_device_roi_builder_matches_restatement(mode):
    from rroi_align._ext import rroi_align as ext
    for integer in (True, False):
        q = random_quads(2000, seed=11 + mode, integer=integer)
        bidx = (np.arange(2000) % 4).astype(np.float32)
        want, wgw = RB.rois_from_quads(q, bidx, mode=mode)
        got, ggw = ext.quads_to_rois(torch.from_numpy(q).cuda(), torch.from_numpy(bidx).cuda(), mode)
        got, ggw = got.cpu().numpy(), ggw.cpu().numpy()
        # fp64 sqrt is correctly rounded on both sides; atan2 comes from ocml on the device and from
        # glibc in the reference's Python -- both within an ulp of the DOUBLE result, which the
        # rounding of the angle to fp32 absorbs: every field bit-identical
        assert np.array_equal(got, want)
        assert np.array_equal(ggw, wgw)


@pytest.mark.gpu
def test_device_roi_builders_match_the_references_own_output():
    """The device kernels against `roi_build.npz` (the reference's own statements, see above)."""
    from rroi_align._ext import rroi_align as ext
    z = np.load(GOLD)
    for tag in ("int", "real"):
        q = torch.from_numpy(z["quads_" + tag]).cuda()
        got, ggw = ext.quads_to_rois(q, None, 0)
        assert np.array_equal(got.cpu().numpy(), z["m0_rois_" + tag])
        assert np.array_equal(ggw.cpu().numpy(), z["m0_gw_" + tag])
        for j in (0, -2, 2):
            jit = torch.full((64,), float(j), device="cuda")
            rois, ratio = ext.gt_quads_to_rois(q[:64], None, jit if j else None)
            want = z["m1_rois_%s_j%d" % (tag, j)]
            assert np.array_equal(rois.cpu().numpy(), want)
            assert math.ceil(11 * float(ratio.item())) == int(z["m1_pw_%s_j%d" % (tag, j)])


@pytest.mark.gpu
def test_ground_truth_module_and_degenerate_jitter(oracle):
    """src/ocr_process.py:196-221, :253-267 through `GroundTruthRRoiAlign`: crops equal the oracle on
    the module's rows and width; a jitter that drives h negative gives all-zero crops (the op's mask,
    kernel.cu:107); h == 0 raises, as the reference's math.ceil(inf) does."""
    from rroi_align.batched import GroundTruthRRoiAlign
    rng = np.random.default_rng(9)
    feats_np = rng.standard_normal((2, 64, 120, 160), dtype=np.float32)
    feats = torch.from_numpy(feats_np).cuda()
    q = random_quads(40, seed=13, size=(640, 480))
    q[3] = np.asarray([100, 101, 100, 100, 160, 100, 160, 101], np.float32)   # h = 1: jitter -2 -> h = -1
    bidx = (np.arange(40) % 2).astype(np.float32)
    jit = np.where(np.arange(40) == 3, -2.0, 1.0).astype(np.float32)
    m = GroundTruthRRoiAlign(11, 0.25)
    crops, rois = m(feats, torch.from_numpy(q).cuda(), torch.from_numpy(bidx).cuda(), torch.from_numpy(jit).cuda())
    want_rois, _ = RB.rois_from_quads(q[:32], bidx[:32], mode=1, jitter=jit[:32])
    assert np.array_equal(rois.cpu().numpy(), want_rois) and rois.shape[0] == 32  # :253-255 keeps 32 rows
    pw = RB.train_pooled_width(want_rois)
    assert crops.shape == (32, 64, 11, pw)
    assert np.array_equal(crops.cpu().numpy(), oracle.forward_c(feats_np, want_rois, 11, pw, 0.25, threads=8))
    assert want_rois[3, 3] == -1.0 and not crops[3].any()
    jit[3] = -1.0                                                              # h == 0 -> w / h = inf
    with pytest.raises(ValueError):
        m(feats, torch.from_numpy(q).cuda(), torch.from_numpy(bidx).cuda(), torch.from_numpy(jit).cuda())
    # the reference's own order -- filter ('##' labels, boxes that leave the image, :212-219),
    # THEN keep 32 rows (:253-255) -- and its one jitter draw PER IMAGE (:204), looked up through batch_index
    keep = np.ones(40, bool)
    keep[[0, 3, 5, 17]] = False
    per_image = np.asarray([2.0, -1.0], np.float32)
    crops2, rois2 = m(feats, torch.from_numpy(q).cuda(), torch.from_numpy(bidx).cuda(), torch.from_numpy(per_image).cuda(),
                      keep=torch.from_numpy(keep).cuda())
    want2, _ = RB.rois_from_quads(q[keep][:32], bidx[keep][:32], mode=1, jitter=per_image[bidx.astype(int)][keep][:32])
    assert np.array_equal(rois2.cpu().numpy(), want2) and rois2.shape[0] == 32
    pw2 = RB.train_pooled_width(want2)
    assert np.array_equal(crops2.cpu().numpy(), oracle.forward_c(feats_np, want2, 11, pw2, 0.25, threads=8))
    # as many boxes as images -- the length alone cannot decide: per box by default, with a warning.  Two boxes, both of image 1:
    # read per image the jitter of both is per_image[1], read per box it is per_image[0] and per_image[1]
    q2, b2 = q[[10, 11]], np.asarray([1.0, 1.0], np.float32)
    args = (feats, torch.from_numpy(q2).cuda(), torch.from_numpy(b2).cuda(), torch.from_numpy(per_image).cuda())
    import rroi_align.batched as _B
    _B._warned_ambiguous_jitter = False
    with pytest.warns(UserWarning, match="per BOX"):     # the per-box default stays, with one warning
        _, r_default = m(*args)
    _, r_img = m(*args, per_image_jitter=True)
    _, r_box = m(*args, per_image_jitter=False)
    assert np.array_equal(r_img.cpu().numpy(), RB.rois_from_quads(q2, b2, mode=1, jitter=per_image[[1, 1]])[0])
    assert np.array_equal(r_box.cpu().numpy(), RB.rois_from_quads(q2, b2, mode=1, jitter=per_image)[0])
    assert not np.array_equal(r_img.cpu().numpy(), r_box.cpu().numpy())
    assert np.array_equal(r_default.cpu().numpy(), r_box.cpu().numpy())


@pytest.mark.gpu
def test_callers_follow_the_features_layout():
    """Round 5: the two callers' modules hand the crops over in the FEATURES' layout by default
    -- channels_last features give channels_last crops (same values, element for element), autograd brings the
    gradient back channels_last and the feature gradient is written channels_last in place: no copy on either side.
    NCHW features keep the reference's contract.  True / False force the layout."""
    from rroi_align.batched import BatchedRRoiAlign, GroundTruthRRoiAlign
    rng = np.random.default_rng(21)
    f = rng.standard_normal((2, 64, 60, 80), dtype=np.float32)
    q = torch.from_numpy(random_quads(24, seed=5, size=(320, 240))).cuda()
    b = torch.from_numpy((np.arange(24) % 2).astype(np.float32)).cuda()
    res = {}
    for tag, fmt in (("nchw", torch.contiguous_format), ("cl", torch.channels_last)):
        F = torch.from_numpy(f).cuda().contiguous(memory_format=fmt).requires_grad_(True)
        crops, rois = GroundTruthRRoiAlign(11, 0.25)(F, q, b)
        assert crops.is_contiguous(memory_format=fmt), tag
        (crops * crops).sum().backward()
        assert F.grad.is_contiguous(memory_format=fmt), tag
        res[tag] = (crops.detach().clone(), F.grad.detach().clone())
        c2, gw = BatchedRRoiAlign(11, 0.25, pooled_width=96)(F.detach(), q[:, :8], b)
        assert c2.is_contiguous(memory_format=fmt), tag
    assert torch.equal(res["nchw"][0], res["cl"][0])
    scale = max(1.0, float(res["nchw"][1].abs().max()))
    assert float((res["nchw"][1] - res["cl"][1]).abs().max()) <= 1e-4 * scale
    Fc = torch.from_numpy(f).cuda().contiguous(memory_format=torch.channels_last)
    forced, _ = GroundTruthRRoiAlign(11, 0.25, channels_last_out=False)(Fc, q, b)
    assert forced.is_contiguous() and torch.equal(forced, res["nchw"][0])


@pytest.mark.gpu
def test_batched_launch_equals_per_box_launches(oracle):
    """One launch for all boxes of an image == the reference's loop of R = 1 launches
    (tools/ocr_utils.py:147-177): identical on each box's own width target_gw[i].  Beyond it the
    batched crop continues up to the box's roi_pooled_width (the per-box width rounds down to a
    multiple of 32 and may cut the word) and is zero after that (kernel.cu:107)."""
    from rroi_align.batched import BatchedRRoiAlign
    from rroi_align.modules.rroi_align import _RRoiAlign
    rng = np.random.default_rng(5)
    feats_np = rng.standard_normal((1, 64, 176, 320), dtype=np.float32)   # focr: 64 ch at 1/4 of 1280x704
    feats = torch.from_numpy(feats_np).cuda()
    q = random_quads(24, seed=7)
    crops, gw = BatchedRRoiAlign(11, 1.0 / 4)(feats, torch.from_numpy(q).cuda())
    rois, wgw = RB.rois_from_quads(q, mode=0)
    assert crops.shape == (24, 64, 11, int(wgw.max()))
    for i in range(24):
        single = _RRoiAlign(11, int(wgw[i]), 1.0 / 4)(feats, torch.from_numpy(rois[i:i + 1]).cuda())
        assert torch.equal(crops[i:i + 1, :, :, :int(wgw[i])], single)
        rpw = 11 * rois[i, 4] / rois[i, 3]
        assert not crops[i, :, :, int(math.floor(rpw)) + 1:].any()
    # and against the oracle, element for element
    want = oracle.forward_c(feats_np, rois, 11, int(wgw.max()), 0.25, threads=4)
    assert np.array_equal(crops.cpu().numpy(), want)
