#!/usr/bin/env python3
"""VERDICT r05 item 1 / missing items 2-3: the hand-off layouts the FOTS pipeline can actually reach.

The reference's backbone emits NCHW features (tools/models.py:387-457) and its recognition head consumes the crops
directly (src/ocr_process.py:266-267, :295).  This probe times, on one GPU:
  (a) the op alone, forward and backward, in the four layout pairs (features / feature gradient x crops / top_diff):
      NCHW-NCHW (the reference contract), NCHW features + channels-last crops ("mixed"), channels-last both;
      at configs[2]'s shape and at the training call's shapes (C = 64, two 120 x 160 maps, 11 x PW, R = 32 / 512);
  (b) the recognition head (forward_ocr forward + backward, CTC loss) on NCHW and on channels-last crops, with the head's
      weights in either format;
  (c) the training caller's step as the reference runs it (train.py:79-119 -> src/ocr_process.py:259-301): op forward ->
      forward_ocr -> CTC -> backward through the head and the op, per layout of the crops.
Prints one JSON object.  Usage: python tools/mixed_layout_probe.py [--quick]
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fots.pytorch_amd"))
from rroi_align._ext import rroi_align as ext  # noqa: E402
from rroi_align.modules.rroi_align import _RRoiAlign  # noqa: E402
from fots_e2e.model import FOTSNet  # noqa: E402
from fots_e2e.weights import deterministic_init  # noqa: E402

dev = torch.device("cuda", 0)
NCHW, NHWC = ext.LAYOUT_NCHW, ext.LAYOUT_NHWC
CL = torch.channels_last


def event_loop(fn, warm, timed):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(timed):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / timed


def settled(fn, timed):
    per = max(event_loop(fn, 5, 30), 1e-3)
    return sorted([event_loop(fn, int(20.0 / per) + 1, timed), event_loop(fn, 0, timed), event_loop(fn, 0, timed)])[1]


def make(B, C, H, W, R, PH, PW, seed, bench_draw=False):
    rng = np.random.default_rng(seed)
    feats = torch.from_numpy(rng.standard_normal((B, C, H, W), dtype=np.float32)).to(dev)
    h = rng.uniform(16, 64, R)
    if bench_draw:
        rois = np.stack([np.zeros(R), rng.uniform(0, 4 * W, R), rng.uniform(0, 4 * H, R), h, h * rng.uniform(4, 8, R),
                         rng.uniform(-90, 90, R)], 1)
    else:
        rois = np.stack([rng.integers(0, B, R), rng.uniform(0, 4 * W, R), rng.uniform(0, 4 * H, R), h,
                         h * rng.uniform(2, PW / float(PH), R), rng.uniform(-45, 45, R)], 1)
    return feats, torch.from_numpy(rois.astype(np.float32)).to(dev)


def op_alone(tag, B, C, H, W, R, PH, PW, bench_draw=False):
    feats, rois = make(B, C, H, W, R, PH, PW, 1000 + R + PW, bench_draw)
    stream = torch.cuda.current_stream().cuda_stream
    feats_cl = feats.contiguous(memory_format=CL)
    out = torch.empty((R, C, PH, PW), device=dev)
    out_cl = torch.empty((R, C, PH, PW), device=dev).contiguous(memory_format=CL)
    gout = torch.randn_like(out)
    gout_cl = gout.contiguous(memory_format=CL)
    gin = torch.empty_like(feats)
    nf = max(ext._lib.rroi_align_forward_workspace_bytes(B, C, H, W, R, NCHW), ext._lib.rroi_align_forward_workspace_bytes(B, C, H, W, R, NHWC))
    nb = ext._lib.rroi_align_backward_workspace_bytes(B, C, H, W, R, PH, PW)
    ws = torch.empty(max(nf, nb, 1), dtype=torch.uint8, device=dev)
    row = {}
    for name, fl, tl in (("nchw", NCHW, NCHW), ("mixed", NCHW, NHWC), ("channels_last", NHWC, NHWC)):
        f_t, o_t, g_t = (feats if fl == NCHW else feats_cl), (out if tl == NCHW else out_cl), (gout if tl == NCHW else gout_cl)

        def fwd():
            st = ext._lib.rroi_align_forward_layout_hip(f_t.data_ptr(), fl, tl, 0.25, B, R, H, W, C, PH, PW, rois.data_ptr(),
                                                        o_t.data_ptr(), ws.data_ptr(), ws.numel(), ext.PATH_AUTO, stream)
            assert st == 1, st

        def bwd():
            st = ext._lib.rroi_align_backward_layout_hip(g_t.data_ptr(), tl, fl, 0.25, B, R, H, W, C, PH, PW, rois.data_ptr(),
                                                         gin.data_ptr(), ws.data_ptr(), ws.numel(), ext.PATH_AUTO, stream)
            assert st == 1, st
        row[name] = {"forward_us": round(settled(fwd, 200) * 1e3, 2), "backward_us": round(settled(bwd, 100) * 1e3, 2)}
        row[name]["sum_us"] = round(row[name]["forward_us"] + row[name]["backward_us"], 2)
    # parity of the mixed pair against the NCHW pair (same values, element for element; backward to fp32 sum order)
    ext._lib.rroi_align_forward_layout_hip(feats.data_ptr(), NCHW, NCHW, 0.25, B, R, H, W, C, PH, PW, rois.data_ptr(), out.data_ptr(),
                                           ws.data_ptr(), ws.numel(), ext.PATH_AUTO, stream)
    ext._lib.rroi_align_forward_layout_hip(feats.data_ptr(), NCHW, NHWC, 0.25, B, R, H, W, C, PH, PW, rois.data_ptr(), out_cl.data_ptr(),
                                           ws.data_ptr(), ws.numel(), ext.PATH_AUTO, stream)
    row["mixed_forward_equals_nchw"] = bool(torch.equal(out, out_cl.contiguous()))
    print(tag, json.dumps(row), flush=True)
    return row


def head_and_step(R, PW, quick):
    """forward_ocr + CTC, forward and backward, on crops of the training shape; and the whole recognition branch of a
    training step (op forward -> head -> CTC -> head backward -> op backward)."""
    B, C, H, W, PH = 2, 64, 128, 128, 11      # train.py: batch_size 2, input_size 512 -> a 64 x 128 x 128 map
    feats, rois = make(B, C, H, W, R, PH, PW, 77 + R + PW)
    nclass = 87
    T = PW
    rng = np.random.default_rng(5)
    lens = torch.from_numpy(rng.integers(3, 10, R).astype(np.int64))
    targets = torch.from_numpy(rng.integers(1, nclass, int(lens.sum())).astype(np.int64))
    res = {}
    for wname, wfmt in (("weights_nchw", torch.contiguous_format), ("weights_channels_last", CL)):
        net = deterministic_init(FOTSNet(nclass)).to(dev).train()
        net = net.to(memory_format=wfmt)
        params = [p for n, p in net.named_parameters() if n.split(".")[0] in ("conv5", "conv6", "conv7", "conv8", "conv9", "conv10_s", "conv11",
                                                                               "batch5", "batch7", "batch10_s")]

        def ctc(preds):
            p = preds.permute(2, 0, 1)      # (T, N, nclass), as src/ocr_process.py:296
            return F.ctc_loss(p, targets, torch.full((R,), T, dtype=torch.int64), lens, blank=0, reduction="sum") / R

        for cname, cl in (("crops_nchw", False), ("crops_channels_last", True)):
            crops0 = _RRoiAlign(PH, PW, 0.25, channels_last_out=cl)(feats, rois).detach()

            def head():
                x = crops0.clone().requires_grad_(True)   # the clone keeps the layout
                loss = ctc(net.forward_ocr(x))
                loss.backward()
                for p in params:
                    p.grad = None
                return x.grad
            g = head()
            gl = "channels_last" if (not g.is_contiguous() and g.is_contiguous(memory_format=CL)) else "nchw"

            f_leaf = feats.clone().requires_grad_(True)

            def step():
                f_leaf.grad = None
                crops = _RRoiAlign(PH, PW, 0.25, channels_last_out=cl)(f_leaf, rois)
                loss = ctc(net.forward_ocr(crops))
                loss.backward()
                for p in params:
                    p.grad = None
            n = 20 if quick else 60
            res["%s.%s" % (wname, cname)] = {"head_fwd_bwd_us": round(settled(head, n) * 1e3, 1), "grad_of_crops_layout": gl,
                                            "train_step_us": round(settled(step, n) * 1e3, 1)}
            print(R, PW, wname, cname, json.dumps(res["%s.%s" % (wname, cname)]), flush=True)
        del net
    return res


def main():
    quick = "--quick" in sys.argv
    out = {"what": __doc__.split("\n")[0], "device": torch.cuda.get_device_name(0), "library": ext.version()}
    out["op_alone"] = {"cfg3_C256_8x64_R512": op_alone("cfg3", 1, 256, 160, 160, 512, 8, 64, bench_draw=True)}
    for PW in (83, 100, 96):
        for R in (32, 512):
            out["op_alone"]["C64_11x%d_R%d" % (PW, R)] = op_alone("train 11x%d R%d" % (PW, R), 2, 64, 120, 160, R, 11, PW)
    out["head_and_step"] = {}
    for R, PW in ((32, 96), (32, 83)) if not quick else ((32, 96),):
        out["head_and_step"]["R%d_11x%d" % (R, PW)] = head_and_step(R, PW, quick)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
