#!/usr/bin/env python3
"""GPU-box exploration: device write/copy ceilings, per-stage timings of our path, the
reference's own kernels (oracle/_ref, hipify-perl build) timed and compared.  Writes
gpurun_out/explore.json.  Measurement tooling, not product code."""
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fots.pytorch_amd"), os.path.join(ROOT, "tests")]
import workloads as Wk  # noqa: E402
from rroi_align._ext import rroi_align as ext  # noqa: E402

res = {}


def timeit(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    t = np.array([a.elapsed_time(b) for a, b in evs]) * 1e3
    return {"avg_us": float(t.mean()), "p50_us": float(np.median(t)), "min_us": float(t.min())}


dev = torch.device("cuda:0")
p = torch.cuda.get_device_properties(0)
res["device"] = {"name": p.name, "cus": p.multi_processor_count, "mem_GB": p.total_memory / 2**30}

# --- ceilings ---------------------------------------------------------------------------
n = 64 * 1024 * 1024  # 256 MiB of fp32
a = torch.empty(n, device=dev)
b = torch.empty(n, device=dev)
r = timeit(lambda: a.fill_(1.0))
res["fill_256MiB"] = {**r, "GBs": n * 4 / r["avg_us"] / 1e3}
r = timeit(lambda: a.zero_())
res["memset_256MiB"] = {**r, "GBs": n * 4 / r["avg_us"] / 1e3}
r = timeit(lambda: b.copy_(a))
res["copy_256MiB"] = {**r, "GBs_rw": 2 * n * 4 / r["avg_us"] / 1e3}
big = torch.empty(4 * n, device=dev)
r = timeit(lambda: big.fill_(1.0), iters=20)
res["fill_1GiB"] = {**r, "GBs": 4 * n * 4 / r["avg_us"] / 1e3}
del a, b, big

# --- our path ---------------------------------------------------------------------------
f, rois = Wk.bench_inputs()
F, R = torch.from_numpy(f).to(dev), torch.from_numpy(rois).to(dev)
out = torch.empty((512, 256, 8, 64), device=dev)
nb = ext._lib.rroi_align_forward_workspace_bytes(1, 256, 160, 160, 512, 0)
ws = torch.empty(nb, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream


def stage(s, path=ext.PATH_TILED):
    rc = ext._lib.rroi_align_forward_stages_hip(F.data_ptr(), 0, 0.25, 1, 512, 160, 160, 256, 8, 64,
                                                 R.data_ptr(), out.data_ptr(), ws.data_ptr(), nb, path, s, st)
    assert rc == 1, rc


res["fwd_prologue"] = timeit(lambda: stage(1))
res["fwd_gather"] = timeit(lambda: stage(2), iters=100)
res["fwd_all"] = timeit(lambda: stage(3), iters=100)
res["fwd_direct"] = timeit(lambda: stage(2, ext.PATH_DIRECT), iters=20)
ours = out.clone()

# sensitivity points of SURVEY.md 8(d): same map, ROI set varied (gather kernel only, 300 launches)
def variant(name, mod):
    rv = rois.copy()
    mod(rv)
    Rv = torch.from_numpy(rv).to(dev)

    def run():
        rc = ext._lib.rroi_align_forward_stages_hip(F.data_ptr(), 0, 0.25, 1, 512, 160, 160, 256, 8, 64,
                                                     Rv.data_ptr(), out.data_ptr(), ws.data_ptr(), nb,
                                                     ext.PATH_TILED, 2, st)
        assert rc == 1, rc
    rc = ext._lib.rroi_align_forward_stages_hip(F.data_ptr(), 0, 0.25, 1, 512, 160, 160, 256, 8, 64, Rv.data_ptr(),
                                                 out.data_ptr(), ws.data_ptr(), nb, ext.PATH_TILED, 1, st)
    assert rc == 1, rc
    r = timeit(run, iters=300, warm=100)
    r["nonzero_bin_fraction"] = float((out[:, 0] != 0).float().mean().item())
    res["fwd_gather_" + name] = r


def _all_active(rv): rv[:, 4] = rv[:, 3] * 8.0           # w/h = 8: roi_pooled_width = 64, no masked bins
def _axis(rv): rv[:, 5] = 0.0
def _vertical(rv): rv[:, 5] = 90.0
def _inside(rv):                                            # every box well inside the map
    rv[:, 1] = 200 + (rv[:, 1] % 240)
    rv[:, 2] = 200 + (rv[:, 2] % 240)
    rv[:, 3] = np.minimum(rv[:, 3], 32)
    rv[:, 4] = np.minimum(rv[:, 4], 200)


variant("bench_rois", lambda rv: None)
variant("all_active_w_over_h_8", _all_active)
variant("axis_aligned_angle_0", _axis)
variant("vertical_angle_90", _vertical)
variant("all_inside_map", _inside)
stage(1)

Fcl = F.contiguous(memory_format=torch.channels_last)
res["fwd_channels_last_zero_copy"] = timeit(lambda: ext.forward(Fcl, R, 8, 64, 0.25))

g = torch.randn_like(out)
res["bwd_tiled"] = timeit(lambda: ext.backward(g, R, f.shape, 0.25, path=ext.PATH_TILED), iters=20)
res["bwd_direct"] = timeit(lambda: ext.backward(g, R, f.shape, 0.25, path=ext.PATH_DIRECT), iters=10)

# small-R regimes of the callers
for name, (Rn, C, H, W, ph, pw) in {"infer_R1_c64_11x64": (1, 64, 176, 320, 11, 64),
                                      "train_R32_c64_11x96": (32, 64, 160, 160, 11, 96)}.items():
    ff, rr = Wk.bench_inputs(R=Rn, C=C, H=H, W=W, img=4 * W, seed=3)
    Ft, Rt = torch.from_numpy(ff).to(dev), torch.from_numpy(rr).to(dev)
    for pname, pth in (("direct", ext.PATH_DIRECT), ("tiled", ext.PATH_TILED)):
        res[f"{name}_{pname}"] = timeit(lambda: ext.forward(Ft, Rt, ph, pw, 0.25, path=pth), iters=30)

# --- inference: one launch per word (tools/ocr_utils.py:131-177) vs one per image ---------------
from rroi_align.batched import BatchedRRoiAlign  # noqa: E402
from rroi_align.modules.rroi_align import _RRoiAlign  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_roi_build import random_quads  # noqa: E402
from oracle import roi_build_oracle as RB  # noqa: E402  (host-side restatement = what the reference does per box)
focr = torch.randn(1, 64, 176, 320, device=dev)
quads_np = random_quads(24, seed=7)
quads = torch.from_numpy(quads_np).to(dev)
batched = BatchedRRoiAlign(11, 0.25)


def per_box_loop():
    outs = []
    for i in range(24):  # host numpy ROI, upload, R = 1 launch -- the reference's structure
        roi, gwi = RB.rois_from_quads(quads_np[i:i + 1], mode=0)
        outs.append(_RRoiAlign(11, int(gwi[0]), 0.25)(focr, torch.from_numpy(roi).to(dev)))
    return outs


import time  # noqa: E402
for name, fn in (("infer_24_boxes_per_box_loop", per_box_loop), ("infer_24_boxes_batched", lambda: batched(focr, quads))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    res[name] = {"wall_us_per_image": (time.perf_counter() - t0) / 20 * 1e6}

# --- the reference's own kernels (oracle/_ref) --------------------------------------------
ref_path = os.path.join(ROOT, "oracle", "_ref", "librroi_ref_hip.so")
if os.path.exists(ref_path):
    ref = ctypes.CDLL(ref_path)
    vp, fl, it = ctypes.c_void_p, ctypes.c_float, ctypes.c_int
    ref.RROIAlignForwardLaucher.argtypes = [vp, fl, it, it, it, it, it, it, vp, vp, vp, vp, vp]
    ref.RROIAlignBackwardLaucher.argtypes = [vp, fl, it, it, it, it, it, it, it, vp, vp, vp, vp, vp]
    top, ix, iy = (torch.zeros_like(out) for _ in range(3))

    def ref_fwd(with_memsets):
        if with_memsets:  # functions/rroi_align.py:17-20
            top.zero_(); ix.zero_(); iy.zero_()
        ref.RROIAlignForwardLaucher(F.data_ptr(), 0.25, 512, 160, 160, 256, 8, 64, R.data_ptr(),
                                    top.data_ptr(), ix.data_ptr(), iy.data_ptr(), st)

    res["ref_fwd_kernel_only"] = timeit(lambda: ref_fwd(False), iters=10, warm=2)
    res["ref_fwd_with_memsets"] = timeit(lambda: ref_fwd(True), iters=10, warm=2)
    ref_fwd(True)
    torch.cuda.synchronize()
    diff = (top != ours)
    nd = int(diff.sum())
    res["ref_vs_ours_forward"] = {"elements": top.numel(), "differ": nd,
                                  "bins_differ": int(diff.any(1).sum()),
                                  "max_abs": float((top - ours).abs().max())}
    gin = torch.zeros_like(F)

    def ref_bwd():
        gin.zero_()
        ref.RROIAlignBackwardLaucher(g.data_ptr(), 0.25, 1, 512, 160, 160, 256, 8, 64, R.data_ptr(),
                                     gin.data_ptr(), ix.data_ptr(), iy.data_ptr(), st)

    res["ref_bwd_with_memset"] = timeit(ref_bwd, iters=5, warm=1)
    mine = ext.backward(g, R, f.shape, 0.25)
    ref_bwd()
    torch.cuda.synchronize()
    res["ref_vs_ours_backward"] = {"max_abs": float((gin - mine).abs().max()), "scale": float(gin.abs().max())}

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "explore.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
