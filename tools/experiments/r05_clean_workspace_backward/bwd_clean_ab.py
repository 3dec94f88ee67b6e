#!/usr/bin/env python3
"""RROI_PATH_WS_CLEAN A/B: the backward with the caller-zeroed workspace (two launches for few ROIs) against the default
(three launches), us per call between HIP events, the product library through the C-ABI."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fots.pytorch_amd"), os.path.join(ROOT, "tests")]
import ctypes
from rroi_align._ext import rroi_align as ext
st = torch.cuda.current_stream().cuda_stream
prev = None   # an older build of the library for a same-box comparison (tools/_explore/librroi_align_hip_prev.so)
_pp = os.environ.get("RROI_PREV_LIB") or os.path.join(ROOT, "tools", "_explore", "librroi_align_hip_prev.so")
if os.path.exists(_pp):
    prev = ctypes.CDLL(_pp)
    prev.rroi_align_backward_hip.argtypes = ext._lib.rroi_align_backward_hip.argtypes
    prev.rroi_align_backward_workspace_bytes.restype = ctypes.c_size_t
    prev.rroi_align_backward_workspace_bytes.argtypes = [ctypes.c_int] * 7


def timeit(fn, warm=30, iters=150):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


shapes = [(2, 64, 120, 160, 8, 11, 96), (2, 64, 120, 160, 32, 11, 83), (2, 64, 120, 160, 32, 11, 96), (2, 64, 120, 160, 32, 11, 100),
          (2, 64, 120, 160, 64, 11, 96), (2, 64, 120, 160, 128, 11, 96), (2, 64, 120, 160, 512, 11, 96), (1, 64, 176, 320, 24, 11, 128),
          (1, 256, 160, 160, 32, 8, 64), (1, 256, 160, 160, 512, 8, 64)]
for (B, C, H, W, R, ph, pw) in shapes:
    rng = np.random.default_rng(1000 + R + pw)
    h = rng.uniform(16, 64, R)
    rois = torch.from_numpy(np.stack([rng.integers(0, B, R), rng.uniform(0, 4 * W, R), rng.uniform(0, 4 * H, R), h,
                                      h * rng.uniform(2, pw / float(ph), R), rng.uniform(-45, 45, R)], 1).astype(np.float32)).cuda()
    g = torch.randn((R, C, ph, pw), device="cuda")
    gin = torch.empty((B, C, H, W), device="cuda")
    nb = ext._lib.rroi_align_backward_workspace_bytes(B, C, H, W, R, ph, pw)
    ncl = ext._lib.rroi_align_backward_clean_bytes(B, C, H, W, R, ph, pw)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    ws[:ncl].zero_()
    def call(flag):
        assert ext._lib.rroi_align_backward_hip(g.data_ptr(), 0.25, B, R, H, W, C, ph, pw, rois.data_ptr(), gin.data_ptr(),
                                                ws.data_ptr(), nb, ext.PATH_AUTO | flag, st) == 1
    row = []
    if prev is not None:
        nbp = prev.rroi_align_backward_workspace_bytes(B, C, H, W, R, ph, pw)
        wsp = torch.empty(nbp, dtype=torch.uint8, device="cuda")
        def callp():
            assert prev.rroi_align_backward_hip(g.data_ptr(), 0.25, B, R, H, W, C, ph, pw, rois.data_ptr(), gin.data_ptr(),
                                                wsp.data_ptr(), nbp, ext.PATH_AUTO, st) == 1
        row.append(f"prev {timeit(callp):6.1f}")
    for flag in (0, ext.PATH_WS_CLEAN, 0, ext.PATH_WS_CLEAN):
        if not flag:
            t = timeit(lambda: call(0))
            ws[:ncl].zero_()
        else:
            t = timeit(lambda: call(flag))
        row.append(f"{'clean' if flag else 'plain'} {t:6.1f}")
    print(f"B={B} C={C:3d} {H}x{W} R={R:3d} {ph}x{pw:3d}  " + "  ".join(row), flush=True)
