#!/usr/bin/env python3
"""K3f (RROI_PATH_FEWROI, the one-launch backward for few ROIs) against the shipped paths: max-abs error against the oracle
and us per call between HIP events, NCHW and channels-last top_diff, by ROI count.  Exploration build (tools/build_explore.sh)
for the channels-per-workgroup knob; falls back to the product library (cpw = its constant)."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fots.pytorch_amd"))
sys.path.insert(0, ROOT)
from rroi_align._ext import rroi_align as ext  # noqa: E402
from oracle import rroi_align_oracle as O  # noqa: E402  (checker)

explore = os.path.join(ROOT, "tools", "_explore", "librroi_align_hip_explore.so")
lib = ctypes.CDLL(explore) if os.path.exists(explore) else ext._lib
vp, fl, it, sz = ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_size_t
lib.rroi_align_backward_layout_hip.argtypes = [vp, it, it, fl, it, it, it, it, it, it, it, vp, vp, vp, sz, it, vp]
lib.rroi_align_backward_workspace_bytes.restype = sz
lib.rroi_align_backward_workspace_bytes.argtypes = [it] * 7
has_knob = hasattr(lib, "rroi_align_debug_set_bwd_fewroi")
st = torch.cuda.current_stream().cuda_stream
FEWROI = 8


def timeit(fn, iters=200):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def loop(n):
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    per = max(loop(30), 1.0)
    loop(int(20000.0 / per) + 1)
    return sorted(loop(iters) for _ in range(3))[1]


def case(B, C, H, W, R, PH, PW, seed, check=True):
    rng = np.random.default_rng(seed)
    f_shape = (B, C, H, W)
    h = rng.uniform(16, 64, R)
    rois = np.stack([rng.integers(0, B, R), rng.uniform(0, 4 * W, R), rng.uniform(0, 4 * H, R), h,
                     h * rng.uniform(2, PW / float(PH), R), rng.uniform(-45, 45, R)], 1).astype(np.float32)
    gout = rng.standard_normal((R, C, PH, PW), dtype=np.float32)
    G, Rr = torch.from_numpy(gout).cuda(), torch.from_numpy(rois).cuda()
    Gcl = G.contiguous(memory_format=torch.channels_last)
    gin = torch.empty(f_shape, device="cuda")
    nb = lib.rroi_align_backward_workspace_bytes(B, C, H, W, R, PH, PW)
    ws = torch.empty(max(nb, 1), dtype=torch.uint8, device="cuda")

    def call(path, nhwc=False):
        src = Gcl if nhwc else G
        s_ = lib.rroi_align_backward_layout_hip(src.data_ptr(), 1 if nhwc else 0, 0, 0.25, B, R, H, W, C, PH, PW, Rr.data_ptr(),
                                                gin.data_ptr(), ws.data_ptr(), nb, path, st)
        assert s_ == 1, (path, nhwc, s_)
    row = ["B%d C%d %dx%d R%-3d %dx%d" % (B, C, H, W, R, PH, PW)]
    if check:
        want = O.backward_c(gout, rois, f_shape, 0.25, threads=8)
        for nhwc in ((False, True) if C % 4 == 0 else (False,)):
            gin.fill_(float("nan"))
            call(FEWROI, nhwc)
            err = float(np.abs(gin.cpu().numpy() - want).max())
            row.append("err%s %.1e" % ("_cl" if nhwc else "", err))
            assert err <= 1e-4 * max(1.0, float(np.abs(want).max())), err
    if has_knob:
        lib.rroi_align_debug_set_bwd_fewroi(0, -1, -1)
    row.append("auto %5.1f" % timeit(lambda: call(0)))
    if C % 4 == 0:
        row.append("auto_cl %5.1f" % timeit(lambda: call(0, True)))
    row.append("direct %6.1f" % timeit(lambda: call(1), 50))
    for cpw in ((32, 64) if has_knob else (0,)):
        if has_knob:
            lib.rroi_align_debug_set_bwd_fewroi(-1, -1, cpw)
        row.append("k3f/%d %5.1f" % (cpw, timeit(lambda: call(FEWROI))))
        if C % 4 == 0:
            row.append("k3f_cl/%d %5.1f" % (cpw, timeit(lambda: call(FEWROI, True))))
    print("  ".join(row), flush=True)


if __name__ == "__main__":
    for R in (1, 4, 8, 16, 32, 48, 64, 96, 128):
        case(2, 64, 120, 160, R, 11, 96, 100 + R)
    for PW in (83, 100):
        case(2, 64, 120, 160, 32, 11, PW, 7 + PW)
    case(2, 64, 128, 128, 32, 11, 96, 3)          # train.py: batch 2, input 512
    case(1, 64, 176, 320, 24, 11, 128, 4)         # one inference image's map
    for R in (4, 16, 32):
        case(1, 256, 160, 160, R, 8, 64, 50 + R)
    case(8, 64, 160, 160, 32, 11, 100, 9)
    case(1, 36, 50, 70, 9, 8, 33, 11)             # odd sizes: C % 32 != 0, W % 4 != 0
    case(1, 3, 64, 128, 4, 8, 32, 12)
