#!/usr/bin/env python3
"""K3f ablations (exploration build): gradient loads dropped (1), LDS atomics dropped (2), both (3); us per call."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_explore", "librroi_align_hip_explore.so"))
vp, fl, it, sz = ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_size_t
lib.rroi_align_backward_layout_hip.argtypes = [vp, it, it, fl, it, it, it, it, it, it, it, vp, vp, vp, sz, it, vp]
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, warm=300, iters=300):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
B, C, H, W, PH, PW = 2, 64, 120, 160, 11, 96
for R in (1, 8, 32):
    rng = np.random.default_rng(100 + R)
    h = rng.uniform(16, 64, R)
    rois = torch.from_numpy(np.stack([rng.integers(0, B, R), rng.uniform(0, 4 * W, R), rng.uniform(0, 4 * H, R), h,
                                      h * rng.uniform(2, PW / float(PH), R), rng.uniform(-45, 45, R)], 1).astype(np.float32)).cuda()
    G = torch.randn((R, C, PH, PW), device="cuda")
    gin = torch.empty((B, C, H, W), device="cuda")
    def call():
        assert lib.rroi_align_backward_layout_hip(G.data_ptr(), 0, 0, 0.25, B, R, H, W, C, PH, PW, rois.data_ptr(), gin.data_ptr(), None, 0, 8, st) == 1
    row = []
    for cpw in (32, 64):
        lib.rroi_align_debug_set_bwd_fewroi(-1, -1, cpw)
        for dbg in (0, 1, 2, 3):
            lib.rroi_align_debug_set_bwd_fewroi_dbg(dbg)
            row.append("cpw%d dbg%d %5.1f" % (cpw, dbg, timeit(call)))
    lib.rroi_align_debug_set_bwd_fewroi_dbg(0)
    print("R=%d  " % R + "  ".join(row), flush=True)
