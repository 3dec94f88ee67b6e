#!/usr/bin/env python3
"""VERDICT r05 item 4 (few-ROI forward <= 7.5 us at R = 32; 8.9-9.9 shipped): what could ANY re-shaping of K2p's memory
instructions gain?  k2p_ablate.patch (exploration build, wrong results on purpose) leaves instructions out wave-uniformly:
dbg 1 = no stores (the upper bound of "four bins per lane, one 16-byte store": stores for free), 2 = no bottom-row loads
(the upper bound of halving the loads, e.g. by staging the source through LDS), 6 = no loads at all, 7 = neither loads nor
stores (geometry + blend + the launch).  us per call between HIP events (400 calls after 60), two passes."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_explore", "librroi_align_hip_explore.so"))
vp, fl, it, sz = ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_size_t
lib.rroi_align_forward_hip.argtypes = [vp, it, fl, it, it, it, it, it, it, it, vp, vp, vp, sz, it, vp]
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, warm=60, iters=400):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
shapes = [(2, 64, 120, 160, 32, 11, pw) for pw in (96, 83, 100)] + [(2, 64, 120, 160, R, 11, 96) for R in (1, 8, 16)] + [(1, 64, 176, 320, 24, 11, 128)]
for (B, C, H, W, R, ph, pw) in shapes:
    rng = np.random.default_rng(1000 + R + pw)
    F = torch.from_numpy(rng.standard_normal((B, C, H, W), dtype=np.float32)).cuda()
    h = rng.uniform(16, 64, R)
    Rt = torch.from_numpy(np.stack([rng.integers(0, B, R), rng.uniform(0, 4 * W, R), rng.uniform(0, 4 * H, R), h,
                                    h * rng.uniform(2, pw / float(ph), R), rng.uniform(-45, 45, R)], 1).astype(np.float32)).cuda()
    top = torch.empty((R, C, ph, pw), device="cuda")
    def call():
        assert lib.rroi_align_forward_hip(F.data_ptr(), 0, 0.25, B, R, H, W, C, ph, pw, Rt.data_ptr(), top.data_ptr(), None, 0, 1, st) == 1
    for rep in range(2):
        row = []
        for dbg in (0, 1, 2, 6, 7):
            lib.rroi_align_debug_set_fwd_dbg(dbg)
            row.append(f"dbg{dbg}: {timeit(call):5.2f}")
        lib.rroi_align_debug_set_fwd_dbg(0)
        print(f"B={B} C={C} {H}x{W} R={R:2d} {ph}x{pw:3d}  " + "  ".join(row), flush=True)
