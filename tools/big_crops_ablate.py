#!/usr/bin/env python3
"""Crops beyond the memory-side cache: the gather launch with its stores dropped (dbg 1), its tap loads dropped (dbg 2), both (3),
for the aligned and the unaligned pooled widths -- which stream is the slow one?"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_explore", "librroi_align_hip_explore.so"))
vp, fl, it, sz = ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_size_t
lib.rroi_align_forward_stages_hip.argtypes = [vp, it, fl, it, it, it, it, it, it, it, vp, vp, vp, sz, it, it, vp]
lib.rroi_align_forward_workspace_bytes.restype = sz
lib.rroi_align_forward_workspace_bytes.argtypes = [it] * 6
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, warm=10, iters=40):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (B, C, H, W, R, pw) in ((8, 64, 160, 160, 2048, 100), (2, 64, 120, 160, 2048, 83), (1, 256, 160, 160, 1024, 50), (8, 64, 160, 160, 2048, 96), (2, 64, 120, 160, 512, 100), (2, 64, 120, 160, 512, 96)):
    rng = np.random.default_rng(1000 + R + pw)
    F = torch.from_numpy(rng.standard_normal((B, C, H, W), dtype=np.float32)).cuda()
    h = rng.uniform(16, 64, R)
    Rt = torch.from_numpy(np.stack([rng.integers(0, B, R), rng.uniform(0, 4 * W, R), rng.uniform(0, 4 * H, R), h,
                                    h * rng.uniform(2, pw / 11.0, R), rng.uniform(-45, 45, R)], 1).astype(np.float32)).cuda()
    nb = lib.rroi_align_forward_workspace_bytes(B, C, H, W, R, 0)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    top = torch.empty((R, C, 11, pw), device="cuda")
    def call(stages):
        assert lib.rroi_align_forward_stages_hip(F.data_ptr(), 0, 0.25, B, R, H, W, C, 11, pw, Rt.data_ptr(), top.data_ptr(), ws.data_ptr(), nb, 2, stages, st) == 1
    mb = R * C * 11 * pw * 4 / 2**20
    for (g, m) in ((1, 0), (0, 0)):
        lib.rroi_align_debug_set_fwd_groups(g); lib.rroi_align_debug_set_fwd_merge(m)
        row = []
        for dbg in (0, 1, 2, 3):
            lib.rroi_align_debug_set_fwd_dbg(dbg)
            call(3)
            row.append(f"dbg{dbg}: {timeit(lambda: call(2)):6.1f}")
        lib.rroi_align_debug_set_fwd_dbg(0)
        print(f"B={B} C={C} {H}x{W} R={R} 11x{pw:3d} {mb:6.0f} MB  g{g}m{m}  gather alone  " + "  ".join(row), flush=True)
