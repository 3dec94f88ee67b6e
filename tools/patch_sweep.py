#!/usr/bin/env python3
"""K2p (the direct path's patch kernel, round 5): target waves x channels per wave x unroll, us per call."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
shapes = [(2, 64, 120, 160, 32, 11, 96), (2, 64, 120, 160, 32, 11, 83), (2, 64, 120, 160, 32, 11, 100), (2, 64, 120, 160, 8, 11, 96),
          (2, 64, 120, 160, 64, 11, 96), (1, 64, 176, 320, 24, 11, 128), (1, 256, 160, 160, 16, 8, 64), (1, 64, 176, 320, 1, 11, 64)]
forms = [(w, c, 4) for w in (2048, 4096, 8192) for c in (8, 16, 32)]
print("form (waves/cwave/unroll): " + " ".join(f"{w//1024}k/{c}/{u}" for w, c, u in forms))
for (B, C, H, W, R, ph, pw) in shapes:
    rng = np.random.default_rng(1000 + R + pw)
    F = torch.from_numpy(rng.standard_normal((B, C, H, W), dtype=np.float32)).cuda()
    h = rng.uniform(16, 64, R)
    Rt = torch.from_numpy(np.stack([rng.integers(0, B, R), rng.uniform(0, 4 * W, R), rng.uniform(0, 4 * H, R), h,
                                    h * rng.uniform(2, pw / float(ph), R), rng.uniform(-45, 45, R)], 1).astype(np.float32)).cuda()
    top = torch.empty((R, C, ph, pw), device="cuda")
    def call():
        assert lib.rroi_align_forward_hip(F.data_ptr(), 0, 0.25, B, R, H, W, C, ph, pw, Rt.data_ptr(), top.data_ptr(), None, 0, 1, st) == 1
    row = []
    for w, c, u in forms:
        lib.rroi_align_debug_set_fwd_patch(1, w, c)
        row.append(f"{timeit(call):5.1f}")
    print(f"C={C:3d} R={R:3d} {ph}x{pw:3d}  " + " ".join(row), flush=True)
