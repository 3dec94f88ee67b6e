#!/usr/bin/env python3
"""(experiment) where the line-aligned windows start to pay: crops of 150-400 MB, shipped rule / lines forced / (C <= 64) groups forced + SHIFT."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_explore", "librroi_align_hip_explore.so"))
vp, fl, it, sz = ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_size_t
lib.rroi_align_forward_stages_hip.argtypes = [vp, it, fl, it, it, it, it, it, it, it, vp, vp, vp, sz, it, it, vp]
lib.rroi_align_forward_workspace_bytes.restype = sz
lib.rroi_align_forward_workspace_bytes.argtypes = [it] * 6
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, warm=20, iters=60):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
shapes = ((2, 64, 120, 160, 700, 11, 83), (2, 64, 120, 160, 900, 11, 83), (2, 64, 120, 160, 1024, 11, 83), (2, 64, 120, 160, 1200, 11, 83), (2, 64, 120, 160, 1024, 11, 100),
          (2, 64, 120, 160, 1500, 11, 83), (1, 256, 160, 160, 300, 11, 50), (1, 256, 160, 160, 400, 11, 50), (1, 256, 160, 160, 512, 11, 50),
          (1, 256, 160, 160, 250, 11, 100), (1, 256, 160, 160, 300, 11, 100), (1, 128, 160, 160, 600, 11, 83), (1, 128, 160, 160, 400, 11, 83), (8, 64, 160, 160, 1024, 11, 100))
for (B, C, H, W, R, ph, pw) in shapes:
    rng = np.random.default_rng(1000 + R + pw)
    F = torch.from_numpy(rng.standard_normal((B, C, H, W), dtype=np.float32)).cuda()
    h = rng.uniform(16, 64, R)
    Rt = torch.from_numpy(np.stack([rng.integers(0, B, R), rng.uniform(0, 4 * W, R), rng.uniform(0, 4 * H, R), h,
                                    h * rng.uniform(2, pw / float(ph), R), rng.uniform(-45, 45, R)], 1).astype(np.float32)).cuda()
    nb = lib.rroi_align_forward_workspace_bytes(B, C, H, W, R, 0)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    top = torch.empty((R, C, ph, pw), device="cuda")
    def call(stages):
        assert lib.rroi_align_forward_stages_hip(F.data_ptr(), 0, 0.25, B, R, H, W, C, ph, pw, Rt.data_ptr(), top.data_ptr(), ws.data_ptr(), nb, 2, stages, st) == 1
    row = []
    for (lines, g) in ((0, 1), (2, 1), (0, 2), (2, 2), (0, 1), (2, 1), (0, 2), (2, 2)):
        lib.rroi_align_debug_set_fwd_shift_lines(lines, 0); lib.rroi_align_debug_set_fwd_groups(g)
        call(3)
        row.append(f"{'ship' if lines == 0 else 'lines'}{'+g' if g == 2 else ''}: {timeit(lambda: call(3)):6.1f}")
    lib.rroi_align_debug_set_fwd_shift_lines(0, 0); lib.rroi_align_debug_set_fwd_groups(1)
    print(f"B={B} C={C:3d} {H}x{W} R={R:4d} {ph}x{pw:3d} {R * C * ph * pw * 4 / 2**20:6.0f} MB  " + "  ".join(row), flush=True)
