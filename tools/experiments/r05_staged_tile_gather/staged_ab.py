#!/usr/bin/env python3
"""The staged tile gather (RROI_PATH_TILED_STAGED: no pixel-major copy, no lists) against what AUTO runs: us per backward call
between HIP events, max |difference| of the two gradients."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = "librroi_align_hip_explore.so" if os.environ.get("EXPLORE") == "1" else None
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_explore", name) if name else
                  os.path.join(ROOT, "fots.pytorch_amd", "rroi_align", "_ext", "rroi_align", "librroi_align_hip.so"))
vp, fl, it, sz = ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_size_t
lib.rroi_align_backward_hip.argtypes = [vp, fl, it, it, it, it, it, it, it, vp, vp, vp, sz, it, vp]
lib.rroi_align_backward_workspace_bytes.restype = sz
lib.rroi_align_backward_workspace_bytes.argtypes = [it] * 7
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, warm=30, iters=100):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
shapes = [(1, 3, 64, 128, 4, 8, 32, 1.0), (2, 64, 120, 160, 32, 11, 96, 0.25), (2, 64, 120, 160, 32, 11, 83, 0.25), (2, 64, 120, 160, 128, 11, 96, 0.25),
          (2, 64, 120, 160, 512, 11, 96, 0.25), (2, 64, 120, 160, 512, 11, 83, 0.25), (8, 64, 160, 160, 512, 11, 100, 0.25),
          (1, 64, 176, 320, 24, 11, 96, 0.25), (1, 32, 160, 160, 512, 8, 64, 0.25), (1, 128, 160, 160, 512, 8, 64, 0.25),
          (1, 256, 160, 160, 512, 8, 64, 0.25), (1, 256, 160, 160, 32, 8, 64, 0.25), (1, 40, 50, 70, 100, 7, 33, 0.25), (1, 512, 160, 160, 512, 8, 64, 0.25)]
for (B, C, H, W, R, ph, pw, scale) in shapes:
    rng = np.random.default_rng(1000 + R + pw)
    h = rng.uniform(16, 64, R) * (scale / 0.25) ** -1 if scale != 1.0 else rng.uniform(6, 20, R)
    rois = torch.from_numpy(np.stack([rng.integers(0, B, R), rng.uniform(0, W / scale, R), rng.uniform(0, H / scale, R), h,
                                      h * rng.uniform(2, pw / float(ph), R), rng.uniform(-45, 45, R)], 1).astype(np.float32)).cuda()
    g = torch.randn((R, C, ph, pw), device="cuda")
    nb = lib.rroi_align_backward_workspace_bytes(B, C, H, W, R, ph, pw)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    outs, row = {}, []
    for path in (0, 8, 0, 8):
        gin = torch.full((B, C, H, W), float("nan"), device="cuda")
        def call():
            s = lib.rroi_align_backward_hip(g.data_ptr(), scale, B, R, H, W, C, ph, pw, rois.data_ptr(), gin.data_ptr(), ws.data_ptr(), nb, path, st)
            assert s == 1, s
        call(); torch.cuda.synchronize()
        outs[path] = gin.clone()
        row.append(f"{'auto' if path == 0 else 'staged'}: {timeit(call):6.1f}")
    d = float((outs[0] - outs[8]).abs().max())
    nan = int(torch.isnan(outs[8]).sum())
    print(f"B={B} C={C:3d} {H}x{W} R={R:4d} {ph}x{pw:3d}  " + "  ".join(row) + f"  max|d| = {d:.2e}  max|g| = {float(outs[0].abs().max()):.1f}  nan = {nan}", flush=True)
