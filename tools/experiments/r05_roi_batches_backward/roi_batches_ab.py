#!/usr/bin/env python3
"""(experiment) The backward with its ROIs in N batches, each a whole backward call that adds to bottom_diff: a batch's
pixel-major copy (a quarter of 268 MB at configs[2]) fits the 256 MB memory-side cache between the launch that writes it and the
one that reads it.  us per call between HIP events, max |difference| against the one-batch result; the exploration build."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_explore", "librroi_align_hip_explore.so"))
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
shapes = [(1, 256, 160, 160, 512, 8, 64), (1, 256, 160, 160, 2048, 8, 64), (2, 64, 120, 160, 512, 11, 96), (2, 64, 120, 160, 512, 11, 83),
          (8, 64, 160, 160, 512, 11, 100), (1, 128, 160, 160, 512, 8, 64), (1, 512, 160, 160, 512, 8, 64)]
for (B, C, H, W, R, ph, pw) in shapes:
    rng = np.random.default_rng(1000 + R + pw)
    h = rng.uniform(16, 64, R)
    rois = torch.from_numpy(np.stack([rng.integers(0, B, R), rng.uniform(0, 4 * W, R), rng.uniform(0, 4 * H, R), h,
                                      h * rng.uniform(2, pw / float(ph), R), rng.uniform(-45, 45, R)], 1).astype(np.float32)).cuda()
    g = torch.randn((R, C, ph, pw), device="cuda")
    gin = torch.empty((B, C, H, W), device="cuda")
    nb = lib.rroi_align_backward_workspace_bytes(B, C, H, W, R, ph, pw)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    def call():
        assert lib.rroi_align_backward_hip(g.data_ptr(), 0.25, B, R, H, W, C, ph, pw, rois.data_ptr(), gin.data_ptr(), ws.data_ptr(), nb, 0, st) == 1
    row, ref = [], None
    for nbat in (0, 2, 4, 8, 0, 2, 4, 8):
        lib.rroi_align_debug_set_bwd_roi_batches(nbat)
        gin.fill_(float("nan")); call()
        if ref is None: ref = gin.clone()
        row.append(f"{nbat}: {timeit(call):6.1f} (d {float((gin - ref).abs().max()):.1e})")
    lib.rroi_align_debug_set_bwd_roi_batches(0)
    print(f"B={B} C={C:3d} {H}x{W} R={R:4d} {ph}x{pw:3d}  " + "  ".join(row), flush=True)
