#!/usr/bin/env python3
"""The forward's prologue launch alone (relayout of the feature maps + affine table) by shape, against the bytes it
moves: back-to-back launches between HIP events (explore build's stage switch).  us, TB/s of map read + copy written."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import workloads as Wk
lib = ctypes.CDLL(os.environ.get("RROI_EXPLORE_LIB") or os.path.join(ROOT, "tools", "_explore", "librroi_align_hip_explore.so"))
vp, fl, it, sz = ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_size_t
lib.rroi_align_forward_stages_hip.argtypes = [vp, it, fl, it, it, it, it, it, it, it, vp, vp, vp, sz, it, it, vp]
lib.rroi_align_forward_workspace_bytes.restype = sz
lib.rroi_align_forward_workspace_bytes.argtypes = [it] * 6
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, warm=100, iters=400):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (B, C, H, W, R, ph, pw) in ((1, 256, 160, 160, 512, 8, 64), (2, 64, 120, 160, 512, 11, 96), (2, 64, 120, 160, 32, 11, 96),
                                (2, 64, 120, 160, 1, 11, 96), (1, 64, 176, 320, 128, 11, 96), (8, 64, 160, 160, 64, 11, 100),
                                (1, 64, 160, 160, 512, 8, 64), (1, 128, 160, 160, 512, 8, 64), (1, 32, 160, 160, 512, 8, 64)):
    f, r = Wk.bench_inputs(R=R, C=C, H=H, W=W, img=4 * W, seed=1, batch=B)
    F, Rt = torch.from_numpy(f).cuda(), torch.from_numpy(r).cuda()
    top = torch.empty((R, C, ph, pw), device="cuda")
    nb = lib.rroi_align_forward_workspace_bytes(B, C, H, W, R, 0)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    def call(stages):
        assert lib.rroi_align_forward_stages_hip(F.data_ptr(), 0, 0.25, B, R, H, W, C, ph, pw, Rt.data_ptr(), top.data_ptr(), ws.data_ptr(), nb, 2, stages, st) == 1
    out = []
    for blocks in (3, 1, 2, 6):
        lib.rroi_align_debug_set_prologue_blocks(blocks)
        out.append(f"{blocks}/CU {timeit(lambda: call(1)):5.1f}")
    lib.rroi_align_debug_set_prologue_blocks(3)
    mb = 2 * B * C * H * W * 4 / 1e6
    t = timeit(lambda: call(1))
    print(f"B={B} C={C:3d} {H}x{W} R={R:3d}: {mb:6.1f} MB  " + "  ".join(out) + f"   -> {mb / t / 1e3:4.2f} TB/s   whole call {timeit(lambda: call(3)):6.1f}", flush=True)
