#!/usr/bin/env python3
"""Crops beyond the 256 MB memory-side cache whose rows are not whole sectors: one group + SHIFT (what
ships) against XCD groups forced on, with the SHIFT and with the merging form; whole call (gather alone), outputs compared."""
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
for (B, C, H, W, R, pw) in ((8, 64, 160, 160, 2048, 100), (8, 64, 160, 160, 1100, 96), (8, 32, 160, 160, 2500, 96), (8, 64, 160, 160, 2048, 83), (8, 64, 160, 160, 2048, 96), (2, 64, 120, 160, 2048, 83),
                            (2, 64, 120, 160, 1024, 100), (2, 64, 120, 160, 4096, 83), (2, 32, 120, 160, 4096, 83), (1, 256, 160, 160, 1024, 50)):
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
    row, ref = [], None
    for (g, m) in ((1, 0), (2, 0), (2, 2), (1, 0), (2, 0), (2, 2)):
        lib.rroi_align_debug_set_fwd_groups(g); lib.rroi_align_debug_set_fwd_merge(m)
        call(3)
        if ref is None: ref = top.clone()
        same = torch.equal(ref.view(torch.int32), top.view(torch.int32))
        row.append(f"g{g}m{m}: {timeit(lambda: call(3)):6.1f} ({timeit(lambda: call(2)):6.1f}){'' if same else ' DIFFERENT'}")
    del ref
    mb = R * C * 11 * pw * 4 / 2**20
    print(f"B={B} C={C} {H}x{W} R={R} 11x{pw:3d} {mb:6.0f} MB  " + "  ".join(row), flush=True)
