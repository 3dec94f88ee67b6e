#!/usr/bin/env python3
"""Gather-kernel ablations at the training shape (C = 64, two 120 x 160 maps, R = 512): output stores dropped (1), every tap
out of range (2), both (3) -- gather launch alone, us between HIP events, the exploration build."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_explore", "librroi_align_hip_explore.so"))
vp, fl, it, sz = ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_size_t
lib.rroi_align_forward_stages_hip.argtypes = [vp, it, fl, it, it, it, it, it, it, it, vp, vp, vp, sz, it, it, vp]
lib.rroi_align_forward_workspace_bytes.restype = sz
lib.rroi_align_forward_workspace_bytes.argtypes = [it] * 6
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, warm=60, iters=300):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
B, C, H, W, R = 2, 64, 120, 160, 512
for pw in (96, 83, 100):
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
    call(3)
    row = []
    for dbg in (0, 1, 2, 3):
        lib.rroi_align_debug_set_fwd_dbg(dbg)
        row.append(f"dbg{dbg}: {timeit(lambda: call(2)):5.1f}")
    lib.rroi_align_debug_set_fwd_dbg(0)
    for wpc in (8, 10, 12, 14, 16):
        lib.rroi_align_debug_set_split_wgs_per_cu(wpc); lib.rroi_align_debug_set_fwd_shift(-1, wpc, 0)
        row.append(f"wg{wpc}: {timeit(lambda: call(2)):5.1f}")
    lib.rroi_align_debug_set_split_wgs_per_cu(12); lib.rroi_align_debug_set_fwd_shift(-1, 0, 0)
    print(f"11x{pw:3d} out {R*C*11*pw*4/1e6:6.1f} MB  " + "  ".join(row), flush=True)
