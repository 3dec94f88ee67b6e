#!/usr/bin/env python3
"""VERDICT r05 item 6 (line-aligned windows at 4/3 instead of 2 x the gather work per byte): the UPPER BOUND, measured before
building.  lines_upper_bound.patch (exploration build only, results wrong on purpose) makes every third item of a workgroup
free in phases A and B (no geometry, no tap loads, no blend) while its tile is still stored: the store stream of all items over
the gather work of two thirds of them -- 2 x 2/3 = 4/3 per stored byte, which is what a 128-gathered / 96-stored tile would
cost at best (its larger drain and its lower occupancy not counted).
us per call: whole call (prologue + gather) and the gather launch alone; dbg 0 = shipped, 8 = the bound, 3 / 11 = the same with
every memory access dropped (the instruction stream alone)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_explore", "librroi_align_hip_explore.so"))
vp, fl, it, sz = ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_size_t
lib.rroi_align_forward_stages_hip.argtypes = [vp, it, fl, it, it, it, it, it, it, it, vp, vp, vp, sz, it, it, vp]
lib.rroi_align_forward_workspace_bytes.restype = sz
lib.rroi_align_forward_workspace_bytes.argtypes = [it] * 6
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, warm=15, iters=50):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
SHAPES = ((1, 256, 160, 160, 600, 11, 100), (2, 64, 120, 160, 2048, 11, 83), (8, 64, 160, 160, 2048, 11, 100), (1, 256, 160, 160, 1024, 11, 50))
for (B, C, H, W, R, ph, pw) in SHAPES:
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
    crops = R * C * ph * pw * 4
    for rep in range(2):
        row = []
        for dbg in (0, 8, 3, 11):
            lib.rroi_align_debug_set_fwd_dbg(dbg)
            call(3)
            whole, alone = timeit(lambda: call(3)), timeit(lambda: call(2))
            row.append(f"dbg{dbg:2d}: {whole:6.1f} / {alone:6.1f}" + (f" ({crops / whole / 1e6:4.2f} TB/s)" if dbg in (0, 8) else ""))
        lib.rroi_align_debug_set_fwd_dbg(0)
        print(f"B={B} C={C} {H}x{W} R={R} {ph}x{pw:3d} {crops / 1e6:6.0f} MB  whole / gather alone  " + "  ".join(row), flush=True)
