#!/usr/bin/env python3
"""Rows that are not whole sectors with XCD groups: the SHIFT form (fwd_merge 0) against strided tiles whose partial sectors the
XCD's L2 merges (plain stores; 2 = wherever groups apply, 1 = the host's rule); whole call and gather alone, outputs compared."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_explore", "librroi_align_hip_explore.so"))
vp, fl, it, sz = ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_size_t
lib.rroi_align_forward_stages_hip.argtypes = [vp, it, fl, it, it, it, it, it, it, it, vp, vp, vp, sz, it, it, vp]
lib.rroi_align_forward_workspace_bytes.restype = sz
lib.rroi_align_forward_workspace_bytes.argtypes = [it] * 6
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, warm=100, iters=300):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (B, C, H, W, R, pw) in ((2, 64, 120, 160, 512, 83), (2, 64, 120, 160, 512, 100), (2, 64, 120, 160, 512, 84), (2, 64, 120, 160, 512, 91), (2, 64, 120, 160, 128, 83), (2, 64, 120, 160, 64, 83), (2, 32, 120, 160, 512, 83), (1, 64, 176, 320, 128, 77)):
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
    row, outs = [], []
    for merge in (0, 2, 1, 0, 2, 1):
        lib.rroi_align_debug_set_fwd_merge(merge)
        call(3); outs.append(top.clone())
        row.append(f"{ {0: 'shift', 2: 'merge', 1: 'rule'}[merge]}: {timeit(lambda: call(3)):5.1f} ({timeit(lambda: call(2)):5.1f})")
    lib.rroi_align_debug_set_fwd_merge(1)
    print(f"B={B} C={C} R={R} 11x{pw:3d}  " + "  ".join(row) + f"  identical={all(torch.equal(o, outs[0]) for o in outs)}", flush=True)
# the merging form wherever groups apply (fwd_merge 2) against the SHIFT form on random problems: same bits
rng = np.random.default_rng(5)
bad = 0
for trial in range(int(os.environ.get("TRIALS", 150))):
    B, C, H, W = int(rng.integers(1, 5)), int(rng.integers(1, 65)), int(rng.integers(8, 100)), int(rng.integers(8, 120))
    R, ph, pw = int(rng.integers(64, 700)), int(rng.integers(1, 13)), int(rng.integers(3, 130))
    F = torch.from_numpy(rng.standard_normal((B, C, H, W), dtype=np.float32)).cuda()
    h = rng.uniform(4, 64, R)
    Rt = torch.from_numpy(np.stack([rng.integers(0, B, R), rng.uniform(-8, 4 * W + 8, R), rng.uniform(-8, 4 * H + 8, R), h,
                                    h * rng.uniform(1, 9, R), rng.uniform(-90, 90, R)], 1).astype(np.float32)).cuda()
    nb = lib.rroi_align_forward_workspace_bytes(B, C, H, W, R, 0)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    outs = []
    for merge in (0, 2):
        lib.rroi_align_debug_set_fwd_merge(merge)
        top = torch.full((R, C, ph, pw), float("nan"), device="cuda")
        assert lib.rroi_align_forward_stages_hip(F.data_ptr(), 0, 0.25, B, R, H, W, C, ph, pw, Rt.data_ptr(), top.data_ptr(), ws.data_ptr(), nb, 2, 3, st) == 1
        outs.append(top)
    if not torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32)):
        bad += 1; print("MISMATCH", B, C, H, W, R, ph, pw)
lib.rroi_align_debug_set_fwd_merge(1)
print(f"merge vs shift on random problems: {bad} mismatches")
