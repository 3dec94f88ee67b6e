#!/usr/bin/env python3
"""XCD groups of the forward (round 5) A/B in one process: the explore build with fwd_groups 0 / 1, whole call and its
two launches, the reference's training shapes (C = 64, two 120 x 160 maps) and neighbours; outputs compared bit for bit.
us per call between HIP events."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
lib = ctypes.CDLL(os.environ.get("RROI_EXPLORE_LIB") or os.path.join(ROOT, "tools", "_explore", "librroi_align_hip_explore.so"))
vp, fl, it, sz = ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_size_t
lib.rroi_align_forward_stages_hip.argtypes = [vp, it, fl, it, it, it, it, it, it, it, vp, vp, vp, sz, it, it, vp]
lib.rroi_align_forward_workspace_bytes.restype = sz
lib.rroi_align_forward_workspace_bytes.argtypes = [it] * 6
st = torch.cuda.current_stream().cuda_stream
lib.rroi_align_debug_set_fwd_groups_min_rois(1)   # the A/B decides, not the host rule


def timeit(fn, warm=60, iters=300):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def rois_for(rng, R, B, H, W, PW, spread):
    h = rng.uniform(16, 64, R)
    cx, cy = rng.uniform(0, 4 * W, R), rng.uniform(0, 4 * H, R)
    b = rng.integers(0, B, R)
    if spread == "skew":       # everything in the top third of image 0
        cy, b = rng.uniform(0, 4 * H / 3, R), np.zeros(R)
    return np.stack([b, cx, cy, h, h * rng.uniform(2, PW / 11.0, R), rng.uniform(-45, 45, R)], 1).astype(np.float32)


shapes = [(2, 64, 120, 160, 512, 11, 96), (2, 64, 120, 160, 512, 11, 83), (2, 64, 120, 160, 512, 11, 100),
          (2, 64, 120, 160, 32, 11, 96), (2, 64, 120, 160, 64, 11, 96), (2, 64, 120, 160, 128, 11, 96), (1, 64, 176, 320, 128, 11, 96),
          (8, 64, 160, 160, 512, 11, 100), (8, 64, 160, 160, 2048, 11, 100), (1, 32, 160, 160, 512, 8, 64)]
if len(sys.argv) > 1: shapes = shapes[:int(sys.argv[1])]
for (B, C, H, W, R, ph, pw) in shapes:
    for spread in ("uniform", "skew"):
        rng = np.random.default_rng(1000 + R + pw)
        F = torch.from_numpy(rng.standard_normal((B, C, H, W), dtype=np.float32)).cuda()
        Rt = torch.from_numpy(rois_for(rng, R, B, H, W, pw if ph == 11 else 88, spread)).cuda()
        nb = lib.rroi_align_forward_workspace_bytes(B, C, H, W, R, 0)
        ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
        outs, row = [], []
        for groups in (0, 1, 0, 1):
            lib.rroi_align_debug_set_fwd_groups(groups)
            top = torch.empty((R, C, ph, pw), device="cuda")
            def call(stages):
                assert lib.rroi_align_forward_stages_hip(F.data_ptr(), 0, 0.25, B, R, H, W, C, ph, pw, Rt.data_ptr(), top.data_ptr(),
                                                         ws.data_ptr(), nb, 2, stages, st) == 1
            t_all = timeit(lambda: call(3))
            t_pro = timeit(lambda: call(1))
            t_gat = timeit(lambda: call(2))
            row.append(f"g{groups}: {t_all:6.1f} (pro {t_pro:4.1f} gat {t_gat:5.1f})")
            call(3)
            outs.append(top.clone())
        lib.rroi_align_debug_set_fwd_groups(1)
        same = all(bool(((o == outs[0]) | (o.isnan() & outs[0].isnan())).all()) for o in outs[1:])
        print(f"B={B} C={C:3d} {H}x{W} R={R:3d} {ph}x{pw:3d} {spread:7s} " + "  ".join(row) + f"  identical={same}", flush=True)
