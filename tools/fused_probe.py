#!/usr/bin/env python3
"""The forward for few ROIs by path: direct (thread per bin), fused (one launch: the tiled gather reading the NCHW map itself,
round 5), tiled (prologue + gather) and what AUTO picks; us per call between HIP events, output preallocated; the three
outputs compared bit for bit."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
lib = ctypes.CDLL(os.environ.get("RROI_EXPLORE_LIB") or os.path.join(ROOT, "tools", "_explore", "librroi_align_hip_explore.so"))
vp, fl, it, sz = ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_size_t
lib.rroi_align_forward_hip.argtypes = [vp, it, fl, it, it, it, it, it, it, it, vp, vp, vp, sz, it, vp]
lib.rroi_align_forward_workspace_bytes.restype = sz
lib.rroi_align_forward_workspace_bytes.argtypes = [it] * 6
st = torch.cuda.current_stream().cuda_stream


def timeit(fn, warm=60, iters=400):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


shapes = [(2, 64, 120, 160, R, 11, pw) for pw in (83, 96, 100) for R in (32,)] + \
         [(2, 64, 120, 160, R, 11, 96) for R in (1, 2, 4, 8, 16, 48, 64, 96, 128)] + \
         [(1, 64, 176, 320, R, 11, 128) for R in (1, 8, 24)] + [(1, 256, 160, 160, R, 8, 64) for R in (1, 4, 8, 16, 32, 64)] + \
         [(1, 3, 64, 128, 4, 8, 32), (8, 64, 160, 160, 32, 11, 100), (1, 32, 160, 160, 64, 8, 64), (1, 128, 160, 160, 32, 8, 64)]
for (B, C, H, W, R, ph, pw) in shapes:
    rng = np.random.default_rng(1000 + R + pw)
    F = torch.from_numpy(rng.standard_normal((B, C, H, W), dtype=np.float32)).cuda()
    h = rng.uniform(16, 64, R)
    rois = np.stack([rng.integers(0, B, R), rng.uniform(0, 4 * W, R), rng.uniform(0, 4 * H, R), h,
                     h * rng.uniform(2, pw / float(ph), R), rng.uniform(-45, 45, R)], 1).astype(np.float32)
    Rt = torch.from_numpy(rois).cuda()
    nb = lib.rroi_align_forward_workspace_bytes(B, C, H, W, R, 0)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    top = torch.empty((R, C, ph, pw), device="cuda")
    def call(path):
        assert lib.rroi_align_forward_hip(F.data_ptr(), 0, 0.25, B, R, H, W, C, ph, pw, Rt.data_ptr(), top.data_ptr(), ws.data_ptr(), nb, path, st) == 1
    outs, row = [], []
    for name, path in (("old", 1), ("direct", 1), ("fused", 7), ("tiled", 2), ("auto", 0)):
        lib.rroi_align_debug_set_fwd_patch(0 if name == "old" else 1, 0, 0)   # old = rounds 1-4's thread-per-bin kernel
        top.fill_(float("nan"))
        call(path)
        outs.append(top.clone())
        row.append(f"{name} {timeit(lambda: call(path)):5.1f}")
    same = all(bool(((o == outs[0]) | (o.isnan() & outs[0].isnan())).all()) for o in outs[1:])
    print(f"B={B} C={C:3d} {H}x{W} R={R:3d} {ph}x{pw:3d} out {R * C * ph * pw / 1e6:6.2f} M  " + "  ".join(row) + f"  identical={same}", flush=True)
