#!/usr/bin/env python3
"""The one-launch forward for few ROIs (rroi_fwd_patch_kernel, round 5) by patch shape and slab count, against round 1-4's
thread-per-bin kernel and the two-launch tiled path: us per call between HIP events (output preallocated), outputs compared
bit for bit with the tiled path's."""
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


shapes = [(2, 64, 120, 160, 32, 11, 83), (2, 64, 120, 160, 32, 11, 96), (2, 64, 120, 160, 32, 11, 100), (2, 64, 120, 160, 8, 11, 96),
          (2, 64, 120, 160, 64, 11, 96), (2, 64, 120, 160, 128, 11, 96), (1, 64, 176, 320, 1, 11, 64), (1, 64, 176, 320, 24, 11, 128),
          (1, 256, 160, 160, 16, 8, 64), (1, 256, 160, 160, 32, 8, 64), (1, 256, 160, 160, 64, 8, 64)]
# (name, patch rows log2 (-1 = old kernel), target waves, sort, unroll, channels per wave at most)
forms = [("old", -1, 4096, 0, 4, 16), ("p4x16", 2, 4096, 0, 4, 16), ("s4x16", 2, 4096, 1, 4, 16), ("s4x16u8", 2, 4096, 1, 8, 16),
         ("s4x16u8c32", 2, 4096, 1, 8, 32), ("s4x16c8", 2, 8192, 1, 4, 8), ("s8x8", 3, 4096, 1, 4, 16), ("s8x8u8", 3, 4096, 1, 8, 16)]
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
    call(2)
    want = top.clone()
    row = [f"tiled {timeit(lambda: call(2)):5.1f}"]
    ok = True
    for name, prl, waves, srt, unroll, cw in forms:
        lib.rroi_align_debug_set_fwd_patch(prl, waves)
        lib.rroi_align_debug_set_fwd_patch_sort(srt)
        lib.rroi_align_debug_set_fwd_patch_shape(unroll, cw)
        lib.rroi_align_debug_set_fwd_patch_ablate(int(os.environ.get("ABLATE", "0")))
        lib.rroi_align_debug_set_fwd_patch_xcd(int(os.environ.get("XCD", "1")))
        top.fill_(float("nan"))
        call(1)
        ok = ok and bool(((top == want) | (top.isnan() & want.isnan())).all())
        row.append(f"{name} {timeit(lambda: call(1)):5.1f}")
    lib.rroi_align_debug_set_fwd_patch(2, 4096)
    lib.rroi_align_debug_set_fwd_patch_sort(1)
    lib.rroi_align_debug_set_fwd_patch_shape(4, 16)
    print(f"B={B} C={C:3d} {H}x{W} R={R:3d} {ph}x{pw:3d}  " + "  ".join(row) + f"  identical={ok}", flush=True)
