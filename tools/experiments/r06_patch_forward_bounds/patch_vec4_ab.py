#!/usr/bin/env python3
"""Round 6: K2p with the stores of four channels of a quad of bins as ONE 16-byte instruction (quad transpose by DPP) against
its dword stores -- exploration build, arms interleaved, us per call between HIP events (400 calls after 100), crops compared
bit for bit."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_explore", "librroi_align_hip_explore.so"))
vp, fl, it, sz = ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_size_t
lib.rroi_align_forward_hip.argtypes = [vp, it, fl, it, it, it, it, it, it, it, vp, vp, vp, sz, it, vp]
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, warm=100, iters=400):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
shapes = [(2, 64, 120, 160, 32, 11, pw) for pw in (96, 83, 100)] + [(2, 64, 120, 160, R, 11, 96) for R in (1, 8, 16, 48)] + \
         [(1, 64, 176, 320, 24, 11, 128), (1, 64, 176, 320, 24, 11, 64), (1, 256, 160, 160, 8, 8, 64), (1, 3, 64, 128, 4, 8, 32), (1, 33, 50, 70, 20, 7, 30), (2, 64, 120, 160, 32, 11, 6)]
for (B, C, H, W, R, ph, pw) in shapes:
    rng = np.random.default_rng(1000 + R + pw)
    F = torch.from_numpy(rng.standard_normal((B, C, H, W), dtype=np.float32)).cuda()
    h = rng.uniform(16, 64, R)
    Rt = torch.from_numpy(np.stack([rng.integers(0, B, R), rng.uniform(0, 4 * W, R), rng.uniform(0, 4 * H, R), h,
                                    h * rng.uniform(2, max(2.5, pw / float(ph)), R), rng.uniform(-45, 45, R)], 1).astype(np.float32)).cuda()
    top = torch.empty((R, C, ph, pw), device="cuda")
    def call():
        assert lib.rroi_align_forward_hip(F.data_ptr(), 0, 0.25, B, R, H, W, C, ph, pw, Rt.data_ptr(), top.data_ptr(), None, 0, 1, st) == 1
    outs, row = {}, []
    for rep in range(2):
        for v in (0, 1):
            lib.rroi_align_debug_set_fwd_patch_vec4(v)
            top.fill_(float("nan")); call(); torch.cuda.synchronize()
            outs[v] = top.clone()
            row.append(f"{'vec4' if v else 'dword'} {timeit(call):5.2f}")
    same = bool(((outs[0] == outs[1]) | (outs[0].isnan() & outs[1].isnan())).all()) and not bool(outs[1].isnan().any())
    print(f"B={B} C={C:3d} {H}x{W} R={R:2d} {ph}x{pw:3d}  " + "  ".join(row) + f"  identical={same}", flush=True)
lib.rroi_align_debug_set_fwd_patch_vec4(1)
