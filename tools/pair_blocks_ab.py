#!/usr/bin/env python3
"""Pair blocks per CU of the backward's list launch (1 = rounds 2-4, 0 = round 5's 256 / C rule, 2 / 4 / 8 fixed): us per
backward call between HIP events, the exploration build."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_explore", "librroi_align_hip_explore.so"))
vp, fl, it, sz = ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_size_t
lib.rroi_align_backward_hip.argtypes = [vp, fl, it, it, it, it, it, it, it, vp, vp, vp, sz, it, vp]
lib.rroi_align_backward_layout_hip.argtypes = [vp, it, it, fl, it, it, it, it, it, it, it, vp, vp, vp, sz, it, vp]
CL = os.environ.get("CL") == "1"   # channels-last top_diff and bottom_diff (no relayout in the list launch)
lib.rroi_align_backward_workspace_bytes.restype = sz
lib.rroi_align_backward_workspace_bytes.argtypes = [it] * 7
st = torch.cuda.current_stream().cuda_stream
prev = None   # an older build for a same-box comparison
if os.path.exists(os.path.join(ROOT, "tools", "_explore", "librroi_align_hip_prev.so")):
    prev = ctypes.CDLL(os.path.join(ROOT, "tools", "_explore", "librroi_align_hip_prev.so"))
    prev.rroi_align_backward_hip.argtypes = lib.rroi_align_backward_hip.argtypes
    prev.rroi_align_backward_workspace_bytes.restype = sz
    prev.rroi_align_backward_workspace_bytes.argtypes = [it] * 7


def timeit(fn, warm=20, iters=100):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


shapes = [(2, 64, 120, 160, 512, 11, 96), (2, 64, 120, 160, 512, 11, 83), (2, 64, 120, 160, 128, 11, 96), (2, 64, 120, 160, 32, 11, 96),
          (1, 64, 176, 320, 128, 11, 96), (8, 64, 160, 160, 512, 11, 100), (1, 32, 160, 160, 512, 8, 64), (1, 128, 160, 160, 512, 8, 64),
          (1, 256, 160, 160, 512, 8, 64), (1, 256, 160, 160, 2048, 8, 64), (1, 512, 160, 160, 512, 8, 64)]
for (B, C, H, W, R, ph, pw) in shapes:
    rng = np.random.default_rng(1000 + R + pw)
    h = rng.uniform(16, 64, R)
    rois = torch.from_numpy(np.stack([rng.integers(0, B, R), rng.uniform(0, 4 * W, R), rng.uniform(0, 4 * H, R), h,
                                      h * rng.uniform(2, pw / float(ph), R), rng.uniform(-45, 45, R)], 1).astype(np.float32)).cuda()
    g = torch.randn((R, C, ph, pw), device="cuda")
    gin = torch.empty((B, C, H, W), device="cuda")
    nb = lib.rroi_align_backward_workspace_bytes(B, C, H, W, R, ph, pw)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    if CL:
        g = g.contiguous(memory_format=torch.channels_last)
        gin = gin.contiguous(memory_format=torch.channels_last)
    def call():
        if CL:
            assert lib.rroi_align_backward_layout_hip(g.data_ptr(), 1, 1, 0.25, B, R, H, W, C, ph, pw, rois.data_ptr(), gin.data_ptr(), ws.data_ptr(), nb, 0, st) == 1
        else:
            assert lib.rroi_align_backward_hip(g.data_ptr(), 0.25, B, R, H, W, C, ph, pw, rois.data_ptr(), gin.data_ptr(), ws.data_ptr(), nb, 0, st) == 1
    row, ref = [], None
    if prev is not None:
        nbp = prev.rroi_align_backward_workspace_bytes(B, C, H, W, R, ph, pw)
        wsp = torch.empty(nbp, dtype=torch.uint8, device="cuda")
        prev.rroi_align_backward_layout_hip.argtypes = lib.rroi_align_backward_layout_hip.argtypes
        def callp():
            if CL:
                assert prev.rroi_align_backward_layout_hip(g.data_ptr(), 1, 1, 0.25, B, R, H, W, C, ph, pw, rois.data_ptr(), gin.data_ptr(), wsp.data_ptr(), nbp, 0, st) == 1
                return
            assert prev.rroi_align_backward_hip(g.data_ptr(), 0.25, B, R, H, W, C, ph, pw, rois.data_ptr(), gin.data_ptr(), wsp.data_ptr(), nbp, 0, st) == 1
        callp(); torch.cuda.synchronize()
        ref = gin.clone()
        row.append(f"prev: {timeit(callp):6.1f}")
    for per in (1, 0, 2, 4, 8, 16, -8):   # (-n: n blocks per CU, one atomic per pair -- no aggregation)
        lib.rroi_align_debug_set_bwd_pair_aggregate(0 if per < 0 else 1)
        lib.rroi_align_debug_set_bwd_pair_blocks(abs(per))
        call(); torch.cuda.synchronize()
        if ref is None: ref = gin.clone()
        ok = float((gin - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))
        row.append(f"{per if per else 'auto'}: {timeit(call):6.1f}{'' if ok else ' WRONG'}")
    lib.rroi_align_debug_set_bwd_pair_blocks(0)
    lib.rroi_align_debug_set_bwd_pair_aggregate(1)
    print(f"B={B} C={C:3d} {H}x{W} R={R:4d} {ph}x{pw:3d}  " + "  ".join(row), flush=True)
