#!/usr/bin/env python3
"""Upper bound of VERDICT r05 item 3 (right-hand corners of a bin handed over from the neighbouring lane): the gather launch with the
four right-hand corner evaluations REMOVED (dbg 8: wrong results, what a free hand-over would cost) and with the whole corner / min-max /
rounding block removed (dbg 16), next to the shipped kernel (dbg 0), and the same three with loads and stores dropped (+3).
Exploration build with geometry_upper_bound.patch applied; us per launch of the gather alone and of the whole call."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_explore", "librroi_align_hip_explore.so"))
vp, fl, it, sz = ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_size_t
lib.rroi_align_forward_stages_hip.argtypes = [vp, it, fl, it, it, it, it, it, it, it, vp, vp, vp, sz, it, it, vp]
lib.rroi_align_forward_workspace_bytes.restype = sz
lib.rroi_align_forward_workspace_bytes.argtypes = [it] * 6
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, warm=200, iters=300):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (B, C, H, W, R, ph, pw) in ((2, 64, 120, 160, 512, 11, 96), (2, 64, 120, 160, 512, 11, 83), (1, 256, 160, 160, 512, 8, 64)):
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
    call(3)
    row = []
    for rep in range(2):
        for dbg in (0, 8, 16, 3, 11, 19):
            lib.rroi_align_debug_set_fwd_dbg(dbg)
            row.append("dbg%-2d gather %5.2f call %5.2f" % (dbg, timeit(lambda: call(2)), timeit(lambda: call(3))))
    lib.rroi_align_debug_set_fwd_dbg(0)
    print("C=%d %dx%d R=%d\n  " % (C, ph, pw, R) + "\n  ".join(row), flush=True)
