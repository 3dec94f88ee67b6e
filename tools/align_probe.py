#!/usr/bin/env python3
"""Crops whose rows are not multiples of 64 bytes (PH*PW % 16 != 0): the forward gather with the SHIFT kernels where
they pay (1), with the strided items only (0) and with SHIFT forced (2), by pooled size, channel count and ROI count
(explore build: rroi_align_debug_set_fwd_shift).  us per gather launch."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import workloads as Wk
lib = ctypes.CDLL(os.environ.get("RROI_EXPLORE_LIB") or os.path.join(ROOT, "tools", "_explore", "librroi_align_hip_explore.so"))
vp, fl, it, sz = ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_size_t
lib.rroi_align_forward_stages_hip.argtypes = [vp, it, fl, it, it, it, it, it, it, it, vp, vp, vp, sz, it, it, vp]
lib.rroi_align_forward_workspace_bytes.restype = sz
lib.rroi_align_forward_workspace_bytes.argtypes = [it] * 6
st = torch.cuda.current_stream().cuda_stream
def timed(fn, warm=100, n=300):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def case(C, ph, pw, R=512, H=160, W=160):
    f, r = Wk.bench_inputs(R=R, C=C, H=H, W=W, img=4 * W, seed=1)
    F, Rt = torch.from_numpy(f).cuda(), torch.from_numpy(r).cuda()
    out = torch.empty((R, C, ph, pw), device="cuda")
    nb = lib.rroi_align_forward_workspace_bytes(1, C, H, W, R, 0)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    def go(stage):
        assert lib.rroi_align_forward_stages_hip(F.data_ptr(), 0, 0.25, 1, R, H, W, C, ph, pw, Rt.data_ptr(), out.data_ptr(), ws.data_ptr(), nb, 2, stage, st) == 1
    go(3)
    res = []
    lib.rroi_align_debug_set_fwd_shift(1, -1, -1)
    go(3)
    a = out.clone()
    lib.rroi_align_debug_set_fwd_shift(0, -1, -1)
    go(3)
    res.append("equal" if torch.equal(a, out) else "DIFFERENT (%d)" % int((a != out).sum()))
    for shift in (1, 0, 2):
        lib.rroi_align_debug_set_fwd_shift(shift, -1, -1)
        res.append(f"shift {shift}: {timed(lambda: go(2)):6.1f}")
    lib.rroi_align_debug_set_fwd_shift(1, -1, -1)
    print(f"R={R:4d} C={C:3d} {ph}x{pw} rows of {ph * pw * 4} B (mod 64 = {ph * pw * 4 % 64:2d}), out {R * C * ph * pw * 4 / 1e6:6.1f} MB | " + "  ".join(res))
if os.environ.get("RROI_ALIGN_PMC"):   # a few launches of four cases for a counter pass
    def timed(fn, warm=2, n=3):   # noqa: F811
        for _ in range(warm + n): fn()
        torch.cuda.synchronize()
        return 0.0
    if os.environ["RROI_ALIGN_PMC"] == "256":
        case(256, 11, 96); case(256, 11, 100)
    else:
        case(64, 11, 96); case(64, 11, 100); case(256, 11, 96); case(256, 11, 100)
    sys.exit(0)
print("shift 1 = what the library does (SHIFT kernels where they pay), 0 = strided items only, 2 = SHIFT forced")
for C in (64, 256):
    for ph, pw in ((11, 96), (11, 100), (11, 104), (8, 62), (11, 83), (11, 85), (7, 50), (3, 21)):
        case(C, ph, pw)
for Rn in (8, 32, 128, 2048):
    case(64, 11, 83, R=Rn); case(64, 11, 100, R=Rn)
