#!/usr/bin/env python3
"""Round 4: where the SHIFT form of the split gather spends its time.  One shape (default C = 256, 11 x 100, R = 512),
the gather launch alone, explore build: ablations (stores dropped by the descriptor check / every tap out of range),
workgroups per CU, against the strided form of the same shape.   python tools/shift_ablate.py [C ph pw R]"""
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


def setup(C, ph, pw, R, H=160, W=160):
    f, r = Wk.bench_inputs(R=R, C=C, H=H, W=W, img=4 * W, seed=1)
    F, Rt = torch.from_numpy(f).cuda(), torch.from_numpy(r).cuda()
    out = torch.empty((R, C, ph, pw), device="cuda")
    nb = lib.rroi_align_forward_workspace_bytes(1, C, H, W, R, 0)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    def go(stage):
        assert lib.rroi_align_forward_stages_hip(F.data_ptr(), 0, 0.25, 1, R, H, W, C, ph, pw, Rt.data_ptr(), out.data_ptr(), ws.data_ptr(), nb, 2, stage, st) == 1
    go(3)
    return go, (F, Rt, out, ws)


C, ph, pw, R = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (256, 11, 100, 512)
go, keep = setup(C, ph, pw, R)
mb = R * C * ph * pw * 4 / 1e6
print(f"C={C} {ph}x{pw} R={R}: {mb:.1f} MB of crops")
for shift, name in ((2, "SHIFT"), (0, "strided")):
    for dbg in (0, 1, 2, 3):
        lib.rroi_align_debug_set_fwd_shift(shift, 0, 0)
        lib.rroi_align_debug_set_fwd_dbg(dbg)
        t = timed(lambda: go(2))
        print(f"  {name:8s} ablation {dbg} (1: stores dropped, 2: taps out of range): {t:7.1f} us  {mb / t / 1e3:5.2f} TB/s")
lib.rroi_align_debug_set_fwd_shift(2, 0, 0)
lib.rroi_align_debug_set_fwd_dbg(64)
print(f"  SHIFT, a workgroup takes consecutive items (dbg 64): {timed(lambda: go(2)):7.1f} us")
lib.rroi_align_debug_set_fwd_dbg(0)
print(f"  SHIFT, items every n-th (ships):                     {timed(lambda: go(2)):7.1f} us")
for wgs in (0, 10, 8):
    lib.rroi_align_debug_set_fwd_shift(2, wgs, 0)
    t = timed(lambda: go(2))
    print(f"  SHIFT, workgroups per CU {wgs or 'default (12)'}: {t:7.1f} us")
lib.rroi_align_debug_set_fwd_shift(1, 0, 0)
