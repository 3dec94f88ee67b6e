#!/usr/bin/env python3
"""Round 3: the prologue's blocks per CU and store policy after the vmcnt(4) fix of its loop (explore build):
prologue alone and whole call, HIP events."""
import ctypes, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import workloads as Wk
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_explore", "librroi_align_hip_explore.so"))
vp, fl, it, sz = ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_size_t
lib.rroi_align_forward_stages_hip.argtypes = [vp, it, fl, it, it, it, it, it, it, it, vp, vp, vp, sz, it, it, vp]
lib.rroi_align_forward_workspace_bytes.restype = sz
lib.rroi_align_forward_workspace_bytes.argtypes = [it] * 6
f, r = Wk.bench_inputs()
F, R = torch.from_numpy(f).cuda(), torch.from_numpy(r).cuda()
n, C, H, W = 512, 256, 160, 160
top = torch.empty((n, C, 8, 64), device="cuda")
nb = lib.rroi_align_forward_workspace_bytes(1, C, H, W, n, 0)
ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def call(stages):
    assert lib.rroi_align_forward_stages_hip(F.data_ptr(), 0, 0.25, 1, n, H, W, C, 8, 64, R.data_ptr(), top.data_ptr(), ws.data_ptr(), nb, 2, stages, st) == 1
def timeit(fn, warm=200, iters=500):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / iters * 1e3, 2)
res = {}
for rnd in range(2):
    for blocks in (2, 3, 4, 5, 6, 8):
        lib.rroi_align_debug_set_prologue_blocks(blocks)
        for aux in (0, 16):
            lib.rroi_align_debug_set_prologue_aux(aux)
            res.setdefault(f"b{blocks}_aux{aux}", []).append([timeit(lambda: call(1)), timeit(lambda: call(3))])
print(json.dumps(res))
