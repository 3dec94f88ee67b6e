#!/usr/bin/env python3
"""Round 3: channels-last crops (R, PH, PW, C storage) written by the one-wave kernel or by the split kernel;
explore build.  Equality and timing at BASELINE configs[1] sizes, NCHW and channels-last features."""
import ctypes, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import workloads as Wk
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_explore", "librroi_align_hip_explore.so"))
vp, fl, it, sz = ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_size_t
lib.rroi_align_forward_layout_hip.argtypes = [vp, it, it, fl, it, it, it, it, it, it, it, vp, vp, vp, sz, it, vp]
lib.rroi_align_forward_workspace_bytes.restype = sz
lib.rroi_align_forward_workspace_bytes.argtypes = [it] * 6
f, r = Wk.bench_inputs()
R = torch.from_numpy(r).cuda()
n, C, H, W = 512, 256, 160, 160
st = torch.cuda.current_stream().cuda_stream
res = {}
for lay, F in ((0, torch.from_numpy(f).cuda()), (1, torch.from_numpy(f).cuda().contiguous(memory_format=torch.channels_last))):
    nb = lib.rroi_align_forward_workspace_bytes(1, C, H, W, n, lay)
    ws = torch.empty(max(nb, 1), dtype=torch.uint8, device="cuda")
    a, b = torch.empty((n, 8, 64, C), device="cuda"), torch.empty((n, 8, 64, C), device="cuda")
    def call(out):
        assert lib.rroi_align_forward_layout_hip(F.data_ptr(), lay, 1, 0.25, 1, n, H, W, C, 8, 64, R.data_ptr(), out.data_ptr(), ws.data_ptr(), nb, 2, st) == 1
    lib.rroi_align_debug_set_fwd_split(0, -1); call(a)
    lib.rroi_align_debug_set_fwd_split(1, -1); call(b)
    torch.cuda.synchronize()
    res[f"features_layout{lay}_equal"] = bool(torch.equal(a, b))
    def timeit(fn, warm=200, iters=500):
        for _ in range(warm): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        return round(e0.elapsed_time(e1) / iters * 1e3, 2)
    for rnd in range(2):
        for name, on in (("one_wave", 0), ("split", 1)):
            lib.rroi_align_debug_set_fwd_split(on, -1)
            res.setdefault(f"features_layout{lay}_{name}", []).append(timeit(lambda: call(a)))
print(json.dumps(res))
