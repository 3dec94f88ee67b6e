#!/usr/bin/env python3
"""Round 3: fixed cost of the forward gather launch (ramp at the start + uneven last round + launch gap): the gather
alone and the whole call at R = 256 ... 2048 ROIs of the benchmark's distribution; a straight line T = a + b R,
a = what a call pays that does not scale with the work.  Product library, HIP events."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fots.pytorch_amd"), os.path.join(ROOT, "tests")]
import workloads as Wk
from rroi_align._ext import rroi_align as ext
res = {}
f, r_all = Wk.bench_inputs(R=2048)
F = torch.from_numpy(f).cuda()
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, warm=150, iters=400):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
xs, g, w = [], [], []
for R in (256, 512, 768, 1024, 1536, 2048):
    Rr = torch.from_numpy(r_all[:R].copy()).cuda()
    out = torch.empty((R, 256, 8, 64), device="cuda")
    nb = ext._lib.rroi_align_forward_workspace_bytes(1, 256, 160, 160, R, 0)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    def call(stages):
        assert ext._lib.rroi_align_forward_stages_hip(F.data_ptr(), 0, 0.25, 1, R, 160, 160, 256, 8, 64, Rr.data_ptr(), out.data_ptr(), ws.data_ptr(), nb, ext.PATH_TILED, stages, st) == 1
    call(3)
    xs.append(R); g.append(timeit(lambda: call(2))); w.append(timeit(lambda: call(3)))
    res[f"R{R}"] = {"gather_us": round(g[-1], 2), "call_us": round(w[-1], 2)}
for name, y in (("gather", g), ("call", w)):
    b, a = np.polyfit(xs, y, 1)
    res[name + "_fit"] = {"fixed_us": round(float(a), 2), "us_per_512_rois": round(float(b) * 512, 2)}
print(json.dumps(res))
