#!/usr/bin/env python3
"""What ANY backward of the training shape pays before it looks at a ROI: the feature gradient (2 x 64 x 120 x 160 fp32 = 9.8 MB) has to
be written once.  us per call, back-to-back between HIP events: torch zero_() of that tensor, the shipped AUTO backward at R = 1."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, "fots.pytorch_amd"))
from rroi_align._ext import rroi_align as ext
def timeit(fn, warm=500, iters=500):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
gin = torch.empty((2, 64, 120, 160), device="cuda")
small = torch.empty((64,), device="cuda")
print("zero_ of 9.8 MB: %.1f us; zero_ of 256 B (launch rate): %.1f us" % (timeit(lambda: gin.zero_()), timeit(lambda: small.zero_())))
for R in (1, 32):
    rois = torch.tensor([[0, 300, 200, 32, 200, 10]] * R, dtype=torch.float32, device="cuda")
    g = torch.randn((R, 64, 11, 96), device="cuda")
    print("AUTO backward R=%d: %.1f us" % (R, timeit(lambda: ext.backward(g, rois, (2, 64, 120, 160), 0.25), 200, 200)))
