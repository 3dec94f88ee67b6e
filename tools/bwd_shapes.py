#!/usr/bin/env python3
"""Backward by pooled size (rows of top_diff that are / are not whole sectors) and map width: us per call."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fots.pytorch_amd"), os.path.join(ROOT, "tests")]
import workloads as Wk
from rroi_align._ext import rroi_align as ext
def timed(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for (R, C, H, W, ph, pw) in ((512, 64, 160, 160, 11, 96), (512, 64, 160, 160, 11, 100), (512, 64, 160, 160, 11, 83),
                            (512, 256, 160, 160, 8, 64), (512, 256, 160, 160, 8, 62), (512, 256, 160, 160, 7, 61),
                            (512, 64, 150, 157, 11, 96), (512, 256, 150, 157, 8, 64)):
    f, r = Wk.bench_inputs(R=R, C=C, H=H, W=W, img=4 * W, seed=2)
    Rt = torch.from_numpy(r).cuda()
    g = torch.randn(R, C, ph, pw, device="cuda")
    t = timed(lambda: ext.backward(g, Rt, f.shape, 0.25))
    mb = (R * C * ph * pw + C * H * W) * 4 / 1e6
    print(f"R={R} C={C} map {H}x{W} pooled {ph}x{pw} (rows {ph*pw*4} B, mod 64 = {ph*pw*4 % 64}): {t:7.1f} us  ({mb:6.1f} MB algorithmic -> {mb / t * 1e-3:5.2f} TB/s)")
