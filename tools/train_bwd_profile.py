#!/usr/bin/env python3
"""The backward of the reference's training call (C = 64, two 120 x 160 maps, 11 x 83, R = 512 and R = 32), 60 calls each --
run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fots.pytorch_amd"), os.path.join(ROOT, "tests")]
import workloads as Wk
from rroi_align._ext import rroi_align as ext
for R in (512, 32):
    f, r = Wk.bench_inputs(R=R, C=64, H=120, W=160, img=640, seed=3, batch=2)
    Rt = torch.from_numpy(r).cuda()
    g = torch.randn(R, 64, 11, 83, device="cuda")
    for _ in range(60):
        ext.backward(g, Rt, f.shape, 0.25)
    torch.cuda.synchronize()
