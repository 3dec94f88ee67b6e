#!/usr/bin/env python3
"""Backward at BASELINE cfg3 sizes, 20 calls -- run under `rocprofv3 --kernel-trace --stats`."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fots.pytorch_amd"), os.path.join(ROOT, "tests")]
import workloads as Wk  # noqa: E402
from rroi_align._ext import rroi_align as ext  # noqa: E402

f, r = Wk.bench_inputs()
R = torch.from_numpy(r).cuda()
g = torch.randn(512, 256, 8, 64, device="cuda")
for _ in range(20):
    ext.backward(g, R, f.shape, 0.25, path=ext.PATH_TILED)
torch.cuda.synchronize()
