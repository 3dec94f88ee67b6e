#!/usr/bin/env python3
"""The training shape's backward (C = 64, two 120 x 160 maps, 11 x PW, R ROIs) by path, 40 calls each -- run under rocprofv3
--kernel-trace: <2, 16> = pairs || relayout in one launch (buckets), <0, 16> = the relayout alone (in-kernel lists).
argv: R PW"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fots.pytorch_amd"), os.path.join(ROOT, "tests")]
from rroi_align._ext import rroi_align as ext
R = int(sys.argv[1]) if len(sys.argv) > 1 else 512
PW = int(sys.argv[2]) if len(sys.argv) > 2 else 96
B, C, H, W = 2, 64, 120, 160
rng = np.random.default_rng(1000 + R + PW)
h = rng.uniform(16, 64, R)
rois = torch.from_numpy(np.stack([rng.integers(0, B, R), rng.uniform(0, 4 * W, R), rng.uniform(0, 4 * H, R), h,
                                  h * rng.uniform(2, PW / 11.0, R), rng.uniform(-45, 45, R)], 1).astype(np.float32)).cuda()
g = torch.randn(R, C, 11, PW, device="cuda")
for path in (ext.PATH_TILED_BUCKETS, ext.PATH_TILED_INKERNEL, ext.PATH_TILED_LISTS):
    for _ in range(40):
        ext.backward(g, rois, (B, C, H, W), 0.25, path=path)
    torch.cuda.synchronize()
