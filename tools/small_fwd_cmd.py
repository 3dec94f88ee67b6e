#!/usr/bin/env python3
"""One small forward problem (the reference's training call: C = 64, two 120 x 160 maps, 11 x 96), 40 calls per path --
run under rocprofv3 (--kernel-trace / --pmc) for true kernel durations and counters.  argv: R [path ...]"""
import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fots.pytorch_amd"), os.path.join(ROOT, "tests")]
from rroi_align._ext import rroi_align as ext
R = int(sys.argv[1]) if len(sys.argv) > 1 else 32
paths = [int(v) for v in sys.argv[2:]] or [1, 2]
B, C, H, W, ph, pw = 2, 64, 120, 160, 11, 96
rng = np.random.default_rng(1000 + R + pw)
F = torch.from_numpy(rng.standard_normal((B, C, H, W), dtype=np.float32)).cuda()
h = rng.uniform(16, 64, R)
rois = np.stack([rng.integers(0, B, R), rng.uniform(0, 4 * W, R), rng.uniform(0, 4 * H, R), h,
                 h * rng.uniform(2, pw / float(ph), R), rng.uniform(-45, 45, R)], 1).astype(np.float32)
Rt = torch.from_numpy(rois).cuda()
out = torch.empty((R, C, ph, pw), device="cuda")
nb = ext._lib.rroi_align_forward_workspace_bytes(B, C, H, W, R, 0)
ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for p in paths:
    for _ in range(40):
        assert ext._lib.rroi_align_forward_hip(F.data_ptr(), 0, 0.25, B, R, H, W, C, ph, pw, Rt.data_ptr(), out.data_ptr(),
                                               ws.data_ptr(), nb, p, st) == 1
    torch.cuda.synchronize()
