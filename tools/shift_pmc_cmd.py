#!/usr/bin/env python3
"""A few launches of the forward on crops whose rows are not whole sectors (R = 512, C = 64: 11 x 96 aligned for
comparison, 11 x 100 -> SHIFT = 1, 11 x 83 -> SHIFT = 2), for counter passes (tools/run_pmc.sh "python tools/shift_pmc_cmd.py")."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fots.pytorch_amd"), os.path.join(ROOT, "tests")]
import workloads as Wk
from rroi_align._ext import rroi_align as ext
f, r = Wk.bench_inputs(R=512, C=64, H=160, W=160, img=640, seed=1)
F, R = torch.from_numpy(f).cuda(), torch.from_numpy(r).cuda()
for pw in (96, 100, 83):
    for _ in range(4):
        ext.forward(F, R, 11, pw, 0.25, path=ext.PATH_TILED)
torch.cuda.synchronize()
