#!/usr/bin/env python3
"""cfg3: the backward call over the bucket lists (relayout || pair pass in one launch, <2, 16>) and over the lists built
inside the gather (the relayout launch alone, <0, 16>) -- run under rocprofv3 --kernel-trace --stats to read what the
pair pass adds to the relayout it shares a launch with."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fots.pytorch_amd"), os.path.join(ROOT, "tests")]
import workloads as Wk
from rroi_align._ext import rroi_align as ext
f, r = Wk.bench_inputs(C=256)
R = torch.from_numpy(r).cuda()
g = torch.randn(512, 256, 8, 64, device="cuda")
for path in (ext.PATH_TILED_BUCKETS, ext.PATH_TILED_INKERNEL):
    for _ in range(60):
        ext.backward(g, R, f.shape, 0.25, path=path)
    torch.cuda.synchronize()
