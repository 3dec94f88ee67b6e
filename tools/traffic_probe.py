#!/usr/bin/env python3
"""The forward call of BASELINE configs[1] (default) or the backward call of configs[2] (argument "backward") a few
times and nothing else -- the process bench.py runs under `rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE` (one
counter per pass) to put the HBM-side bytes of the kernels, measured on THIS box in THIS run, into `roofline.traffic`
and `extra.backward.roofline`.  The product library only."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fots.pytorch_amd"), os.path.join(ROOT, "tests")]
import workloads as Wk  # noqa: E402
from rroi_align._ext import rroi_align as ext  # noqa: E402

f, r = Wk.bench_inputs()                     # 1 x 256 x 160 x 160, 512 ROIs: configs[1]
F, R = torch.from_numpy(f).cuda(), torch.from_numpy(r).cuda()
B, C, H, W = f.shape
out = torch.empty((len(r), C, 8, 64), device="cuda")
nb = ext._lib.rroi_align_forward_workspace_bytes(B, C, H, W, len(r), 0)
ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
calls = int(os.environ.get("RROI_TRAFFIC_CALLS", "6"))
if len(sys.argv) > 1 and sys.argv[1] == "backward":
    g = torch.randn(len(r), C, 8, 64, device="cuda")
    gin = torch.empty((B, C, H, W), device="cuda")
    nbb = ext._lib.rroi_align_backward_workspace_bytes(B, C, H, W, len(r), 8, 64)
    wsb = torch.empty(nbb, dtype=torch.uint8, device="cuda")
    for _ in range(calls):
        rc = ext._lib.rroi_align_backward_hip(g.data_ptr(), 0.25, B, len(r), H, W, C, 8, 64, R.data_ptr(), gin.data_ptr(),
                                              wsb.data_ptr(), nbb, ext.PATH_TILED, st)
        assert rc == 1, rc
    torch.cuda.synchronize()
    sys.exit(0)
for _ in range(calls):
    rc = ext._lib.rroi_align_forward_hip(F.data_ptr(), 0, 0.25, B, len(r), H, W, C, 8, 64, R.data_ptr(), out.data_ptr(),
                                         ws.data_ptr(), nb, ext.PATH_TILED, st)
    assert rc == 1, rc
torch.cuda.synchronize()
