#!/usr/bin/env python3
"""The forward of the reference's training call (C = 64, two 120 x 160 maps, R = 512 / 32; 11 x 96, 11 x 83), 60 calls each --
run under `rocprofv3 --kernel-trace` for the per-kernel timeline."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fots.pytorch_amd"), os.path.join(ROOT, "tests")]
import workloads as Wk
from rroi_align._ext import rroi_align as ext
st = torch.cuda.current_stream().cuda_stream
for (R, pw) in ((512, 96), (512, 83), (32, 96), (24, 128)):
    f, r = Wk.bench_inputs(R=R, C=64, H=120, W=160, img=640, seed=3, batch=2)
    F, Rt = torch.from_numpy(f).cuda(), torch.from_numpy(r).cuda()
    out = torch.empty((R, 64, 11, pw), device="cuda")
    nb = ext._lib.rroi_align_forward_workspace_bytes(2, 64, 120, 160, R, 0)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    for _ in range(60):
        assert ext._lib.rroi_align_forward_hip(F.data_ptr(), 0, 0.25, 2, R, 120, 160, 64, 11, pw, Rt.data_ptr(), out.data_ptr(),
                                               ws.data_ptr(), nb, ext.PATH_AUTO, st) == 1
    torch.cuda.synchronize()
