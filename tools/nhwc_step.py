#!/usr/bin/env python3
"""The forward call when the producer hands over channels-last features (consumed in place, no
prologue): 1000 back-to-back calls of rroi_align_forward_hip with layout NHWC, bench shapes."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fots.pytorch_amd"), os.path.join(ROOT, "tests")]
import workloads as Wk
from rroi_align._ext import rroi_align as ext
f, r = Wk.bench_inputs()
F = torch.from_numpy(f).cuda().contiguous(memory_format=torch.channels_last)  # storage (B, H, W, C)
R = torch.from_numpy(r).cuda()
out = torch.empty((512, 256, 8, 64), device="cuda")
nb = ext._lib.rroi_align_forward_workspace_bytes(1, 256, 160, 160, 512, ext.LAYOUT_NHWC)
ws = torch.empty(max(nb, 1), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def call():
    rc = ext._lib.rroi_align_forward_hip(F.data_ptr(), ext.LAYOUT_NHWC, 0.25, 1, 512, 160, 160, 256, 8, 64,
                                         R.data_ptr(), out.data_ptr(), ws.data_ptr(), nb, ext.PATH_TILED, st)
    assert rc == 1, rc
for _ in range(300): call()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(1000): call()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 1000
Fn = torch.from_numpy(f).cuda()
nbn = ext._lib.rroi_align_forward_workspace_bytes(1, 256, 160, 160, 512, ext.LAYOUT_NCHW)
wsn = torch.empty(nbn, dtype=torch.uint8, device="cuda")
ocl = torch.empty((512, 256, 8, 64), device="cuda", memory_format=torch.channels_last)
for name, feats, lay, wsx, nbx in (("NCHW features -> channels-last crops", Fn, ext.LAYOUT_NCHW, wsn, nbn),
                                    ("channels-last features -> channels-last crops", F, ext.LAYOUT_NHWC, ws, nb)):
    def c2():
        rc = ext._lib.rroi_align_forward_layout_hip(feats.data_ptr(), lay, ext.LAYOUT_NHWC, 0.25, 1, 512, 160, 160,
                                                    256, 8, 64, R.data_ptr(), ocl.data_ptr(), wsx.data_ptr(), nbx,
                                                    ext.PATH_TILED, st)
        assert rc == 1, rc
    for _ in range(300): c2()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(1000): c2()
    torch.cuda.synchronize(); d2 = (time.perf_counter() - t0) / 1000
    print(f"{name}: {d2 * 1e6:.2f} us -> {512 / d2 / 1e6:.2f} M ROIs/s")
ref = ext.forward(torch.from_numpy(f).cuda(), R, 8, 64, 0.25)
print("channels-last crops identical:", torch.equal(ocl, ref))
print(f"channels-last forward call: {dt * 1e6:.2f} us -> {512 / dt / 1e6:.2f} M ROIs/s, "
      f"{294332416 / dt / 1e12:.2f} TB/s of algorithmic bytes; identical to the NCHW result: {torch.equal(out, ref)}")
