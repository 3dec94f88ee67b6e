#!/usr/bin/env python3
"""The reference's launcher ABI at BASELINE configs[1] / configs[2] sizes:
`RROIAlignForwardLaucher` with and without con_idx tensors, `RROIAlignBackwardLaucher`."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fots.pytorch_amd"), os.path.join(ROOT, "tests")]
import workloads as Wk  # noqa: E402
from rroi_align._ext import rroi_align as ext  # noqa: E402

f, r = Wk.bench_inputs()
F, R = torch.from_numpy(f).cuda(), torch.from_numpy(r).cuda()
out, ix, iy = (torch.empty(512, 256, 8, 64, device="cuda") for _ in range(3))
gin = torch.zeros_like(F)
st = torch.cuda.current_stream().cuda_stream


def timed(fn, warm=50, n=200):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def fwd(with_idx):
    rc = ext._lib.RROIAlignForwardLaucher(F.data_ptr(), 0.25, 512, 160, 160, 256, 8, 64, R.data_ptr(), out.data_ptr(),
                                          ix.data_ptr() if with_idx else None, iy.data_ptr() if with_idx else None, st)
    assert rc == 1, rc


def bwd():
    rc = ext._lib.RROIAlignBackwardLaucher(out.data_ptr(), 0.25, 1, 512, 160, 160, 256, 8, 64, R.data_ptr(), gin.data_ptr(),
                                           ix.data_ptr(), iy.data_ptr(), st)
    assert rc == 1, rc


nbytes = ext._lib.rroi_align_forward_workspace_bytes(1, 256, 160, 160, 512, 0)
ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")


def native():
    rc = ext._lib.rroi_align_forward_hip(F.data_ptr(), 0, 0.25, 1, 512, 160, 160, 256, 8, 64, R.data_ptr(), out.data_ptr(),
                                         ws.data_ptr(), nbytes, ext.PATH_TILED, st)
    assert rc == 1, rc


for _ in range(2):   # interleaved, same clock: the native entry point (caller's workspace) and the launcher symbol
    print("rroi_align_forward_hip (native, for scale)  : %.1f us per call" % timed(native, 200, 500))
    print("RROIAlignForwardLaucher, con_idx NULL      : %.1f us per call" % timed(lambda: fwd(False), 200, 500))
print("RROIAlignForwardLaucher, con_idx_x/y filled: %.1f us per call" % timed(lambda: fwd(True)))
print("RROIAlignBackwardLaucher                   : %.1f us per call" % timed(bwd, 10, 50))
ref = ext.forward(F, R, 8, 64, 0.25)
fwd(False)
print("forward identical to the native entry point:", bool(torch.equal(out, ref)))
