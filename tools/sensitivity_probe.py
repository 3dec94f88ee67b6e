#!/usr/bin/env python3
"""SURVEY 8(d) sensitivity points of the forward call, interleaved with the default draw: every bin active
(w = 8 h), axis-aligned (angle = 0), both; whole call (prologue + gather) and gather alone, us."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fots.pytorch_amd"), os.path.join(ROOT, "tests")]
import workloads as Wk  # noqa: E402
from rroi_align._ext import rroi_align as ext  # noqa: E402

f, r0 = Wk.bench_inputs()
F = torch.from_numpy(f).cuda()
out = torch.empty((512, 256, 8, 64), device="cuda")
nb = ext._lib.rroi_align_forward_workspace_bytes(1, 256, 160, 160, 512, ext.LAYOUT_NCHW)
ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream


def variant(name):
    r = r0.copy()
    if "active" in name:
        r[:, 4] = r[:, 3] * 8
    if "axis" in name:
        r[:, 5] = 0
    if "a45" in name:
        r[:, 5] = 45
    if "a90" in name:
        r[:, 5] = 90
    return torch.from_numpy(r).cuda()


def timed(R, stage, warm=50, n=200):
    def go():
        assert ext._lib.rroi_align_forward_stages_hip(F.data_ptr(), ext.LAYOUT_NCHW, 0.25, 1, 512, 160, 160, 256, 8, 64,
                                                      R.data_ptr(), out.data_ptr(), ws.data_ptr(), nb, ext.PATH_TILED, stage, st) == 1
    for _ in range(warm):
        go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        go()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


timed(variant("default"), ext.STAGE_ALL, 300, int(os.environ.get("RROI_PROBE_SOAK", "300")))
for rep in range(2):
    for name in ("default", "active", "axis", "active+axis", "a45", "a90", "default"):
        R = variant(name)
        print(f"{name:12s} whole call {timed(R, ext.STAGE_ALL):6.2f} us   gather alone {timed(R, ext.STAGE_GATHER):6.2f} us")
if os.environ.get("RROI_PROBE_INPLACE"):
    R = variant("default")
    keep = R.clone()
    for name, edit in (("default", lambda r: None), ("active", lambda r: r[:, 4].copy_(r[:, 3] * 8.0)), ("axis", lambda r: r[:, 5].zero_())):
        R.copy_(keep)
        edit(R)
        print(f"in place {name:10s} whole call {timed(R, ext.STAGE_ALL):6.2f} us", "equal to the fresh tensor:", bool(torch.equal(R, variant(name))))
