#!/usr/bin/env python3
"""Backward at BASELINE cfg3 sizes: wall time per call of every path, then 20 calls of the default
tiled path -- run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fots.pytorch_amd"), os.path.join(ROOT, "tests")]
import workloads as Wk  # noqa: E402
from rroi_align._ext import rroi_align as ext  # noqa: E402

CH = int(os.environ.get("RROI_BWD_C", "256"))   # channel count (configs[2]: 256)
f, r = Wk.bench_inputs(C=CH)
R = torch.from_numpy(r).cuda()
g = torch.randn(512, CH, 8, 64, device="cuda")
B, C, H, W = f.shape
nb = ext._lib.rroi_align_backward_workspace_bytes(B, C, H, W, 512, 8, 64)
ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
gin = torch.empty((B, C, H, W), device="cuda")
st = torch.cuda.current_stream().cuda_stream


def call(path):
    rc = ext._lib.rroi_align_backward_hip(g.data_ptr(), 0.25, B, 512, H, W, C, 8, 64, R.data_ptr(),
                                          gin.data_ptr(), ws.data_ptr(), nb, path, st)
    assert rc == 1, rc


if os.environ.get("RROI_BWD_ONLY"):   # the default path alone (for counter passes: one kind of launch per kernel name)
    for _ in range(30):
        call(ext.PATH_TILED)
    torch.cuda.synchronize()
    sys.exit(0)
for name, path, n in (("tiled (default: lists in HBM for C > 64)", ext.PATH_TILED, 50),
                      ("tiled_lists (count / scan / fill)", ext.PATH_TILED_LISTS, 50),
                      ("tiled_buckets (one-pass buckets + overflow chains)", ext.PATH_TILED_BUCKETS, 50),
                      ("tiled_inkernel (lists built in the gather kernel)", ext.PATH_TILED_INKERNEL, 50),
                      ("tiled_atomic (scatter)", ext.PATH_TILED_ATOMIC, 20),
                      ("direct", ext.PATH_DIRECT, 5)):
    for _ in range(3):
        call(path)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        call(path)
    torch.cuda.synchronize()
    print(f"backward cfg3 {name}: {(time.perf_counter() - t0) / n * 1e6:.1f} us  (workspace {nb / 1e6:.1f} MB)")
g_cl = g.contiguous(memory_format=torch.channels_last)


def call_cl():
    rc = ext._lib.rroi_align_backward_layout_hip(g_cl.data_ptr(), ext.LAYOUT_NHWC, ext.LAYOUT_NCHW, 0.25, B, 512, H, W, C, 8, 64,
                                                 R.data_ptr(), gin.data_ptr(), ws.data_ptr(), nb, ext.PATH_TILED, st)
    assert rc == 1, rc


for _ in range(3):
    call_cl()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    call_cl()
torch.cuda.synchronize()
print(f"backward cfg3 tiled (gather), channels-last top_diff consumed in place: {(time.perf_counter() - t0) / 50 * 1e6:.1f} us")
if os.environ.get("RROI_BWD_SWEEP") and hasattr(ext._lib, "rroi_align_debug_set_bwd_relayout_aux"):  # make EXPLORE=1
    for raux in (0, 2, 16):
        ext._lib.rroi_align_debug_set_bwd_relayout_aux(raux)
        for _ in range(5):
            call(ext.PATH_TILED)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            call(ext.PATH_TILED)
        torch.cuda.synchronize()
        print(f"backward sweep: relayout store aux={raux}: {(time.perf_counter() - t0) / 50 * 1e6:.1f} us")
    ext._lib.rroi_align_debug_set_bwd_relayout_aux(2)
for _ in range(20):
    call(ext.PATH_TILED)
torch.cuda.synchronize()
