#!/usr/bin/env python3
"""Backward pixel lists: exact lists (count / scan / fill, two passes over the ROI geometry)
against one-pass buckets with overflow chains (RROI_PATH_TILED_BUCKETS), interleaved, at cfg3 and
at densities around the bucket size.  Prints us per call and the largest difference."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fots.pytorch_amd"), os.path.join(ROOT, "tests")]
import workloads as Wk  # noqa: E402
from rroi_align._ext import rroi_align as ext  # noqa: E402


def timed(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def case(label, R, C, H, W, img, ph=8, pw=64, batch=1, cl=False):
    f, r = Wk.bench_inputs(R=R, C=C, H=H, W=W, img=img, seed=3, batch=batch)
    Rt = torch.from_numpy(r).cuda()
    g = torch.randn(R, C, ph, pw, device="cuda")
    if cl:
        g = g.contiguous(memory_format=torch.channels_last)
    res = {}
    outs = {}
    for rep in range(3):
        for name, path in (("lists", ext.PATH_TILED_LISTS), ("buckets", ext.PATH_TILED_BUCKETS),
                           ("inkernel", ext.PATH_TILED_INKERNEL), ("auto", ext.PATH_AUTO)) + (
                               (("direct", ext.PATH_DIRECT),) if R <= 64 and not cl else ()):
            fn = lambda: ext.backward(g, Rt, f.shape, 0.25, path=path, channels_last_grad=cl)  # noqa: E731
            try:
                t = timed(fn)
            except ValueError:
                t = float("nan")
                continue
            res.setdefault(name, []).append(t)
            outs[name] = fn()
    d = float((outs["lists"] - outs["buckets"]).abs().max()) if "buckets" in outs else float("nan")
    print(f"{label:34s} " + "  ".join(f"{k} {min(v):7.1f} us" for k, v in res.items()) + f"   max|lists-buckets| {d:.3g}")


case("cfg3 512x256 160x160 NCHW", 512, 256, 160, 160, 640)
case("cfg3 channels-last grad", 512, 256, 160, 160, 640, cl=True)
case("train 32x64 120x160 b2 11x83", 32, 64, 120, 160, 640, ph=11, pw=83, batch=2)
case("dense 512x128 64x64", 512, 128, 64, 64, 256)
case("sparse 64x256 256x256", 64, 256, 256, 256, 1024)
case("2048x64 160x160", 2048, 64, 160, 160, 640)
case("denser 512x128 32x32 (avg 512)", 512, 128, 32, 32, 128)
case("denser 1024x64 32x32 (avg 1024)", 1024, 64, 32, 32, 128)
case("densest 512x64 16x16 (avg 2048)", 512, 64, 16, 16, 64)
case("small 16x256 160x160", 16, 256, 160, 160, 640)
case("small 4x64 176x320", 4, 64, 176, 320, 1280, ph=11, pw=96)
case("cfg4 on one GPU: 4096x256 160x160", 4096, 256, 160, 160, 640)
case("2048x256 160x160", 2048, 256, 160, 160, 640)
case("few ROIs, 16 maps: 8x256 160x160 b16", 8, 256, 160, 160, 640, batch=16)
case("few ROIs, 8 maps: 4x64 160x160 b8", 4, 64, 160, 160, 640, ph=11, pw=100, batch=8)
case("few ROIs: 32x64 176x320", 32, 64, 176, 320, 1280, ph=11, pw=96)
