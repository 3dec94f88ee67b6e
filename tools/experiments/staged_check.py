#!/usr/bin/env python3
"""Round 3: the staged forward (one launch, NCHW in place) against the tiled and direct paths --
bit for bit on a spread of shapes -- and timed against prologue + tiled gather at BASELINE configs[1].
    python tools/staged_check.py [quick]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fots.pytorch_amd"), os.path.join(ROOT, "tests")]
import workloads as Wk  # noqa: E402
from rroi_align._ext import rroi_align as ext  # noqa: E402


def same(a, b):
    return bool(((a == b) | (a.isnan() & b.isnan())).all())


def check():
    bad = 0
    cases = [
        ("cfg2", Wk.bench_inputs(), 8, 64, 0.25),
        ("cfg1", Wk.cfg1_inputs(), 8, 32, 1.0),
        ("train", Wk.bench_inputs(R=32, C=64, H=120, W=160, img=640, seed=5, batch=2), 11, 83, 0.25),
        ("image", Wk.bench_inputs(R=3, C=3, H=276, W=500, img=500, seed=6), 44, 349, 1.0),
        ("c33", Wk.bench_inputs(R=40, C=33, H=64, W=96, img=384, seed=7), 8, 64, 0.25),
        ("sparse", Wk.bench_inputs(R=64, C=64, H=176, W=320, img=1280, seed=8), 8, 64, 0.25),
        ("allact", Wk.bench_inputs(R=64, C=32, seed=9, all_active=True), 8, 64, 0.25),
        ("axis", Wk.bench_inputs(R=64, C=32, seed=10, axis_aligned=True), 8, 64, 0.25),
    ]
    f0 = Wk.bench_inputs(R=1, C=8, seed=3)[0]
    cases.append(("edge", (f0, Wk.edge_rois()), 8, 64, 0.25))
    cases.append(("degen", (f0, Wk.degenerate_rois()), 8, 64, 0.25))
    cases.append(("ties", (f0, Wk.tie_rois()), 8, 64, 0.25))
    cases.append(("ph1", (f0, Wk.edge_rois()), 1, 7, 0.25))
    cases.append(("pw100", (f0, Wk.edge_rois()), 11, 100, 0.25))
    for name, (f, r), ph, pw, s in cases:
        F, R = torch.from_numpy(f).cuda(), torch.from_numpy(r).cuda()
        want = ext.forward(F, R, ph, pw, s, path=ext.PATH_DIRECT)
        got = ext.forward(F, R, ph, pw, s, path=ext.PATH_STAGED)
        torch.cuda.synchronize()
        ok = same(got, want)
        nd = int((~((got == want) | (got.isnan() & want.isnan()))).sum())
        print(f"{name:8s} {tuple(got.shape)} staged == direct: {ok} ({nd} differ)", flush=True)
        bad += not ok
    return bad


def timeit(fn, warm=100, iters=400):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def bench():
    f, r = Wk.bench_inputs()
    F, R = torch.from_numpy(f).cuda(), torch.from_numpy(r).cuda()
    n, C, H, W = R.shape[0], F.shape[1], F.shape[2], F.shape[3]
    top = torch.empty((n, C, 8, 64), device="cuda")
    nbytes = ext._lib.rroi_align_forward_workspace_bytes(1, C, H, W, n, ext.LAYOUT_NCHW)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def call(path):
        st = ext._lib.rroi_align_forward_hip(F.data_ptr(), ext.LAYOUT_NCHW, 0.25, 1, n, H, W, C, 8, 64, R.data_ptr(),
                                             top.data_ptr(), ws.data_ptr(), nbytes, path, stream)
        assert st == 1, st
    out = {}
    for name, path in (("tiled", ext.PATH_TILED), ("staged", ext.PATH_STAGED), ("tiled2", ext.PATH_TILED),
                       ("staged2", ext.PATH_STAGED)):
        out[name + "_us"] = round(timeit(lambda: call(path)), 2)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    bad = check()
    print("mismatching cases:", bad)
    if bad == 0 and "quick" not in sys.argv:
        bench()
