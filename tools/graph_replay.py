#!/usr/bin/env python3
"""The library's launches under HIP graph capture (torch.cuda.CUDAGraph): correctness of the replay
and the wall time per call, eager vs replay, for the launch-bound inference regime (24 boxes, 64 ch)
and for the bench shapes."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fots.pytorch_amd"), os.path.join(ROOT, "tests")]
import workloads as Wk
from rroi_align._ext import rroi_align as ext


def run(name, R, C, H, W, ph, pw, path):
    f, r = Wk.bench_inputs(R=R, C=C, H=H, W=W, img=4 * W, seed=R)
    F, Rr = torch.from_numpy(f).cuda(), torch.from_numpy(r).cuda()
    out = torch.empty((R, C, ph, pw), device="cuda")
    nb = ext._lib.rroi_align_forward_workspace_bytes(1, C, H, W, R, 0)
    ws = torch.empty(max(nb, 1), dtype=torch.uint8, device="cuda")

    def call():
        st = torch.cuda.current_stream().cuda_stream
        rc = ext._lib.rroi_align_forward_hip(F.data_ptr(), 0, 0.25, 1, R, H, W, C, ph, pw, Rr.data_ptr(),
                                             out.data_ptr(), ws.data_ptr(), nb, path, st)
        assert rc == 1, rc
    call(); torch.cuda.synchronize(); want = out.clone()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        call()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        call()
    out.zero_(); g.replay(); torch.cuda.synchronize()
    ok = torch.equal(out, want)

    def T(fn, n=500):
        for _ in range(50): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
    print(f"{name}: eager {T(call):.1f} us, graph replay {T(g.replay):.1f} us, replay identical: {ok}")


run("inference, 24 boxes x 64 ch x 11x256, tiled (2 launches)", 24, 64, 176, 320, 11, 256, ext.PATH_TILED)
run("inference, 24 boxes x 64 ch x 11x256, direct (1 launch)", 24, 64, 176, 320, 11, 256, ext.PATH_DIRECT)
run("bench shapes, tiled", 512, 256, 160, 160, 8, 64, ext.PATH_TILED)
run("bench shapes, tiled (again)", 512, 256, 160, 160, 8, 64, ext.PATH_TILED)
