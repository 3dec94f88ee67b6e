#!/usr/bin/env python3
"""Direct vs tiled paths by ROI count (validates the AUTO rule `pick_tiled`): wall time per call."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fots.pytorch_amd"), os.path.join(ROOT, "tests")]
import workloads as Wk
from rroi_align._ext import rroi_align as ext

def T(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6

for C, H, W, ph, pw, B in ((256, 160, 160, 8, 64, 1), (64, 176, 320, 11, 96, 1), (64, 160, 160, 11, 100, 8)):
    for R in (4, 16, 32, 64, 128, 256, 512):
        f, r = Wk.bench_inputs(R=R, C=C, H=H, W=W, img=4 * W, seed=R, batch=B)
        F, Rr = torch.from_numpy(f).cuda(), torch.from_numpy(r).cuda()
        g = torch.randn(R, C, ph, pw, device="cuda")
        fd = T(lambda: ext.forward(F, Rr, ph, pw, 0.25, path=ext.PATH_DIRECT))
        ft = T(lambda: ext.forward(F, Rr, ph, pw, 0.25, path=ext.PATH_TILED))
        fa = T(lambda: ext.forward(F, Rr, ph, pw, 0.25))
        bd = T(lambda: ext.backward(g, Rr, f.shape, 0.25, path=ext.PATH_DIRECT), 20)
        bt = T(lambda: ext.backward(g, Rr, f.shape, 0.25, path=ext.PATH_TILED), 20)
        ba = T(lambda: ext.backward(g, Rr, f.shape, 0.25), 20)
        bl = T(lambda: ext.backward(g, Rr, f.shape, 0.25, path=ext.PATH_TILED_LISTS), 20)
        bk = T(lambda: ext.backward(g, Rr, f.shape, 0.25, path=ext.PATH_TILED_INKERNEL), 20)
        bb = T(lambda: ext.backward(g, Rr, f.shape, 0.25, path=ext.PATH_TILED_BUCKETS), 20)
        ratio = R * C * ph * pw / (B * C * H * W)
        print(f"B={B} C={C} {H}x{W} {ph}x{pw} R={R:4d} out/map={ratio:6.2f}  fwd direct {fd:7.1f} tiled {ft:7.1f} auto {fa:7.1f} | bwd direct {bd:8.1f} tiled {bt:7.1f} (exact lists {bl:7.1f}, buckets {bb:7.1f}, in-kernel {bk:7.1f}) auto {ba:8.1f}")
