import os, sys, time
import numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "fots.pytorch_amd"), os.path.join(ROOT, "tests")]
from rroi_align.batched import BatchedRRoiAlign, rois_from_quads
from rroi_align._ext import rroi_align as ext
from test_roi_build import random_quads
dev = torch.device("cuda")
focr = torch.randn(1, 64, 176, 320, device=dev)
quads = torch.from_numpy(random_quads(24, seed=7)).to(dev)
def T(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
b_auto = BatchedRRoiAlign(11, 0.25)
b_fix = BatchedRRoiAlign(11, 0.25, pooled_width=256)
print("batched, width from gw.max().item():", T(lambda: b_auto(focr, quads)))
print("batched, fixed width 256          :", T(lambda: b_fix(focr, quads)))
rois, gw = rois_from_quads(quads)
print("quads_to_rois only                :", T(lambda: rois_from_quads(quads)))
print("forward only (auto path) w=256    :", T(lambda: ext.forward(focr, rois, 11, 256, 0.25)))
print("forward only direct               :", T(lambda: ext.forward(focr, rois, 11, 256, 0.25, path=ext.PATH_DIRECT)))
print("forward only tiled                :", T(lambda: ext.forward(focr, rois, 11, 256, 0.25, path=ext.PATH_TILED)))
print("torch.empty x3                    :", T(lambda: (torch.empty(24*64*11*256, device=dev), torch.empty(1000, device=dev), torch.empty(100, device=dev))))
