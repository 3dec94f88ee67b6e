#!/usr/bin/env python3
"""Round 6: configs[1] with TWO calls in flight -- calls alternate between two streams, each with its own crops and
workspace, so that one call's prologue can run beside the other's gather.  Is the sequential call's 53.4 us a property of
the two launches' ORDER (then overlap helps) or of the bytes they move (372 MB at ~7 TB/s of fabric traffic: then it does not)?
us per call between events on the default stream around N calls (all streams joined at both ends)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fots.pytorch_amd")]
import bench
from rroi_align._ext import rroi_align as ext
c = bench.CFG
dev = torch.device("cuda", 0)
f, r = bench.make_inputs(512)
feats, rois = torch.from_numpy(f).to(dev), torch.from_numpy(r).to(dev)
nb = ext._lib.rroi_align_forward_workspace_bytes(1, c["C"], c["H"], c["W"], 512, ext.LAYOUT_NCHW)
def run(nstreams, calls=600, warm=300):
    streams = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]
    outs = [torch.empty((512, c["C"], c["PH"], c["PW"]), device=dev) for _ in range(nstreams)]
    wss = [torch.empty(nb, dtype=torch.uint8, device=dev) for _ in range(nstreams)]
    def call(i):
        s = i % nstreams
        st = ext._lib.rroi_align_forward_stages_hip(feats.data_ptr(), ext.LAYOUT_NCHW, c["scale"], 1, 512, c["H"], c["W"], c["C"], c["PH"], c["PW"],
                                                    rois.data_ptr(), outs[s].data_ptr(), wss[s].data_ptr(), nb, ext.PATH_TILED, ext.STAGE_ALL, streams[s].cuda_stream)
        assert st == 1
    for i in range(warm): call(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(calls): call(i)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / calls * 1e6
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    return wall, same
for rep in range(3):
    row = []
    for n in (1, 2, 3, 1):
        w, same = run(n)
        row.append(f"{n} stream(s): {w:6.2f} us per call (crops identical: {same})")
    print("  ".join(row), flush=True)
