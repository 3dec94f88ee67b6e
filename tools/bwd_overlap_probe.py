#!/usr/bin/env python3
"""Round 4: would the backward's phases overlap productively?  Two complete backward calls (configs[2]) on two
streams with their own workspaces, issued alternately, against the same number of calls on one stream: if the pair is
faster than two calls in sequence, a bandwidth-bound phase (the relayout of top_diff) is running beside a phase that is
bound by something else (the gather over the copy the memory-side cache holds).
    python tools/bwd_overlap_probe.py"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fots.pytorch_amd"), os.path.join(ROOT, "tests")]
import workloads as Wk  # noqa: E402
from rroi_align._ext import rroi_align as ext  # noqa: E402


def main():
    f, r = Wk.bench_inputs()
    R = torch.from_numpy(r).cuda()
    n, C, H, W = R.shape[0], 256, 160, 160
    nb = ext._lib.rroi_align_backward_workspace_bytes(1, C, H, W, n, 8, 64)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    gout = [torch.randn((n, C, 8, 64), device="cuda") for _ in range(2)]
    gin = [torch.empty((1, C, H, W), device="cuda") for _ in range(2)]
    ws = [torch.empty(nb, dtype=torch.uint8, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()

    def call(i, s):
        st = ext._lib.rroi_align_backward_hip(gout[i].data_ptr(), 0.25, 1, n, H, W, C, 8, 64, R.data_ptr(), gin[i].data_ptr(),
                                              ws[i].data_ptr(), nb, ext.PATH_TILED, s.cuda_stream)
        assert st == 1, st

    def run(two_streams, iters=100):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(10):
            call(0, streams[0])
            call(1, streams[1 if two_streams else 0])
        torch.cuda.synchronize()
        e0.record(streams[0])
        for _ in range(iters):
            call(0, streams[0])
            call(1, streams[1 if two_streams else 0])
        if two_streams:
            streams[0].wait_stream(streams[1])
        e1.record(streams[0])
        torch.cuda.synchronize()
        return round(e0.elapsed_time(e1) / (2 * iters) * 1e3, 2)

    res = {"one_stream_us_per_call": [], "two_streams_us_per_call": []}
    for _ in range(3):
        res["one_stream_us_per_call"].append(run(False))
        res["two_streams_us_per_call"].append(run(True))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
