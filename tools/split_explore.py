#!/usr/bin/env python3
"""Round 3: A/B of the forward gather with loads and stores in one wave (rroi_fwd_tiled_kernel) against
loads and stores in different waves (rroi_fwd_split_kernel), at BASELINE configs[1], through the explore
build of the library (tools/build_explore.sh).  Whole call (prologue + gather) and gather alone, HIP events
around back-to-back launches, interleaved rounds.  Prints one JSON object.
    python tools/split_explore.py [rounds]"""
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import workloads as Wk  # noqa: E402

lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_explore", "librroi_align_hip_explore.so"))
vp, fl, it, sz = ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_size_t
lib.rroi_align_forward_stages_hip.argtypes = [vp, it, fl, it, it, it, it, it, it, it, vp, vp, vp, sz, it, it, vp]
lib.rroi_align_forward_workspace_bytes.restype = sz
lib.rroi_align_forward_workspace_bytes.argtypes = [it] * 6
PATH_TILED = 2


def timeit(fn, warm=200, iters=500):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / iters * 1e3, 2)  # us


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    f, r = Wk.bench_inputs()
    if os.environ.get("RROI_ALL_ACTIVE"):   # SURVEY 8(d) sensitivity point: w = 8 h, no masked bins
        r[:, 4] = r[:, 3] * 8
    F, R = torch.from_numpy(f).cuda(), torch.from_numpy(r).cuda()
    n, C, H, W = R.shape[0], F.shape[1], F.shape[2], F.shape[3]
    top = torch.empty((n, C, 8, 64), device="cuda")
    ref = torch.empty_like(top)
    nbytes = lib.rroi_align_forward_workspace_bytes(1, C, H, W, n, 0)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def call(stages, out=top):
        st = lib.rroi_align_forward_stages_hip(F.data_ptr(), 0, 0.25, 1, n, H, W, C, 8, 64, R.data_ptr(), out.data_ptr(),
                                               ws.data_ptr(), nbytes, PATH_TILED, stages, stream)
        assert st == 1, st
    lib.rroi_align_debug_set_fwd_split(0, -1)
    call(3, ref)
    lib.rroi_align_debug_set_fwd_split(1, -1)
    call(3, top)
    torch.cuda.synchronize()
    res = {"split_equals_tiled": bool(torch.equal(top, ref))}
    for rnd in range(rounds):
        for name, on, wgs in (("tiled", 0, -1), ("split10", 1, 10), ("e0o6h3w12", 2, 12), ("e1o6h3w12", 3, 12), ("e2o6h3w12", 4, 12), ("e2o5h3w10", 5, 10), ("e0o6h1w12", 6, 12), ("split8", 1, 8), ("split12", 1, 12)):
            lib.rroi_align_debug_set_fwd_split(on, wgs)
            res.setdefault(name + "_step", []).append(timeit(lambda: call(3)))
            res.setdefault(name + "_gather", []).append(timeit(lambda: call(2)))
    lib.rroi_align_debug_set_fwd_split(1, 10)
    for dbg in (0,):
        lib.rroi_align_debug_set_fwd_dbg(dbg)
        res.setdefault(f"split10_dbg{dbg}_gather", []).append(timeit(lambda: call(2)))
        res.setdefault(f"split10_dbg{dbg}_step", []).append(timeit(lambda: call(3)))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
