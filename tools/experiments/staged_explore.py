#!/usr/bin/env python3
"""Round 3: ablations and knob sweeps of the staged forward at BASELINE configs[1], through the
explore build of the library (tools/build_explore.sh).  Prints one JSON object.
    python tools/staged_explore.py"""
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
lib.rroi_align_forward_hip.argtypes = [vp, it, fl, it, it, it, it, it, it, it, vp, vp, vp, sz, it, vp]
lib.rroi_align_forward_workspace_bytes.restype = sz
lib.rroi_align_forward_workspace_bytes.argtypes = [it] * 6
PATH_TILED, PATH_STAGED = 2, 6


def timeit(fn, warm=60, iters=200):
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
    kw = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
    f, r = Wk.bench_inputs(**{k: int(v) for k, v in kw.items() if k in ("R", "C", "H", "W", "img", "seed")})
    if "angle" in kw:
        r[:, 5] = float(kw["angle"])
    if "hmax" in kw:   # smaller boxes: h ~ U[16, hmax)
        rng = np.random.default_rng(1)
        ratio = r[:, 4] / r[:, 3]
        r[:, 3] = rng.uniform(16, float(kw["hmax"]), len(r)).astype(np.float32)
        r[:, 4] = r[:, 3] * ratio
    F, R = torch.from_numpy(f).cuda(), torch.from_numpy(r).cuda()
    n, C, H, W = R.shape[0], F.shape[1], F.shape[2], F.shape[3]
    top = torch.empty((n, C, 8, 64), device="cuda")
    nbytes = lib.rroi_align_forward_workspace_bytes(1, C, H, W, n, 0)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def call(path):
        st = lib.rroi_align_forward_hip(F.data_ptr(), 0, 0.25, 1, n, H, W, C, 8, 64, R.data_ptr(), top.data_ptr(),
                                        ws.data_ptr(), nbytes, path, stream)
        assert st == 1, st
    res = {}
    if "short" in sys.argv:
        res["tiled"] = timeit(lambda: call(PATH_TILED))
        for dbg in (0, 1, 2, 4):
            lib.rroi_align_debug_set_fwd_dbg(dbg)
            res[f"staged_dbg{dbg}"] = timeit(lambda: call(PATH_STAGED))
        print(json.dumps(res))
        return
    res["tiled"] = timeit(lambda: call(PATH_TILED))
    res["staged"] = timeit(lambda: call(PATH_STAGED))
    for wgs in (4, 6, 7, 8, 9, 10):
        lib.rroi_align_debug_set_staged(wgs, -1)
        res[f"staged_wgs{wgs}"] = timeit(lambda: call(PATH_STAGED))
    lib.rroi_align_debug_set_staged(8, -1)
    for aux in (0, 16, 17, 2):
        lib.rroi_align_debug_set_staged(-1, aux)
        res[f"staged_aux{aux}"] = timeit(lambda: call(PATH_STAGED))
    for dbg in (1, 2, 3, 4, 5, 6, 7, 0):
        lib.rroi_align_debug_set_fwd_dbg(dbg)
        res[f"staged_dbg{dbg}"] = timeit(lambda: call(PATH_STAGED))
    res["tiled_again"] = timeit(lambda: call(PATH_TILED))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
