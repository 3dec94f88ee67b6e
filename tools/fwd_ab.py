#!/usr/bin/env python3
"""Round 4: A/B of the forward call's round-4 candidates at BASELINE configs[1], through the explore build
(tools/build_explore.sh): the gather's left-over items claimed dynamically (dyn_tail) against dealt to the first slots;
the gather's AQL packet with / without the barrier bit (hipExtAnyOrderLaunch); prologue stores plain / write-through.
HIP events around back-to-back calls, interleaved rounds; every variant's crops compared with the first call's.
    python tools/fwd_ab.py [rounds] [all_active]
Needs tools/experiments/r04_forward_tail_experiments.patch applied (`git apply` it, `sh tools/build_explore.sh`): the
dynamic tail and the any-order launch were measured and NOT shipped (profiles/r04_forward_floor.md)."""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import workloads as Wk  # noqa: E402

lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_explore", "librroi_align_hip_explore.so"))
vp, fl, it, sz = ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_size_t
lib.rroi_align_forward_stages_hip.argtypes = [vp, it, fl, it, it, it, it, it, it, it, vp, vp, vp, sz, it, it, vp]
lib.rroi_align_forward_workspace_bytes.restype = sz
lib.rroi_align_forward_workspace_bytes.argtypes = [it] * 6


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
    f, r = Wk.bench_inputs(all_active=len(sys.argv) > 2 and sys.argv[2] == "all_active")
    F, R = torch.from_numpy(f).cuda(), torch.from_numpy(r).cuda()
    n, C, H, W = R.shape[0], F.shape[1], F.shape[2], F.shape[3]
    top = torch.empty((n, C, 8, 64), device="cuda")
    ref = torch.empty_like(top)
    nbytes = lib.rroi_align_forward_workspace_bytes(1, C, H, W, n, 0)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def call(stages=3, out=top):
        st = lib.rroi_align_forward_stages_hip(F.data_ptr(), 0, 0.25, 1, n, H, W, C, 8, 64, R.data_ptr(), out.data_ptr(),
                                               ws.data_ptr(), nbytes, 2, stages, stream)
        assert st == 1, st
    lib.rroi_align_debug_set_fwd_dyn_tail(0)
    call(3, ref)
    torch.cuda.synchronize()
    res = {}
    for rnd in range(rounds):
        for name, dyn, anyo, paux in (("static", 0, 0, 0), ("dyn_tail", 1, 0, 0), ("static_anyorder", 0, 1, 0),
                                      ("static_free_first_item", 0, 0, 0)):
            lib.rroi_align_debug_set_fwd_dbg(256 if name == "static_free_first_item" else 0)
            lib.rroi_align_debug_set_fwd_dyn_tail(dyn)
            lib.rroi_align_debug_set_fwd_anyorder(anyo)
            lib.rroi_align_debug_set_prologue_aux(paux)
            res.setdefault(name + "_step", []).append(timeit(call))
            res.setdefault(name + "_gather", []).append(timeit(lambda: call(2)))
            # equality after a fresh map (the copy is rewritten by every call): 20 calls, compare the last
            top.zero_()
            for _ in range(20):
                call()
            torch.cuda.synchronize()
            res.setdefault(name + "_equal", []).append(bool(torch.equal(top, ref)))
    lib.rroi_align_debug_set_fwd_anyorder(0)
    lib.rroi_align_debug_set_prologue_aux(0)
    lib.rroi_align_debug_set_fwd_dyn_tail(1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
