#!/usr/bin/env python3
"""Round 4: where the gather launch's fixed cost sits.  Per-workgroup time stamps of rroi_fwd_split_kernel
(explore build: storer entry, the gatherer's first loads, the storer's first store, storer exit; 100 MHz clock)
for one launch inside a run of back-to-back calls at BASELINE configs[1]; prints the distribution of each stamp
relative to the first workgroup's entry.
    python tools/wg_trace.py"""
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
lib.rroi_align_debug_set_wg_trace.argtypes = [vp]


def main():
    f, r = Wk.bench_inputs()
    F, R = torch.from_numpy(f).cuda(), torch.from_numpy(r).cuda()
    n, C, H, W = R.shape[0], F.shape[1], F.shape[2], F.shape[3]
    top = torch.empty((n, C, 8, 64), device="cuda")
    nbytes = lib.rroi_align_forward_workspace_bytes(1, C, H, W, n, 0)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    trace = torch.zeros(8 * 4096, dtype=torch.int32, device="cuda")

    def call(stages=3):
        st = lib.rroi_align_forward_stages_hip(F.data_ptr(), 0, 0.25, 1, n, H, W, C, 8, 64, R.data_ptr(), top.data_ptr(),
                                               ws.data_ptr(), nbytes, 2, stages, stream)
        assert st == 1, st
    out = {}
    for what, stages in (("in_step", 3), ("gather_alone", 2)):
        for _ in range(300):
            call(stages)
        torch.cuda.synchronize()
        lib.rroi_align_debug_set_wg_trace(trace.data_ptr())
        for _ in range(5):
            call(stages)          # the LAST launch's stamps stay in the buffer
        torch.cuda.synchronize()
        lib.rroi_align_debug_set_wg_trace(None)
        raw = trace.cpu().numpy().astype(np.int64).reshape(-1, 8)[:3072]
        t = (raw[:, :4] - raw[:, 0].min()) / 100.0      # us
        t5 = (raw[:, 6] - raw[:, 0].min()) / 100.0
        hw, xcc, items = raw[:, 4] & 0xffffffff, raw[:, 5] & 0xf, raw[:, 7]
        # gfx9 HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13 (CDNA: se bits wider), ...
        cu_key = (xcc << 16) | ((hw >> 8) & 0xff) | (((hw >> 13) & 0x7) << 8)
        np.save(os.path.join(ROOT, "gpurun_out", "r04", "wg_trace_%s.npy" % what), raw)
        q = lambda a: [round(float(np.percentile(a, p)), 2) for p in (0, 10, 50, 90, 100)]
        out[what] = {"storer_entry_us[min,p10,p50,p90,max]": q(t[:, 0]), "first_loads_us": q(t[:, 1]),
                     "first_store_us": q(t[:, 2]), "exit_us": q(t[:, 3]),
                     "entry_to_first_store_us": q(t[:, 2] - t[:, 0]),
                     "busy_us(exit-entry)": q(t[:, 3] - t[:, 0]),
                     "exit_by_xcc_median": [round(float(np.median(t[xcc == x, 3])), 2) for x in range(8)],
                     "five_items_drained_us": q(t5), "items": [int(items.min()), int(items.max())],
                     "distinct_cu_keys": int(len(set(cu_key.tolist()))),
                     "wgs_per_cu_key[min,max]": [int(np.bincount(np.unique(cu_key, return_inverse=True)[1]).min()),
                                                 int(np.bincount(np.unique(cu_key, return_inverse=True)[1]).max())],
                     "exit_spread_within_cu_us(median over CUs of max-min)": round(float(np.median(
                         [np.ptp(t[cu_key == c, 3]) for c in np.unique(cu_key)])), 2),
                     "cu_mean_exit_us": q(np.array([t[cu_key == c, 3].mean() for c in np.unique(cu_key)])),
                     "exit_by_items": {"11 items (slots < 256 of each chunk)": q(t[(np.arange(3072) // 8) < 256, 3]),
                                       "10 items": q(t[(np.arange(3072) // 8) >= 256, 3])}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
