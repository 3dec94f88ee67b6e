#!/usr/bin/env python3
"""A few launches of the C = 64 training-shape forward (two 120 x 160 maps, R = 512) for counter passes: 11 x 96 (whole-sector
rows, strided tiles), 11 x 83 and 11 x 100 (the merging form ships for these; their SHIFT form runs with RROI_MERGE=0 through the
exploration build): bash tools/run_pmc_small.sh "python tools/merge_pmc_cmd.py"."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_explore", "librroi_align_hip_explore.so"))
vp, fl, it, sz = ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_size_t
lib.rroi_align_forward_hip.argtypes = [vp, it, fl, it, it, it, it, it, it, it, vp, vp, vp, sz, it, vp]
lib.rroi_align_forward_workspace_bytes.restype = sz
lib.rroi_align_forward_workspace_bytes.argtypes = [it] * 6
lib.rroi_align_debug_set_fwd_merge(int(os.environ.get("RROI_MERGE", "1")))
st = torch.cuda.current_stream().cuda_stream
B, C, H, W, R = 2, 64, 120, 160, 512
for pw in (96, 83, 100):
    rng = np.random.default_rng(1000 + R + pw)
    F = torch.from_numpy(rng.standard_normal((B, C, H, W), dtype=np.float32)).cuda()
    h = rng.uniform(16, 64, R)
    Rt = torch.from_numpy(np.stack([rng.integers(0, B, R), rng.uniform(0, 4 * W, R), rng.uniform(0, 4 * H, R), h,
                                    h * rng.uniform(2, pw / 11.0, R), rng.uniform(-45, 45, R)], 1).astype(np.float32)).cuda()
    nb = lib.rroi_align_forward_workspace_bytes(B, C, H, W, R, 0)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    top = torch.empty((R, C, 11, pw), device="cuda")
    for _ in range(6):
        assert lib.rroi_align_forward_hip(F.data_ptr(), 0, 0.25, B, R, H, W, C, 11, pw, Rt.data_ptr(), top.data_ptr(), ws.data_ptr(), nb, 0, st) == 1
    torch.cuda.synchronize()
