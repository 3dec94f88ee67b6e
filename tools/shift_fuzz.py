#!/usr/bin/env python3
"""The split kernel's SHIFT forms (sector-aligned store windows; round 5: + the LINE-aligned windows, SHIFT == 2) against its strided form, forced
through the explore build (tools/build_explore.sh) on random shapes: pooled sizes of every residue mod 16, channel
counts that are not multiples of 32, several images, degenerate and non-finite ROIs, crops that start 4 / 8 / 12
bytes off a 16-byte boundary, every cut of the blocks into runs.  Bit for bit, the NaN payloads included.
    python tools/shift_fuzz.py [trials] [seed]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import workloads as Wk  # noqa: E402

lib = ctypes.CDLL(os.environ.get("RROI_EXPLORE_LIB") or os.path.join(ROOT, "tools", "_explore", "librroi_align_hip_explore.so"))
vp, fl, it, sz = ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_size_t
lib.rroi_align_forward_hip.argtypes = [vp, it, fl, it, it, it, it, it, it, it, vp, vp, vp, sz, it, vp]
lib.rroi_align_forward_workspace_bytes.restype = sz
lib.rroi_align_forward_workspace_bytes.argtypes = [it] * 6
st = torch.cuda.current_stream().cuda_stream
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
bad = 0
for tr in range(trials):
    C = int(rng.choice([1, 3, 4, 8, 31, 32, 33, 40, 64, 65, 96, 130, 256]))
    H, W = int(rng.integers(4, 80)), int(rng.integers(4, 120))
    B = int(rng.integers(1, 4))
    ph = int(rng.choice([1, 2, 3, 7, 8, 11, 16]))
    pw = int(rng.integers(1, 140))
    s = float(rng.choice([1.0, 0.5, 0.25]))
    R = int(rng.integers(1, 200))
    f, r = Wk.bench_inputs(R=R, C=C, H=H, W=W, img=max(4, int(W / s)), seed=9000 + tr, batch=B)
    r[:, 2] = rng.uniform(-5, H / s + 5, R)
    r[:, 3] = rng.uniform(0.5, 60, R) / (s * 4)
    r[:, 4] = r[:, 3] * rng.uniform(0.1, 20, R)
    kind = rng.integers(0, 12, R)
    r[kind == 1, 0] = rng.choice([-1.0, float(B), float(B) + 3.0], int((kind == 1).sum()))
    r[kind == 2, 3] = 0.0
    if rng.random() < 0.1:
        r[int(rng.integers(0, R)), int(rng.integers(1, 6))] = rng.choice([np.nan, np.inf, -np.inf, 1e30])
    if rng.random() < 0.1:
        f[0, 0, int(rng.integers(0, H)), int(rng.integers(0, W))] = np.nan
    F, Rt = torch.from_numpy(f).cuda(), torch.from_numpy(r).cuda()
    n_out = R * C * ph * pw
    skew = int(rng.integers(0, 4)) if rng.random() < 0.5 else int(rng.integers(0, 32))   # crops that start anywhere in a 128-byte line
    bufs = [torch.full((n_out + 40,), float("nan"), device="cuda") for _ in range(3)]
    outs = [b[skew:skew + n_out] for b in bufs]
    nb = lib.rroi_align_forward_workspace_bytes(B, C, H, W, R, 0)
    ws = torch.empty(max(nb, 1), dtype=torch.uint8, device="cuda")
    parts = int(rng.integers(0, 6))
    wgs = int(rng.choice([0, 0, 2, 4]))
    for mode, lines, o in ((0, 0, outs[0]), (2, 0, outs[1]), (0, 2, outs[2])):
        lib.rroi_align_debug_set_fwd_shift(mode, wgs, parts)
        lib.rroi_align_debug_set_fwd_shift_lines(lines, 0)
        rc = lib.rroi_align_forward_hip(F.data_ptr(), 0, s, B, R, H, W, C, ph, pw, Rt.data_ptr(), o.data_ptr(), ws.data_ptr(), nb, 2, st)
        assert rc == 1, rc
    torch.cuda.synchronize()
    a, b, c = (x.cpu().numpy().view(np.uint32) for x in bufs)
    if not np.array_equal(a, c):
        bad += 1
        d = np.nonzero(a != c)[0]
        print(f"trial {tr} LINES: C={C} {H}x{W} B={B} {ph}x{pw} R={R} skew={skew}: {d.size} words differ, first at {d[:5]}")
    if not np.array_equal(a, b):   # the guard floats around the crops included
        bad += 1
        d = np.nonzero(a != b)[0]
        print(f"trial {tr}: C={C} {H}x{W} B={B} {ph}x{pw} R={R} skew={skew} parts={parts} wgs={wgs}: {d.size} words differ, first at {d[:5]}")
lib.rroi_align_debug_set_fwd_shift(1, 0, 0)
lib.rroi_align_debug_set_fwd_shift_lines(1, 0)
print(f"shift fuzz: {trials} trials, {bad} mismatches")
