#!/usr/bin/env python3
"""One-off extended fuzz on the GPU box (not part of the test suite): N random problems, every
path against the oracle -- forward bit-exact, backward within 1e-4 of the gradient scale.
    python tools/fuzz_gpu.py [trials] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fots.pytorch_amd"), os.path.join(ROOT, "tests")]
import workloads as Wk  # noqa: E402
from oracle import rroi_align_oracle as O  # noqa: E402  (tools/ is measurement/test tooling)
from rroi_align._ext import rroi_align as ext  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
bad = 0
LARGE = bool(os.environ.get("FUZZ_LARGE"))   # FUZZ_LARGE=1: maps up to 330 x 420, up to 260 ROIs
for t in range(trials):
    C = int(rng.choice([1, 2, 3, 4, 7, 8, 31, 32, 33, 40, 64, 65, 128, 257, 300]))
    H, W = int(rng.integers(2, 70)), int(rng.integers(2, 100))
    if LARGE:   # maps of many key tiles, several scan blocks, long pixel-major copies
        H, W = int(rng.integers(60, 330)), int(rng.integers(60, 420))
    B = int(rng.integers(1, 5))
    ph = int(rng.choice([1, 2, 3, 7, 8, 11, 16]))
    pw = int(rng.integers(1, 100))
    s = float(rng.choice([1.0, 0.5, 0.25, 0.125, 0.3]))
    R = int(rng.integers(1, 60))
    if LARGE:
        R = int(rng.integers(1, 260))
        C = int(rng.choice([8, 32, 40, 64, 96, 128, 160, 256, 300]))
    f, r = Wk.bench_inputs(R=R, C=C, H=H, W=W, img=max(4, int(W / s)), seed=5000 + t, batch=B)
    r[:, 2] = rng.uniform(-5, H / s + 5, R)
    r[:, 1] = rng.uniform(-5, W / s + 5, R)
    r[:, 3] = rng.uniform(0.5, 60, R) / (s * 4)
    r[:, 4] = r[:, 3] * rng.uniform(0.1, 20, R)
    kind = rng.integers(0, 12, R)
    r[kind == 0, 5] = rng.choice([0.0, 90.0, -90.0, 180.0, 45.0], int((kind == 0).sum()))
    r[kind == 1, 0] = rng.choice([-1.0, float(B), float(B) + 3.0], int((kind == 1).sum()))  # bad batch index
    r[kind == 2, 3] = 0.0
    r[kind == 3, 4] = 0.0
    r[kind == 4, 1:3] = np.round(r[kind == 4, 1:3] * s) / s  # integer feature-space centres: rounding ties
    if rng.random() < 0.1:
        r[int(rng.integers(0, R)), int(rng.integers(1, 6))] = rng.choice([np.nan, np.inf, -np.inf, 1e30])
    if os.environ.get("FUZZ_VERBOSE"):
        print(f"trial {t}: C={C} {H}x{W} B={B} {ph}x{pw} s={s} R={R}", flush=True)
    # an out-of-range batch index is undefined behaviour in the reference (and in the oracle, which
    # follows it); the library defines zeros / no gradient for such ROIs
    bi = r[:, 0]
    badb = ~((bi > -1) & (bi < B))   # (int) truncation: -0.5 -> 0 is valid
    r_o = r.copy()
    r_o[badb, 0] = 0
    want = O.forward_c(f, r_o, ph, pw, s, threads=16)
    want[badb] = 0
    F, Rr = torch.from_numpy(f).cuda(), torch.from_numpy(r).cuda()
    for p in (ext.PATH_DIRECT, ext.PATH_TILED, ext.PATH_FUSED, ext.PATH_AUTO):   # (round 5: + the one-launch fused form, AUTO)
        got = ext.forward(F, Rr, ph, pw, s, path=p).cpu().numpy()
        nb = int((~((got == want) | (np.isnan(got) & np.isnan(want)))).sum())
        if nb:
            bad += 1
            print(f"FWD MISMATCH trial {t} path {p}: C={C} {H}x{W} B={B} {ph}x{pw} s={s} R={R}: {nb} differ")
    if C % 4 == 0:
        got = ext.forward(F.contiguous(memory_format=torch.channels_last), Rr, ph, pw, s).cpu().numpy()
        if int((~((got == want) | (np.isnan(got) & np.isnan(want)))).sum()):
            bad += 1
            print(f"FWD MISMATCH trial {t} channels_last")
    gout = np.random.default_rng(t).standard_normal(want.shape).astype(np.float32)
    gout_o = gout.copy()
    gout_o[badb] = 0
    gw = O.backward_c(gout_o, r_o, f.shape, s)
    sc = max(1.0, float(np.abs(gw).max()))
    G = torch.from_numpy(gout).cuda()
    for p in (ext.PATH_DIRECT, ext.PATH_TILED, ext.PATH_TILED_LISTS, ext.PATH_TILED_BUCKETS, ext.PATH_TILED_INKERNEL, ext.PATH_TILED_ATOMIC):
        g = ext.backward(G, Rr, f.shape, s, path=p).cpu().numpy()
        e = float(np.abs(g - gw).max())
        if not e <= 1e-4 * sc:
            bad += 1
            print(f"BWD MISMATCH trial {t} path {p}: C={C} {H}x{W} B={B} {ph}x{pw} s={s} R={R}: err {e} scale {sc}")
    if C % 4 == 0:
        got = ext.forward(F, Rr, ph, pw, s, channels_last_out=True).cpu().numpy()
        if int((~((got == want) | (np.isnan(got) & np.isnan(want)))).sum()):
            bad += 1
            print(f"FWD MISMATCH trial {t} channels_last_out")
        Gcl = G.contiguous(memory_format=torch.channels_last)
        g = ext.backward(G, Rr, f.shape, s, channels_last_grad=True).cpu().numpy()
        if not float(np.abs(g - gw).max()) <= 1e-4 * sc:
            bad += 1
            print(f"BWD MISMATCH trial {t} channels_last_grad")
        g = ext.backward(Gcl, Rr, f.shape, s, channels_last_grad=bool(t & 1)).cpu().numpy()
        e = float(np.abs(g - gw).max())
        if not e <= 1e-4 * sc:
            bad += 1
            print(f"BWD MISMATCH trial {t} channels-last grad: C={C} {H}x{W} B={B} {ph}x{pw} s={s} R={R}: err {e} scale {sc}")
print(f"fuzz: {trials} trials, {bad} mismatches")
