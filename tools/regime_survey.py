#!/usr/bin/env python3
"""Forward call across regimes: prologue / gather / whole call (us) next to what the call's algorithmic bytes
(crops written + map read once) cost at 6.6 TB/s, the rate the cfg2 gather streams at."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fots.pytorch_amd"), os.path.join(ROOT, "tests")]
import workloads as Wk  # noqa: E402
from rroi_align._ext import rroi_align as ext  # noqa: E402

st = torch.cuda.current_stream().cuda_stream


def timed(fn, warm=100, n=300):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def case(label, R, C, H, W, ph, pw, B=1, scale=0.25, presort=None):
    f, r = Wk.bench_inputs(R=R, C=C, H=H, W=W, img=int(W / scale), seed=1, batch=B)
    if presort == "image":
        r = r[np.argsort(r[:, 0], kind="stable")]
    elif presort == "image+cy":
        r = r[np.lexsort((r[:, 2], r[:, 0]))]
    F, Rt = torch.from_numpy(f).cuda(), torch.from_numpy(r).cuda()
    out = torch.empty((R, C, ph, pw), device="cuda")
    nb = ext._lib.rroi_align_forward_workspace_bytes(B, C, H, W, R, ext.LAYOUT_NCHW)
    ws = torch.empty(max(nb, 1), dtype=torch.uint8, device="cuda")

    def go(stage, path=ext.PATH_TILED):
        assert ext._lib.rroi_align_forward_stages_hip(F.data_ptr(), ext.LAYOUT_NCHW, scale, B, R, H, W, C, ph, pw, Rt.data_ptr(),
                                                      out.data_ptr(), ws.data_ptr(), nb, path, stage, st) == 1
    pro, gat, allt = timed(lambda: go(1)), timed(lambda: go(2)), timed(lambda: go(3))
    direct = timed(lambda: go(3, ext.PATH_DIRECT), 20, 50) if R * C * ph * pw < 40e6 else float("nan")
    out_b, map_b = R * C * ph * pw * 4, B * C * H * W * 4
    floor = (out_b + map_b) / 6.6e6
    print(f"{label:36s} out {out_b / 1e6:7.1f} MB map {map_b / 1e6:6.1f} MB | prologue {pro:6.1f} gather {gat:6.1f} call {allt:6.1f} "
          f"direct {direct:6.1f} | bytes at 6.6 TB/s {floor:6.1f} us -> call = {allt / floor:4.2f} x")


if os.environ.get("RROI_SURVEY") == "align":
    case("512x64 160x160 11x100 (4400 B rows)", 512, 64, 160, 160, 11, 100)
    case("512x64 160x160 11x96  (4224 B rows)", 512, 64, 160, 160, 11, 96)
    case("512x64 160x160 11x104 (4576 B rows)", 512, 64, 160, 160, 11, 104)
    case("512x64 160x160 11x128 (5632 B rows)", 512, 64, 160, 160, 11, 128)
    case("512x64 160x160 8x100  (3200 B rows)", 512, 64, 160, 160, 8, 100)
    case("512x64 160x160 8x98   (3136 B rows)", 512, 64, 160, 160, 8, 98)
    case("512x256 160x160 8x64  (2048 B rows)", 512, 256, 160, 160, 8, 64)
    case("512x256 160x160 8x62  (1984 B rows)", 512, 256, 160, 160, 8, 62)
    case("512x256 160x160 8x60  (1920 B rows)", 512, 256, 160, 160, 8, 60)
    case("512x256 160x160 7x64  (1792 B rows)", 512, 256, 160, 160, 7, 64)
    sys.exit(0)
case("cfg2 512x256 160x160 8x64", 512, 256, 160, 160, 8, 64)
case("128x256 160x160 8x64", 128, 256, 160, 160, 8, 64)
case("512x64 176x320 11x96", 512, 64, 176, 320, 11, 96)
case("128x64 176x320 11x96", 128, 64, 176, 320, 11, 96)
case("512x64 8 maps 160x160 11x100", 512, 64, 160, 160, 11, 100, B=8)
case("128x64 8 maps 160x160 11x100", 128, 64, 160, 160, 11, 100, B=8)
case("64x64 8 maps 160x160 11x100", 64, 64, 160, 160, 11, 100, B=8)
case("512x64 160x160 8x64", 512, 64, 160, 160, 8, 64)
case("512x64 160x160 8x100", 512, 64, 160, 160, 8, 100)
case("512x64 160x160 11x64", 512, 64, 160, 160, 11, 64)
case("512x128 160x160 8x64", 512, 128, 160, 160, 8, 64)
case("2048x256 160x160 8x64", 2048, 256, 160, 160, 8, 64)
case("512x64 8 maps, ROIs sorted by image", 512, 64, 160, 160, 11, 100, B=8, presort="image")
case("512x64 8 maps, sorted by image, cy", 512, 64, 160, 160, 11, 100, B=8, presort="image+cy")
case("128x64 8 maps, sorted by image", 128, 64, 160, 160, 11, 100, B=8, presort="image")
case("512x256 8 maps 160x160 8x64", 512, 256, 160, 160, 8, 64, B=8)
case("512x256 8 maps, sorted by image", 512, 256, 160, 160, 8, 64, B=8, presort="image")
case("512x64 176x320 sorted by cy", 512, 64, 176, 320, 11, 96, presort="image+cy")
case("cfg2 sorted by cy", 512, 256, 160, 160, 8, 64, presort="image+cy")
