#!/usr/bin/env python3
"""Product vs the REFERENCE'S OWN KERNELS, live on the GPU box, over millions of bins.

What it quantifies: the one library-dependent step of the arithmetic recipe.  The hipified
reference calls ocml's cos(float) / sin(float) (rroi_align_kernel.cu:73-74); the oracle and the
product evaluate (float)cos((double)angle) (csrc/rroi_device_common.h:77-78).  ocml's float
cosine is not correctly rounded, so over enough angles the two affines differ in the last place
for some ROIs, and where such a difference meets a rounding tie a bin's sample point moves.

    python tools/fuzz_ref.py [rois_per_round] [rounds] [seed]  ->  one JSON line
    RROI_FUZZ_TRIG=fp32     the product's opt-in recipe (trig=ext.TRIG_FP32 on every call) for the campaign
    RROI_FUZZ_POOLED=11x83  another pooled size

Per round: `rois_per_round` random ROIs (every angle in [-180, 180), centres on and off the
half-integer grid, C = 1 so a bin is one output element), pooled 8 x 64 on a 160 x 160 map, through
  * oracle/_ref/librroi_ref_hip_nofma.so   the reference kernels, source semantics
  * ext.forward (tiled and direct)          the product
and con_idx_x / con_idx_y of both.  Reports bins compared, bins whose sample point differs, ROIs whose
affine-dependent centres differ anywhere, output elements that differ.  Measurement / test tooling:
it may load oracle/_ref (the product never does)."""
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fots.pytorch_amd"), os.path.join(ROOT, "tests")]
from rroi_align._ext import rroi_align as ext  # noqa: E402


def load_ref(name="librroi_ref_hip_nofma.so"):
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", name))
    vp, fl, it = ctypes.c_void_p, ctypes.c_float, ctypes.c_int
    lib.RROIAlignForwardLaucher.argtypes = [vp, fl, it, it, it, it, it, it, vp, vp, vp, vp, vp]
    return lib


def random_rois(rng, n, img=640):
    cx, cy = rng.uniform(-20, img + 20, n), rng.uniform(-20, img + 20, n)
    h = rng.uniform(8, 80, n)
    w = h * rng.uniform(0.5, 10, n)
    ang = rng.uniform(-180, 180, n)
    kind = rng.integers(0, 8, n)
    # a quarter of the ROIs sit where rounding ties live: centres on the feature grid, unit-step
    # affines (h = 32, w = 256 at scale .25 and 8 x 64), multiples of 15 degrees
    cx[kind == 0] = np.round(cx[kind == 0] / 4) * 4
    cy[kind == 0] = np.round(cy[kind == 0] / 4) * 4
    h[kind == 1], w[kind == 1] = 32, 256
    cx[kind == 1] = np.round(cx[kind == 1] / 2) * 2
    cy[kind == 1] = np.round(cy[kind == 1] / 2) * 2
    ang[kind <= 1] = rng.integers(-12, 12, int((kind <= 1).sum())) * 15.0
    return np.stack([np.zeros(n), cx, cy, h, w, ang], 1).astype(np.float32)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 11)
    ref = load_ref()
    trig = os.environ.get("RROI_FUZZ_TRIG", "double")
    T = {"double": ext.TRIG_DOUBLE, "fp32": ext.TRIG_FP32}[trig]   # per call (round 5)
    ph, pw, s, H, W = 8, 64, 0.25, 160, 160
    if os.environ.get("RROI_FUZZ_POOLED"):   # e.g. 11x83: rows that are not whole sectors (the SHIFT kernels)
        ph, pw = (int(v) for v in os.environ["RROI_FUZZ_POOLED"].split("x"))
    F = torch.from_numpy(rng.standard_normal((1, 1, H, W), dtype=np.float32)).cuda()
    tot = dict(rois=0, bins=0, bins_centre_differs=0, rois_centre_differs=0, out_differs_tiled=0,
               out_differs_direct=0, out_differs_where_centres_agree=0, max_centre_shift=0.0)
    worst = []
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(rounds):
        r = random_rois(rng, n)
        R = torch.from_numpy(r).cuda()
        want, ix, iy = (torch.zeros((n, 1, ph, pw), device="cuda") for _ in range(3))
        ref.RROIAlignForwardLaucher(F.data_ptr(), s, n, H, W, 1, ph, pw, R.data_ptr(), want.data_ptr(),
                                    ix.data_ptr(), iy.data_ptr(), stream)
        geom = ext.bin_centres(R, ph, pw, s, H, W, trig=T)
        dxy = (geom[..., 0] != ix[:, 0]) | (geom[..., 1] != iy[:, 0])
        shift = torch.maximum((geom[..., 0] - ix[:, 0]).abs(), (geom[..., 1] - iy[:, 0]).abs())
        tot["rois"] += n
        tot["bins"] += n * ph * pw
        tot["bins_centre_differs"] += int(dxy.sum())
        per_roi = dxy.flatten(1).any(1)
        tot["rois_centre_differs"] += int(per_roi.sum())
        tot["max_centre_shift"] = max(tot["max_centre_shift"], float(shift.max()))
        for name, path in (("tiled", ext.PATH_TILED), ("direct", ext.PATH_DIRECT)):
            got = ext.forward(F, R, ph, pw, s, path=path, trig=T)
            d = ~((got == want) | (got.isnan() & want.isnan()))
            tot["out_differs_" + name] += int(d.sum())
            if name == "tiled":
                tot["out_differs_where_centres_agree"] += int((d[:, 0] & ~dxy).sum())
        for i in torch.nonzero(per_roi).flatten()[:4].tolist():
            worst.append([float(v) for v in r[i]] + [int(dxy[i].sum())])
    tot["differing_bins_per_million"] = round(1e6 * tot["bins_centre_differs"] / max(1, tot["bins"]), 3)
    tot["pooled"] = [ph, pw]
    tot["trig_recipe"] = trig
    tot["example_rois"] = worst[:8]
    print(json.dumps(tot))


if __name__ == "__main__":
    main()
