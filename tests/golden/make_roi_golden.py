#!/usr/bin/env python3
"""Pins the callers' ROI construction to the reference's OWN statements.

Runs in the authoring container only (needs /root/reference).  The two places that turn quads
into RoIRotate rows are taken out of their files by line range and executed as they stand, on
seeded quads; only the resulting ARRAYS are stored (`roi_build.npz`), no source text:

  mode 0  `tools/ocr_utils.py:133-150` -- the body of `align_ocr` from `boxr = ...` to the
          `target_gw` rule, one detected box at a time (fp32 quad as `nms.get_boxes` returns it);
  mode 1  `src/ocr_process.py:197-206` -- the ground-truth branch of `process_boxes`, vectorised
          over the quads of an image (float64, `tools/data_gen.py:101`), with `random.randint`
          (the +-2 height jitter of :204) replaced by a constant that is stored alongside, and the
          training rule for the pooled width, `:259-263`, on the fp32 roi tensor.
"""
import math
import os
import random
import sys
import textwrap

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(HERE, ".."), os.path.join(HERE, "..", "..")]
from test_roi_build import random_quads  # noqa: E402  (the seeded quads the tests use)

REF = "/root/reference"


def lines(path, first, last):
    with open(os.path.join(REF, path), encoding="utf-8") as f:
        src = f.readlines()[first - 1:last]
    return textwrap.dedent("".join(src))


ALIGN_OCR = compile(lines("tools/ocr_utils.py", 133, 150), "ocr_utils.py:133-150", "exec")
GT_BRANCH = compile(lines("src/ocr_process.py", 197, 206), "ocr_process.py:197-206", "exec")
# (the container has no GPU: the one `.cuda()` of :259 is dropped, the arithmetic is untouched)
WIDTH_RULE = compile(lines("src/ocr_process.py", 259, 263).replace(".cuda()", ""), "ocr_process.py:259-263", "exec")


def mode0(quads):
    rois, gws = [], []
    for q in quads:
        env = {"math": math, "np": np, "boxo": np.concatenate([q, [0.9]]).astype(np.float32)}
        exec(ALIGN_OCR, env)
        rois.append(torch.tensor(env["rroi"]).to(torch.float).numpy())  # ocr_utils.py:152
        gws.append(env["target_gw"])
    return np.stack(rois), np.asarray(gws, np.int32)


def mode1(quads, jitter):
    env = {"math": math, "np": np, "random": random, "gts": [q.reshape(4, 2).astype(np.float64) for q in quads]}
    real = random.randint
    random.randint = lambda a, b: jitter
    try:
        exec(GT_BRANCH, env)
    finally:
        random.randint = real
    n = len(quads)
    rrois = [[0, env["center"][i][0], env["center"][i][1], env["h"][i], env["w"][i], env["angle_gt"][i]]
             for i in range(n)]                                           # ocr_process.py:219
    env2 = {"math": math, "torch": torch, "rrois": rrois}
    exec(WIDTH_RULE, env2)
    return env2["rois"].numpy(), int(env2["pooled_width"])


if __name__ == "__main__":
    out = {}
    for tag, integer in (("int", True), ("real", False)):
        q = random_quads(400, seed=41 + integer, integer=integer)
        out["quads_" + tag] = q
        out["m0_rois_" + tag], out["m0_gw_" + tag] = mode0(q)
        for j in (0, -2, 2):
            r, pwid = mode1(q[:64], j)
            out["m1_rois_%s_j%d" % (tag, j)] = r
            out["m1_pw_%s_j%d" % (tag, j)] = np.int32(pwid)
    np.savez_compressed(os.path.join(HERE, "roi_build.npz"), **out)
    print({k: getattr(v, "shape", v) for k, v in out.items()})
