#!/usr/bin/env python3
"""Pins the detection post-processing to the reference's OWN nms/ (adaptor.cpp + nms.h + its vendored
Clipper), built from its sources by `make -C oracle ref` into oracle/_ref/nms_ref/adaptor.so
(authoring container only).  For every seeded case of tests/nms_cases.py: the boxes the
reference returns, exactly as nms/__init__.py:11-29 drives it (poly_map of -1, thresholds
0.4 / 0.2, / 10000), and a checksum of the generated inputs.  Arrays only."""
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle", "_ref", "nms_ref")]
import adaptor  # noqa: E402  the reference's extension module
from nms_cases import CASES, NUM_HARD, hard_case, quad_class, synth_maps  # noqa: E402

out = {}
for name, (h, w), words, seed, noise in CASES:
    segm, geo, ang = synth_maps(h, w, words, seed, noise)
    poly_map = np.full((h, w), -1, np.int32)
    angle_hw2 = np.ascontiguousarray(ang.swapaxes(0, 1).swapaxes(1, 2))          # nms/__init__.py:22-23
    ret = np.array(adaptor.do_nms(segm, geo, angle_hw2, poly_map, 0.4, 0.2, 0.5), dtype="float32")
    if len(ret) > 0:
        ret[:, :8] /= 10000
    out[name + "_boxes"] = ret.reshape(-1, 9)
    out[name + "_crc"] = np.int64(zlib.crc32(segm.tobytes() + geo.tobytes() + ang.tobytes()))
    out[name + "_pixels"] = np.int32((segm > 0.5).sum())
    print(name, (h, w), "pixels", int(out[name + "_pixels"]), "boxes", len(ret))
# round 3: maps that provoke the clipper's hard cases (merged quads that are not convex / not simple),
# other thresholds; what the reference's build (its vendored Clipper, even-odd fill) returns for them
shapes = [0, 0, 0]
for i in range(NUM_HARD):
    segm, geo, ang, thr, iou1, iou2 = hard_case(i)
    h, w = segm.shape
    angle_hw2 = np.ascontiguousarray(ang.swapaxes(0, 1).swapaxes(1, 2))
    ret = np.array(adaptor.do_nms(segm, geo, angle_hw2, np.full((h, w), -1, np.int32), iou1, iou2, thr), dtype="float32").reshape(-1, 9)
    if len(ret) > 0:
        ret[:, :8] /= 10000
    out["hard%d_boxes" % i] = ret
    out["hard%d_crc" % i] = np.int64(zlib.crc32(segm.tobytes() + geo.tobytes() + ang.tobytes()))
    for b in ret:
        shapes[quad_class(b[:8])] += 1
print("hard cases: boxes convex / concave / self-intersecting:", shapes)
np.savez_compressed(os.path.join(HERE, "nms_cases.npz"), **out)
