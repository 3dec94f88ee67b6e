#!/usr/bin/env python3
"""Detection post-processing: the product's host merge (`rroi_nms_merge_host`: locality-aware merge +
polygon NMS, with its own polygon clipper) against the REFERENCE'S OWN build of nms/ (adaptor.cpp +
nms.h + the vendored Clipper, oracle/_ref/nms_ref/adaptor.so) on maps made to provoke the hard cases:
heavy geometry noise (merged quads go non-convex or self-intersecting), words that overlap at
different angles, thresholds 0.3 ... 0.9.  CPU only.
    python tools/fuzz_nms.py [trials] [seed]  ->  one JSON line
Counts the maps whose boxes differ in any bit, and how many of the boxes the reference returned are
non-convex / self-intersecting (evidence that the hard cases were reached)."""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fots.pytorch_amd"), os.path.join(ROOT, "tests"),
                os.path.join(ROOT, "oracle", "_ref", "nms_ref")]
from oracle import nms_oracle as NO  # noqa: E402  (tools/ is test tooling)


from nms_cases import hard_maps, quad_class as classify  # noqa: E402


def main():
    from rroi_align.nms import CANDIDATE, merge
    adaptor = importlib.import_module("adaptor")
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
    tot = dict(maps=0, candidates=0, boxes=0, maps_differ=0, boxes_concave=0, boxes_self_intersecting=0,
               first_mismatches=[])
    for t in range(trials):
        h, w = int(rng.integers(16, 72)), int(rng.integers(24, 120))
        noise = float(rng.choice([0.0, 0.2, 0.5, 1.0, 2.0]))
        thr = float(rng.choice([0.3, 0.5, 0.7, 0.9]))
        iou1, iou2 = float(rng.choice([0.1, 0.3, 0.4, 0.6])), float(rng.choice([0.05, 0.2, 0.5]))
        segm, geo, ang = hard_maps(h, w, int(rng.integers(1, 10)), rng, noise, float(rng.choice([0.0, 0.5, 0.9])))
        a_hw2 = np.ascontiguousarray(ang.swapaxes(0, 1).swapaxes(1, 2))
        want = np.array(adaptor.do_nms(segm, geo, a_hw2, np.full((h, w), -1, np.int32), iou1, iou2, thr),
                        dtype="float32").reshape(-1, 9)
        if len(want):
            want[:, :8] /= 10000
        polys = NO.decode(segm, geo, a_hw2, thr)
        rec = np.zeros(len(polys), CANDIDATE)
        for i, p in enumerate(polys):
            rec[i]["quad"] = np.asarray(p["poly"], np.int64).reshape(8)
            rec[i]["score"], rec[i]["rdist"], rec[i]["x"], rec[i]["y"] = p["score"], p["rdist"], p["x"], p["y"]
        got = merge(rec, w, h, iou1, iou2)
        tot["maps"] += 1
        tot["candidates"] += len(polys)
        tot["boxes"] += len(want)
        for b in want:
            k = classify(b[:8])
            tot["boxes_concave"] += k == 1
            tot["boxes_self_intersecting"] += k == 2
        if got.shape != want.shape or not np.array_equal(got, want):
            tot["maps_differ"] += 1
            if len(tot["first_mismatches"]) < 5:
                tot["first_mismatches"].append(dict(trial=t, h=h, w=w, noise=noise, thr=thr, iou=[iou1, iou2],
                                                    boxes=[int(len(got)), int(len(want))]))
    print(json.dumps({k: (int(v) if isinstance(v, (np.integer,)) else v) for k, v in tot.items()}))


if __name__ == "__main__":
    main()
