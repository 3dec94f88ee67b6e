"""Benchmark / test INPUT GENERATORS for the end-to-end path (BASELINE configs[4]) -- not product code (moved out of
fots.pytorch_amd/fots_e2e/pipeline.py in round 5).  Used by bench_e2e.py, tests/test_e2e_*.py and
tools/e2e_*.py.  With random detection weights the score map passes no box (or a hundred thousand) through the NMS,
so seeded word-shaped boxes / trained-detector-shaped maps stand in for the detector's output."""
import math

import numpy as np


def synthetic_boxes(n, height, width, seed=0):
    """(n, 9) fp32 [x0,y0,x1,y1,x2,y2,x3,y3,score]: word-shaped rotated rectangles inside a
    height x width image, corner order as `nms.get_boxes` hands them to `align_ocr` (edge 0->1 is
    the short side, 1->2 the long one).  Stands in for the detector's output while its heads carry
    random weights (a random score map passes no box, or a hundred thousand, through the NMS)."""
    rng = np.random.default_rng(seed)
    out = np.zeros((n, 9), np.float32)
    for i in range(n):
        h = rng.uniform(14, 48)
        w = h * rng.uniform(1.5, 9.0)
        a = rng.uniform(-25, 25) / 180 * math.pi
        m = 0.5 * (w + h)
        cx, cy = rng.uniform(m, max(m + 1, width - m)), rng.uniform(m, max(m + 1, height - m))
        ux, uy, vx, vy = math.cos(a), math.sin(a), -math.sin(a), math.cos(a)
        p1 = (cx - ux * w / 2 - vx * h / 2, cy - uy * w / 2 - vy * h / 2)
        p2 = (p1[0] + ux * w, p1[1] + uy * w)
        p0 = (p1[0] + vx * h, p1[1] + vy * h)
        p3 = (p2[0] + vx * h, p2[1] + vy * h)
        out[i] = (*p0, *p1, *p2, *p3, rng.uniform(0.5, 1.0))
    return out


def synthetic_detector_maps(height, width, nwords, seed=0):
    """score (h, w), rbox (4, h, w), angle (2, h, w) fp32 numpy at 1/4 of a height x width image: what a
    TRAINED detector emits for `nwords` rotated words (score inside the shrunk box, distances to the
    four sides, unit direction) -- input for timing `rroi_align.nms.get_boxes`, which random
    detection weights cannot exercise."""
    h, w = height // 4, width // 4
    rng = np.random.default_rng(seed)
    segm, geo, ang = np.zeros((h, w), np.float32), np.zeros((4, h, w), np.float32), np.zeros((2, h, w), np.float32)
    ang[1] = 1
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float32)
    for _ in range(nwords):
        cx, cy = rng.uniform(10, w - 10), rng.uniform(6, h - 6)
        bh = rng.uniform(3, 8)
        bw = bh * rng.uniform(2, 7)
        a = rng.uniform(-0.5, 0.5)
        c, s = np.cos(a), np.sin(a)
        u = (xs + 0.25 - cx) * c + (ys + 0.25 - cy) * s
        v = -(xs + 0.25 - cx) * s + (ys + 0.25 - cy) * c
        inside = (np.abs(u) < bw / 2 * 0.8) & (np.abs(v) < bh / 2 * 0.6)
        segm[inside] = rng.uniform(0.6, 0.99, inside.sum())
        for k, d in enumerate((v + bh / 2, bh / 2 - v, u + bw / 2, bw / 2 - u)):
            geo[k][inside] = np.maximum(d[inside], 0)
        ang[0][inside], ang[1][inside] = s, c
    return segm, geo, ang
