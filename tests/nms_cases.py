"""Seeded detector outputs for the nms tests: score / RBOX / angle maps of rotated word boxes, the
shape a trained FOTS emits at 1/4 resolution (score inside a shrunk box, the four distances to the
box sides, the unit direction vector)."""
import numpy as np


def synth_maps(h, w, nwords, seed, noise=0.0):
    """-> segm (h, w), geo (h, w, 4) [top, bottom, left, right], angle (2, h, w) [sin, cos], fp32."""
    rng = np.random.default_rng(seed)
    segm = np.zeros((h, w), np.float32)
    geo = np.zeros((h, w, 4), np.float32)
    ang = np.zeros((2, h, w), np.float32)
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
        d = np.stack([v + bh / 2, bh / 2 - v, u + bw / 2, bw / 2 - u], -1)
        d = d + rng.normal(0, noise, d.shape)
        geo[inside] = np.maximum(d[inside], 0).astype(np.float32)
        ang[0][inside] = s
        ang[1][inside] = c
    return segm, geo, ang


# (name, h, w, words, seed, noise): small maps, and the 11 example images' size (1280 x 704 -> 176 x 320)
CASES = [("small%d" % i, (44, 80) if i % 2 else (64, 96), 3 + i % 5, i, 0.15 * (i % 3)) for i in range(6)]
CASES += [("img%d" % i, (176, 320), 6 + 2 * i, 100 + i, 0.1 * (i % 4)) for i in range(11)]


def hard_maps(h, w, nwords, rng, noise, overlap):
    """Maps that provoke the hard cases of the polygon clipper: heavy noise on the four
    distances AND on the direction vector (merged quads go non-convex or self-intersecting), words dropped
    on top of each other at other angles (`overlap` = probability), any angle, low scores."""
    segm = np.zeros((h, w), np.float32)
    geo = np.zeros((h, w, 4), np.float32)
    ang = np.zeros((2, h, w), np.float32)
    ang[1] = 1
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float32)
    centres = []
    for i in range(nwords):
        if centres and rng.random() < overlap:
            cx, cy = centres[int(rng.integers(len(centres)))]
            cx, cy = cx + rng.uniform(-4, 4), cy + rng.uniform(-2, 2)
        else:
            cx, cy = rng.uniform(6, w - 6), rng.uniform(4, h - 4)
        centres.append((cx, cy))
        bh = rng.uniform(2, 9)
        bw = bh * rng.uniform(1, 8)
        a = rng.uniform(-1.5, 1.5)
        c, s = np.cos(a), np.sin(a)
        u = (xs + 0.25 - cx) * c + (ys + 0.25 - cy) * s
        v = -(xs + 0.25 - cx) * s + (ys + 0.25 - cy) * c
        inside = (np.abs(u) < bw / 2 * 0.8) & (np.abs(v) < bh / 2 * 0.6)
        segm[inside] = rng.uniform(0.3, 0.99, inside.sum())
        d = np.stack([v + bh / 2, bh / 2 - v, u + bw / 2, bw / 2 - u], -1)
        d = d + rng.normal(0, noise, d.shape)
        geo[inside] = np.maximum(d[inside], 0).astype(np.float32)
        da = rng.normal(0, 0.3 * noise, inside.sum())
        ang[0][inside] = np.sin(a + da)
        ang[1][inside] = np.cos(a + da)
    return segm, geo, ang


def hard_case(i):
    """-> (segm, geo, ang, segm_thresh, iou1, iou2) of frozen hard case i"""
    rng = np.random.default_rng(7000 + i)
    h, w = int(rng.integers(16, 56)), int(rng.integers(24, 88))
    noise = float([0.5, 1.0, 2.0][i % 3])
    thr = float([0.3, 0.5, 0.7, 0.9][i % 4])
    iou1, iou2 = float([0.1, 0.3, 0.4, 0.6][(i // 2) % 4]), float([0.05, 0.2, 0.5][(i // 3) % 3])
    segm, geo, ang = hard_maps(h, w, int(rng.integers(2, 9)), rng, noise, [0.0, 0.5, 0.9][(i // 4) % 3])
    return segm, geo, ang, thr, iou1, iou2


NUM_HARD = 32


def quad_class(q):
    """0 convex, 1 concave (simple), 2 self-intersecting"""
    p = np.asarray(q, np.float64).reshape(4, 2)

    def cross(o, a, b):
        return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])

    def seg(a, b, c, d):
        return cross(a, b, c) * cross(a, b, d) < 0 and cross(c, d, a) * cross(c, d, b) < 0
    if seg(p[0], p[1], p[2], p[3]) or seg(p[1], p[2], p[3], p[0]):
        return 2
    s = [cross(p[i], p[(i + 1) % 4], p[(i + 2) % 4]) for i in range(4)]
    return 0 if all(v >= 0 for v in s) or all(v <= 0 for v in s) else 1
