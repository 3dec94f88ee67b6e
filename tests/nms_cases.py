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
