"""oracle/nms_oracle.py -- CPU restatement of the reference's detection post-processing
(SURVEY.md 8(f) rank 4).  TEST INFRASTRUCTURE ONLY (see rroi_align_oracle.py): only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.  Citations relative to
/root/reference.

  get_boxes        nms/__init__.py:20-29 (angle map to (h, w, 2), thresholds 0.4 / 0.2)
  decode           nms/adaptor.cpp:76-117: every pixel with score > segm_thresh becomes a quad from
                   its 4 RBOX distances and unit direction vector, in fp32 exactly as written,
                   corners rounded (roundf) to 1/10000 px integers; 4 corner confidences
                   exp(-distance / 9) products (:92-98, :107)
  merge_iou        nms/nms.h:149-213, statement for statement -- including that a polygon which
                   merges with nothing is appended TWICE when the list is not empty (:198 and :201)
                   and that the (y-1, x+1) look-up has no bound on x (:184)
  PolyMerger       nms/nms.h:48-113: int64 accumulators updated through fp32 arithmetic
                   (`int64 += int64 * float` evaluates in float and truncates), corner = int64 / float
  standard_nms     nms/nms.h:116-146 (it MERGES the suppressed polygon into the kept one)
  poly_iou         nms/nms.h:24-36: |area(a & b)| / max(|area(a | b)|, 1) with both areas summed in
                   float.  The reference clips with Clipper 6.2.6 (vendored, integer coordinates);
                   here: Sutherland-Hodgman on the integer quads with the intersection points
                   rounded to integers as Clipper rounds them, union = A + B - I.  Equal to
                   Clipper's result to ~1e-7 relative for the convex quads this path produces;
                   pinned by tests/golden/nms_*.npz, which the reference's OWN nms/ (built from its
                   sources by oracle/Makefile: ref) produced.
"""
import ctypes
import ctypes.util
import math

import numpy as np

_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.expf.restype = ctypes.c_float
_libm.expf.argtypes = [ctypes.c_float]

F = np.float32


def _expf(v):
    """the C library's expf, as adaptor.cpp:95-98 calls it (numpy's own fp32 exp may differ in the last place)"""
    return F(_libm.expf(float(v)))
PRECISION = F(10000)
SCALE = F(4)


def _roundf(v):
    """roundf: half away from zero, on an fp32 value."""
    v = float(v)
    return int(math.floor(abs(v) + 0.5)) * (1 if v >= 0 else -1)


def decode(segm, geo, angle, segm_thresh):
    """adaptor.cpp:76-117 -> list of polygons {poly: [[X, Y] * 4] ints, score, probs[4], x, y}
    in raster order."""
    segm, geo, angle = (np.ascontiguousarray(a, np.float32) for a in (segm, geo, angle))
    h, w = segm.shape
    thr = F(segm_thresh)
    polys = []
    for y in range(h):
        for x in range(w):
            if not segm[y, x] > thr:
                continue
            r = geo[y, x]
            acos, asin = angle[y, x, 1], angle[y, x, 0]
            xp, yp = F(x) + F(0.25), F(y) + F(0.25)
            prx = (xp - r[2] * acos) * SCALE
            pry = (yp - r[2] * asin) * SCALE
            pr2x = (xp + r[3] * acos) * SCALE
            pr2y = (yp + r[3] * asin) * SCALE
            nine = F(9)
            p_left, p_top = _expf(-r[2] / nine), _expf(-r[0] / nine)
            p_right, p_bt = _expf(-r[3] / nine), _expf(-r[1] / nine)
            quad = [
                [_roundf(PRECISION * (prx - r[1] * asin * SCALE)), _roundf(PRECISION * (pry + r[1] * acos * SCALE))],
                [_roundf(PRECISION * (prx + r[0] * asin * SCALE)), _roundf(PRECISION * (pry - r[0] * acos * SCALE))],
                [_roundf(PRECISION * (pr2x + r[0] * asin * SCALE)), _roundf(PRECISION * (pr2y - r[0] * acos * SCALE))],
                [_roundf(PRECISION * (pr2x - r[1] * asin * SCALE)), _roundf(PRECISION * (pr2y + r[1] * acos * SCALE))],
            ]
            polys.append(dict(poly=quad, score=F(segm[y, x]),
                              probs=[F(p_left * p_bt), F(p_left * p_top), F(p_right * p_top), F(p_right * p_bt)],
                              rdist=[F(r[0]), F(r[1]), F(r[2]), F(r[3])],   # what the product's record carries
                              x=x, y=y))
    return polys


def _area2(p):
    """twice the signed shoelace area of integer / float vertices, in double"""
    s = 0.0
    n = len(p)
    for i in range(n):
        x0, y0 = p[i]
        x1, y1 = p[(i + 1) % n]
        s += float(x0) * float(y1) - float(x1) * float(y0)
    return s


def _clip(subject, clip):
    """Sutherland-Hodgman: subject polygon against a CONVEX clip polygon given counter-clockwise
    (positive _area2); new vertices are rounded to integers like Clipper's IntersectPoint."""
    out = [(float(x), float(y)) for x, y in subject]
    n = len(clip)
    for i in range(n):
        if not out:
            break
        ax, ay = clip[i]
        bx, by = clip[(i + 1) % n]
        ex, ey = float(bx) - float(ax), float(by) - float(ay)
        inp, out = out, []
        m = len(inp)
        for j in range(m):
            px, py = inp[j]
            qx, qy = inp[(j + 1) % m]
            sp = ex * (py - float(ay)) - ey * (px - float(ax))
            sq = ex * (qy - float(ay)) - ey * (qx - float(ax))
            if sp >= 0:
                out.append((px, py))
            if (sp > 0 and sq < 0) or (sp < 0 and sq > 0):
                t = sp / (sp - sq)
                ix, iy = px + t * (qx - px), py + t * (qy - py)
                out.append((float(math.floor(ix + 0.5)), float(math.floor(iy + 0.5))))
    return out


def _convex(p):
    """all four corner cross products of one strict sign: a convex, simple quad"""
    pos = neg = 0
    for i in range(4):
        (x0, y0), (x1, y1), (x2, y2) = p[i], p[(i + 1) % 4], p[(i + 2) % 4]
        c = (float(x1) - float(x0)) * (float(y2) - float(y1)) - (float(y1) - float(y0)) * (float(x2) - float(x1))
        pos += c > 0
        neg += c < 0
    return pos == 4 or neg == 4


def _inside_evenodd(p, px, py):
    inside = False
    for i in range(4):
        (xi, yi), (xj, yj) = p[i], p[(i + 1) % 4]
        xi, yi, xj, yj = float(xi), float(yi), float(xj), float(yj)
        if (yi > py) != (yj > py):
            if xi + (py - yi) / (yj - yi) * (xj - xi) > px:
                inside = not inside
    return inside


def _evenodd_areas(pa, pb):
    """(area of A and B, area of A or B) under the even-odd fill rule for quads of any shape -- what
    Clipper's Execute(ctIntersection / ctUnion, pftEvenOdd) + Area() return (nms.h:24-36).  Edges are
    cut at their crossings (points rounded to integers, as Clipper stores them); a piece of A bounds
    the intersection when it lies inside B, else the union, and symmetrically; Green's theorem with
    every piece oriented so that its own quad's interior is on its left.  pb = None: A alone."""
    polys = [pa] if pb is None else [pa, pb]
    edges = [(q, i) for q in range(len(polys)) for i in range(4)]
    cuts = {e: [] for e in edges}
    for ei, (q, i) in enumerate(edges):
        for (r, j) in edges[ei + 1:]:
            if q == r and ((i + 1) % 4 == j or (j + 1) % 4 == i):
                continue
            (x1, y1), (x2, y2) = polys[q][i], polys[q][(i + 1) % 4]
            (x3, y3), (x4, y4) = polys[r][j], polys[r][(j + 1) % 4]
            x1, y1, x2, y2, x3, y3, x4, y4 = (float(v) for v in (x1, y1, x2, y2, x3, y3, x4, y4))
            d = (x2 - x1) * (y4 - y3) - (y2 - y1) * (x4 - x3)
            if d == 0:
                continue
            t = ((x3 - x1) * (y4 - y3) - (y3 - y1) * (x4 - x3)) / d
            u = ((x3 - x1) * (y2 - y1) - (y3 - y1) * (x2 - x1)) / d
            if not (0 < t < 1 and 0 < u < 1):
                continue
            pt = (float(math.floor(x1 + t * (x2 - x1) + 0.5)), float(math.floor(y1 + t * (y2 - y1) + 0.5)))
            cuts[(q, i)].append((t, pt))
            cuts[(r, j)].append((u, pt))
    inter = uni = 0.0
    for (q, i) in edges:
        (x1, y1), (x2, y2) = polys[q][i], polys[q][(i + 1) % 4]
        x1, y1, x2, y2 = float(x1), float(y1), float(x2), float(y2)
        ln = math.sqrt((x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1))
        if ln == 0:
            continue
        nx, ny = -(y2 - y1) / ln, (x2 - x1) / ln
        t0, (sx, sy) = 0.0, (x1, y1)
        for t1, (ex, ey) in sorted(cuts[(q, i)], key=lambda c: c[0]) + [(1.0, (x2, y2))]:
            tm = 0.5 * (t0 + t1)
            mx, my = x1 + tm * (x2 - x1), y1 + tm * (y2 - y1)
            left = _inside_evenodd(polys[q], mx + 1e-3 * nx, my + 1e-3 * ny)
            right = _inside_evenodd(polys[q], mx - 1e-3 * nx, my - 1e-3 * ny)
            if left != right:
                cr = 0.5 * (sx * ey - ex * sy) * (1.0 if left else -1.0)
                if pb is not None and _inside_evenodd(polys[1 - q], mx, my):
                    inter += cr
                else:
                    uni += cr
            t0, (sx, sy) = t1, (ex, ey)
    return inter, uni


def poly_iou(a, b):
    """nms.h:24-36.  Two convex quads: Sutherland-Hodgman (crossing points rounded as Clipper rounds them);
    a quad that is not convex or not simple -- merged quads are per-coordinate weighted means with
    different weights for X and Y, nms.h:87-96 -- takes the even-odd arrangement."""
    pa, pb = a["poly"], b["poly"]
    if _convex(pa) and _convex(pb):
        a2, b2 = _area2(pa), _area2(pb)
        clip = pb if b2 >= 0 else pb[::-1]
        inter = abs(_area2(_clip(pa, clip))) / 2.0
        uni = abs(a2) / 2.0 + abs(b2) / 2.0 - inter
    elif all(tuple(u) == tuple(v) for u, v in zip(pa, pb)):
        _, uni = _evenodd_areas(pa, None)      # the same quad twice (nms.h:198/201)
        inter = uni
    else:
        inter, uni = _evenodd_areas(pa, pb)
    inter_f, uni_f = F(inter), F(uni)          # `float area` accumulators of paths_area (:17-22)
    return F(abs(inter_f) / max(abs(uni_f), F(1.0)))


def should_merge(a, b, thr):
    return poly_iou(a, b) > F(thr)


class PolyMerger:
    """nms.h:48-113.  data[] are int64, updated as `data += int64 * float`: the product and the sum
    are fp32, the result truncates back to int64."""

    def __init__(self):
        self.data = [0] * 8
        self.score = F(0)
        self.probs = [F(0)] * 4

    def add(self, p):
        q, pr = p["poly"], p["probs"]
        idx = [(0, 0, 0), (0, 1, 3), (1, 0, 0), (1, 1, 1), (2, 0, 2), (2, 1, 1), (3, 0, 2), (3, 1, 3)]
        for k, (v, c, pi) in enumerate(idx):
            self.data[k] = int(F(self.data[k]) + F(q[v][c]) * pr[pi])   # trunc toward zero
        self.score = F(self.score + p["score"])
        self.probs = [F(self.probs[i] + pr[i]) for i in range(4)]

    def get(self):
        d, pr = self.data, self.probs
        div = lambda a, b: int(F(a) / b)  # noqa: E731  int64 / float -> float -> cInt
        quad = [[div(d[0], pr[0]), div(d[1], pr[3])], [div(d[2], pr[0]), div(d[3], pr[1])],
                [div(d[4], pr[2]), div(d[5], pr[1])], [div(d[6], pr[2]), div(d[7], pr[3])]]
        return dict(poly=quad, score=self.score, probs=list(pr), x=0, y=0)


def _merged(first, second):
    m = PolyMerger()
    m.add(first)
    m.add(second)
    return m.get()


def standard_nms(polys, thr):
    """nms.h:116-146."""
    n = len(polys)
    if n == 0:
        return []
    # std::sort with a strict-weak "score greater" comparator: order of equal scores is
    # implementation-defined in the reference (introsort); a stable descending sort here
    indices = sorted(range(n), key=lambda i: -float(polys[i]["score"]))
    keep = []
    while indices:
        cur = indices[0]
        keep.append(cur)
        rest = []
        for i in indices[1:]:
            if not should_merge(polys[cur], polys[i], thr):
                rest.append(i)
            else:
                polys[cur] = _merged(polys[i], polys[cur])
        indices = rest
    return [polys[i] for i in keep]


def merge_iou(polys_in, w, h, thr1, thr2):
    """nms.h:149-213."""
    poly_map = [-1] * (w * h)
    polys = []
    for poly in polys_in:
        px, py = poly["x"], poly["y"]
        if polys:
            if should_merge(poly, polys[-1], thr1):
                polys[-1] = _merged(polys[-1], poly)
                poly_map[py * w + px] = len(polys) - 1
                continue
            done = False
            if py > 0:
                idx = poly_map[(py - 1) * w + px]
                if idx >= 0:
                    cands = [idx]
                    if px > 0:
                        cands.append(poly_map[(py - 1) * w + px - 1])
                    cands.append(poly_map[(py - 1) * w + px + 1])   # no bound on x + 1 (:184)
                    for k, c in enumerate(cands):
                        if k > 0 and c < 0:
                            continue
                        if should_merge(poly, polys[c], thr1):
                            polys[c] = _merged(polys[c], poly)
                            poly_map[py * w + px] = c
                            done = True
                            break
            if done:
                continue
            polys.append(poly)            # :198
        polys.append(poly)                # :201
        poly_map[py * w + px] = len(polys) - 1
    return standard_nms(polys, thr2)


def get_boxes(iou_map, rbox, angle_pred, segm_thresh=0.5):
    """nms/__init__.py:11-29: iou_map (h, w), rbox (h, w, 4), angle_pred (2, h, w) -> (n, 9) fp32."""
    angle = np.asarray(angle_pred).swapaxes(0, 1).swapaxes(1, 2)
    polys = decode(iou_map, rbox, angle, segm_thresh)
    h, w = np.asarray(iou_map).shape
    out = merge_iou(polys, w, h, 0.4, 0.2)
    ret = np.asarray([[F(v) for pt in p["poly"] for v in pt] + [p["score"]] for p in out], np.float32)
    if len(ret) > 0:
        ret[:, :8] /= 10000
    return ret.reshape(-1, 9)
