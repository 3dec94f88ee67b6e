// rroi_nms_host.h -- detection post-processing, host part: locality-aware merge + polygon NMS
// Included by rroi_align_hip.hip inside its anonymous namespace; plain host C++, no HIP.
#pragma once

// ------------------------------------------------------------------------------------
// nms/nms.h:48-213 of the reference, on the candidate records the device produced.  The merge is a
// sequential scan whose every step depends on the previous one (merge with the last polygon, else
// with the polygon a neighbouring pixel of the previous row went into), so it stays on the host --
// but it now starts from a compact list in raster order instead of three full-resolution maps.
// Arithmetic as the reference's: int64 accumulators updated through fp32 (`int64 += int64 * float`),
// corners = int64 / float truncated, float area sums.  The polygon intersection replaces the
// vendored Clipper by Sutherland-Hodgman on the integer quads with intersection points rounded
// to integers as Clipper rounds them (convex quads: the same area to ~1e-7 relative).
// ------------------------------------------------------------------------------------
struct NmsPoly {
    long long X[4], Y[4];
    float score;
    float probs[4];
    int x, y;
};

inline double nms_area2(const double* px, const double* py, int n)
{
    double s = 0.0;
    for (int i = 0; i < n; ++i) {
        const int j = (i + 1) % n;
        s += px[i] * py[j] - px[j] * py[i];
    }
    return s;
}

// A quad is convex (and simple) iff its four corner cross products have one strict sign.
inline bool nms_quad_convex(const double* x, const double* y)
{
    int pos = 0, neg = 0;
    for (int i = 0; i < 4; ++i) {
        const int j = (i + 1) % 4, k = (i + 2) % 4;
        const double c = (x[j] - x[i]) * (y[k] - y[j]) - (y[j] - y[i]) * (x[k] - x[j]);
        pos += c > 0;
        neg += c < 0;
    }
    return pos == 4 || neg == 4;
}

// even-odd point-in-quad (ray to +x, half-open in y)
inline bool nms_inside_evenodd(const double* x, const double* y, double px, double py)
{
    bool in = false;
    for (int i = 0; i < 4; ++i) {
        const int j = (i + 1) % 4;
        if ((y[i] > py) != (y[j] > py)) {
            const double xi = x[i] + (py - y[i]) / (y[j] - y[i]) * (x[j] - x[i]);
            if (xi > px) in = !in;
        }
    }
    return in;
}

// Areas of (A and B) and (A or B) for two quads of ANY shape under the even-odd fill rule -- what
// ClipperLib's Execute(ctIntersection / ctUnion, ..., pftEvenOdd) + Area() give (nms.h:24-36): merged
// quads are per-coordinate weighted means with different weights for X and Y (nms.h:87-96) and need not
// stay convex or simple.  Every edge is cut at its crossings with the other quad's edges and with its own
// quad's non-adjacent edges, the crossing points ROUNDED to integers as Clipper stores them; a piece of A
// bounds the intersection when it lies inside B (the union: outside B), and symmetrically; each piece is
// oriented so that its own quad's interior is on its left, and Green's theorem sums the area.
// (bx == nullptr: one quad alone; `uni` is its even-odd area.)
inline void nms_evenodd_areas(const double* ax, const double* ay, const double* bx, const double* by,
                              double& inter, double& uni)
{
    const double* X[2] = {ax, bx};
    const double* Y[2] = {ay, by};
    const int nq = bx ? 2 : 1;
    // crossing of edge (q, i) with edge (r, j): parameter on each edge and the shared rounded point
    struct Cut { double t, px, py; };
    Cut cuts[2][4][8];
    int ncut[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int e = 0; e < 4 * nq; ++e)
        for (int f = e + 1; f < 4 * nq; ++f) {
            const int q = e / 4, i = e % 4, r = f / 4, j = f % 4;
            if (q == r && ((i + 1) % 4 == j || (j + 1) % 4 == i)) continue;  // adjacent edges share a vertex
            const double x1 = X[q][i], y1 = Y[q][i], x2 = X[q][(i + 1) % 4], y2 = Y[q][(i + 1) % 4];
            const double x3 = X[r][j], y3 = Y[r][j], x4 = X[r][(j + 1) % 4], y4 = Y[r][(j + 1) % 4];
            const double d = (x2 - x1) * (y4 - y3) - (y2 - y1) * (x4 - x3);
            if (d == 0) continue;  // parallel
            const double t = ((x3 - x1) * (y4 - y3) - (y3 - y1) * (x4 - x3)) / d;
            const double u = ((x3 - x1) * (y2 - y1) - (y3 - y1) * (x2 - x1)) / d;
            if (!(t > 0 && t < 1 && u > 0 && u < 1)) continue;
            const double px = floor(x1 + t * (x2 - x1) + 0.5), py = floor(y1 + t * (y2 - y1) + 0.5);
            cuts[q][i][ncut[q][i]++] = Cut{t, px, py};
            cuts[r][j][ncut[r][j]++] = Cut{u, px, py};
        }
    inter = uni = 0.0;
    for (int q = 0; q < nq; ++q) {
        const double *ox = X[1 - q], *oy = Y[1 - q];
        for (int i = 0; i < 4; ++i) {
            const double x1 = X[q][i], y1 = Y[q][i], x2 = X[q][(i + 1) % 4], y2 = Y[q][(i + 1) % 4];
            Cut* c = cuts[q][i];
            const int n = ncut[q][i];
            std::sort(c, c + n, [](const Cut& a, const Cut& b) { return a.t < b.t; });
            const double len = sqrt((x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1));
            if (len == 0) continue;
            const double nx = -(y2 - y1) / len, ny = (x2 - x1) / len;  // left normal
            double t0 = 0.0, sx = x1, sy = y1;
            for (int k = 0; k <= n; ++k) {
                const double t1 = k < n ? c[k].t : 1.0;
                const double ex = k < n ? c[k].px : x2, ey = k < n ? c[k].py : y2;
                const double tm = 0.5 * (t0 + t1);
                const double mx = x1 + tm * (x2 - x1), my = y1 + tm * (y2 - y1);
                // which side of this piece is its own quad's interior (none: an edge covered twice)
                const double eps = 1e-3;  // coordinates are integers (1/10000 px)
                const bool left = nms_inside_evenodd(X[q], Y[q], mx + eps * nx, my + eps * ny);
                const bool right = nms_inside_evenodd(X[q], Y[q], mx - eps * nx, my - eps * ny);
                if (left != right) {
                    const double cr = 0.5 * (sx * ey - ex * sy) * (left ? 1.0 : -1.0);
                    if (nq == 2 && nms_inside_evenodd(ox, oy, mx, my)) inter += cr;  // bounds A and B
                    else uni += cr;                                                    // bounds A or B
                }
                t0 = t1;
                sx = ex;
                sy = ey;
            }
        }
    }
}

inline float nms_poly_iou(const NmsPoly& a, const NmsPoly& b)
{
    double ax[4], ay[4], bx[4], by[4];
    bool same = true;
    for (int i = 0; i < 4; ++i) {
        ax[i] = (double)a.X[i];
        ay[i] = (double)a.Y[i];
        bx[i] = (double)b.X[i];
        by[i] = (double)b.Y[i];
        same = same && a.X[i] == b.X[i] && a.Y[i] == b.Y[i];
    }
    double inter = 0.0, uni = 0.0;
    if (nms_quad_convex(ax, ay) && nms_quad_convex(bx, by)) {
        // two convex quads: Sutherland-Hodgman, crossing points rounded to integers as Clipper rounds them
        const double a2 = nms_area2(ax, ay, 4);
        double b2 = nms_area2(bx, by, 4);
        if (b2 < 0) {  // clip polygon counter-clockwise
            for (int i = 0; i < 2; ++i) {
                std::swap(bx[i], bx[3 - i]);
                std::swap(by[i], by[3 - i]);
            }
        }
        double ox[16], oy[16], ix[16], iy[16];
        int n = 4;
        for (int i = 0; i < 4; ++i) {
            ox[i] = ax[i];
            oy[i] = ay[i];
        }
        for (int e = 0; e < 4 && n > 0; ++e) {
            const double ex = bx[(e + 1) % 4] - bx[e], ey = by[(e + 1) % 4] - by[e];
            int m = 0;
            for (int j = 0; j < n; ++j) {
                const int k = (j + 1) % n;
                const double sp = ex * (oy[j] - by[e]) - ey * (ox[j] - bx[e]);
                const double sq = ex * (oy[k] - by[e]) - ey * (ox[k] - bx[e]);
                if (sp >= 0) {
                    ix[m] = ox[j];
                    iy[m++] = oy[j];
                }
                if ((sp > 0 && sq < 0) || (sp < 0 && sq > 0)) {
                    const double t = sp / (sp - sq);
                    ix[m] = floor(ox[j] + t * (ox[k] - ox[j]) + 0.5);
                    iy[m++] = floor(oy[j] + t * (oy[k] - oy[j]) + 0.5);
                }
            }
            n = m;
            for (int j = 0; j < n; ++j) {
                ox[j] = ix[j];
                oy[j] = iy[j];
            }
        }
        inter = fabs(nms_area2(ox, oy, n)) / 2.0;
        uni = fabs(a2) / 2.0 + fabs(b2) / 2.0 - inter;
    } else if (same) {
        // the same (non-convex) quad twice (nms.h:198/201 append a polygon twice): every edge coincides
        double none;
        nms_evenodd_areas(ax, ay, nullptr, nullptr, none, uni);
        inter = uni;
    } else {
        nms_evenodd_areas(ax, ay, bx, by, inter, uni);
    }
    const float inter_f = (float)inter, uni_f = (float)uni;  // `float area` of paths_area (nms.h:17-22)
    return fabsf(inter_f) / std::max(fabsf(uni_f), 1.0f);
}

struct NmsMerger {  // nms.h:48-113
    long long data[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float score = 0.0f;
    float probs[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    void add(const NmsPoly& p)
    {
        data[0] = (long long)((float)data[0] + (float)p.X[0] * p.probs[0]);
        data[1] = (long long)((float)data[1] + (float)p.Y[0] * p.probs[3]);
        data[2] = (long long)((float)data[2] + (float)p.X[1] * p.probs[0]);
        data[3] = (long long)((float)data[3] + (float)p.Y[1] * p.probs[1]);
        data[4] = (long long)((float)data[4] + (float)p.X[2] * p.probs[2]);
        data[5] = (long long)((float)data[5] + (float)p.Y[2] * p.probs[1]);
        data[6] = (long long)((float)data[6] + (float)p.X[3] * p.probs[2]);
        data[7] = (long long)((float)data[7] + (float)p.Y[3] * p.probs[3]);
        score += p.score;
        for (int i = 0; i < 4; ++i) probs[i] += p.probs[i];
    }
    NmsPoly get() const
    {
        NmsPoly p;
        p.X[0] = (long long)((float)data[0] / probs[0]);
        p.Y[0] = (long long)((float)data[1] / probs[3]);
        p.X[1] = (long long)((float)data[2] / probs[0]);
        p.Y[1] = (long long)((float)data[3] / probs[1]);
        p.X[2] = (long long)((float)data[4] / probs[2]);
        p.Y[2] = (long long)((float)data[5] / probs[1]);
        p.X[3] = (long long)((float)data[6] / probs[2]);
        p.Y[3] = (long long)((float)data[7] / probs[3]);
        p.score = score;
        for (int i = 0; i < 4; ++i) p.probs[i] = probs[i];
        p.x = p.y = 0;
        return p;
    }
};

inline NmsPoly nms_merged(const NmsPoly& first, const NmsPoly& second)
{
    NmsMerger m;
    m.add(first);
    m.add(second);
    return m.get();
}

// nms.h:149-213 (first pass) + :116-146 (standard NMS, which merges what it suppresses)
inline std::vector<NmsPoly> nms_merge(const NmsCandidate* cand, int n, int w, int h, float thr1, float thr2)
{
    std::vector<int> poly_map((size_t)w * h, -1);
    std::vector<NmsPoly> polys;
    for (int i = 0; i < n; ++i) {
        NmsPoly poly;
        for (int v = 0; v < 4; ++v) {
            poly.X[v] = cand[i].quad[2 * v];
            poly.Y[v] = cand[i].quad[2 * v + 1];
        }
        poly.score = cand[i].score;
        {
            // adaptor.cpp:94-100, 107: ph = phx = 9; the C library's expf, as the reference calls it
            const float* r = cand[i].rdist;
            const float ph = 9, phx = 9;
            const float p_left = expf(-r[2] / phx);
            const float p_top = expf(-r[0] / ph);
            const float p_right = expf(-r[3] / phx);
            const float p_bt = expf(-r[1] / ph);
            poly.probs[0] = p_left * p_bt;
            poly.probs[1] = p_left * p_top;
            poly.probs[2] = p_right * p_top;
            poly.probs[3] = p_right * p_bt;
        }
        poly.x = cand[i].x;
        poly.y = cand[i].y;
        const size_t here = (size_t)poly.y * w + poly.x;
        if (!polys.empty()) {
            if (nms_poly_iou(poly, polys.back()) > thr1) {
                polys.back() = nms_merged(polys.back(), poly);
                poly_map[here] = (int)polys.size() - 1;
                continue;
            }
            bool done = false;
            if (poly.y > 0) {
                const size_t up = (size_t)(poly.y - 1) * w + poly.x;
                const int idx = poly_map[up];
                if (idx >= 0) {
                    // (y-1, x), then (y-1, x-1) if x > 0, then (y-1, x+1) with no bound on x (:184):
                    // at the last column that is (y, 0), still inside the map
                    const int cands[3] = {idx, poly.x > 0 ? poly_map[up - 1] : -1, poly_map[up + 1]};
                    for (int k = 0; k < 3 && !done; ++k) {
                        const int c = cands[k];
                        if (c < 0) continue;
                        if (nms_poly_iou(poly, polys[(size_t)c]) > thr1) {
                            polys[(size_t)c] = nms_merged(polys[(size_t)c], poly);
                            poly_map[here] = c;
                            done = true;
                        }
                    }
                }
            }
            if (done) continue;
            polys.push_back(poly);  // :198 -- the reference appends the polygon here AND below
        }
        polys.push_back(poly);      // :201
        poly_map[here] = (int)polys.size() - 1;
    }
    const size_t np = polys.size();
    std::vector<NmsPoly> ret;
    if (np == 0) return ret;
    std::vector<size_t> indices(np);
    for (size_t i = 0; i < np; ++i) indices[i] = i;
    std::sort(indices.begin(), indices.end(), [&](size_t i, size_t j) { return polys[i].score > polys[j].score; });
    std::vector<size_t> keep;
    while (!indices.empty()) {
        size_t p = 0;
        const size_t cur = indices[0];
        keep.push_back(cur);
        for (size_t i = 1; i < indices.size(); ++i) {
            if (!(nms_poly_iou(polys[cur], polys[indices[i]]) > thr2)) indices[p++] = indices[i];
            else polys[cur] = nms_merged(polys[indices[i]], polys[cur]);
        }
        indices.resize(p);
    }
    for (size_t i : keep) ret.push_back(polys[i]);
    return ret;
}
