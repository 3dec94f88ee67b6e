// tools/m2bench.hip -- round-2 feasibility probe for a relayout-free backward ("tile-stationary"):
// how fast can map tiles pull the bins that reach them straight out of the NCHW top_diff?
// Measurement tooling only (not product).  Host code builds, for the cfg3 workload, the list of
// (roi, ph, 4-bin quad) / (roi, ph, 16-bin piece) records per map tile with plain geometry, device
// skeletons read exactly those bytes with the lane mappings under consideration and reduce them
// trivially; nothing here computes a gradient.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/m2bench tools/m2bench.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <set>
#include <vector>

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int R = 512, C = 256, H = 160, W = 160, PH = 8, PW = 64, NB = PH * PW;

// lane = channel (64 per wave); a record = one aligned 4-bin quad of one (roi, ph) row: each lane
// loads 16 B of its own channel row (64 distinct lines per instruction)
template <int DEPTH>
__global__ __launch_bounds__(64) void k_quads(const float* __restrict__ top, const unsigned* __restrict__ off,
                                              const unsigned* __restrict__ rec, float* __restrict__ sink,
                                              int groups, const unsigned* __restrict__ order)
{
    const unsigned lane = threadIdx.x;
    const unsigned item = order[blockIdx.x];
    const unsigned tile = item / groups, grp = item % groups;
    const unsigned c = grp * 64 + lane;
    const unsigned beg = off[tile], end = off[tile + 1];
    const float* base = top + (size_t)c * NB;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (unsigned i = beg; i < end; i += DEPTH) {
        v4f v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const unsigned r = rec[min(i + d, end - 1)];   // (n << 7) | (ph << 4) | quad
            const size_t o = (size_t)(r >> 7) * C * NB + ((r >> 4) & 7u) * PW + (r & 15u) * 4u;
            v[d] = *reinterpret_cast<const v4f*>(base + o);
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
            if (i + d < end) acc += v[d];
    }
    sink[(size_t)blockIdx.x * 64 + lane] = acc.x + acc.y + acc.z + acc.w;
}

// lane = (channel of 16, 4-bin quad of 4): a record = one aligned 16-bin piece of a (roi, ph) row;
// one instruction reads 64 contiguous bytes of 16 channel rows, four instructions cover 64 channels
template <int DEPTH>
__global__ __launch_bounds__(64) void k_pieces(const float* __restrict__ top, const unsigned* __restrict__ off,
                                               const unsigned* __restrict__ rec, float* __restrict__ sink,
                                               int groups, const unsigned* __restrict__ order)
{
    const unsigned lane = threadIdx.x;
    const unsigned item = order[blockIdx.x];
    const unsigned tile = item / groups, grp = item % groups;
    const unsigned c16 = lane >> 2, j = lane & 3u;
    const unsigned beg = off[tile], end = off[tile + 1];
    const float* base = top + (size_t)(grp * 64 + c16) * NB + j * 4u;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (unsigned i = beg; i < end; i += DEPTH) {
        v4f v[DEPTH][4];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const unsigned r = rec[min(i + d, end - 1)];   // (n << 5) | (ph << 2) | piece
            const size_t o = (size_t)(r >> 5) * C * NB + ((r >> 2) & 7u) * PW + (r & 3u) * 16u;
#pragma unroll
            for (int s = 0; s < 4; ++s) v[d][s] = *reinterpret_cast<const v4f*>(base + o + (size_t)s * 16 * NB);
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
            if (i + d < end) acc += v[d][0] + v[d][1] + v[d][2] + v[d][3];
    }
    sink[(size_t)blockIdx.x * 64 + lane] = acc.x + acc.y + acc.z + acc.w;
}

// the ceiling: every byte of top_diff once, linear
__global__ __launch_bounds__(256) void k_read_all(const float* __restrict__ top, float* __restrict__ sink, size_t n4)
{
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
        acc += reinterpret_cast<const v4f*>(top)[i];
    if (acc.x == 1234.5f) sink[threadIdx.x] = acc.y;
}

struct Timer {
    hipEvent_t a, b;
    Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
    template <class F>
    double us(F&& f, int iters = 30, int warm = 5)
    {
        for (int i = 0; i < warm; ++i) f();
        CK(hipDeviceSynchronize());
        std::vector<float> t(iters);
        for (int i = 0; i < iters; ++i) {
            CK(hipEventRecord(a));
            f();
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            CK(hipEventElapsedTime(&t[i], a, b));
        }
        std::sort(t.begin(), t.end());
        return t[iters / 2] * 1e3;
    }
};

int main()
{
    std::mt19937 rng(0);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    std::vector<float> hr(R * 6);
    for (int i = 0; i < R; ++i) {
        const float h = 16 + 48 * U(rng);
        hr[i * 6 + 0] = 0;
        hr[i * 6 + 1] = 640 * U(rng);
        hr[i * 6 + 2] = 640 * U(rng);
        hr[i * 6 + 3] = h;
        hr[i * 6 + 4] = h * (4 + 4 * U(rng));
        hr[i * 6 + 5] = -90 + 180 * U(rng);
    }
    float *top, *sink;
    CK(hipMalloc(&top, (size_t)R * C * NB * 4));
    CK(hipMalloc(&sink, 64u << 20));
    CK(hipMemset(top, 0, (size_t)R * C * NB * 4));
    Timer T;
    const double MB = (double)R * C * NB * 4 / 1e6;
    printf("%-64s %9.2f us\n", "read all of top_diff once, linear (268 MB)",
           T.us([&] { hipLaunchKernelGGL(k_read_all, dim3(4096), dim3(256), 0, 0, top, sink, (size_t)R * C * NB / 4); }));

    for (int shape = 0; shape < 3; ++shape) {
        const int TW = shape == 0 ? 8 : 16, TH = shape == 0 ? 4 : shape == 1 ? 4 : 8;
        const int TX = W / TW, TY = H / TH, NT = TX * TY;
        // (tile -> set of quads / pieces) from plain geometry (the reference's recipe without its rounding care)
        std::vector<std::set<unsigned>> quads(NT), pieces(NT);
        size_t pairs = 0, bins_live = 0;
        for (int n = 0; n < R; ++n) {
            const float cx = hr[n * 6 + 1], cy = hr[n * 6 + 2], h = hr[n * 6 + 3], w = hr[n * 6 + 4];
            const float ang = hr[n * 6 + 5] / 180.f * 3.1415926535f, sc = 0.25f;
            const float rpw = PH * w / h, dx = -rpw / 2, dy = -PH / 2.f, Sx = w * sc / rpw, Sy = h * sc / PH;
            const float A = cosf(ang), Bt = sinf(ang);
            const float m00 = A * Sx, m01 = Bt * Sy, m02 = m00 * dx + m01 * dy + cx * sc;
            const float m10 = -Bt * Sx, m11 = A * Sy, m12 = m10 * dx + m11 * dy + cy * sc;
            for (int ph = 0; ph < PH; ++ph)
                for (int pw = 0; pw < PW; ++pw) {
                    if ((float)pw > rpw) continue;
                    ++bins_live;
                    float xs[4], ys[4];
                    int k = 0;
                    for (int a = 0; a < 2; ++a)
                        for (int b = 0; b < 2; ++b, ++k) {
                            xs[k] = m00 * (pw + a) + m01 * (ph + b) + m02;
                            ys[k] = m10 * (pw + a) + m11 * (ph + b) + m12;
                        }
                    const float l = std::max(roundf(*std::min_element(xs, xs + 4)), 0.f);
                    const float r = std::min(roundf(*std::max_element(xs, xs + 4)), W - 1.f);
                    const float t = std::max(roundf(*std::min_element(ys, ys + 4)), 0.f);
                    const float bo = std::min(roundf(*std::max_element(ys, ys + 4)), H - 1.f);
                    const float bx = (l + r) / 2, by = (t + bo) / 2;
                    const int x0 = (int)floorf(bx), x1 = (int)ceilf(bx), y0 = (int)floorf(by), y1 = (int)ceilf(by);
                    std::set<std::pair<int, int>> px;
                    for (int yy : {y0, y1})
                        for (int xx : {x0, x1})
                            if (xx > 0 && yy > 0 && xx < W - 1 && yy < H - 1) px.insert({yy, xx});
                    pairs += px.size();
                    for (auto& p : px) {
                        const int tile = (p.first / TH) * TX + p.second / TW;
                        quads[tile].insert((unsigned)n << 7 | (unsigned)ph << 4 | (unsigned)(pw >> 2));
                        pieces[tile].insert((unsigned)n << 5 | (unsigned)ph << 2 | (unsigned)(pw >> 4));
                    }
                }
        }
        auto flatten = [&](std::vector<std::set<unsigned>>& s, std::vector<unsigned>& off, std::vector<unsigned>& rec) {
            off.assign(1, 0);
            rec.clear();
            size_t mx = 0;
            for (auto& t : s) {
                rec.insert(rec.end(), t.begin(), t.end());
                off.push_back((unsigned)rec.size());
                mx = std::max(mx, t.size());
            }
            return mx;
        };
        std::vector<unsigned> qoff, qrec, poff, prec;
        const size_t qmax = flatten(quads, qoff, qrec), pmax = flatten(pieces, poff, prec);
        printf("tile %2d x %d: %d tiles, %zu live bins, %zu pairs; quad records %zu (x%.2f of the %zu live quads, max %zu per tile), "
               "piece records %zu (max %zu)\n",
               TW, TH, NT, bins_live, pairs, qrec.size(), qrec.size() / (bins_live / 4.0), bins_live / 4, qmax, prec.size(), pmax);
        unsigned *d_qoff, *d_qrec, *d_poff, *d_prec, *d_order;
        CK(hipMalloc(&d_qoff, qoff.size() * 4)); CK(hipMalloc(&d_qrec, qrec.size() * 4));
        CK(hipMalloc(&d_poff, poff.size() * 4)); CK(hipMalloc(&d_prec, prec.size() * 4));
        CK(hipMemcpy(d_qoff, qoff.data(), qoff.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_qrec, qrec.data(), qrec.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_poff, poff.data(), poff.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_prec, prec.data(), prec.size() * 4, hipMemcpyHostToDevice));
        const int groups = C / 64, items = NT * groups;
        CK(hipMalloc(&d_order, items * 4));
        for (int ord = 0; ord < 3; ++ord) {
            // 0: block -> item in raster order; 1: the 8 XCDs (block % 8) own 8 map regions (2 x 4), tiles of
            // one region and its channel groups on one XCD; 2: like 1 but longest lists first
            std::vector<unsigned> order(items);
            if (ord == 0) {
                for (int i = 0; i < items; ++i) order[i] = i;
            } else {
                std::vector<std::vector<unsigned>> per(8);
                for (int ty = 0; ty < TY; ++ty)
                    for (int tx = 0; tx < TX; ++tx) {
                        const int reg = (ty * 4 / TY) * 2 + (tx * 2 / TX);
                        for (int g = 0; g < groups; ++g) per[reg].push_back((ty * TX + tx) * groups + g);
                    }
                if (ord == 2)
                    for (auto& v : per)
                        std::stable_sort(v.begin(), v.end(), [&](unsigned a, unsigned b) {
                            return quads[a / groups].size() > quads[b / groups].size(); });
                for (int i = 0; i < items; ++i) order[i] = per[i % 8][i / 8];
            }
            CK(hipMemcpy(d_order, order.data(), items * 4, hipMemcpyHostToDevice));
            char nm[128];
            const char* on[3] = {"raster", "8 regions <-> 8 XCDs", "regions, longest first"};
            snprintf(nm, 128, "  quads  lane=channel          depth 8, order %s", on[ord]);
            printf("%-64s %9.2f us\n", nm, T.us([&] { hipLaunchKernelGGL(k_quads<8>, dim3(items), dim3(64), 0, 0, top, d_qoff, d_qrec, sink, groups, d_order); }));
            snprintf(nm, 128, "  pieces lane=(16 ch, 4 quads) depth 2, order %s", on[ord]);
            printf("%-64s %9.2f us\n", nm, T.us([&] { hipLaunchKernelGGL(k_pieces<2>, dim3(items), dim3(64), 0, 0, top, d_poff, d_prec, sink, groups, d_order); }));
            fflush(stdout);
        }
        snprintf(nullptr, 0, "%f", MB);
        CK(hipFree(d_qoff)); CK(hipFree(d_qrec)); CK(hipFree(d_poff)); CK(hipFree(d_prec)); CK(hipFree(d_order));
    }
    return 0;
}
