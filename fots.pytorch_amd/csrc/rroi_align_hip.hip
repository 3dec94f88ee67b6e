// rroi_align_hip.hip -- RoIRotate (rroi_align) for MI355X / gfx950 (CDNA4).
//
// Written from scratch for wave64 / LDS / per-XCD-L2 hardware; it replaces the
// reference's CUDA kernels (rroi_align/src/rroi_align_kernel.cu:28-162 forward,
// :193-278 backward, launchers :164-187 / :280-312) behind the C-ABI declared in
// include/rroi_align_hip.h.  Build: -ffp-contract=off (the arithmetic recipe of
// the bin geometry is rounding-exact; see DESIGN.md "Arithmetic recipe").
//
// Design in one paragraph.  A bin's sample point depends on (roi, ph, pw) only,
// never on the channel, and its 4 taps are whole pixels.  The op is therefore a
// pixel gather replicated over C channels plus a 256 MiB streaming write.  In
// NCHW the channel vector of a pixel is strided by H*W (a cache line per
// channel), so the hot path first relays the map out pixel-major
// (B,H,W,Cs; one 26 MB pass that the gather then reads from L2), after which a
// tap is ONE contiguous run of channels.  Each wave owns a [32 channel] x
// [64 bin] output tile: lanes = 8 bins x 8 channel-quads fetch taps as 16-byte
// loads (8 lanes cover a pixel's 128-byte line of 32 channels), blend in the
// reference's order, transpose through a wave-private LDS tile and stream the
// tile out as full 256-byte rows of the (R,C,PH,PW) tensor.  Channel chunk k is
// handled by blocks with blockIdx % nchunks == k, i.e. (8 chunks at C=256) by
// one XCD, whose 4 MiB L2 then holds exactly its 3.2 MB slice of the map.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rroi_align_hip.h"

#pragma clang fp contract(off)

namespace {

constexpr int kWave = 64;
constexpr int kChunk = 32;     // channels per work item  (8 lanes x 16 B = one 128 B line)
constexpr int kTileBins = 64;  // bins per work item       (one 256 B output row segment)
constexpr int kTStride = 68;   // LDS tile row stride in dwords: 4*odd -> writes <=2-way, b128 reads aligned
constexpr int kQuads = kChunk / 4;
constexpr int kBinsPerIter = kWave / kQuads;  // 8
constexpr int kIters = kTileBins / kBinsPerIter;  // 8

struct Affine {  // kernel.cu:78-84 (M), :68 (roi_pooled_width), :60 (roi_batch_ind)
    float m00, m01, m02, m10, m11, m12, rpw;
    int batch;
};
static_assert(sizeof(Affine) == 32, "Affine is read as two 16-byte scalars");

__device__ __forceinline__ float as_f(unsigned u) { return __uint_as_float(u); }
__device__ __forceinline__ unsigned as_u(float f) { return __float_as_uint(f); }

// (int)x with the semantics of v_cvt_i32_f32 / cvt.rzi.s32.f32: saturating, NaN -> 0.
__device__ __forceinline__ int f2i_sat(float x)
{
    if (x != x) return 0;
    if (x >= 2147483648.0f) return INT32_MAX;
    if (x <= -2147483648.0f) return INT32_MIN;
    return (int)x;
}

// kernel.cu:58-84.  Every * and + below is one separately rounded fp32
// operation, in source order; the degree->radian conversion is the
// reference's double expression (:65); cos/sin are evaluated in double and
// rounded once to fp32 (recipe shared with oracle/rroi_align_oracle.c).
__device__ __forceinline__ Affine make_affine(const float* __restrict__ roi, int pooled_height,
                                              float spatial_scale)
{
    Affine A;
    A.batch = f2i_sat(roi[0]);
    const float cx = roi[1], cy = roi[2], h = roi[3], w = roi[4];
    const float angle = (float)(((double)roi[5] / 180.0) * 3.1415926535);
    const float rpw = ((float)pooled_height * w) / h;
    const float dx = -rpw / 2.0f;
    const float dy = (float)(-pooled_height / 2.0);
    const float Sx = (w * spatial_scale) / rpw;
    const float Sy = (h * spatial_scale) / (float)pooled_height;
    const float Alpha = (float)cos((double)angle);
    const float Beta = (float)sin((double)angle);
    const float Dx = cx * spatial_scale;
    const float Dy = cy * spatial_scale;
    A.m00 = Alpha * Sx;
    A.m01 = Beta * Sy;
    A.m02 = ((A.m00 * dx) + (A.m01 * dy)) + Dx;
    A.m10 = (-Beta) * Sx;
    A.m11 = Alpha * Sy;
    A.m12 = ((A.m10 * dx) + (A.m11 * dy)) + Dy;
    A.rpw = rpw;
    return A;
}

// kernel.cu:86-107: centre of the rounded+clamped bounding box of the bin's four
// transformed corners; returns in_rroi.
__device__ __forceinline__ bool bin_centre(const Affine& A, int ph, int pw, int height, int width,
                                           float& bin_cx, float& bin_cy)
{
    const float fpw = (float)pw, fph = (float)ph;
    const float fpw1 = (float)(pw + 1), fph1 = (float)(ph + 1);
    const float P0 = ((A.m00 * fpw) + (A.m01 * fph)) + A.m02;
    const float P1 = ((A.m10 * fpw) + (A.m11 * fph)) + A.m12;
    const float P2 = ((A.m00 * fpw) + (A.m01 * fph1)) + A.m02;
    const float P3 = ((A.m10 * fpw) + (A.m11 * fph1)) + A.m12;
    const float P4 = ((A.m00 * fpw1) + (A.m01 * fph)) + A.m02;
    const float P5 = ((A.m10 * fpw1) + (A.m11 * fph)) + A.m12;
    const float P6 = ((A.m00 * fpw1) + (A.m01 * fph1)) + A.m02;
    const float P7 = ((A.m10 * fpw1) + (A.m11 * fph1)) + A.m12;
    // fmaxf/fminf drop a NaN operand, as CUDA's max/min(float,double) do.
    const float leftMost = fmaxf(roundf(fminf(fminf(P0, P2), fminf(P4, P6))), 0.0f);
    const float rightMost = fminf(roundf(fmaxf(fmaxf(P0, P2), fmaxf(P4, P6))), (float)width - 1.0f);
    const float topMost = fmaxf(roundf(fminf(fminf(P1, P3), fminf(P5, P7))), 0.0f);
    const float bottomMost = fminf(roundf(fmaxf(fmaxf(P1, P3), fmaxf(P5, P7))), (float)height - 1.0f);
    bin_cx = (leftMost + rightMost) / 2.0f;
    bin_cy = (topMost + bottomMost) / 2.0f;
    return fpw <= A.rpw;
}

// Tap record of one bin.  Taps are whole pixels: x1 = x0 + dx, y1 = y0 + dy with
// dx,dy in {0,1} (bin centres are multiples of 0.5), so when dx == 0 the
// reference's "right" taps ARE its left taps (same pixel, same validity) and
// need no load of their own.
enum : unsigned {
    kV00 = 1u,   // lt valid: y0>0 && x0>0 && y0<H && x0<W     (kernel.cu:116)
    kV01 = 2u,   // rt                                           (:119)
    kV10 = 4u,   // lb                                           (:122)
    kV11 = 8u,   // rb                                           (:125)
    kDx = 16u,   // x1 != x0
    kDy = 32u,   // y1 != y0
    kActive = 64u,
    // backward's own, stricter bounds (kernel.cu:267-274)
    kB00 = 128u, kB01 = 256u, kB11 = 512u, kB10 = 1024u,
};

struct Taps {
    unsigned o_lt;   // ((y0*W + x0) * pixel_stride) mod 2^32; only dereferenced when valid
    unsigned flags;
    float rx, ry;    // kernel.cu:128-129
};

__device__ __forceinline__ Taps make_taps(float bin_cx, float bin_cy, bool active, int height,
                                          int width, unsigned pixel_stride)
{
    const float fx = floorf(bin_cx), fy = floorf(bin_cy);
    const int x0 = f2i_sat(fx), x1 = f2i_sat(ceilf(bin_cx));
    const int y0 = f2i_sat(fy), y1 = f2i_sat(ceilf(bin_cy));
    Taps t;
    t.rx = bin_cx - fx;
    t.ry = bin_cy - fy;
    unsigned f = 0;
    if (active) {
        f = kActive;
        const bool x0ok = x0 > 0 && x0 < width, x1ok = x1 > 0 && x1 < width;
        const bool y0ok = y0 > 0 && y0 < height, y1ok = y1 > 0 && y1 < height;
        if (y0ok && x0ok) f |= kV00;
        if (y0ok && x1ok) f |= kV01;
        if (y1ok && x0ok) f |= kV10;
        if (y1ok && x1ok) f |= kV11;
        if (x1 != x0) f |= kDx;
        if (y1 != y0) f |= kDy;
        // kernel.cu:267-274, term by term
        if (y0 > 0 && x0 > 0 && y0 < height - 1 && x0 < width - 1) f |= kB00;
        if (y0 > 0 && x1 < width - 1 && y0 < height - 1 && x1 > 0) f |= kB01;
        if (y1 < height - 1 && x1 < width - 1 && y1 > 0 && x1 > 0) f |= kB11;
        if (y1 < height - 1 && x0 > 0 && y1 > 0 && x0 < width - 1) f |= kB10;
    }
    t.flags = f;
    t.o_lt = ((unsigned)y0 * (unsigned)width + (unsigned)x0) * pixel_stride;
    return t;
}

// kernel.cu:131-134 / :248-251.  The reference forms these in double and rounds
// once; rx, ry are 0, 0.5 or NaN, for which the fp32 evaluation is identical.
__device__ __forceinline__ void tap_weights(float rx, float ry, float& wlt, float& wrt, float& wrb,
                                            float& wlb)
{
    wlt = (1.0f - rx) * (1.0f - ry);
    wrt = rx * (1.0f - ry);
    wrb = rx * ry;
    wlb = (1.0f - rx) * ry;
}

// kernel.cu:136-141: inter_val = 0; += lt*wlt; += rt*wrt; += rb*wrb; += lb*wlb.
__device__ __forceinline__ float blend1(float lt, float rt, float rb, float lb, float wlt,
                                        float wrt, float wrb, float wlb)
{
    float v = 0.0f;
    v += lt * wlt;
    v += rt * wrt;
    v += rb * wrb;
    v += lb * wlb;
    return v;
}

// ------------------------------------------------------------------------------------
// K0a: per-ROI affine table (R x 32 B).  One thread per ROI.
// ------------------------------------------------------------------------------------
__global__ void rroi_affine_kernel(const float* __restrict__ rois, int num_rois, int pooled_height,
                                   float spatial_scale, Affine* __restrict__ aff)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < num_rois) aff[n] = make_affine(rois + (size_t)n * 6, pooled_height, spatial_scale);
}

// ------------------------------------------------------------------------------------
// K0b: relayout (B,C,H,W) -> pixel-major (B,H*W,Cs), Cs = C rounded up to 4.
// 256 threads move a [64 channel] x [64 pixel] tile through LDS: coalesced 256 B
// reads along pixels, coalesced 256 B writes along channels.  The last
// gridDim.x - relayout_blocks blocks fill the affine table instead, so the whole
// prologue is one launch.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rroi_prologue_kernel(
    const float* __restrict__ nchw, float* __restrict__ pm, int C, int Cs, int HW, int ptiles,
    int ctiles, int relayout_blocks, const float* __restrict__ rois, int num_rois,
    int pooled_height, float spatial_scale, Affine* __restrict__ aff)
{
    __shared__ float T[64 * 65];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= relayout_blocks) {
        const int n = ((int)blockIdx.x - relayout_blocks) * 256 + tid;
        if (n < num_rois) aff[n] = make_affine(rois + (size_t)n * 6, pooled_height, spatial_scale);
        return;
    }
    int bid = blockIdx.x;
    const int pt = bid % ptiles;
    bid /= ptiles;
    const int ct = bid % ctiles;
    const int b = bid / ctiles;
    const int lane = tid & 63, w = tid >> 6;
    const int p0 = pt * 64, c0 = ct * 64;
    const float* src = nchw + ((size_t)b * C + c0) * HW + p0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = w * 16 + i;
        float v = 0.0f;
        if (c0 + c < C && p0 + lane < HW) v = src[(size_t)c * HW + lane];
        T[c * 65 + lane] = v;
    }
    __syncthreads();
    float* dst = pm + ((size_t)b * HW + p0) * Cs + c0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int p = w * 16 + i;
        if (p0 + p < HW && c0 + lane < Cs) dst[(size_t)p * Cs + lane] = T[lane * 65 + p];
    }
}

// ------------------------------------------------------------------------------------
// K1: the hot kernel.  One wave per block; block -> channel chunk k = blockIdx %
// nchunks (XCD affinity), and a grid-stride loop over (roi, 64-bin tile) items.
// ------------------------------------------------------------------------------------
struct TapRegs {
    float4 lt, rt, lb, rb;
};

typedef float v4f __attribute__((ext_vector_type(4)));  // native vector: nontemporal builtins take it

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4_nt(float* p, const float4& v)
{
    v4f t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(p));
}
__device__ __forceinline__ float4 ld4_nt(const float* p)
{
    const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
    return make_float4(t.x, t.y, t.z, t.w);
}

__device__ __forceinline__ TapRegs load_taps(const float* __restrict__ sp, const Taps& g,
                                             unsigned pixel_stride, unsigned row_stride, bool chok)
{
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    TapRegs r;
    const unsigned f = chok ? g.flags : 0u;
    const unsigned o_rt = g.o_lt + pixel_stride;
    const unsigned o_lb = g.o_lt + row_stride;
    const unsigned o_rb = o_lb + pixel_stride;
    r.lt = z;
    r.rt = z;
    r.lb = z;
    r.rb = z;
    if (f & kV00) r.lt = ld4(sp + g.o_lt);
    if ((f & (kV01 | kDx)) == (kV01 | kDx)) r.rt = ld4(sp + o_rt);
    if ((f & (kV10 | kDy)) == (kV10 | kDy)) r.lb = ld4(sp + o_lb);
    if ((f & (kV11 | kDx | kDy)) == (kV11 | kDx | kDy)) r.rb = ld4(sp + o_rb);
    return r;
}

// Resolve the taps that alias an already loaded pixel (dx == 0 and/or dy == 0).
__device__ __forceinline__ void alias_taps(TapRegs& r, unsigned f)
{
    if (!(f & kDx)) r.rt = r.lt;
    if (!(f & kDy)) r.lb = r.lt;
    if (!(f & kDx))
        r.rb = r.lb;  // x1 == x0: rb is the pixel below lt, i.e. lb (which is lt when dy == 0 too)
    else if (!(f & kDy))
        r.rb = r.rt;  // y1 == y0: rb is rt
}

template <bool VEC_STORE>
__global__ __launch_bounds__(kWave) void rroi_fwd_tiled_kernel(
    const float* __restrict__ pm,      // pixel-major features (B, H*W, Cs)
    const Affine* __restrict__ aff,    // (R)
    float* __restrict__ out,           // (R, C, PH*PW)
    int num_rois, int C, int Cs, int height, int width, int pooled_height, int pooled_width,
    int batch_size, int nchunks, int ntiles)
{
    __shared__ __attribute__((aligned(16))) float T[kChunk * kTStride];
    __shared__ __attribute__((aligned(16))) uint4 G[kTileBins];

    const int lane = threadIdx.x;
    const int k = blockIdx.x % nchunks;
    const int slot = blockIdx.x / nchunks;
    const int nslots = gridDim.x / nchunks;
    const int NB = pooled_height * pooled_width;
    const long items = (long)num_rois * ntiles;
    const unsigned pixel_stride = (unsigned)Cs;
    const unsigned row_stride = (unsigned)width * (unsigned)Cs;

    const int b = lane & (kBinsPerIter - 1), q = lane >> 3;
    const int ch = k * kChunk + q * 4;
    const bool chok = ch < Cs;

    for (long item = slot; item < items; item += nslots) {
        const int n = (int)(item / ntiles);
        const int t = (int)(item - (long)n * ntiles);

        // ---- phase A: lane = bin; geometry -> LDS record -----------------------------
        const Affine A = aff[n];
        {
            const int bin = t * kTileBins + lane;
            const int ph = bin / pooled_width;
            const int pw = bin - ph * pooled_width;
            float bcx, bcy;
            bool active = bin_centre(A, ph, pw, height, width, bcx, bcy);
            active = active && bin < NB && A.batch >= 0 && A.batch < batch_size;
            const Taps tp = make_taps(bcx, bcy, active, height, width, pixel_stride);
            G[lane] = make_uint4(tp.o_lt, tp.flags, as_u(tp.rx), as_u(tp.ry));
        }
        __syncthreads();

        // ---- phase B: lane = (bin b of 8, channel quad q of 8); gather + blend --------
        const int batch = (A.batch >= 0 && A.batch < batch_size) ? A.batch : 0;
        const float* sp = pm + (size_t)batch * height * width * Cs + ch;

        Taps g[kIters];
#pragma unroll
        for (int it = 0; it < kIters; ++it) {
            const uint4 u = G[it * kBinsPerIter + b];
            g[it].o_lt = u.x;
            g[it].flags = u.y;
            g[it].rx = as_f(u.z);
            g[it].ry = as_f(u.w);
        }
        TapRegs cur = load_taps(sp, g[0], pixel_stride, row_stride, chok);
#pragma unroll
        for (int it = 0; it < kIters; ++it) {
            TapRegs nxt;
            if (it + 1 < kIters) nxt = load_taps(sp, g[it + 1], pixel_stride, row_stride, chok);
            alias_taps(cur, g[it].flags);
            float wlt, wrt, wrb, wlb;
            tap_weights(g[it].rx, g[it].ry, wlt, wrt, wrb, wlb);
            const bool act = g[it].flags & kActive;
            const float v0 = act ? blend1(cur.lt.x, cur.rt.x, cur.rb.x, cur.lb.x, wlt, wrt, wrb, wlb) : 0.0f;
            const float v1 = act ? blend1(cur.lt.y, cur.rt.y, cur.rb.y, cur.lb.y, wlt, wrt, wrb, wlb) : 0.0f;
            const float v2 = act ? blend1(cur.lt.z, cur.rt.z, cur.rb.z, cur.lb.z, wlt, wrt, wrb, wlb) : 0.0f;
            const float v3 = act ? blend1(cur.lt.w, cur.rt.w, cur.rb.w, cur.lb.w, wlt, wrt, wrb, wlb) : 0.0f;
            float* tw = T + (q * 4) * kTStride + it * kBinsPerIter + b;
            tw[0 * kTStride] = v0;
            tw[1 * kTStride] = v1;
            tw[2 * kTStride] = v2;
            tw[3 * kTStride] = v3;
            if (it + 1 < kIters) cur = nxt;
        }
        __syncthreads();

        // ---- phase C: lane = (row r of 4, 4 consecutive bins); stream the tile out ----
        const int col = (lane & 15) * 4;
        const int bin0 = t * kTileBins + col;
#pragma unroll
        for (int s = 0; s < kChunk / 4; ++s) {
            const int r = s * 4 + (lane >> 4);
            const int c = k * kChunk + r;
            const float4 v = *reinterpret_cast<const float4*>(T + r * kTStride + col);
            if (c < C) {
                float* op = out + ((size_t)n * C + c) * NB + bin0;
                if (VEC_STORE) {
                    // NB % 4 == 0: the 4 bins are all inside or all outside the row, 16 B aligned
                    if (bin0 < NB) st4_nt(op, v);
                } else {
                    if (bin0 + 0 < NB) __builtin_nontemporal_store(v.x, op + 0);
                    if (bin0 + 1 < NB) __builtin_nontemporal_store(v.y, op + 1);
                    if (bin0 + 2 < NB) __builtin_nontemporal_store(v.z, op + 2);
                    if (bin0 + 3 < NB) __builtin_nontemporal_store(v.w, op + 3);
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------
// K2: direct NCHW forward, no workspace: thread = (roi, bin), loops a channel slab.
// Used for small R (where relaying out the whole map would dominate) and by the
// reference-ABI launcher; optionally writes the reference's con_idx_x / con_idx_y.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rroi_fwd_direct_kernel(
    const float* __restrict__ feat, const float* __restrict__ rois, float* __restrict__ out,
    float* __restrict__ idx_x, float* __restrict__ idx_y, int num_rois, int C, int height,
    int width, int pooled_height, int pooled_width, float spatial_scale, int batch_size,
    int cslab)
{
    const int NB = pooled_height * pooled_width;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)num_rois * NB) return;
    const int n = (int)(gid / NB);
    const int bin = (int)(gid - (long)n * NB);
    const int ph = bin / pooled_width, pw = bin - ph * pooled_width;

    const Affine A = make_affine(rois + (size_t)n * 6, pooled_height, spatial_scale);
    float bcx, bcy;
    const bool in_rroi = bin_centre(A, ph, pw, height, width, bcx, bcy);
    // batch_size < 0: unknown (reference ABI) -> trust the index like the reference does
    const bool batch_ok = batch_size < 0 || (A.batch >= 0 && A.batch < batch_size);
    const bool active = in_rroi && batch_ok;
    const Taps tp = make_taps(bcx, bcy, active, height, width, 1u);
    float wlt, wrt, wrb, wlb;
    tap_weights(tp.rx, tp.ry, wlt, wrt, wrb, wlb);
    const unsigned f = tp.flags;
    const unsigned o_lt = tp.o_lt;
    const unsigned o_rt = o_lt + ((f & kDx) ? 1u : 0u);
    const unsigned o_lb = o_lt + ((f & kDy) ? (unsigned)width : 0u);
    const unsigned o_rb = o_lb + ((f & kDx) ? 1u : 0u);

    const size_t HW = (size_t)height * width;
    const int c_begin = blockIdx.y * cslab;
    const int c_end = min(C, c_begin + cslab);
    const float* plane = feat + ((size_t)(batch_ok ? A.batch : 0) * C + c_begin) * HW;
    size_t o = ((size_t)n * C + c_begin) * NB + bin;
    for (int c = c_begin; c < c_end; ++c, plane += HW, o += NB) {
        float v = 0.0f;
        if (active) {
            const float lt = (f & kV00) ? plane[o_lt] : 0.0f;
            const float rt = (f & kV01) ? plane[o_rt] : 0.0f;
            const float lb = (f & kV10) ? plane[o_lb] : 0.0f;
            const float rb = (f & kV11) ? plane[o_rb] : 0.0f;
            v = blend1(lt, rt, rb, lb, wlt, wrt, wrb, wlb);
        }
        out[o] = v;
        if (idx_x) idx_x[o] = active ? bcx : 0.0f;
        if (idx_y) idx_y[o] = active ? bcy : 0.0f;
    }
}

// ------------------------------------------------------------------------------------
// Backward, tiled: scatter into a pixel-major gradient (B, H*W, Cs) with hardware
// fp32 atomics (lanes of a wave hit consecutive channels of a pixel, so an
// atomic instruction touches whole lines, and with the XCD mapping all atomics
// of a channel chunk resolve in one L2), then relayout to NCHW.
// ------------------------------------------------------------------------------------
template <bool VEC_LOAD>
__global__ __launch_bounds__(kWave) void rroi_bwd_tiled_kernel(
    const float* __restrict__ top_diff,  // (R, C, PH*PW)
    const Affine* __restrict__ aff, float* __restrict__ gpm,  // (B, H*W, Cs) zeroed
    int num_rois, int C, int Cs, int height, int width, int pooled_height, int pooled_width,
    int batch_size, int nchunks, int ntiles)
{
    __shared__ __attribute__((aligned(16))) float T[kChunk * kTStride];
    __shared__ __attribute__((aligned(16))) uint4 G[kTileBins];

    const int lane = threadIdx.x;
    const int k = blockIdx.x % nchunks;
    const int slot = blockIdx.x / nchunks;
    const int nslots = gridDim.x / nchunks;
    const int NB = pooled_height * pooled_width;
    const long items = (long)num_rois * ntiles;
    const unsigned pixel_stride = (unsigned)Cs;
    const unsigned row_stride = (unsigned)width * (unsigned)Cs;
    const int b = lane & (kBinsPerIter - 1), q = lane >> 3;
    const int ch = k * kChunk + q * 4;

    for (long item = slot; item < items; item += nslots) {
        const int n = (int)(item / ntiles);
        const int t = (int)(item - (long)n * ntiles);
        const Affine A = aff[n];
        {
            const int bin = t * kTileBins + lane;
            const int ph = bin / pooled_width;
            const int pw = bin - ph * pooled_width;
            float bcx, bcy;
            // kernel.cu:232-242: the backward reads the centre the forward stored; where the
            // forward's mask (pw <= roi_pooled_width) was false it stored nothing, the
            // buffer holds 0, and a (0,0) centre fails every bound of :267-274.  So the
            // scatter happens exactly where the forward's mask holds.
            bool active = bin_centre(A, ph, pw, height, width, bcx, bcy);
            active = active && bin < NB && A.batch >= 0 && A.batch < batch_size;
            const Taps tp = make_taps(bcx, bcy, active, height, width, pixel_stride);
            G[lane] = make_uint4(tp.o_lt, tp.flags, as_u(tp.rx), as_u(tp.ry));
        }
        // stage the [32 ch][64 bin] slice of top_diff
        const int col = (lane & 15) * 4;
        const int bin0 = t * kTileBins + col;
#pragma unroll
        for (int s = 0; s < kChunk / 4; ++s) {
            const int r = s * 4 + (lane >> 4);
            const int c = k * kChunk + r;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < C) {
                const float* ip = top_diff + ((size_t)n * C + c) * NB + bin0;
                if (VEC_LOAD) {
                    if (bin0 < NB) v = ld4_nt(ip);
                } else {
                    if (bin0 + 0 < NB) v.x = ip[0];
                    if (bin0 + 1 < NB) v.y = ip[1];
                    if (bin0 + 2 < NB) v.z = ip[2];
                    if (bin0 + 3 < NB) v.w = ip[3];
                }
            }
            *reinterpret_cast<float4*>(T + r * kTStride + col) = v;
        }
        __syncthreads();

        const int batch = (A.batch >= 0 && A.batch < batch_size) ? A.batch : 0;
        float* gp = gpm + (size_t)batch * height * width * Cs + ch;
#pragma unroll
        for (int it = 0; it < kIters; ++it) {
            const uint4 u = G[it * kBinsPerIter + b];
            const unsigned f = u.y;
            float wlt, wrt, wrb, wlb;
            tap_weights(as_f(u.z), as_f(u.w), wlt, wrt, wrb, wlb);
            const float* tr = T + (q * 4) * kTStride + it * kBinsPerIter + b;
            const unsigned o_lt = u.x;
            const unsigned o_rt = o_lt + ((f & kDx) ? pixel_stride : 0u);
            const unsigned o_lb = o_lt + ((f & kDy) ? row_stride : 0u);
            const unsigned o_rb = o_lb + ((f & kDx) ? pixel_stride : 0u);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (ch + j >= C) break;
                const float gval = tr[j * kTStride];
                // kernel.cu:260-274: v1..v4 = w * top_diff, four independent atomicAdds
                if (f & kB00) unsafeAtomicAdd(gp + o_lt + j, wlt * gval);
                if (f & kB01) unsafeAtomicAdd(gp + o_rt + j, wrt * gval);
                if (f & kB11) unsafeAtomicAdd(gp + o_rb + j, wrb * gval);
                if (f & kB10) unsafeAtomicAdd(gp + o_lb + j, wlb * gval);
            }
        }
        __syncthreads();
    }
}

// pixel-major (B, H*W, Cs) -> NCHW (B, C, H*W); inverse of the prologue's relayout.
__global__ __launch_bounds__(256) void rroi_pm_to_nchw_kernel(const float* __restrict__ pm,
                                                              float* __restrict__ nchw, int C,
                                                              int Cs, int HW, int ptiles,
                                                              int ctiles)
{
    __shared__ float T[64 * 65];
    const int tid = threadIdx.x;
    int bid = blockIdx.x;
    const int pt = bid % ptiles;
    bid /= ptiles;
    const int ct = bid % ctiles;
    const int b = bid / ctiles;
    const int lane = tid & 63, w = tid >> 6;
    const int p0 = pt * 64, c0 = ct * 64;
    const float* src = pm + ((size_t)b * HW + p0) * Cs + c0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int p = w * 16 + i;
        float v = 0.0f;
        if (p0 + p < HW && c0 + lane < Cs) v = src[(size_t)p * Cs + lane];
        T[lane * 65 + p] = v;
    }
    __syncthreads();
    float* dst = nchw + ((size_t)b * C + c0) * HW + p0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = w * 16 + i;
        if (c0 + c < C && p0 + lane < HW) dst[(size_t)c * HW + lane] = T[c * 65 + lane];
    }
}

// Backward, direct NCHW (small R): thread = (roi, bin), loops a channel slab.
__global__ __launch_bounds__(256) void rroi_bwd_direct_kernel(
    const float* __restrict__ top_diff, const float* __restrict__ rois,
    float* __restrict__ bottom_diff, int num_rois, int C, int height, int width,
    int pooled_height, int pooled_width, float spatial_scale, int batch_size, int cslab)
{
    const int NB = pooled_height * pooled_width;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)num_rois * NB) return;
    const int n = (int)(gid / NB);
    const int bin = (int)(gid - (long)n * NB);
    const int ph = bin / pooled_width, pw = bin - ph * pooled_width;
    const Affine A = make_affine(rois + (size_t)n * 6, pooled_height, spatial_scale);
    if (A.batch < 0 || A.batch >= batch_size) return;
    float bcx, bcy;
    if (!bin_centre(A, ph, pw, height, width, bcx, bcy)) return;  // see rroi_bwd_tiled_kernel
    const Taps tp = make_taps(bcx, bcy, true, height, width, 1u);
    float wlt, wrt, wrb, wlb;
    tap_weights(tp.rx, tp.ry, wlt, wrt, wrb, wlb);
    const unsigned f = tp.flags;
    const unsigned o_lt = tp.o_lt;
    const unsigned o_rt = o_lt + ((f & kDx) ? 1u : 0u);
    const unsigned o_lb = o_lt + ((f & kDy) ? (unsigned)width : 0u);
    const unsigned o_rb = o_lb + ((f & kDx) ? 1u : 0u);
    const size_t HW = (size_t)height * width;
    const int c_begin = blockIdx.y * cslab;
    const int c_end = min(C, c_begin + cslab);
    float* plane = bottom_diff + ((size_t)A.batch * C + c_begin) * HW;
    size_t o = ((size_t)n * C + c_begin) * NB + bin;
    for (int c = c_begin; c < c_end; ++c, plane += HW, o += NB) {
        const float g = top_diff[o];
        if (f & kB00) unsafeAtomicAdd(plane + o_lt, wlt * g);
        if (f & kB01) unsafeAtomicAdd(plane + o_rt, wrt * g);
        if (f & kB11) unsafeAtomicAdd(plane + o_rb, wrb * g);
        if (f & kB10) unsafeAtomicAdd(plane + o_lb, wlb * g);
    }
}

// Backward of the reference ABI: literal per-element body of kernel.cu:207-277,
// reading the bin centre of EVERY element from con_idx_x / con_idx_y.
__global__ __launch_bounds__(256) void rroi_bwd_literal_kernel(
    const float* __restrict__ top_diff, const float* __restrict__ con_idx_x,
    const float* __restrict__ con_idx_y, const float* __restrict__ rois,
    float* __restrict__ bottom_diff, long nthreads, int C, int height, int width,
    int pooled_height, int pooled_width)
{
    for (long index = (long)blockIdx.x * blockDim.x + threadIdx.x; index < nthreads;
         index += (long)blockDim.x * gridDim.x) {
        long n = index;
        const int pw = (int)(n % pooled_width);
        n /= pooled_width;
        n /= pooled_height;
        const int c = (int)(n % C);
        n /= C;
        const float* roi = rois + n * 6;
        const int roi_batch_ind = f2i_sat(roi[0]);
        const float h = roi[3], w = roi[4];
        const float rpw = ((float)pooled_height * w) / h;
        if ((float)pw > rpw) continue;
        const float bcx = con_idx_x[index], bcy = con_idx_y[index];
        const Taps tp = make_taps(bcx, bcy, true, height, width, 1u);
        float wlt, wrt, wrb, wlb;
        tap_weights(tp.rx, tp.ry, wlt, wrt, wrb, wlb);
        const unsigned f = tp.flags;
        const unsigned o_lt = tp.o_lt;
        const unsigned o_rt = o_lt + ((f & kDx) ? 1u : 0u);
        const unsigned o_lb = o_lt + ((f & kDy) ? (unsigned)width : 0u);
        const unsigned o_rb = o_lb + ((f & kDx) ? 1u : 0u);
        float* plane = bottom_diff + ((size_t)roi_batch_ind * C + c) * height * width;
        const float g = top_diff[index];
        if (f & kB00) unsafeAtomicAdd(plane + o_lt, wlt * g);
        if (f & kB01) unsafeAtomicAdd(plane + o_rt, wrt * g);
        if (f & kB11) unsafeAtomicAdd(plane + o_rb, wrb * g);
        if (f & kB10) unsafeAtomicAdd(plane + o_lb, wlb * g);
    }
}

__global__ void rroi_bin_centres_kernel(const float* __restrict__ rois, float* __restrict__ geom,
                                        int num_rois, int height, int width, int pooled_height,
                                        int pooled_width, float spatial_scale)
{
    const int NB = pooled_height * pooled_width;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)num_rois * NB) return;
    const int n = (int)(gid / NB);
    const int bin = (int)(gid - (long)n * NB);
    const int ph = bin / pooled_width, pw = bin - ph * pooled_width;
    const Affine A = make_affine(rois + (size_t)n * 6, pooled_height, spatial_scale);
    float bcx, bcy;
    const bool in_rroi = bin_centre(A, ph, pw, height, width, bcx, bcy);
    geom[gid * 2 + 0] = in_rroi ? bcx : 0.0f;
    geom[gid * 2 + 1] = in_rroi ? bcy : 0.0f;
}

__global__ void rroi_sincos_probe_kernel(const float* __restrict__ deg, int n, float* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float angle = (float)(((double)deg[i] / 180.0) * 3.1415926535);
    out[2 * i + 0] = (float)cos((double)angle);
    out[2 * i + 1] = (float)sin((double)angle);
}

// ------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------
inline int status_of(hipError_t e) { return e == hipSuccess ? 1 : -(int)e; }
inline int launch_status() { return status_of(hipGetLastError()); }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
inline int round_up4(int c) { return (c + 3) & ~3; }
inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

struct DeviceShape {
    int cus = 256;
    bool init = false;
};
DeviceShape g_dev;

int num_cus()
{
    if (!g_dev.init) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess &&
            p.multiProcessorCount > 0)
            g_dev.cus = p.multiProcessorCount;
        g_dev.init = true;
    }
    return g_dev.cus;
}

bool shape_ok(int batch_size, int num_rois, int height, int width, int channels, int pooled_height,
              int pooled_width)
{
    if (batch_size <= 0 || num_rois < 0 || height <= 0 || width <= 0 || channels <= 0 ||
        pooled_height <= 0 || pooled_width <= 0)
        return false;
    // per-image pixel-major offsets are 32-bit; bins per roi are int
    if ((long)height * width * round_up4(channels) >= (1L << 31)) return false;
    if ((long)pooled_height * pooled_width >= (1L << 31)) return false;
    return true;
}

// grid for the tiled kernels: one wave per block, up to 16 waves per CU, a
// multiple of lcm(nchunks, 8) so that blockIdx % nchunks is also stable per XCD.
int tiled_grid(long items, int nchunks)
{
    long want = items * nchunks;
    const long cap = (long)num_cus() * 16;
    if (want > cap) want = cap;
    long unit = nchunks;
    while (unit % 8) unit += nchunks;  // lcm(nchunks, 8) for nchunks <= ...; bounded by 8*nchunks
    long g = (want + unit - 1) / unit * unit;
    if (g < nchunks) g = nchunks;
    return (int)g;
}

void direct_grid(int num_rois, int NB, int channels, dim3& grid, int& cslab)
{
    const long threads = (long)num_rois * NB;
    const int bx = ceil_div(threads, 256);
    // enough blocks to fill the chip a few times over, but keep slabs >= 4 channels
    int slabs = 1;
    const long target = (long)num_cus() * 8;
    while ((long)bx * slabs < target && channels / (slabs * 2) >= 4) slabs *= 2;
    cslab = ceil_div(channels, slabs);
    grid = dim3(bx, ceil_div(channels, cslab), 1);
}

struct FwdWorkspace {
    Affine* aff;
    float* pm;
    size_t bytes;
};

FwdWorkspace carve_fwd(void* ws, int batch_size, int channels, int height, int width, int num_rois,
                       int layout)
{
    FwdWorkspace w;
    const size_t aff_bytes = align_up((size_t)(num_rois > 0 ? num_rois : 1) * sizeof(Affine), 256);
    const size_t pm_bytes =
        layout == RROI_LAYOUT_NHWC && channels % 4 == 0
            ? 0
            : align_up((size_t)batch_size * height * width * round_up4(channels) * sizeof(float), 256);
    w.aff = reinterpret_cast<Affine*>(ws);
    w.pm = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + aff_bytes);
    w.bytes = aff_bytes + pm_bytes;
    return w;
}

// AUTO: the tiled path pays one pass over the whole map (read + write B*C*H*W);
// the direct path pays ~4 uncoalesced taps per output.  Tiled wins once the
// output is a few times larger than the map.
bool pick_tiled(int batch_size, int channels, int height, int width, int num_rois, int NB)
{
    const double out_elems = (double)num_rois * channels * NB;
    const double map_elems = (double)batch_size * channels * height * width;
    return out_elems >= 2.0 * map_elems;
}

}  // namespace

// ====================================================================================
extern "C" {

const char* rroi_align_hip_version(void) { return "rroi_align_hip 0.1.0 gfx950"; }

size_t rroi_align_forward_workspace_bytes(int batch_size, int channels, int height, int width,
                                          int num_rois, int feature_layout)
{
    if (batch_size <= 0 || channels <= 0 || height <= 0 || width <= 0 || num_rois < 0) return 0;
    return carve_fwd(nullptr, batch_size, channels, height, width, num_rois, feature_layout).bytes;
}

size_t rroi_align_backward_workspace_bytes(int batch_size, int channels, int height, int width,
                                           int num_rois)
{
    if (batch_size <= 0 || channels <= 0 || height <= 0 || width <= 0 || num_rois < 0) return 0;
    return carve_fwd(nullptr, batch_size, channels, height, width, num_rois, RROI_LAYOUT_NCHW).bytes;
}

int rroi_align_forward_hip(const float* features, int feature_layout, float spatial_scale,
                           int batch_size, int num_rois, int height, int width, int channels,
                           int pooled_height, int pooled_width, const float* rois,
                           float* top_data, void* workspace, size_t workspace_bytes, int path,
                           void* stream_)
{
    return rroi_align_forward_stages_hip(features, feature_layout, spatial_scale, batch_size,
                                         num_rois, height, width, channels, pooled_height,
                                         pooled_width, rois, top_data, workspace, workspace_bytes,
                                         path, RROI_STAGE_ALL, stream_);
}

int rroi_align_forward_stages_hip(const float* features, int feature_layout, float spatial_scale,
                                  int batch_size, int num_rois, int height, int width,
                                  int channels, int pooled_height, int pooled_width,
                                  const float* rois, float* top_data, void* workspace,
                                  size_t workspace_bytes, int path, int stages, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if ((stages & ~RROI_STAGE_ALL) || stages == 0) return 0;
    if (!shape_ok(batch_size, num_rois, height, width, channels, pooled_height, pooled_width))
        return 0;
    if (feature_layout != RROI_LAYOUT_NCHW && feature_layout != RROI_LAYOUT_NHWC) return 0;
    if (path != RROI_PATH_AUTO && path != RROI_PATH_DIRECT && path != RROI_PATH_TILED) return 0;
    if (num_rois == 0) return 1;
    if (!features || !rois || !top_data) return 0;
    const int NB = pooled_height * pooled_width;

    bool tiled;
    if (path == RROI_PATH_AUTO)
        tiled = feature_layout == RROI_LAYOUT_NHWC ||
                pick_tiled(batch_size, channels, height, width, num_rois, NB);
    else
        tiled = path == RROI_PATH_TILED;
    if (!tiled && feature_layout != RROI_LAYOUT_NCHW) return 0;  // direct path reads NCHW only

    if (!tiled) {
        if (!(stages & RROI_STAGE_GATHER)) return 1;  // the direct path has no prologue
        dim3 grid;
        int cslab;
        direct_grid(num_rois, NB, channels, grid, cslab);
        hipLaunchKernelGGL(rroi_fwd_direct_kernel, grid, dim3(256), 0, stream, features, rois,
                           top_data, (float*)nullptr, (float*)nullptr, num_rois, channels, height,
                           width, pooled_height, pooled_width, spatial_scale, batch_size, cslab);
        return launch_status();
    }

    const FwdWorkspace ws =
        carve_fwd(workspace, batch_size, channels, height, width, num_rois, feature_layout);
    if (!workspace || workspace_bytes < ws.bytes) return 0;
    const int Cs = round_up4(channels);
    const int HW = height * width;
    const bool zero_copy = feature_layout == RROI_LAYOUT_NHWC && channels % 4 == 0;
    const float* pm = zero_copy ? features : ws.pm;

    // prologue: relayout + affine table in one launch
    if (stages & RROI_STAGE_PROLOGUE) {
        const int ptiles = ceil_div(HW, 64), ctiles = ceil_div(Cs, 64);
        int relayout_blocks = zero_copy ? 0 : ptiles * ctiles * batch_size;
        const int aff_blocks = ceil_div(num_rois, 256);
        if (feature_layout == RROI_LAYOUT_NHWC && !zero_copy) {
            // channels-last storage with C % 4 != 0: repack by treating it as a strided copy
            // (rare: C = 3 image inputs).  Handled by a plain 2D memcpy.
            hipError_t e = hipMemcpy2DAsync(ws.pm, (size_t)Cs * sizeof(float), features,
                                            (size_t)channels * sizeof(float),
                                            (size_t)channels * sizeof(float),
                                            (size_t)batch_size * HW, hipMemcpyDeviceToDevice, stream);
            if (e != hipSuccess) return status_of(e);
            relayout_blocks = 0;
        }
        hipLaunchKernelGGL(rroi_prologue_kernel, dim3(relayout_blocks + aff_blocks), dim3(256), 0,
                           stream, features, ws.pm, channels, Cs, HW, ptiles, ctiles,
                           relayout_blocks, rois, num_rois, pooled_height, spatial_scale, ws.aff);
        const int st = launch_status();
        if (st != 1) return st;
    }
    if (stages & RROI_STAGE_GATHER) {
        const int nchunks = ceil_div(channels, kChunk);
        const int ntiles = ceil_div(NB, kTileBins);
        const int grid = tiled_grid((long)num_rois * ntiles, nchunks);
        if (NB % 4 == 0)
            hipLaunchKernelGGL(rroi_fwd_tiled_kernel<true>, dim3(grid), dim3(kWave), 0, stream, pm,
                               ws.aff, top_data, num_rois, channels, Cs, height, width,
                               pooled_height, pooled_width, batch_size, nchunks, ntiles);
        else
            hipLaunchKernelGGL(rroi_fwd_tiled_kernel<false>, dim3(grid), dim3(kWave), 0, stream, pm,
                               ws.aff, top_data, num_rois, channels, Cs, height, width,
                               pooled_height, pooled_width, batch_size, nchunks, ntiles);
    }
    return launch_status();
}

int rroi_align_backward_hip(const float* top_diff, float spatial_scale, int batch_size,
                            int num_rois, int height, int width, int channels,
                            int pooled_height, int pooled_width, const float* rois,
                            float* bottom_diff, void* workspace, size_t workspace_bytes, int path,
                            void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!shape_ok(batch_size, num_rois, height, width, channels, pooled_height, pooled_width))
        return 0;
    if (path != RROI_PATH_AUTO && path != RROI_PATH_DIRECT && path != RROI_PATH_TILED) return 0;
    if (!bottom_diff) return 0;
    const int NB = pooled_height * pooled_width;
    const size_t HW = (size_t)height * width;
    const size_t in_bytes = (size_t)batch_size * channels * HW * sizeof(float);
    if (num_rois == 0) return status_of(hipMemsetAsync(bottom_diff, 0, in_bytes, stream));
    if (!top_diff || !rois) return 0;

    const bool tiled = path == RROI_PATH_AUTO
                           ? pick_tiled(batch_size, channels, height, width, num_rois, NB)
                           : path == RROI_PATH_TILED;
    if (!tiled) {
        hipError_t e = hipMemsetAsync(bottom_diff, 0, in_bytes, stream);
        if (e != hipSuccess) return status_of(e);
        dim3 grid;
        int cslab;
        direct_grid(num_rois, NB, channels, grid, cslab);
        hipLaunchKernelGGL(rroi_bwd_direct_kernel, grid, dim3(256), 0, stream, top_diff, rois,
                           bottom_diff, num_rois, channels, height, width, pooled_height,
                           pooled_width, spatial_scale, batch_size, cslab);
        return launch_status();
    }

    const FwdWorkspace ws =
        carve_fwd(workspace, batch_size, channels, height, width, num_rois, RROI_LAYOUT_NCHW);
    if (!workspace || workspace_bytes < ws.bytes) return 0;
    const int Cs = round_up4(channels);
    hipError_t e = hipMemsetAsync(ws.pm, 0, (size_t)batch_size * HW * Cs * sizeof(float), stream);
    if (e != hipSuccess) return status_of(e);
    hipLaunchKernelGGL(rroi_affine_kernel, dim3(ceil_div(num_rois, 256)), dim3(256), 0, stream,
                       rois, num_rois, pooled_height, spatial_scale, ws.aff);
    int st = launch_status();
    if (st != 1) return st;
    const int nchunks = ceil_div(channels, kChunk);
    const int ntiles = ceil_div(NB, kTileBins);
    const int grid = tiled_grid((long)num_rois * ntiles, nchunks);
    if (NB % 4 == 0)
        hipLaunchKernelGGL(rroi_bwd_tiled_kernel<true>, dim3(grid), dim3(kWave), 0, stream,
                           top_diff, ws.aff, ws.pm, num_rois, channels, Cs, height, width,
                           pooled_height, pooled_width, batch_size, nchunks, ntiles);
    else
        hipLaunchKernelGGL(rroi_bwd_tiled_kernel<false>, dim3(grid), dim3(kWave), 0, stream,
                           top_diff, ws.aff, ws.pm, num_rois, channels, Cs, height, width,
                           pooled_height, pooled_width, batch_size, nchunks, ntiles);
    st = launch_status();
    if (st != 1) return st;
    const int ptiles = ceil_div((long)HW, 64), ctiles = ceil_div(Cs, 64);
    hipLaunchKernelGGL(rroi_pm_to_nchw_kernel, dim3(ptiles * ctiles * batch_size), dim3(256), 0,
                       stream, ws.pm, bottom_diff, channels, Cs, (int)HW, ptiles, ctiles);
    return launch_status();
}

int rroi_align_bin_centres_hip(float spatial_scale, int num_rois, int height, int width,
                               int pooled_height, int pooled_width, const float* rois,
                               float* geom, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (num_rois < 0 || height <= 0 || width <= 0 || pooled_height <= 0 || pooled_width <= 0)
        return 0;
    if (num_rois == 0) return 1;
    if (!rois || !geom) return 0;
    const long threads = (long)num_rois * pooled_height * pooled_width;
    hipLaunchKernelGGL(rroi_bin_centres_kernel, dim3(ceil_div(threads, 256)), dim3(256), 0, stream,
                       rois, geom, num_rois, height, width, pooled_height, pooled_width,
                       spatial_scale);
    return launch_status();
}

int rroi_align_sincos_probe_hip(const float* angle_deg, int n, float* out, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (n < 0) return 0;
    if (n == 0) return 1;
    if (!angle_deg || !out) return 0;
    hipLaunchKernelGGL(rroi_sincos_probe_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, stream,
                       angle_deg, n, out);
    return launch_status();
}

// ---- the reference's launcher ABI (rroi_align_kernel.h:8-18) ------------------------
int RROIAlignForwardLaucher(const float* bottom_data, const float spatial_scale,
                            const int num_rois, const int height, const int width,
                            const int channels, const int pooled_height, const int pooled_width,
                            const float* bottom_rois, float* top_data, float* con_idx_x,
                            float* con_idx_y, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!shape_ok(1, num_rois, height, width, channels, pooled_height, pooled_width)) return 0;
    if (num_rois == 0) return 1;
    if (!bottom_data || !bottom_rois || !top_data) return 0;
    dim3 grid;
    int cslab;
    direct_grid(num_rois, pooled_height * pooled_width, channels, grid, cslab);
    hipLaunchKernelGGL(rroi_fwd_direct_kernel, grid, dim3(256), 0, stream, bottom_data, bottom_rois,
                       top_data, con_idx_x, con_idx_y, num_rois, channels, height, width,
                       pooled_height, pooled_width, spatial_scale, /*batch_size unknown*/ -1,
                       cslab);
    return launch_status();
}

int RROIAlignBackwardLaucher(const float* top_diff, const float spatial_scale,
                             const int batch_size, const int num_rois, const int height,
                             const int width, const int channels, const int pooled_height,
                             const int pooled_width, const float* bottom_rois,
                             float* bottom_diff, const float* con_idx_x, const float* con_idx_y,
                             void* stream_)
{
    (void)spatial_scale;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!shape_ok(batch_size, num_rois, height, width, channels, pooled_height, pooled_width))
        return 0;
    if (num_rois == 0) return 1;
    if (!top_diff || !bottom_rois || !bottom_diff || !con_idx_x || !con_idx_y) return 0;
    const long nthreads = (long)num_rois * pooled_height * pooled_width * channels;
    long blocks = (nthreads + 255) / 256;
    const long cap = (long)num_cus() * 32;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(rroi_bwd_literal_kernel, dim3((unsigned)blocks), dim3(256), 0, stream,
                       top_diff, con_idx_x, con_idx_y, bottom_rois, bottom_diff, nthreads, channels,
                       height, width, pooled_height, pooled_width);
    return launch_status();
}

}  // extern "C"
