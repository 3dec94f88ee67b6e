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
// NCHW the channels of a pixel are H*W apart (one cache line per channel), so
// the hot path first relays the map out CHUNK-MAJOR: (B, C/32, H*W+1, 32) --
// 32 channels of a pixel are one 128-byte line, consecutive pixels of a chunk
// are consecutive lines (all L2 channels of an XCD are used evenly), and every
// (image, chunk) slice ends in a zero pixel that invalid taps point at.  Each
// wave then owns a [32 channel] x [64 bin] output tile: lanes = 8 bins x 8
// channel-quads fetch taps as 16-byte loads, blend in the reference's order,
// transpose through a wave-private LDS tile and stream the tile out as full
// 256-byte row segments of the (R,C,PH,PW) tensor.  Channel chunk k is handled
// by blocks with blockIdx % nchunks == k, i.e. (8 chunks at C=256) by one XCD,
// whose 4 MiB L2 then holds exactly its 3.2 MB slice of the map.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rroi_align_hip.h"

#pragma clang fp contract(off)

namespace {

constexpr int kWave = 64;
constexpr int kChunk = 32;     // channels per slice / work item (8 lanes x 16 B = one 128 B line)
constexpr int kTileBins = 64;  // bins per work item              (one 256 B output row segment)
constexpr int kTStride = 68;   // LDS tile row stride in dwords: 4*odd -> writes <=2-way, b128 reads aligned
constexpr int kQuads = kChunk / 4;
constexpr int kBinsPerIter = kWave / kQuads;      // 8
constexpr int kIters = kTileBins / kBinsPerIter;  // 8
constexpr unsigned kLineBytes = kChunk * 4;       // 128

typedef float v4f __attribute__((ext_vector_type(4)));

struct Affine {  // kernel.cu:78-84 (M), :68 (roi_pooled_width), :60 (roi_batch_ind)
    float m00, m01, m02, m10, m11, m12, rpw;
    int batch;
};
static_assert(sizeof(Affine) == 32, "Affine is read as two 16-byte scalars");

struct FastDiv {  // Granlund-Montgomery unsigned division by an invariant, exact for all 32-bit x
    unsigned m, sh1, sh2;
};

__device__ __forceinline__ unsigned fdiv(unsigned x, const FastDiv& f)
{
    const unsigned t = __umulhi(f.m, x);
    return (t + ((x - t) >> f.sh1)) >> f.sh2;
}

// Ordering point for LDS traffic between the lanes of ONE wave (the tiled kernels run one
// wave per workgroup).  LDS instructions of a wave execute in issue order, so a later
// ds_read sees an earlier ds_write of another lane without any wait; all that is needed is
// to stop the compiler from reordering them.  (__syncthreads() would also emit
// s_waitcnt vmcnt(0), i.e. drain the tile's global stores.)
__device__ __forceinline__ void lds_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float as_f(unsigned u) { return __uint_as_float(u); }
__device__ __forceinline__ unsigned as_u(float f) { return __float_as_uint(f); }

// (int)x as the reference's device code performs it (cvt.rzi.s32.f32): truncating, saturating,
// NaN -> 0.  That is exactly v_cvt_i32_f32; it is emitted directly because a C cast leaves the
// out-of-range cases undefined, and the equivalent compare chain costs 12 instructions and three
// branches per conversion in the gather kernel's geometry phase.
__device__ __forceinline__ int f2i_sat(float x)
{
    int r;
    asm("v_cvt_i32_f32_e32 %0, %1" : "=v"(r) : "v"(x));
    return r;
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

// Taps are whole pixels: x1 = x0 + dx, y1 = y0 + dy with dx, dy in {0, 1} (bin
// centres are multiples of 0.5), so when dx == 0 the reference's "right" taps
// ARE its left taps (same pixel, same validity) and need no load of their own.
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
    // "issue a load for this tap" (tiled forward)
    kL0 = 1u << 16, kL1 = 1u << 17, kL2 = 1u << 18, kL3 = 1u << 19,
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
    const float ux = 1.0f - rx, uy = 1.0f - ry;
    wlt = ux * uy;
    wrt = rx * uy;
    wrb = rx * ry;
    wlb = ux * ry;
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

// Where the sampled map lives for the tiled kernels.  A "slice" is the 32-channel
// chunk k of image b; pixel p of a slice starts at slice_base + p * px_bytes.
//   chunk-major copy  : px_bytes = 128, row pitch Wp >= W pixels, chunk_stride = (H*Wp+1)*32,
//                       img_stride = nchunks*chunk_stride.  Wp is chosen so that vertically
//                       adjacent pixels do not fall on the same L2 channel (W = 160 lines is
//                       a multiple of the 16-channel interleave: a 90-degree ROI would queue
//                       all 8 lines of a load instruction on one channel).
//   channels-last user tensor (zero copy): px_bytes = C*4, chunk_stride = 32, img_stride = HW*C
struct SliceLayout {
    unsigned px_bytes;
    unsigned row_bytes;     // pitch of one map row inside a slice (chunk-major rows are padded)
    unsigned slice_bytes;   // extent of one slice from its base (range of the buffer descriptor)
    unsigned chunk_stride;  // floats
    unsigned img_stride;    // floats  (fits: shape_ok bounds it)
};

// Buffer addressing: every tap load and every output store goes through a raw buffer
// descriptor (base, num_records) whose range check does the predication in hardware -- a lane
// whose byte offset is >= num_records reads zeros / stores nothing and costs no memory access.
// An invalid tap (kernel.cu:116-126 yields 0.0 for it), a tap the bin does not need, a channel
// quad beyond C and a bin beyond PH*PW are all just "offset = kOOB".  The hot loop therefore
// has no branches and no exec masking, and the compiler's s_waitcnt counts are exact.
typedef unsigned v4u __attribute__((ext_vector_type(4)));
constexpr unsigned kOOB = 0x80000000u;          // > any slice / tile size (shape_ok: < 2 GiB)
constexpr unsigned kRsrcWord3 = 0x00020000u;    // raw buffer, 32-bit data format (gfx9 family)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, kRsrcWord3);
}
__device__ __forceinline__ v4f buf_load(__amdgpu_buffer_rsrc_t r, unsigned byte_off)
{
    return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}
// Cache policy of the output stream (gfx940-family bits: 1 = sc0, 2 = nt, 16 = sc1).  The
// 256 MiB of crops must not displace the 3.3 MB map slice from the XCD's 4 MiB L2: with plain
// stores every written line is kept in L2 and 47 % of the tap reads missed L2 (gather kernel
// 59 us).  nt (streaming) and sc1 (write-through, line dropped) both avoid that.  A store-only
// kernel runs at 49.5 us with nt and 40 us with sc1 -- the nt write path is narrower -- but over
// a whole step (prologue + gather, the bench's unit) write-through costs more than it saves: it
// displaces the feature map the next prologue reads (57.1 us with nt, 63.6 us with sc0 sc1).
// A MIX wins on both counts: one of a tile's eight stores write-through, seven nt -- kernel
// 46.9 us, step 54.5 us (two of eight: 55.9; one of sixteen: 56.8; the position in the tile is
// irrelevant).  profiles/r01_micro_store_policy.txt.
constexpr int kMinorAux = 17;
template <int AUX>
__device__ __forceinline__ void buf_store(__amdgpu_buffer_rsrc_t r, unsigned byte_off, v4f v)
{
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), r, byte_off, 0, AUX);
}
template <int AUX>
__device__ __forceinline__ void buf_store1(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float v)
{
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, byte_off, 0, AUX);
}

// ------------------------------------------------------------------------------------
// K0: forward prologue, one launch:
//   blocks [0, relayout_blocks)      NCHW -> chunk-major: a [32 ch] x [128 px] tile goes
//                                    through LDS; reads are 512 B runs of a channel row,
//                                    writes are one contiguous 16 KiB run of the slice;
//   then zero_blocks                 the zero pixel that ends every slice;
//   then the rest                    per-ROI affine table (R x 32 B).
// ------------------------------------------------------------------------------------
constexpr int kRelayoutPx = 128;

// The relayout proper, shared by the forward prologue (feature map) and the backward (top_diff,
// R "images" of PH x PW "pixels").  MASK: image b is ROI b and pixels (ph, pw) with
// pw > roi_pooled_width (or every pixel of a ROI with an invalid batch index) are bins the
// forward masks -- nothing reads them again, so they are neither loaded nor written.
constexpr int kTP = kRelayoutPx + 4;

template <int AUX, bool MASK>
__device__ __forceinline__ void relayout_run(float* __restrict__ T, const float* __restrict__ nchw,
                                               float* __restrict__ cm, int C, int HW, int width, int pitch,
                                               FastDiv div_w, int nchunks, int ptiles, int first_tile,
                                               int tile_stride, int relayout_tiles,
                                               const Affine* __restrict__ mask_aff, int mask_batches)
{
    // [32 ch][128 px] tile, 132-float pitch (16-byte aligned rows for the b128 writes); the
    // pixel index of rows 8m..8m+7 is XORed with 4m so that the transposed ds_read_b32 of
    // phase 2 (8 channel quads x 4 pixels per 32-lane group) hits 32 different banks.
    const int tid = threadIdx.x;
    const size_t zp_index = (size_t)(HW / width) * pitch;  // pixel index of the zero pixel
    const size_t slice_stride = (zp_index + 1) * kChunk;
    const int lane = tid & 63, w = tid >> 6;
    // phase 1 mapping: lane -> 4 consecutive pixels (x4) of channel row (csub); a wave
    // instruction reads two 512-byte runs.  phase 2 mapping: lane -> (channel quad, pixel).
    const int x4 = lane & 31, csub = lane >> 5;
    const int cq = lane & 7, pl = lane >> 3;
    // rows of 16-byte aligned float4 (p0 is a multiple of 128): needs HW % 4 == 0 and an aligned base
    const bool vec_ok = (HW & 3) == 0 && (reinterpret_cast<uintptr_t>(nchw) & 15) == 0;

    // (MASK) highest live pooled column of image b: pw <= rpw  <=>  pw <= floor(rpw) for integer pw
    auto live_limit = [&](int b) -> float {
        const Affine A = mask_aff[b];
        return (A.batch >= 0 && A.batch < mask_batches) ? A.rpw : -1.0f;
    };

    v4f r[4];
    auto load_tile = [&](int tile) {
        // chunk index fastest: with the grid a multiple of nchunks a block always relays out
        // the same chunk, i.e. (8 chunks, blocks dealt round-robin to the 8 XCDs) slice k is
        // written through the L2 of the XCD whose gather blocks will read it
        const int k = tile % nchunks;
        const int pt = (tile / nchunks) % ptiles;
        const int b = tile / (ptiles * nchunks);
        const int p0 = pt * kRelayoutPx, c0 = k * kChunk;
        const float* src = nchw + ((size_t)b * C + c0) * HW + p0;
        const int p = 4 * x4;
        bool live = true;
        if (MASK) {
            // the four pixels of this lane are dead when the first one is (same row), or when the
            // run starts in a dead tail and ends in the next row's live head: keep it then
            const unsigned gp = (unsigned)(p0 + p);
            const unsigned y = fdiv(gp, div_w);
            const unsigned x = gp - y * (unsigned)width;
            const float lim = live_limit(b);
            live = !((float)x > lim) || x + 3u >= (unsigned)width;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = w * 8 + i * 2 + csub;
            v4f v = {0.f, 0.f, 0.f, 0.f};
            if (c0 + c < C && live) {
                const float* sp = src + (size_t)c * HW + p;
                if (vec_ok && p0 + p + 3 < HW) {
                    v = *reinterpret_cast<const v4f*>(sp);
                } else {
                    if (p0 + p + 0 < HW) v.x = sp[0];
                    if (p0 + p + 1 < HW) v.y = sp[1];
                    if (p0 + p + 2 < HW) v.z = sp[2];
                    if (p0 + p + 3 < HW) v.w = sp[3];
                }
            }
            r[i] = v;
        }
    };
    // grid-stride over tiles, software-pipelined: the loads of the next tile are in flight
    // while the current tile goes through LDS and out to the chunk-major copy
    int tile = first_tile;
    if (tile < relayout_tiles) load_tile(tile);
    while (tile < relayout_tiles) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = w * 8 + i * 2 + csub;
            *reinterpret_cast<v4f*>(T + c * kTP + ((4 * x4) ^ ((c >> 3) * 4))) = r[i];
        }
        __syncthreads();
        const int cur = tile;
        tile += tile_stride;
        if (tile < relayout_tiles) load_tile(tile);
        {
            const int k = cur % nchunks;
            const int pt = (cur / nchunks) % ptiles;
            const int b = cur / (ptiles * nchunks);
            const int p0 = pt * kRelayoutPx;
            float* dst = cm + ((size_t)b * nchunks + k) * slice_stride;
            const float lim = MASK ? live_limit(b) : 0.0f;
            // wave w writes pixels 32w..32w+31: per instruction 8 pixels x 128 B = 1 KiB contiguous
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int p = w * 32 + j * 8 + pl;
                const float* tr = T + (cq * 4) * kTP + (p ^ ((cq >> 1) * 4));
                v4f v = {tr[0], tr[kTP], tr[2 * kTP], tr[3 * kTP]};
                const unsigned gp = (unsigned)(p0 + p);
                const unsigned y = fdiv(gp, div_w);
                const unsigned x = gp - y * (unsigned)width;
                const size_t pix = (size_t)y * pitch + x;
                if (p0 + p < HW && !(MASK && (float)x > lim)) {
                    if (AUX == 0) {
                        *reinterpret_cast<v4f*>(dst + pix * kChunk + cq * 4) = v;
                    } else {
                        const __amdgpu_buffer_rsrc_t ws = make_rsrc(dst, (unsigned)(slice_stride * 4));
                        buf_store<AUX>(ws, (unsigned)((pix * kChunk + cq * 4) * 4), v);
                    }
                }
            }
        }
        __syncthreads();
    }
}

template <int AUX>
__global__ __launch_bounds__(256) void rroi_prologue_kernel(
    const float* __restrict__ nchw, float* __restrict__ cm, int C, int HW, int width, int pitch,
    FastDiv div_w, int nchunks, int ptiles, int relayout_blocks, int relayout_tiles, int zero_blocks,
    int batch_size, const float* __restrict__ rois, int num_rois, int pooled_height,
    float spatial_scale, Affine* __restrict__ aff)
{
    __shared__ __attribute__((aligned(16))) float T[kChunk * kTP];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= relayout_blocks + zero_blocks) {
        const int n = ((int)blockIdx.x - relayout_blocks - zero_blocks) * 256 + tid;
        if (n < num_rois) aff[n] = make_affine(rois + (size_t)n * 6, pooled_height, spatial_scale);
        return;
    }
    if ((int)blockIdx.x >= relayout_blocks) {
        const size_t zp_index = (size_t)(HW / width) * pitch;  // pixel index of the zero pixel
        const size_t slice_stride = (zp_index + 1) * kChunk;
        const int i = ((int)blockIdx.x - relayout_blocks) * 256 + tid;  // (slice, channel-in-chunk)
        if (i < batch_size * nchunks * kChunk)
            cm[(size_t)(i / kChunk) * slice_stride + zp_index * kChunk + (i % kChunk)] = 0.0f;
        return;
    }
    relayout_run<AUX, false>(T, nchw, cm, C, HW, width, pitch, div_w, nchunks, ptiles, (int)blockIdx.x,
                               relayout_blocks, relayout_tiles, nullptr, 0);
}

__global__ void rroi_affine_kernel(const float* __restrict__ rois, int num_rois, int pooled_height,
                                   float spatial_scale, Affine* __restrict__ aff)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < num_rois) aff[n] = make_affine(rois + (size_t)n * 6, pooled_height, spatial_scale);
}

// ------------------------------------------------------------------------------------
// K1: the hot kernel.  One wave per block; block -> channel chunk k = blockIdx %
// nchunks (XCD affinity) and a grid-stride loop over (roi, 64-bin tile) items.
//   phase A  lane = bin: geometry -> one 16-byte tap record per bin in LDS, sorted by
//            class (bins with <= 2 distinct taps, bins with 4); an invalid tap is the
//            out-of-range offset kOOB, which the buffer descriptor turns into 0.0.
//   phase B  lane = (bin b of 8, channel quad q of 8): 2 or 4 buffer loads per group
//            of 8 bins, blend, transpose through LDS; depth-2 software pipeline.
//   phase C  the [32 ch][64 bin] tile leaves LDS as 16-byte streaming stores, 256 B
//            per row.
// The three phases of consecutive items are interleaved around the store burst, see the
// loop at the end.
// ------------------------------------------------------------------------------------
template <bool VEC_STORE, int AUX>
__global__ __launch_bounds__(kWave) void rroi_fwd_tiled_kernel(
    const float* __restrict__ map, const Affine* __restrict__ aff, float* __restrict__ out,
    int num_rois, int C, int height, int width, int pooled_width, int NB, int batch_size,
    int nchunks, int ntiles, SliceLayout lay, FastDiv div_tiles, FastDiv div_pw, int dbg)
{
    // A tile's bins are processed in CLASS-SORTED groups of 8, because the texture addresser
    // charges 16 cycles for every dwordx4 wave instruction whatever the number of lanes that
    // really fetch (measured: 16.1 / 15.6 / 15.2 clk with 0 / 50 / 87 % of the lanes out of
    // range).  Issuing all 4 taps for all 64 bins costs 32 load instructions per tile although
    // only ~1.3 taps per bin are distinct pixels.  Sorted:
    //   LO  bins with at most two distinct taps (lt, and rt OR lb): 2 loads per group;
    //   HI  bins with four distinct taps (dx and dy):                4 loads per group;
    //   masked bins (pw > roi_pooled_width) are in no group -- phase C writes their zeros.
    // Typical tile: 5 LO + 2 HI groups = 18 load instructions instead of 32.
    constexpr int kMaxGroups = kIters + 2;  // two classes, each padded to a multiple of 8
    constexpr unsigned kPadPos = kTileBins;  // "bin position" of a padding record
    // T: rows 0..31 are the tile; the tail absorbs the writes of padding records (4 rows at the
    // tile's pitch, 32 columns).  LDS is granted in 1280-byte granules on gfx950: the block must
    // stay <= 12800 B for 12 waves per CU.
    __shared__ __attribute__((aligned(16))) float T[kChunk * kTStride + 3 * kTStride + 32];
    // tap records of two items: item i+1 is sampled out of one set while the other is being
    // built for item i+2
    constexpr int kRecs = kMaxGroups * kBinsPerIter;
    __shared__ __attribute__((aligned(16))) uint4 Gbuf[2 * kRecs];
    __shared__ unsigned char HPbuf[2 * kRecs];

    const unsigned lane = threadIdx.x;
    const unsigned k = blockIdx.x % (unsigned)nchunks;
    const unsigned slot = blockIdx.x / (unsigned)nchunks;
    const unsigned nslots = gridDim.x / (unsigned)nchunks;
    const unsigned items = (unsigned)num_rois * (unsigned)ntiles;
    const unsigned px_bytes = lay.px_bytes;
    const unsigned row_bytes = lay.row_bytes;

    // lane = q + 8*b: the 8 lanes that fetch the 8 channel quads of ONE pixel (one 128-byte
    // line) are consecutive, so the texture addresser merges them into two 64-byte
    // accesses.  (With the quads strided over the wave every lane costs its own access:
    // measured 43 vs 16 TCP accesses per load instruction.)
    const unsigned q = lane & (kQuads - 1), b = lane >> 3;
    // a channel quad wholly beyond C never loads (its rows are not stored either)
    const unsigned q_bytes = ((dbg & 2) || k * kChunk + q * 4 >= (unsigned)C) ? kOOB : q * 16u;
    // LDS tile: row r = channel, 68-dword pitch; the column of rows 8m..8m+7 is XORed with
    // 4*m so that the 32 lanes of a store group (8 quads x 4 bins) spread over the banks
    // while rows stay 16-byte aligned for the ds_read_b128 of phase C.
    const unsigned wswz = (q >> 1) * 4u;  // rows 4q..4q+3 -> m = q >> 1
    const unsigned col = (lane & 15) * 4, row0 = lane >> 4;
    const unsigned chans_here = min((unsigned)kChunk, (unsigned)C - k * kChunk);  // rows of this chunk < C
    const v4f z4 = {0.f, 0.f, 0.f, 0.f};

    unsigned g_lo = 0, g_hi = 0;          // groups of the current item (wave-uniform)
    unsigned long long act_mask = 0;      // bins of the current item that are in a group

    // phase A of one item: lane = bin, geometry -> sorted 16-byte tap records in LDS:
    //   LO: {off_lt, off_2nd, w_lt, bin position}      (w_2nd = 1 - w_lt, see blend_lo)
    //   HI: {off_lt, off_rt, off_lb, off_rb}, bin position in HP[]   (all four weights are 1/4)
    // Offsets are byte offsets into the slice; kOOB reads as 0.0, which is what
    // kernel.cu:116-126 substitutes for a tap outside the map.
    auto geometry = [&](const Affine& A, unsigned t, unsigned p, unsigned& n_lo_groups,
                        unsigned& n_hi_groups, unsigned long long& amask) {
        uint4* const G = Gbuf + p * kRecs;
        unsigned char* const HP = HPbuf + p * kRecs;
        const bool batch_ok = A.batch >= 0 && A.batch < batch_size;
        const unsigned bin = t * kTileBins + lane;
        const unsigned ph = fdiv(bin, div_pw);
        const unsigned pw = bin - ph * (unsigned)pooled_width;
        float bcx, bcy;
        bool active = bin_centre(A, (int)ph, (int)pw, height, width, bcx, bcy);
        active = active && bin < (unsigned)NB && batch_ok;
        const float fx = floorf(bcx), fy = floorf(bcy);
        const int x0 = f2i_sat(fx), x1 = f2i_sat(ceilf(bcx));
        const int y0 = f2i_sat(fy), y1 = f2i_sat(ceilf(bcy));
        const bool x0ok = x0 > 0 && x0 < width, x1ok = x1 > 0 && x1 < width;
        const bool y0ok = y0 > 0 && y0 < height, y1ok = y1 > 0 && y1 < height;
        const bool dx = active && x1 != x0, dy = active && y1 != y0;
        // kernel.cu:116-126 validity; a tap that aliases lt (dx == 0 / dy == 0) is not loaded
        const unsigned o00 = (unsigned)y0 * row_bytes + (unsigned)x0 * px_bytes;
        const unsigned o_lt = (active && y0ok && x0ok) ? o00 : kOOB;
        const unsigned o_rt = (dx && y0ok && x1ok) ? o00 + px_bytes : kOOB;
        const unsigned o_lb = (dy && y1ok && x0ok) ? o00 + row_bytes : kOOB;
        const unsigned o_rb = (dx && dy && y1ok && x1ok) ? o00 + row_bytes + px_bytes : kOOB;
        const bool hi = dx && dy, lo = active && !hi;
        const unsigned long long m_lo = __ballot(lo), m_hi = __ballot(hi);
        const unsigned n_lo = __popcll(m_lo), n_hi = __popcll(m_hi);
        // at least one LO group (all padding if need be): its loads are issued unconditionally,
        // one item ahead, before the previous item's stores
        n_lo_groups = n_lo ? (n_lo + kBinsPerIter - 1) / kBinsPerIter : 1u;
        n_hi_groups = (n_hi + kBinsPerIter - 1) / kBinsPerIter;
        amask = m_lo | m_hi;
        const unsigned hi_base = n_lo_groups * kBinsPerIter;
        const unsigned long long below = (1ull << lane) - 1ull;
        const unsigned idx = lo ? __popcll(m_lo & below) : hi_base + __popcll(m_hi & below);
        const float rx = bcx - fx, ry = bcy - fy;
        const float wlt = (1.0f - rx) * (1.0f - ry);  // kernel.cu:131
        if (active) {
            // LO: the one other distinct tap is rt (dx) or lb (dy); neither -> kOOB, weight 0
            G[idx] = make_uint4(o_lt, dx ? o_rt : o_lb, hi ? o_lb : as_u(wlt), hi ? o_rb : lane);
            HP[idx] = (unsigned char)lane;
        }
        // pad both classes to whole groups with records that load nothing and store nowhere
        const unsigned pad_lo = hi_base - n_lo, pad_hi = n_hi_groups * kBinsPerIter - n_hi;
        if (lane < pad_lo + pad_hi) {
            const bool plo = lane < pad_lo;
            const unsigned pidx = plo ? n_lo + lane : hi_base + n_hi + (lane - pad_lo);
            G[pidx] = make_uint4(kOOB, kOOB, plo ? 0u : kOOB, plo ? kPadPos : kOOB);
            HP[pidx] = (unsigned char)kPadPos;
        }
    };
    uint4 ra[2];
    unsigned hpos[2];
    v4f lt[2], rt[2], lb[2], rbv[2];
    auto fetch_lo = [&](unsigned p, unsigned grp, int s) { ra[s] = Gbuf[p * kRecs + grp * kBinsPerIter + b]; };
    auto fetch_hi = [&](unsigned p, unsigned grp, int s) {
        ra[s] = Gbuf[p * kRecs + grp * kBinsPerIter + b];
        hpos[s] = HPbuf[p * kRecs + grp * kBinsPerIter + b];
    };
    auto issue_lo = [&](__amdgpu_buffer_rsrc_t rs, int s) {
        // kOOB + q_bytes (or anything + kOOB) stays out of range: no wrap below 2^32
        lt[s] = buf_load(rs, ra[s].x + q_bytes);
        rt[s] = buf_load(rs, ra[s].y + q_bytes);  // the bin's one other distinct tap, if any
    };
    auto issue_hi = [&](__amdgpu_buffer_rsrc_t rs, int s) {
        lt[s] = buf_load(rs, ra[s].x + q_bytes);
        rt[s] = buf_load(rs, ra[s].y + q_bytes);
        lb[s] = buf_load(rs, ra[s].z + q_bytes);
        rbv[s] = buf_load(rs, ra[s].w + q_bytes);
    };
    float* const t_row = T + (q * 4) * kTStride;
    float* const t_pad = T + kChunk * kTStride + (lane & 31u);
    auto put = [&](unsigned pos, v4f v) {
        float* tw = pos < (unsigned)kTileBins ? t_row + (pos ^ wswz) : t_pad;
        tw[0 * kTStride] = v.x;
        tw[1 * kTStride] = v.y;
        tw[2 * kTStride] = v.z;
        tw[3 * kTStride] = v.w;
    };
    auto blend_lo = [&](int s) {
        // At most two distinct pixels p (= lt) and p2, with weights w and 1 - w
        // (w = 1: p alone; w = 1/2: p and its right OR lower neighbour; kernel.cu:131-134 with
        // rx, ry in {0, 1/2}).  The reference adds all four terms (:138-141); the two that
        // re-read p or p2 carry weight exactly 0.  So
        //   taps finite      -> those terms add +-0 and the sum is  (0 + p*w) + p2*(1-w);
        //   a tap non-finite -> the reference's 0 * tap is NaN, and so is its sum.
        // The two-term sum is finite exactly when both taps are (both weights are non-zero and
        // at most 1), so adding  v - v  (0, or NaN when v is not finite) reproduces the
        // reference bit for bit in both cases.  A NaN weight (centre at infinity) gives NaN
        // either way.
        const float w = as_f(ra[s].z), w2 = 1.0f - w;
        v4f v = z4;
        v += lt[s] * w;
        v += rt[s] * w2;
        v += v - v;
        put(ra[s].w, v);
    };
    auto blend_hi = [&](int s) {
        // four distinct pixels: dx and dy, so rx = ry = 1/2 and every weight is 1/4
        v4f v = z4;  // kernel.cu:136-141, four channels at a time
        v += lt[s] * 0.25f;
        v += rt[s] * 0.25f;
        v += rbv[s] * 0.25f;
        v += lb[s] * 0.25f;
        put(hpos[s], v);
    };
    // An empty asm that "rewrites" the current group's taps: placed right after the next
    // group's loads are issued, it pins the first use of the current taps (and with it the
    // s_waitcnt) BEHIND that issue.  Without it the compiler hoists the first multiplies of the
    // blend above the "more groups?" branch and waits before anything new is in flight.
    auto pin_lo = [&](int s) { asm volatile("" : "+v"(lt[s]), "+v"(rt[s])); };
    auto pin_hi = [&](int s) { asm volatile("" : "+v"(lt[s]), "+v"(rt[s]), "+v"(lb[s]), "+v"(rbv[s])); };

    // phase C of one item: [rows < C] x [64 bins] -> 256-byte row segments
    // `live` = false turns every store into an out-of-range one (dropped by the descriptor
    // check) instead of branching around them: the instruction stream of the loop must be the
    // same on every path, or the compiler's s_waitcnt counts -- which take the most
    // conservative value where paths merge -- degrade to vmcnt(0) and every blend waits for
    // the store acknowledgements.  (Also the ablation knob: dbg & 1 drops the output stores.)
    auto store_tile = [&](unsigned n, unsigned t, unsigned long long cur_mask, bool live) {
        live = live && !(dbg & 1);
        // descriptor over this (roi, chunk) block of the output: rows >= C fall out of range
        float* obase = out + ((size_t)n * C + k * kChunk) * NB;
        const __amdgpu_buffer_rsrc_t ws = make_rsrc(obase, chans_here * (unsigned)NB * 4u);
        const unsigned bin0 = t * kTileBins + col;
        // bins that were in no group (masked by pw > roi_pooled_width) are zero
        const unsigned nib = (unsigned)(cur_mask >> col) & 15u;
        const bool a0 = nib & 1u, a1 = nib & 2u, a2 = nib & 4u, a3 = nib & 8u;
        // two halves of 4 row groups: 16 instead of 32 registers live across the LDS reads
#pragma unroll
        for (int hs = 0; hs < 2; ++hs) {
            v4f v[kChunk / 8];
#pragma unroll
            for (int s4 = 0; s4 < kChunk / 8; ++s4) {
                const unsigned r = (hs * (kChunk / 8) + s4) * 4 + row0;
                v[s4] = *reinterpret_cast<const v4f*>(T + r * kTStride + (col ^ ((r >> 3) * 4u)));
            }
#pragma unroll
            for (int s4 = 0; s4 < kChunk / 8; ++s4) {
                const unsigned r = (hs * (kChunk / 8) + s4) * 4 + row0;
                const unsigned off = (r * (unsigned)NB + bin0) * 4u;
                const v4f o = {a0 ? v[s4].x : 0.f, a1 ? v[s4].y : 0.f, a2 ? v[s4].z : 0.f, a3 ? v[s4].w : 0.f};
                if (VEC_STORE) {  // NB % 4 == 0: the 4 bins are all inside or all outside the row
                    // AUX == 2 (the shipped policy): the first of the tile's eight stores goes out
                    // write-through (sc0 sc1), the other seven streaming (nt) -- see buf_store
                    if (AUX == 2 && hs == 0 && s4 == 0)
                        buf_store<kMinorAux>(ws, (live && bin0 < (unsigned)NB) ? off : kOOB, o);
                    else
                    buf_store<AUX>(ws, (live && bin0 < (unsigned)NB) ? off : kOOB, o);
                } else {
                    buf_store1<AUX>(ws, (live && bin0 + 0 < (unsigned)NB) ? off + 0 : kOOB, o.x);
                    buf_store1<AUX>(ws, (live && bin0 + 1 < (unsigned)NB) ? off + 4 : kOOB, o.y);
                    buf_store1<AUX>(ws, (live && bin0 + 2 < (unsigned)NB) ? off + 8 : kOOB, o.z);
                    buf_store1<AUX>(ws, (live && bin0 + 3 < (unsigned)NB) ? off + 12 : kOOB, o.w);
                }
            }
        }
    };

    // Software pipeline over the items of this wave.  gfx950 counts loads and stores with ONE
    // in-order counter, so a load issued after a tile's stores cannot be consumed before those
    // stores are acknowledged by the memory system (microseconds, with 256 MiB streaming out).
    // Per iteration, with the tile of item i-1 complete in T and the records of item i in set p:
    //   1. the first loads of item i are issued (they are AHEAD of the stores in the counter);
    //   2. tile i-1 leaves: LDS -> registers -> 8 x 1 KiB streaming stores;
    //   3. geometry of item i+1 -> record set p^1: ~250 instructions that depend on no memory
    //      access, run while the stores drain;
    //   4. phase B of item i -> T (its later groups do wait for the store acknowledgements).
    unsigned cur = slot;
    if (cur >= items) return;
    unsigned n = fdiv(cur, div_tiles);
    unsigned t = cur - n * (unsigned)ntiles;
    unsigned p = 0;
    unsigned n_prev = 0, t_prev = 0;
    unsigned long long mask_prev = 0;
    bool have_prev = false;
    unsigned g_lo_next = 0, g_hi_next = 0;
    unsigned long long mask_next = 0;
    {
        const Affine A = aff[n];
        geometry(A, t, 0, g_lo, g_hi, act_mask);
    }
    int batch = aff[n].batch;
    lds_wave_sync();

    for (;;) {
        const unsigned nxt = cur + nslots;
        const bool has_next = nxt < items;
        const unsigned n_next = has_next ? fdiv(nxt, div_tiles) : n;
        const unsigned t_next = nxt - n_next * (unsigned)ntiles;
        const Affine A_next = aff[n_next];  // scalar loads, in flight during steps 1-2

        const bool batch_ok = batch >= 0 && batch < batch_size;
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(
            map + (size_t)(batch_ok ? batch : 0) * lay.img_stride + (size_t)k * lay.chunk_stride, lay.slice_bytes);
        fetch_lo(p, 0, 0);
        issue_lo(rs, 0);  // LO group 0 (there always is one)
        store_tile(n_prev, t_prev, mask_prev, have_prev);
        lds_wave_sync();  // T has been read: free for this item's blends
        if (has_next) geometry(A_next, t_next, p ^ 1u, g_lo_next, g_hi_next, mask_next);

        // ---- phase B: LO groups (group 0 is already in flight), then HI groups; the loads of
        // group g+1 are issued before group g is blended.  The loops are unrolled with an early
        // exit, and the two exit paths end in different (empty) asm statements so that the
        // compiler cannot merge their tails: each blend then has ONE predecessor and its
        // s_waitcnt knows exactly how many younger loads are in flight.
#pragma unroll
        for (int it = 0; it < kIters; ++it) {
            const int s = it & 1;
            if ((unsigned)(it + 1) < g_lo) {
                fetch_lo(p, it + 1, s ^ 1);
                issue_lo(rs, s ^ 1);
                pin_lo(s);
                blend_lo(s);
                asm volatile("; lo: more groups follow");
            } else {
                blend_lo(s);
                asm volatile("; lo: last group");
                break;
            }
        }
        if (g_hi > 0) {
            fetch_hi(p, g_lo, 0);
            issue_hi(rs, 0);
#pragma unroll
            for (int it = 0; it < kIters; ++it) {
                const int s = it & 1;
                if ((unsigned)(it + 1) < g_hi) {
                    fetch_hi(p, g_lo + it + 1, s ^ 1);
                    issue_hi(rs, s ^ 1);
                    pin_hi(s);
                    blend_hi(s);
                    asm volatile("; hi: more groups follow");
                } else {
                    blend_hi(s);
                    asm volatile("; hi: last group");
                    break;
                }
            }
        }
        lds_wave_sync();  // T complete; record set p^1 complete
        if (!has_next) {
            store_tile(n, t, act_mask, true);
            break;
        }
        n_prev = n;
        t_prev = t;
        mask_prev = act_mask;
        have_prev = true;
        cur = nxt;
        n = n_next;
        t = t_next;
        batch = A_next.batch;
        g_lo = g_lo_next;
        g_hi = g_hi_next;
        act_mask = mask_next;
        p ^= 1u;
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
// K3g: backward as a GATHER (the default tiled backward).  The scatter of K3 is bound by the
// fp32 atomic rate (113 M lane-atomics at cfg3 -> 0.65 ms).  The (bin, tap) -> pixel relation
// does not depend on the channel, so it is inverted ONCE per call:
//   pairs   count pass + exclusive scan + fill pass: for every map pixel the list of
//           (bin, weight) that the reference's four atomicAdds (kernel.cu:267-274) send to it
//           -- 442 K pairs of 8 bytes at cfg3, integer atomics on 25.6 K counters;
//   relayout top_diff (R, C, PH*PW) -> chunk-major (R, C/32, PH*PW + 1, 32) with the forward's
//           prologue kernel, so that the 32 channels of one bin are one 128-byte line;
//   gather  one (sub-)wave per pixel walks its list: a 16-byte load per lane and pair, all
//           channels of the pixel accumulated in registers, one store.  No float atomics, no
//           memset of the gradient.
// Taps of a bin that alias one pixel (dx == 0 / dy == 0) become ONE pair: the reference adds
// w*g and 0*g separately, which for finite g is w*g and for non-finite g is NaN either way;
// the pair carries an "add 0*g as well" flag (sign bit of the weight) so that both cases are
// reproduced.
// ------------------------------------------------------------------------------------
// Pixel keys of the lists are TILED: a 128-byte line of counters holds an 8 x 4 pixel block
// (key = ((b*Ht + y/4)*Wt + x/8)*32 + (y%4)*8 + x%8).  Device-scope atomics are bound by the
// number of line REQUESTS (measured ~10-13 G/s chip-wide, however many lanes a request carries):
// the 64 bins of a wave lie along a line segment of the map, which crosses ~3x fewer 8 x 4
// blocks than 32 x 1 row segments.
struct KeyLayout {
    unsigned Wt, Ht;   // blocks per row / per column
    unsigned keys;     // batch * Ht * Wt * 32
};

__device__ __forceinline__ unsigned pixel_key(const KeyLayout& L, unsigned b, unsigned y, unsigned x)
{
    return (((b * L.Ht + (y >> 2)) * L.Wt + (x >> 3)) << 5) + ((y & 3u) << 3) + (x & 7u);
}

// The (pixel, weight) pairs of one bin: the taps that pass kernel.cu:267-274, one pair per
// DISTINCT pixel.  `emit(key, w)`: w carries the "reference also adds 0*g here" flag in its sign.
template <class Emit>
__device__ __forceinline__ void bin_pairs(const Affine& A, unsigned ph, unsigned pw, int height, int width,
                                          int batch_size, const KeyLayout& L, Emit emit)
{
    float bcx, bcy;
    bool active = bin_centre(A, (int)ph, (int)pw, height, width, bcx, bcy);
    active = active && A.batch >= 0 && A.batch < batch_size;
    const Taps tp = make_taps(bcx, bcy, active, height, width, 1u);
    const unsigned f = tp.flags;
    if (!(f & kActive)) return;
    float wlt, wrt, wrb, wlb;
    tap_weights(tp.rx, tp.ry, wlt, wrt, wrb, wlb);
    const bool dx = f & kDx, dy = f & kDy;
    // a passing tap has 0 < x, y < W-1, H-1: the coordinates are small non-negative integers
    const unsigned x0 = (unsigned)f2i_sat(floorf(bcx)), y0 = (unsigned)f2i_sat(floorf(bcy));
    const unsigned b = (unsigned)A.batch;
    const float alias = (dx && dy) ? 1.0f : -1.0f;  // not all four taps distinct: some pixel also gets 0*g
    // an aliased tap has the bounds of the tap it aliases; weights are positive (NaN only when
    // every bound has failed)
    if (f & kB00) emit(pixel_key(L, b, y0, x0), wlt * alias);
    if (dx && (f & kB01)) emit(pixel_key(L, b, y0, x0 + 1u), wrt * alias);
    if (dy && (f & kB10)) emit(pixel_key(L, b, y0 + 1u, x0), wlb * alias);
    if (dx && dy && (f & kB11)) emit(pixel_key(L, b, y0 + 1u, x0 + 1u), wrb * alias);
}

constexpr unsigned kScanBlock = 4096;  // keys per block of the first scan level

// list offset of key i after the two-level scan
__device__ __forceinline__ unsigned list_offset(const unsigned* __restrict__ off, const unsigned* __restrict__ bsum, unsigned i)
{
    return off[i] + bsum[i / kScanBlock];
}

// FILL == false: cnt[key] += 1 per pair.  FILL == true: cnt counts back down, handing out the
// slots of the key's segment.
template <bool FILL>
__device__ __forceinline__ void pairs_body(unsigned idx, const Affine* __restrict__ aff, int num_rois,
                                           int height, int width, int pooled_width, int NB, int batch_size,
                                           unsigned lines_per_roi, FastDiv div_nb, FastDiv div_pw,
                                           const KeyLayout& L, int* __restrict__ cnt,
                                           const unsigned* __restrict__ off, const unsigned* __restrict__ bsum,
                                           uint2* __restrict__ pairs)
{
    const unsigned n = fdiv(idx, div_nb);
    if (n >= (unsigned)num_rois) return;
    const unsigned j = idx - n * (unsigned)NB;
    const unsigned ph = fdiv(j, div_pw);
    const unsigned pw = j - ph * (unsigned)pooled_width;
    const Affine A = aff[n];
    bin_pairs(A, ph, pw, height, width, batch_size, L, [&](unsigned key, float w) {
        if (!FILL) {
            atomicAdd(cnt + key, 1);
        } else {
            const int slot = atomicAdd(cnt + key, -1) - 1;
            // line index of (roi n, bin j) in chunk 0 of the relaid-out top_diff
            pairs[list_offset(off, bsum, key) + (unsigned)slot] = make_uint2(n * lines_per_roi + j, as_u(w));
        }
    });
}

// One launch, two kinds of blocks: [0, pair_blocks) count (FILL = false) or write (FILL = true)
// the pair lists -- bound by the atomic request rate -- and the rest relay out tiles
// [tile_begin, tile_end) of top_diff -- bound by HBM.  They share the chip instead of running
// one after the other; the host gives each of the two launches half of the tiles.
template <bool FILL, int SAUX>
__global__ __launch_bounds__(256) void rroi_bwd_pairs_relayout_kernel(
    const Affine* __restrict__ aff, int num_rois, int height, int width, int pooled_width, int NB,
    int batch_size, unsigned lines_per_roi, FastDiv div_nb, FastDiv div_pw, KeyLayout L,
    int* __restrict__ cnt, const unsigned* __restrict__ off, const unsigned* __restrict__ bsum,
    uint2* __restrict__ pairs, int pair_blocks, const float* __restrict__ top_diff,
    float* __restrict__ tdT, int C, int nchunks, int ptiles, int relayout_blocks, int tile_begin,
    int tile_end)
{
    __shared__ __attribute__((aligned(16))) float T[kChunk * kTP];
    if ((int)blockIdx.x < pair_blocks) {
        const unsigned total = (unsigned)num_rois * (unsigned)NB;
        for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += (unsigned)pair_blocks * 256u)
            pairs_body<FILL>(idx, aff, num_rois, height, width, pooled_width, NB, batch_size, lines_per_roi,
                             div_nb, div_pw, L, cnt, off, bsum, pairs);
        return;
    }
    relayout_run<SAUX, true>(T, top_diff, tdT, C, NB, pooled_width, pooled_width, div_pw, nchunks, ptiles,
                          tile_begin + (int)blockIdx.x - pair_blocks, relayout_blocks, tile_end, aff,
                          batch_size);
}

// Exclusive scan of cnt[0..N) (N = keys + 1, the last element reads as 0), two levels:
// level 1: every block scans kScanBlock keys -> off[] (block-local) and its total -> bsum[block];
// level 2: one block scans the totals in place.  Readers add the two (list_offset).
__device__ __forceinline__ unsigned block_exclusive_scan_1024(unsigned mine, unsigned* wsum, unsigned& total)
{
    const unsigned tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    unsigned incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned o = __shfl_up(incl, d, 64);
        if (lane >= (unsigned)d) incl += o;
    }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    unsigned wbase = 0, tot = 0;
    for (unsigned k = 0; k < 16; ++k) {
        const unsigned v = wsum[k];
        if (k < wv) wbase += v;
        tot += v;
    }
    total = tot;
    __syncthreads();
    return wbase + incl - mine;
}

__global__ __launch_bounds__(1024) void rroi_scan1_kernel(const int* __restrict__ cnt, unsigned* __restrict__ off,
                                                          unsigned* __restrict__ bsum, unsigned keys)
{
    __shared__ unsigned wsum[16];
    const unsigned i0 = blockIdx.x * kScanBlock + threadIdx.x * 4u;
    unsigned v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = i0 + e < keys ? (unsigned)cnt[i0 + e] : 0u;
    unsigned total;
    unsigned run = block_exclusive_scan_1024(v[0] + v[1] + v[2] + v[3], wsum, total);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (i0 + e <= keys) off[i0 + e] = run;
        run += v[e];
    }
    if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}

__global__ __launch_bounds__(1024) void rroi_scan2_kernel(unsigned* __restrict__ bsum, unsigned nblocks)
{
    __shared__ unsigned wsum[16];
    __shared__ unsigned carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (unsigned base = 0; base < nblocks; base += 1024u) {
        const unsigned i = base + threadIdx.x;
        const unsigned v = i < nblocks ? bsum[i] : 0u;
        unsigned total;
        const unsigned ex = block_exclusive_scan_1024(v, wsum, total);
        const unsigned carry = carry_s;
        if (i < nblocks) bsum[i] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + total;
        __syncthreads();
    }
}

// gather: `sub` = 8 * nchunks_pass lanes serve one pixel (lane -> chunk, channel quad); 64 / sub
// pixels per wave; one pixel group per thread group, so the hardware's block dispatch balances
// the (very uneven) list lengths.  The 16-byte loads of eight pairs are in flight together.
__global__ __launch_bounds__(256) void rroi_bwd_gather_kernel(
    const float* __restrict__ tdT, const unsigned* __restrict__ off, const unsigned* __restrict__ bsum,
    const uint2* __restrict__ pairs, float* __restrict__ gcm, int C, int height, int width, int pitch,
    int nchunks, unsigned lines_per_chunk, unsigned sub_shift, KeyLayout L, FastDiv div_bt, FastDiv div_wt)
{
    const unsigned tid = blockIdx.x * 256u + threadIdx.x;
    const unsigned sub = 1u << sub_shift;              // lanes per pixel (8..64)
    const unsigned sl = tid & (sub - 1u);              // lane within the pixel's group
    const unsigned key = tid >> sub_shift;
    if (key >= L.keys) return;
    // key -> (b, y, x)
    const unsigned blk = key >> 5, in = key & 31u;
    const unsigned b = fdiv(blk, div_bt);              // / (Ht*Wt)
    const unsigned r = blk - b * (L.Ht * L.Wt);
    const unsigned by = fdiv(r, div_wt);
    const unsigned y = by * 4u + (in >> 3), x = (r - by * L.Wt) * 8u + (in & 7u);
    if (y >= (unsigned)height || x >= (unsigned)width) return;  // padding of the key space
    const unsigned beg = list_offset(off, bsum, key), end = list_offset(off, bsum, key + 1u);
    const unsigned quad = sl & 7u;
    const unsigned slice_px = (unsigned)height * (unsigned)pitch;
    const v4f z4 = {0.f, 0.f, 0.f, 0.f};
    constexpr int kDepth = 8;
    // channel passes of `sub / 8` chunks each (one pass when C <= 256)
    for (unsigned k0 = 0; k0 < (unsigned)nchunks; k0 += sub >> 3) {
        const unsigned k = k0 + (sl >> 3);
        const bool c_ok = k < (unsigned)nchunks && k * kChunk + quad * 4u < (unsigned)C;
        const float* src = tdT + ((size_t)k * lines_per_chunk) * kChunk + quad * 4u;
        v4f acc = z4;
        for (unsigned i = beg; i < end; i += kDepth) {
            uint2 e[kDepth];
            v4f g[kDepth];
#pragma unroll
            for (int d = 0; d < kDepth; ++d) e[d] = i + d < end ? pairs[i + d] : make_uint2(0u, 0u);
#pragma unroll
            for (int d = 0; d < kDepth; ++d)
                g[d] = (c_ok && i + d < end) ? *reinterpret_cast<const v4f*>(src + (size_t)e[d].x * kChunk) : z4;
#pragma unroll
            for (int d = 0; d < kDepth; ++d) {
                if (i + d < end) {
                    // kernel.cu:260-263: v_k = w_k * top_diff, then one add per tap
                    acc += g[d] * as_f(e[d].y & 0x7fffffffu);
                    if (e[d].y & 0x80000000u) acc += g[d] * 0.0f;
                }
            }
        }
        if (c_ok) {
            float* dst = gcm + (((size_t)b * nchunks + k) * slice_px + (size_t)y * pitch + x) * kChunk + quad * 4u;
            *reinterpret_cast<v4f*>(dst) = acc;
        }
    }
}

// ------------------------------------------------------------------------------------
// K3: backward as a SCATTER (RROI_PATH_TILED_ATOMIC; the first tiled backward, kept for
// comparison and for problems whose pair lists do not fit 32-bit indices): into a zeroed
// chunk-major gradient (B, C/32, H*Wp, 32) with hardware fp32 atomics, then relayout to NCHW.
// Same item decomposition as the forward.
// Measured on MI355X (tools/kbench): an atomic wave instruction that covers 2 full 128-byte
// lines sustains 325 G lane-atomics/s, one that touches 8 lines at a 16-byte stride only
// 80 G/s.  So the (bin, tap) contributions of a tile are first COMPACTED into a list (only
// the taps that pass the reference's bounds, kernel.cu:267-274), and the scatter loop takes
// two list entries per instruction: lanes 0-31 add the 32 channels of one pixel, lanes 32-63
// those of another.
// ------------------------------------------------------------------------------------
template <bool VEC_LOAD>
__global__ __launch_bounds__(kWave) void rroi_bwd_tiled_kernel(
    const float* __restrict__ top_diff, const Affine* __restrict__ aff, float* __restrict__ gcm,
    int num_rois, int C, int height, int width, int pitch, int pooled_width, int NB, int batch_size,
    int nchunks, int ntiles, FastDiv div_tiles, FastDiv div_pw)
{
    __shared__ __attribute__((aligned(16))) float T[kChunk * kTStride];
    __shared__ __attribute__((aligned(16))) uint4 P[kTileBins * 4];  // {float offset of the pixel, weight, bin, -}

    const unsigned lane = threadIdx.x;
    const unsigned k = blockIdx.x % (unsigned)nchunks;
    const unsigned slot = blockIdx.x / (unsigned)nchunks;
    const unsigned nslots = gridDim.x / (unsigned)nchunks;
    const unsigned items = (unsigned)num_rois * (unsigned)ntiles;
    const unsigned slice_px = (unsigned)height * (unsigned)pitch;
    const unsigned col = (lane & 15) * 4, row0 = lane >> 4;
    const unsigned c = lane & 31, half = lane >> 5;   // scatter phase: channel within the chunk, list parity
    const bool c_ok = k * kChunk + c < (unsigned)C;
    const unsigned long long below = (1ull << lane) - 1ull;

    for (unsigned item = slot; item < items; item += nslots) {
        const unsigned n = fdiv(item, div_tiles);
        const unsigned t = item - n * (unsigned)ntiles;
        const Affine A = aff[n];
        const bool batch_ok = A.batch >= 0 && A.batch < batch_size;
        unsigned npairs;
        {
            const unsigned bin = t * kTileBins + lane;
            const unsigned ph = fdiv(bin, div_pw);
            const unsigned pw = bin - ph * (unsigned)pooled_width;
            float bcx, bcy;
            // kernel.cu:232-242: the backward reads the centre the forward stored; where the
            // forward's mask (pw <= roi_pooled_width) was false it stored nothing, the
            // buffer holds 0, and a (0,0) centre fails every bound of :267-274.  So the
            // scatter happens exactly where the forward's mask holds.
            bool active = bin_centre(A, (int)ph, (int)pw, height, width, bcx, bcy);
            active = active && bin < (unsigned)NB && batch_ok;
            const Taps tp = make_taps(bcx, bcy, active, height, width, 1u);
            float wlt, wrt, wrb, wlb;
            tap_weights(tp.rx, tp.ry, wlt, wrt, wrb, wlb);
            const unsigned f = tp.flags;
            // pixel index on the padded row pitch of the chunk-major gradient, as a float offset
            const int x0 = f2i_sat(floorf(bcx)), y0 = f2i_sat(floorf(bcy));
            const unsigned o_lt = ((unsigned)y0 * (unsigned)pitch + (unsigned)x0) * kChunk;
            const unsigned o_rt = o_lt + ((f & kDx) ? (unsigned)kChunk : 0u);
            const unsigned o_lb = o_lt + ((f & kDy) ? (unsigned)pitch * kChunk : 0u);
            const unsigned o_rb = o_lb + ((f & kDx) ? (unsigned)kChunk : 0u);
            // compaction: list order = all lt entries, then rt, rb, lb (kernel.cu:267-274 order)
            const unsigned long long m0 = __ballot(f & kB00), m1 = __ballot(f & kB01);
            const unsigned long long m2 = __ballot(f & kB11), m3 = __ballot(f & kB10);
            const unsigned n0 = __popcll(m0), n1 = __popcll(m1), n2 = __popcll(m2);
            npairs = n0 + n1 + n2 + (unsigned)__popcll(m3);
            if (f & kB00) P[__popcll(m0 & below)] = make_uint4(o_lt, as_u(wlt), lane, 0u);
            if (f & kB01) P[n0 + __popcll(m1 & below)] = make_uint4(o_rt, as_u(wrt), lane, 0u);
            if (f & kB11) P[n0 + n1 + __popcll(m2 & below)] = make_uint4(o_rb, as_u(wrb), lane, 0u);
            if (f & kB10) P[n0 + n1 + n2 + __popcll(m3 & below)] = make_uint4(o_lb, as_u(wlb), lane, 0u);
        }
        // stage the [32 ch][64 bin] slice of top_diff
        {
            const float* ibase = top_diff + ((size_t)n * C + k * kChunk) * NB + (size_t)t * kTileBins;
            const unsigned bin0 = t * kTileBins + col;
#pragma unroll
            for (int s = 0; s < kChunk / 4; ++s) {
                const unsigned r = s * 4 + row0;
                v4f v = {0.f, 0.f, 0.f, 0.f};
                if (k * kChunk + r < (unsigned)C) {
                    const float* ip = ibase + (size_t)(r * (unsigned)NB + col);
                    if (VEC_LOAD) {
                        if (bin0 < (unsigned)NB) v = *reinterpret_cast<const v4f*>(ip);
                    } else {
                        if (bin0 + 0 < (unsigned)NB) v.x = ip[0];
                        if (bin0 + 1 < (unsigned)NB) v.y = ip[1];
                        if (bin0 + 2 < (unsigned)NB) v.z = ip[2];
                        if (bin0 + 3 < (unsigned)NB) v.w = ip[3];
                    }
                }
                *reinterpret_cast<v4f*>(T + r * kTStride + (col ^ ((r >> 3) * 4u))) = v;
            }
        }
        lds_wave_sync();

        // scatter: two list entries per atomic instruction, 32 consecutive floats each
        float* gp = gcm + ((size_t)(batch_ok ? A.batch : 0) * nchunks + k) * ((size_t)slice_px * kChunk) + c;
        const float* trow = T + c * kTStride;
        const unsigned cswz = (c >> 3) * 4u;
        for (unsigned i = half; i < npairs; i += 2) {
            const uint4 e = P[i];
            // kernel.cu:260-263: v_k = w_k * top_diff_of_bin, one fp32 multiply
            const float contrib = as_f(e.y) * trow[e.z ^ cswz];
            if (c_ok) unsafeAtomicAdd(gp + e.x, contrib);
        }
        lds_wave_sync();
    }
}

// chunk-major gradient (B, nchunks, HW, 32) -> NCHW (B, C, HW); inverse of the prologue's tile.
__global__ __launch_bounds__(256) void rroi_cm_to_nchw_kernel(const float* __restrict__ cm,
                                                              float* __restrict__ nchw, int C,
                                                              int HW, int width, int pitch,
                                                              FastDiv div_w, int nchunks, int ptiles)
{
    __shared__ float T[kChunk * (kRelayoutPx + 1)];
    const int tid = threadIdx.x;
    int bid = blockIdx.x;
    const int pt = bid % ptiles;
    bid /= ptiles;
    const int k = bid % nchunks;
    const int b = bid / nchunks;
    const int lane = tid & 63, w = tid >> 6;
    const int p0 = pt * kRelayoutPx, c0 = k * kChunk;
    const float* src = cm + ((size_t)b * nchunks + k) * ((size_t)(HW / width) * pitch * kChunk);
    const int cq = lane & 7, pl = lane >> 3;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = w * 32 + j * 8 + pl;
        v4f v = {0.f, 0.f, 0.f, 0.f};
        const unsigned gp = (unsigned)(p0 + p);
        const unsigned y = fdiv(gp, div_w);
        const size_t pix = (size_t)y * pitch + (gp - y * (unsigned)width);
        if (p0 + p < HW) v = *reinterpret_cast<const v4f*>(src + pix * kChunk + cq * 4);
        float* tw = T + (cq * 4) * (kRelayoutPx + 1) + p;
        tw[0] = v.x;
        tw[kRelayoutPx + 1] = v.y;
        tw[2 * (kRelayoutPx + 1)] = v.z;
        tw[3 * (kRelayoutPx + 1)] = v.w;
    }
    __syncthreads();
    float* dst = nchw + ((size_t)b * C + c0) * HW + p0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = w * 8 + i;
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf) {
            const int p = hlf * 64 + lane;
            if (c0 + c < C && p0 + p < HW) dst[(size_t)c * HW + p] = T[c * (kRelayoutPx + 1) + p];
        }
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

// ------------------------------------------------------------------------------------
// Callers' side of the path: detected / annotated quads -> the op's
// (R, 6) ROI rows, on the device, for a whole image batch at once -- so that inference can
// issue ONE RoIRotate launch per image instead of one per word (tools/ocr_utils.py:131-177).
//   mode 0  tools/ocr_utils.py:133-150: fp32 edge vectors, fp32 squared length, sqrt and atan2 in
//           double, centre truncated to int, angle of edge 1->2
//   mode 1  src/ocr_process.py:196-206: everything in double, angle = mean of edges 1->2 and 0->3
// Also emits each box's pooled width by the inference rule (ocr_utils.py:147-150).
// ------------------------------------------------------------------------------------
__global__ void rroi_quads_to_rois_kernel(const float* __restrict__ quads, const float* __restrict__ bidx,
                                          int n, int mode, int target_h, float* __restrict__ rois,
                                          int* __restrict__ gw)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* b = quads + (size_t)i * 8;
    const float x0 = b[0], y0 = b[1], x1 = b[2], y1 = b[3], x2 = b[4], y2 = b[5], x3 = b[6], y3 = b[7];
    double w, h, angle, cx, cy;
    if (mode == 0) {
        const float ccx = (((x0 + x1) + x2) + x3) / 4.0f, ccy = (((y0 + y1) + y2) + y3) / 4.0f;
        const float dwx = x2 - x1, dwy = y2 - y1, dhx = x1 - x0, dhy = y1 - y0;
        w = sqrt((double)((dwx * dwx) + (dwy * dwy)));
        h = sqrt((double)((dhx * dhx) + (dhy * dhy)));
        angle = atan2((double)(y2 - y1), (double)(x2 - x1));
        cx = (double)(int)ccx;  // int(center[0]): truncation toward zero
        cy = (double)(int)ccy;
    } else {
        const double X0 = x0, Y0 = y0, X1 = x1, Y1 = y1, X2 = x2, Y2 = y2, X3 = x3, Y3 = y3;
        cx = (((X0 + X1) + X2) + X3) / 4.0;
        cy = (((Y0 + Y1) + Y2) + Y3) / 4.0;
        const double dwx = X2 - X1, dwy = Y2 - Y1, dhx = X1 - X0, dhy = Y1 - Y0;
        w = sqrt(dwx * dwx + dwy * dwy);
        h = sqrt(dhx * dhx + dhy * dhy);
        angle = (atan2(Y2 - Y1, X2 - X1) + atan2(Y3 - Y0, X3 - X0)) / 2.0;
    }
    angle = -angle / 3.1415926535 * 180.0;
    float* r = rois + (size_t)i * 6;
    r[0] = bidx ? bidx[i] : 0.0f;
    r[1] = (float)cx;
    r[2] = (float)cy;
    r[3] = (float)h;
    r[4] = (float)w;
    r[5] = (float)angle;
    if (gw) {
        const double scale = (double)target_h / (h > 1.0 ? h : 1.0);  // max(1, h)
        const int t = (int)(w * scale) + target_h;
        const int g = t / 32;  // t >= target_h > 0: floor division == truncation
        gw[i] = (g > 2 ? g : 2) * 32;
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

// ------------------------------------------------------------------------------------
// Greedy CTC decode of the recognition logits that the crops turn into:
// tools/ocr_utils.py:183-186 takes `labels_pred.max(1)` -- arg max over the class axis of
// (N, nclass, T) -- and src/utils.py:87-97 keeps label t iff it is not the blank (0) and differs
// from label t-1.  One wave per sequence, lanes = time steps (the class loop reads rows that
// are contiguous in t); arg max = first index of the largest value, NaN counting as largest
// (torch.max); the kept labels are compacted with a ballot + popcount prefix.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWave) void rroi_ctc_greedy_kernel(
    const float* __restrict__ logits, int nclass, int T, const int* __restrict__ lengths,
    int* __restrict__ labels, int* __restrict__ decoded, int* __restrict__ decoded_len)
{
    const unsigned n = blockIdx.x, lane = threadIdx.x;
    int len = lengths ? lengths[n] : T;
    len = len < 0 ? 0 : (len > T ? T : len);
    const float* row = logits + (size_t)n * nclass * T;
    int* lab = labels ? labels + (size_t)n * T : nullptr;
    int* dec = decoded + (size_t)n * T;
    unsigned out = 0;
    int prev_last = -1;  // label of time step t0 - 1 (none before the first)
    for (int t0 = 0; t0 < T; t0 += kWave) {
        const int t = t0 + (int)lane;
        int best = 0;
        if (t < T) {
            float bv = row[t];
            bool bnan = bv != bv;
            for (int k = 1; k < nclass; ++k) {
                const float v = row[(size_t)k * T + t];
                const bool vnan = v != v;
                if (!bnan && (vnan || v > bv)) {
                    bv = v;
                    best = k;
                    bnan = vnan;
                }
            }
            if (lab) lab[t] = best;
        }
        int prev = __shfl_up(best, 1, kWave);
        if (lane == 0) prev = prev_last;
        const bool keep = t < len && best != 0 && best != prev;
        const unsigned long long m = __ballot(keep);
        if (keep) dec[out + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = best;
        out += (unsigned)__popcll(m);
        prev_last = __shfl(best, kWave - 1, kWave);
    }
    for (unsigned i = out + lane; i < (unsigned)T; i += kWave) dec[i] = 0;  // padding
    if (lane == 0) decoded_len[n] = (int)out;
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

// Row pitch (pixels = 128-byte lines) of the chunk-major copy: W plus a pad that makes the
// pitch odd, so that the lines of vertically adjacent pixels differ in their low address
// bits and spread over the L2 channels.
int g_row_pad = -1;  // exploration knob: -1 = automatic
int row_pitch(int width)
{
    if (g_row_pad >= 0) return width + g_row_pad;
    return width | 1;
}

bool shape_ok(int batch_size, int num_rois, int height, int width, int channels, int pooled_height,
              int pooled_width)
{
    if (batch_size <= 0 || num_rois < 0 || height <= 0 || width <= 0 || channels <= 0 ||
        pooled_height <= 0 || pooled_width <= 0)
        return false;
    // tap offsets inside a slice are 32-bit BYTE offsets; the image stride is a 32-bit float count
    const long nchunks = (channels + kChunk - 1) / kChunk;
    const long slice_px = (long)height * (width + 16) + 1;
    if (slice_px * kLineBytes >= (1L << 31)) return false;                          // chunk-major (< kOOB)
    if ((long)height * width * channels * 4 >= (1L << 31)) return false;            // channels-last
    if ((long)kChunk * pooled_height * pooled_width * 4 >= (1L << 31)) return false; // one output block
    if (slice_px * kChunk * nchunks >= (1L << 32)) return false;                    // img_stride
    if ((long)pooled_height * pooled_width >= (1L << 31)) return false;
    return true;
}

// grid for the tiled kernels: one wave per block, 12 waves per CU -- what the forward
// kernel's 12.4 KB of LDS admits (LDS is granted in 1280-byte granules; a grid larger than
// the resident set would run its surplus blocks as a second, mostly empty round) -- and a
// multiple of lcm(nchunks, 8) so that blockIdx % nchunks is also stable per XCD.
int g_waves_per_cu = 12;

int tiled_grid(long items, int nchunks)
{
    long want = items * nchunks;
    const long cap = (long)num_cus() * g_waves_per_cu;
    if (want > cap) want = cap;
    long unit = nchunks;
    while (unit % 8) unit += nchunks;  // lcm(nchunks, 8)
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

FastDiv make_fastdiv(unsigned d)
{
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    FastDiv f;
    f.m = (unsigned)((((1ull << l) - d) << 32) / d + 1);
    f.sh1 = l < 1 ? l : 1;
    f.sh2 = l > 0 ? l - 1 : 0;
    return f;
}

struct Workspace {
    Affine* aff;
    float* cm;
    size_t cm_bytes;
    size_t bytes;
};

// [affine table | chunk-major copy (B, nchunks, HW+1, 32)]; the copy is absent when
// channels-last features with C % 4 == 0 are consumed in place.
Workspace carve(void* ws, int batch_size, int channels, int height, int width, int num_rois,
                int layout)
{
    Workspace w;
    const size_t aff_bytes = align_up((size_t)(num_rois > 0 ? num_rois : 1) * sizeof(Affine), 256);
    const size_t nchunks = (channels + kChunk - 1) / kChunk;
    w.cm_bytes = layout == RROI_LAYOUT_NHWC
                     ? 0
                     : align_up((size_t)batch_size * nchunks * ((size_t)height * row_pitch(width) + 1) * kLineBytes, 256);
    w.aff = reinterpret_cast<Affine*>(ws);
    w.cm = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + aff_bytes);
    w.bytes = aff_bytes + w.cm_bytes;
    return w;
}

// backward: [affine | chunk-major gradient | pixel counters | pixel offsets | pairs |
//            chunk-major top_diff (R, nchunks, NB + 1, 32)]
struct BwdWorkspace {
    Affine* aff;
    float* gcm;
    int* cnt;
    unsigned* off;
    unsigned* bsum;
    uint2* pairs;
    float* tdT;
    KeyLayout keys;
    unsigned scan_blocks;
    size_t gcm_bytes, cnt_bytes;
    size_t bytes;
    bool gather_ok;  // the gather formulation's 32-bit indices hold for this problem
};

BwdWorkspace carve_bwd(void* ws, int batch_size, int channels, int height, int width, int num_rois, int NB)
{
    BwdWorkspace w;
    const size_t nchunks = (channels + kChunk - 1) / kChunk;
    const size_t R = num_rois > 0 ? num_rois : 1;
    w.keys.Wt = (unsigned)((width + 7) / 8);
    w.keys.Ht = (unsigned)((height + 3) / 4);
    const size_t nkeys = (size_t)batch_size * w.keys.Ht * w.keys.Wt * 32;
    w.keys.keys = (unsigned)nkeys;
    w.scan_blocks = (unsigned)((nkeys + 1 + kScanBlock - 1) / kScanBlock);
    const size_t aff_bytes = align_up(R * sizeof(Affine), 256);
    w.gcm_bytes = align_up((size_t)batch_size * nchunks * height * row_pitch(width) * kLineBytes, 256);
    w.cnt_bytes = align_up(nkeys * sizeof(int), 256);
    const size_t off_bytes = align_up((nkeys + 1) * sizeof(unsigned), 256);
    const size_t bsum_bytes = align_up((size_t)w.scan_blocks * sizeof(unsigned), 256);
    const size_t pair_bytes = align_up(4 * R * NB * sizeof(uint2), 256);
    const size_t td_bytes = align_up(R * nchunks * ((size_t)NB + 1) * kLineBytes, 256);
    // pair slots, top_diff line indices and keys are 32-bit; the gather launches one thread group per key
    w.gather_ok = 4 * R * NB < (1ull << 32) && R * nchunks * ((size_t)NB + 1) < (1ull << 32) &&
                  nkeys * 64 < (1ull << 40) && nkeys < (1ull << 31);
    char* b = reinterpret_cast<char*>(ws);
    w.aff = reinterpret_cast<Affine*>(b);
    b += aff_bytes;
    w.gcm = reinterpret_cast<float*>(b);
    b += w.gcm_bytes;
    w.cnt = reinterpret_cast<int*>(b);
    b += w.cnt_bytes;
    w.off = reinterpret_cast<unsigned*>(b);
    b += off_bytes;
    w.bsum = reinterpret_cast<unsigned*>(b);
    b += bsum_bytes;
    w.pairs = reinterpret_cast<uint2*>(b);
    b += pair_bytes;
    w.tdT = reinterpret_cast<float*>(b);
    b += td_bytes;
    w.bytes = aff_bytes + w.gcm_bytes +
              (w.gather_ok ? w.cnt_bytes + off_bytes + bsum_bytes + pair_bytes + td_bytes : 0);
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

int g_store_aux = 2;   // cache policy of the output stores (see buf_store); 16 / 0 for exploration
int g_fwd_dbg = 0;     // ablation: 1 = skip output stores, 2 = all taps out of range
int g_prologue_blocks_per_cu = 3;
int g_prologue_aux = 0;
// store policy of the backward's top_diff relayout: streaming (nt) 189.5 us per call, sc1 191.3,
// plain 194.0 (tools/bwd_profile.py with RROI_BWD_SWEEP=1); non-temporal LOADS in the gather: +16 us
int g_bwd_relayout_aux = 2;

}  // namespace

// ====================================================================================
extern "C" {

const char* rroi_align_hip_version(void) { return "rroi_align_hip 0.2.0 gfx950"; }

size_t rroi_align_forward_workspace_bytes(int batch_size, int channels, int height, int width,
                                          int num_rois, int feature_layout)
{
    if (batch_size <= 0 || channels <= 0 || height <= 0 || width <= 0 || num_rois < 0) return 0;
    return carve(nullptr, batch_size, channels, height, width, num_rois, feature_layout).bytes;
}

size_t rroi_align_backward_workspace_bytes(int batch_size, int channels, int height, int width,
                                           int num_rois, int pooled_height, int pooled_width)
{
    if (batch_size <= 0 || channels <= 0 || height <= 0 || width <= 0 || num_rois < 0 ||
        pooled_height <= 0 || pooled_width <= 0)
        return 0;
    return carve_bwd(nullptr, batch_size, channels, height, width, num_rois,
                     pooled_height * pooled_width).bytes;
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

    if (feature_layout == RROI_LAYOUT_NHWC && channels % 4 != 0) return 0;  // repack to NCHW first
    const Workspace ws = carve(workspace, batch_size, channels, height, width, num_rois, feature_layout);
    if (!workspace || workspace_bytes < ws.bytes) return 0;
    const int HW = height * width;
    const int nchunks = ceil_div(channels, kChunk);
    const bool zero_copy = feature_layout == RROI_LAYOUT_NHWC;
    const float* map = zero_copy ? features : ws.cm;
    const int pitch = row_pitch(width);

    // prologue: relayout + zero pixels + affine table in one launch
    if (stages & RROI_STAGE_PROLOGUE) {
        const int ptiles = ceil_div(HW, kRelayoutPx);
        const int relayout_tiles = zero_copy ? 0 : ptiles * nchunks * batch_size;
        // ~3 resident blocks per CU, each streaming several tiles with the next tile prefetched
        int relayout_blocks = relayout_tiles;
        if (relayout_blocks > num_cus() * g_prologue_blocks_per_cu) {
            relayout_blocks = num_cus() * g_prologue_blocks_per_cu;
            long unit = nchunks;
            while (unit % 8) unit += nchunks;  // lcm(nchunks, 8): keeps block -> chunk -> XCD stable
            if (relayout_blocks >= unit) relayout_blocks = (int)(relayout_blocks / unit * unit);
        }
        const int zero_blocks = zero_copy ? 0 : ceil_div((long)batch_size * nchunks * kChunk, 256);
        const int aff_blocks = ceil_div(num_rois, 256);
#define RROI_LAUNCH_PRO(AUX)                                                                        \
    hipLaunchKernelGGL(rroi_prologue_kernel<AUX>, dim3(relayout_blocks + zero_blocks + aff_blocks),   \
                       dim3(256), 0, stream, features, ws.cm, channels, HW, width, pitch,             \
                       make_fastdiv((unsigned)width), nchunks, ptiles, relayout_blocks,               \
                       relayout_tiles, zero_blocks, batch_size, rois, num_rois, pooled_height,        \
                       spatial_scale, ws.aff)
        if (g_prologue_aux == 16) RROI_LAUNCH_PRO(16);
        else RROI_LAUNCH_PRO(0);
#undef RROI_LAUNCH_PRO
        const int st = launch_status();
        if (st != 1) return st;
    }
    if (stages & RROI_STAGE_GATHER) {
        const int ntiles = ceil_div(NB, kTileBins);
        if ((long)num_rois * ntiles >= (1L << 31)) return 0;
        const int grid = tiled_grid((long)num_rois * ntiles, nchunks);
        SliceLayout lay;
        if (zero_copy) {
            lay.px_bytes = (unsigned)channels * 4u;
            lay.row_bytes = (unsigned)width * lay.px_bytes;
            lay.slice_bytes = (unsigned)HW * lay.px_bytes;  // to the end of the image (base = chunk k of pixel 0)
            lay.chunk_stride = kChunk;
            lay.img_stride = (unsigned)HW * (unsigned)channels;
        } else {
            lay.px_bytes = kLineBytes;
            lay.row_bytes = (unsigned)pitch * kLineBytes;
            lay.slice_bytes = (unsigned)height * lay.row_bytes;
            lay.chunk_stride = ((unsigned)height * (unsigned)pitch + 1u) * kChunk;
            lay.img_stride = lay.chunk_stride * (unsigned)nchunks;
        }
        const FastDiv dt = make_fastdiv((unsigned)ntiles), dp = make_fastdiv((unsigned)pooled_width);
#define RROI_LAUNCH_FWD(VEC, AUX)                                                                    \
    hipLaunchKernelGGL((rroi_fwd_tiled_kernel<VEC, AUX>), dim3(grid), dim3(kWave), 0, stream, map,   \
                       ws.aff, top_data, num_rois, channels, height, width, pooled_width, NB,        \
                       batch_size, nchunks, ntiles, lay, dt, dp, g_fwd_dbg)
        if (NB % 4 != 0) RROI_LAUNCH_FWD(false, 2);
        else if (g_store_aux == 0) RROI_LAUNCH_FWD(true, 0);    // exploration only
        else if (g_store_aux == 16) RROI_LAUNCH_FWD(true, 16);  // exploration only
        else if (g_store_aux == 3) RROI_LAUNCH_FWD(true, 3);    // exploration only: pure nt (sc0 nt)
        else RROI_LAUNCH_FWD(true, 2);
#undef RROI_LAUNCH_FWD
    }
    return launch_status();
}

// Exploration knobs for the sweeps and ablations quoted in DESIGN.md: compiled only with
// -DRROI_EXPLORE (tools/kbench.hip, `make EXPLORE=1`).  The product library does not export them;
// the defaults above are the shipped configuration.
#ifdef RROI_EXPLORE
int rroi_align_debug_set_store_aux(int v)
{
    const int old = g_store_aux;
    g_store_aux = v;
    return old;
}
int rroi_align_debug_set_bwd_relayout_aux(int v)
{
    const int old = g_bwd_relayout_aux;
    g_bwd_relayout_aux = v;
    return old;
}
int rroi_align_debug_set_fwd_dbg(int v)
{
    const int old = g_fwd_dbg;
    g_fwd_dbg = v;
    return old;
}
int rroi_align_debug_set_prologue_blocks(int v)
{
    const int old = g_prologue_blocks_per_cu;
    if (v >= 1 && v <= 64) g_prologue_blocks_per_cu = v;
    return old;
}
int rroi_align_debug_set_prologue_aux(int v)
{
    const int old = g_prologue_aux;
    g_prologue_aux = v;
    return old;
}
int rroi_align_debug_set_row_pad(int v)
{
    const int old = g_row_pad;
    g_row_pad = v;
    return old;
}
int rroi_align_debug_set_waves_per_cu(int v)
{
    const int old = g_waves_per_cu;
    if (v >= 1 && v <= 64) g_waves_per_cu = v;
    return old;
}
#endif  // RROI_EXPLORE

int rroi_align_backward_hip(const float* top_diff, float spatial_scale, int batch_size,
                            int num_rois, int height, int width, int channels,
                            int pooled_height, int pooled_width, const float* rois,
                            float* bottom_diff, void* workspace, size_t workspace_bytes, int path,
                            void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!shape_ok(batch_size, num_rois, height, width, channels, pooled_height, pooled_width))
        return 0;
    if (path != RROI_PATH_AUTO && path != RROI_PATH_DIRECT && path != RROI_PATH_TILED &&
        path != RROI_PATH_TILED_ATOMIC)
        return 0;
    if (!bottom_diff) return 0;
    const int NB = pooled_height * pooled_width;
    const size_t HW = (size_t)height * width;
    const size_t in_bytes = (size_t)batch_size * channels * HW * sizeof(float);
    if (num_rois == 0) return status_of(hipMemsetAsync(bottom_diff, 0, in_bytes, stream));
    if (!top_diff || !rois) return 0;

    const bool tiled = path == RROI_PATH_AUTO
                           ? pick_tiled(batch_size, channels, height, width, num_rois, NB)
                           : path != RROI_PATH_DIRECT;
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

    const BwdWorkspace ws = carve_bwd(workspace, batch_size, channels, height, width, num_rois, NB);
    if (!workspace || workspace_bytes < ws.bytes) return 0;
    const int nchunks = ceil_div(channels, kChunk);
    const int pitch = row_pitch(width);
    const int ptiles = ceil_div((long)HW, kRelayoutPx);
    hipLaunchKernelGGL(rroi_affine_kernel, dim3(ceil_div(num_rois, 256)), dim3(256), 0, stream,
                       rois, num_rois, pooled_height, spatial_scale, ws.aff);
    int st = launch_status();
    if (st != 1) return st;

    if (path != RROI_PATH_TILED_ATOMIC && ws.gather_ok) {
        // (1) pixel -> (bin, weight) lists: count, scan, fill
        const KeyLayout KL = ws.keys;
        hipError_t e = hipMemsetAsync(ws.cnt, 0, (size_t)KL.keys * sizeof(int), stream);
        if (e != hipSuccess) return status_of(e);
        const unsigned lines_per_chunk = (unsigned)NB + 1u;
        const unsigned lines_per_roi = lines_per_chunk * (unsigned)nchunks;
        // one pair block per CU, looping over the bins: the pair passes need outstanding atomics,
        // not CU slots -- more blocks only take residency from the relayout (207 -> 197 us per call)
        int pblocks = ceil_div((long)num_rois * NB, 256);
        if (pblocks > num_cus()) pblocks = num_cus();
        const FastDiv dnb = make_fastdiv((unsigned)NB), dpw = make_fastdiv((unsigned)pooled_width);
        // count || first half of the relayout;  scan;  fill || second half.  The relayout is the
        // forward's, with R "images" of PH x PW "pixels" and the masked bins skipped:
        // top_diff (R, C, NB) -> chunk-major (R, nchunks, NB + 1, 32)
        const int tt = ceil_div(NB, kRelayoutPx);
        const long tiles = (long)tt * nchunks * num_rois;
        if (tiles >= (1L << 31)) return 0;
        long unit = nchunks;
        while (unit % 8) unit += nchunks;  // whole groups of chunks per launch and per grid step
        const long half = (tiles / 2 + unit - 1) / unit * unit < tiles ? (tiles / 2 + unit - 1) / unit * unit : tiles;
        auto relayout_grid = [&](long n) {
            const long cap = (long)num_cus() * 8;
            if (n <= cap) return n;
            return cap >= unit ? cap / unit * unit : cap;
        };
#define RROI_LAUNCH_PR(FILL, SAUX, BLOCKS, T0, T1)                                                   \
    hipLaunchKernelGGL((rroi_bwd_pairs_relayout_kernel<FILL, SAUX>), dim3((unsigned)(pblocks + (BLOCKS))), \
                       dim3(256), 0, stream, ws.aff, num_rois, height, width, pooled_width, NB,          \
                       batch_size, lines_per_roi, dnb, dpw, KL, ws.cnt, ws.off, ws.bsum, ws.pairs,       \
                       pblocks, top_diff, ws.tdT, channels, nchunks, tt, (int)(BLOCKS), (int)(T0), (int)(T1))
        {
            const long blocks = relayout_grid(half);
            if (g_bwd_relayout_aux == 16) RROI_LAUNCH_PR(false, 16, blocks, 0, half);
            else if (g_bwd_relayout_aux == 2) RROI_LAUNCH_PR(false, 2, blocks, 0, half);
            else RROI_LAUNCH_PR(false, 0, blocks, 0, half);
        }
        hipLaunchKernelGGL(rroi_scan1_kernel, dim3(ws.scan_blocks), dim3(1024), 0, stream, ws.cnt, ws.off,
                           ws.bsum, KL.keys);
        hipLaunchKernelGGL(rroi_scan2_kernel, dim3(1), dim3(1024), 0, stream, ws.bsum, ws.scan_blocks);
        {
            const long blocks = relayout_grid(tiles - half);
            if (g_bwd_relayout_aux == 16) RROI_LAUNCH_PR(true, 16, blocks, half, tiles);
            else if (g_bwd_relayout_aux == 2) RROI_LAUNCH_PR(true, 2, blocks, half, tiles);
            else RROI_LAUNCH_PR(true, 0, blocks, half, tiles);
        }
#undef RROI_LAUNCH_PR
        st = launch_status();
        if (st != 1) return st;
        // (3) gather: one thread group per key, no grid-stride
        unsigned sub_shift = 3;  // 8 lanes = one chunk
        while ((1u << sub_shift) < 8u * (unsigned)nchunks && sub_shift < 6) ++sub_shift;
        const unsigned groups_per_block = 256u >> sub_shift;
        const long gblocks = ceil_div((long)KL.keys, (long)groups_per_block);
        hipLaunchKernelGGL(rroi_bwd_gather_kernel, dim3((unsigned)gblocks), dim3(256), 0, stream, ws.tdT,
                           ws.off, ws.bsum, ws.pairs, ws.gcm, channels, height, width, pitch, nchunks,
                           lines_per_chunk, sub_shift, KL, make_fastdiv(KL.Ht * KL.Wt),
                           make_fastdiv(KL.Wt));
        st = launch_status();
        if (st != 1) return st;
    } else {
        hipError_t e = hipMemsetAsync(ws.gcm, 0, (size_t)batch_size * nchunks * height * pitch * kLineBytes, stream);
        if (e != hipSuccess) return status_of(e);
        const int ntiles = ceil_div(NB, kTileBins);
        if ((long)num_rois * ntiles >= (1L << 31)) return 0;
        const int grid = tiled_grid((long)num_rois * ntiles, nchunks);
        const FastDiv dt = make_fastdiv((unsigned)ntiles), dp = make_fastdiv((unsigned)pooled_width);
        if (NB % 4 == 0)
            hipLaunchKernelGGL(rroi_bwd_tiled_kernel<true>, dim3(grid), dim3(kWave), 0, stream,
                               top_diff, ws.aff, ws.gcm, num_rois, channels, height, width, pitch,
                               pooled_width, NB, batch_size, nchunks, ntiles, dt, dp);
        else
            hipLaunchKernelGGL(rroi_bwd_tiled_kernel<false>, dim3(grid), dim3(kWave), 0, stream,
                               top_diff, ws.aff, ws.gcm, num_rois, channels, height, width, pitch,
                               pooled_width, NB, batch_size, nchunks, ntiles, dt, dp);
        st = launch_status();
        if (st != 1) return st;
    }
    hipLaunchKernelGGL(rroi_cm_to_nchw_kernel, dim3(ptiles * nchunks * batch_size), dim3(256), 0,
                       stream, ws.gcm, bottom_diff, channels, (int)HW, width, pitch,
                       make_fastdiv((unsigned)width), nchunks, ptiles);
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

int rroi_align_quads_to_rois_hip(const float* quads, const float* batch_index, int n, int mode,
                                 int target_h, float* rois, int* target_gw, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (n < 0 || (mode != 0 && mode != 1) || target_h <= 0) return 0;
    if (n == 0) return 1;
    if (!quads || !rois) return 0;
    hipLaunchKernelGGL(rroi_quads_to_rois_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, stream, quads,
                       batch_index, n, mode, target_h, rois, target_gw);
    return launch_status();
}

int rroi_ctc_greedy_decode_hip(const float* logits, int num_seqs, int num_classes, int num_steps,
                               const int* lengths, int* labels, int* decoded, int* decoded_len,
                               void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (num_seqs < 0 || num_classes <= 0 || num_steps < 0) return 0;
    if ((long)num_seqs * num_classes * (long)num_steps >= (1L << 40)) return 0;
    if (num_seqs == 0) return 1;
    if (!decoded_len || (num_steps > 0 && (!logits || !decoded))) return 0;
    hipLaunchKernelGGL(rroi_ctc_greedy_kernel, dim3(num_seqs), dim3(kWave), 0, stream, logits,
                       num_classes, num_steps, lengths, labels, decoded, decoded_len);
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
