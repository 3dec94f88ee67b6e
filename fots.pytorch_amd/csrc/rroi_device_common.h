// rroi_device_common.h -- constants, the arithmetic recipe of the bin geometry (kernel.cu:58-141), buffer-descriptor helpers
// Part of the single translation unit rroi_align_hip.hip (included inside its anonymous
// namespace, in this order: rroi_device_common.h, rroi_forward_kernels.h,
// rroi_backward_kernels.h, rroi_callers_kernels.h); not a standalone header.
#pragma once


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

// The one library-dependent step of the arithmetic recipe (kernel.cu:73-74: `cos(angle)`, `sin(angle)` of a float).
//   RROI_TRIG_DOUBLE (0, default)  (float)cos((double)angle): the correctly rounded cosine up to a double rounding --
//                what oracle/rroi_align_oracle.c evaluates with the host's libm, identical on the device (2.7 M
//                angles, test_sincos_recipe_matches_host_libm);
//   RROI_TRIG_FP32 (1, opt-in)     cosf(angle) / sinf(angle) in fp32, i.e. this toolchain's device library (ocml) --
//                what the reference's OWN sources become when they are built for this GPU.  Neither float cosine is
//                correctly rounded, so recipe 0 and a float-cosine build differ in the last place for some angles,
//                and where that meets a rounding tie a bin's sample point moves (15-17 bins per million,
//                profiles/r03_fuzz_ref*.json); with recipe 1 the product and the reference's build for gfx950
//                agree in EVERY bin (tests/test_gpu_vs_reference.py::test_ocml_trig_recipe_moves_no_bin).
// PER CALL since round 5 (the RROI_PATH_TRIG_FP32 bit of an entry point's `path`): a kernel argument of every kernel
// that derives an affine from ROIs -- stream-ordered, capturable, no device-wide state.

// kernel.cu:58-84.  Every * and + below is one separately rounded fp32
// operation, in source order; the degree->radian conversion is the
// reference's double expression (:65); cos/sin are evaluated in double and
// rounded once to fp32 (recipe shared with oracle/rroi_align_oracle.c) unless
// the call asked for the fp32 recipe (`trig`, see above).
__device__ __forceinline__ Affine make_affine(const float* __restrict__ roi, int pooled_height,
                                              float spatial_scale, int trig)
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
    float Alpha, Beta;
    if (trig == 1) {
        Alpha = cosf(angle);
        Beta = sinf(angle);
    } else {
        Alpha = (float)cos((double)angle);
        Beta = (float)sin((double)angle);
    }
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
constexpr unsigned kOOB = 0x80000000u;          // > any slice / tile size (shape_ok)
constexpr unsigned kQuadOOB = 0x40000000u;      // "this lane's channel quad is beyond C": slices are < 1 GiB
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

