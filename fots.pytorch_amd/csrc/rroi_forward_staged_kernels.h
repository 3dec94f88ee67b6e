// rroi_forward_staged_kernels.h -- forward, ONE launch, NCHW read in place (round 3).
// Part of the single translation unit rroi_align_hip.hip (included inside its anonymous namespace
// after rroi_forward_kernels.h); not a standalone header.
//
// The tiled forward (rroi_forward_kernels.h) pays for a chunk-major copy of the whole map: a second
// launch, 52 MB of extra traffic and a re-fetch of the copy.  This kernel samples the (B, C, H, W)
// tensor as it is.  What makes that possible: for a fixed (roi, channel) the 512 bins of an 8 x 64
// crop are ONE contiguous 2 KiB run of the output, and the pixels they read are a small parallelogram
// of ONE channel plane -- rows of consecutive floats.  So per (roi, 512-bin block, 32-channel chunk):
//
//   geometry   once per item (not once per channel chunk pass over a 64-bin tile): every bin's
//              integer tap coordinates, validity and weight codes -> an 8-byte record in LDS;
//   spans      the taps of a PASS (8 or 4 slots of 64 bins) are bucketed by map row with LDS integer
//              min/max: row r of the pass needs pixels [xlo[r], xhi[r]].  The rows are laid out in LDS
//              at one odd pitch P = max row length (a skewed, parallelogram-shaped image of the map):
//              LDS pixel (y - ymin) * P + x - xlo[y - ymin];
//   loader     wave 2 of the workgroup: per channel PAIR, lane i of load m fetches pixel m * 64 + i of
//              that image from the two channel planes (consecutive lanes = consecutive floats of a map
//              row: coalesced), and writes the pair as one 8-byte LDS word.  It only ever loads, so its
//              vmcnt never waits for a store;
//   samplers   waves 0 and 1, four slots of the item each: lane = bin (consecutive lanes = consecutive pooled columns).  Per channel pair
//              and 64-bin slot: four ds_read_b64 (two channels of one tap each), the reference's
//              four-term blend (kernel.cu:136-141) as packed fp32 multiplies and adds, two 256-byte
//              stores.  It only ever stores, so it never waits for a store acknowledgement either
//              (gfx950 counts loads and stores with ONE in-order vmcnt: in the tiled kernel every
//              load issued behind a tile's stores waits for their acknowledgement);
//   hand-over  two LDS buffers + the loader's registers: while the sampler reads pair c from one buffer,
//              the loader writes pair c + 1 into the other and has the loads of pair c + 2 in flight;
//              one s_barrier per pair (an s_barrier does not drain vmcnt).
//
// A pass whose image does not fit the buffer (or would cost more loads than four taps per bin) is
// split (8 slots -> 2 x 4) and, failing that, staged DIRECTLY: load m = (slot, tap) fetches every
// lane's own tap pixel and the sampler reads LDS lane-linearly -- the sparse regime (bin spacing of
// several pixels) of FOTS's own crops.  Invalid taps (kernel.cu:116-126) read a zero pixel in LDS.
//
// Channel chunk k is served by workgroups with blockIdx % nchunks == k, i.e. (8 chunks) by one XCD
// whose L2 then holds exactly the 32 planes (3.3 MB at 160 x 160) it samples -- the affinity of the
// tiled kernel, without the copy.
#pragma once

constexpr int kStCap = 1024;       // pixels per staging buffer = per pass (LDS: 6 three-wave workgroups per CU)
constexpr int kStRows = 128;       // map rows a pass may touch in span mode
constexpr int kStSlots = 8;        // 64-bin slots per item
constexpr int kStMaxLoads = (kStCap + 63) / 64;            // loads per channel and pass
static_assert(kStCap % 256 == 0 && kStMaxLoads >= 16, "loads are issued in groups of four; direct staging needs 16");
constexpr unsigned kStZeroOff = (unsigned)kStCap * 8u;     // byte offset of the zero pixel in a buffer
constexpr int kStBig = 0x3fffffff;

typedef float v2f __attribute__((ext_vector_type(2)));

enum : unsigned {  // record flags (word 1)
    kRv00 = 1u, kRv01 = 2u, kRv11 = 4u, kRv10 = 8u,  // tap valid: lt, rt, rb, lb (kernel.cu:116-126)
    kRdx = 16u, kRdy = 32u,                          // x1 != x0, y1 != y0
    kRryShift = 6,                                   // weight code of ry (2 bits)
    kRactive = 256u,
};
enum : int { kStSpan = 0, kStDirect = 1, kStFail = 2 };

// Workgroup barrier that orders LDS traffic only: s_barrier does not wait for vector memory, and
// unlike __syncthreads() nothing here makes the compiler emit s_waitcnt vmcnt(0).
__device__ __forceinline__ void st_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// rx, ry are 0, 1/2 or NaN: the bin centre is half the sum of two integer-valued floats (or of
// infinities), kernel.cu:97-105 / :128-129
__device__ __forceinline__ unsigned st_wcode(float r) { return r == 0.0f ? 0u : r == 0.5f ? 1u : 2u; }
__device__ __forceinline__ float st_wvalue(unsigned c) { return c == 0u ? 0.0f : c == 1u ? 0.5f : __builtin_nanf(""); }

__device__ __forceinline__ uint2 st_record(const Affine& A, bool batch_ok, unsigned bin, int NB, FastDiv div_pw,
                                           int pooled_width, int height, int width)
{
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
    unsigned f = 0;
    if (active) {
        f = kRactive;
        if (y0ok && x0ok) f |= kRv00;
        if (y0ok && x1ok) f |= kRv01;
        if (y1ok && x1ok) f |= kRv11;
        if (y1ok && x0ok) f |= kRv10;
        if (x1 != x0) f |= kRdx;
        if (y1 != y0) f |= kRdy;
    }
    f |= st_wcode(bcy - fy) << kRryShift;
    // a valid tap implies 0 <= x0 <= x1 < W and 0 <= y0 <= y1 < H (15 bits each, staged_ok)
    const bool any = f & (kRv00 | kRv01 | kRv11 | kRv10);
    const unsigned xy = any ? ((unsigned)x0 & 0x7fffu) | (((unsigned)y0 & 0x7fffu) << 15) : 0u;
    return make_uint2(xy | (st_wcode(bcx - fx) << 30), f);
}

__device__ __forceinline__ int st_wave_min(int v)
{
#pragma unroll
    for (int o = 32; o; o >>= 1) v = min(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ int st_wave_max(int v)
{
#pragma unroll
    for (int o = 32; o; o >>= 1) v = max(v, __shfl_xor(v, o));
    return v;
}

struct StShared {
    v2f (*sbuf)[kStCap + 1];  // two staging buffers of (pixel, channel pair) words, each ending in the zero pixel
    uint2* srec;              // the item's bin records
    int *sxlo, *sxhi;         // per map row of the pass: lowest / highest pixel a tap needs
    int* syr;                 // lowest / highest map row of the pass
};

// One role of the workgroup (LOADER: wave 2; samplers: waves 0 and 1).  The two roles are separate
// instantiations so that their registers are allocated separately; they meet at the barriers, which
// both execute in the same number and order.
template <int AUX, bool LOADER>
__device__ __forceinline__ void st_run(
    const StShared sh, const float* __restrict__ feat, const float* __restrict__ rois, float* __restrict__ out,
    int num_rois, int C, int height, int width, int pooled_height, int pooled_width, int NB, float spatial_scale,
    int batch_size, int nchunks, int nblk, FastDiv div_blk, FastDiv div_pw, int dbg)
{
    v2f (*const sbuf)[kStCap + 1] = sh.sbuf;
    uint2* const srec = sh.srec;
    int* const sxlo = sh.sxlo;
    int* const sxhi = sh.sxhi;
    int* const syr = sh.syr;
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    constexpr bool loader = LOADER;
    const unsigned k = blockIdx.x % (unsigned)nchunks;
    const unsigned first = blockIdx.x / (unsigned)nchunks;
    const unsigned stride = gridDim.x / (unsigned)nchunks;
    const unsigned items = (unsigned)num_rois * (unsigned)nblk;
    const unsigned HW = (unsigned)height * (unsigned)width;
    const unsigned chans_here = min((unsigned)kChunk, (unsigned)C - k * kChunk);
    // dbg & 4 (ablation): one channel pair per pass -- what is left is geometry + pass set-up
    const unsigned npairs = (dbg & 4) ? 1u : (chans_here + 1u) / 2u;
    const unsigned nb4 = (unsigned)NB * 4u;

    if (threadIdx.x < 2) {
        sbuf[threadIdx.x][kStCap] = v2f{0.0f, 0.0f};  // the zero pixels: never written again
        syr[threadIdx.x] = threadIdx.x ? -1 : kStBig;
    }

    for (unsigned item = first; item < items; item += stride) {
        const unsigned n = fdiv(item, div_blk);
        const unsigned blk = item - n * (unsigned)nblk;
        const Affine A = make_affine(rois + (size_t)n * 6, pooled_height, spatial_scale);
        // the affine is computed by every lane (same values); the image index must be an SGPR, or the
        // map's buffer descriptor would be `divergent` and every load a waterfall loop
        const int batch = __builtin_amdgcn_readfirstlane(A.batch);
        const bool batch_ok = batch >= 0 && batch < batch_size;
        const unsigned bin_base = blk * (unsigned)(kStSlots * kWave);
        const unsigned nslots = min((unsigned)kStSlots, ((unsigned)NB - bin_base + 63u) / 64u);

        // ---- geometry: the three waves share the slots -> records (slots beyond the crop hold inactive bins)
        for (unsigned s = wave; s < (unsigned)kStSlots; s += 3u)
            srec[s * kWave + lane] = st_record(A, batch_ok, bin_base + s * kWave + lane, NB, div_pw, pooled_width,
                                               height, width);
        st_barrier();

        const float* plane0 = feat + ((size_t)(batch_ok ? batch : 0) * C + k * kChunk) * HW;
        const __amdgpu_buffer_rsrc_t rs_map = make_rsrc(plane0, chans_here * HW * 4u);
        float* obase = out + ((size_t)n * C + k * kChunk) * NB;
        const __amdgpu_buffer_rsrc_t rs_out = make_rsrc(obase, chans_here * nb4);

        // ---- one pass over slots [j0, j0 + 2 * SPS): sampler w takes slots j0 + w * SPS .. + SPS - 1.
        // Returns false when span mode does not fit and direct staging is not allowed (the caller then
        // splits the pass).  Every wave takes the same path: the decision is a function of the tables.
        auto pass = [&](auto stag, unsigned j0, bool allow_direct) __attribute__((always_inline)) -> bool {
            constexpr int SPS = decltype(stag)::value;
            const unsigned js = j0 + wave * SPS;  // first slot of this sampler
            // -- (a) samplers: the map rows their taps touch; loader: clear the row tables
            if (!loader) {
                int ylo = kStBig, yhi = -1;
#pragma unroll
                for (int jj = 0; jj < SPS; ++jj) {
                    const uint2 rec = srec[(js + jj) * kWave + lane];
                    const unsigned f = rec.y;
                    const int y0 = (int)((rec.x >> 15) & 0x7fffu), y1 = y0 + ((f & kRdy) ? 1 : 0);
                    const bool r0 = f & (kRv00 | kRv01), r1 = f & (kRv10 | kRv11);
                    ylo = min(ylo, r0 ? y0 : r1 ? y1 : kStBig);
                    yhi = max(yhi, r1 ? y1 : r0 ? y0 : -1);
                }
                ylo = st_wave_min(ylo);
                yhi = st_wave_max(yhi);
                if (lane == 0) {
                    atomicMin(&syr[0], ylo);
                    atomicMax(&syr[1], yhi);
                }
            } else {
#pragma unroll
                for (int r = 0; r < kStRows / kWave; ++r) {
                    sxlo[r * kWave + lane] = kStBig;
                    sxhi[r * kWave + lane] = -1;
                }
            }
            st_barrier();
            const int ymin = __builtin_amdgcn_readfirstlane(syr[0]);
            const int ymax = __builtin_amdgcn_readfirstlane(syr[1]);
            const int nrows = ymax >= ymin ? ymax - ymin + 1 : 0;
            const bool rows_fit = nrows <= kStRows;
            // -- (b) samplers: per row, the lowest and highest pixel a tap needs
            if (!loader && rows_fit && nrows > 0) {
#pragma unroll
                for (int jj = 0; jj < SPS; ++jj) {
                    const uint2 rec = srec[(js + jj) * kWave + lane];
                    const unsigned f = rec.y;
                    const int x0 = (int)(rec.x & 0x7fffu), x1 = x0 + ((f & kRdx) ? 1 : 0);
                    const int y0 = (int)((rec.x >> 15) & 0x7fffu);
                    if (f & (kRv00 | kRv01)) {
                        atomicMin(&sxlo[y0 - ymin], (f & kRv00) ? x0 : x1);
                        atomicMax(&sxhi[y0 - ymin], (f & kRv01) ? x1 : x0);
                    }
                    if ((f & kRdy) && (f & (kRv10 | kRv11))) {
                        atomicMin(&sxlo[y0 + 1 - ymin], (f & kRv10) ? x0 : x1);
                        atomicMax(&sxhi[y0 + 1 - ymin], (f & kRv11) ? x1 : x0);
                    }
                }
            }
            st_barrier();
            // everybody holds the row range in registers: reset it for the next pass (whose samplers touch it
            // only behind another barrier)
            if (loader && lane < 2) syr[lane] = lane ? -1 : kStBig;
            // -- (c) every wave: pitch and size of the image, the same arithmetic on the same tables
            int mode = kStSpan, P = 1, total = 0;
            if (!rows_fit) {
                mode = kStFail;
            } else if (nrows > 0) {
                int len = 1;
#pragma unroll
                for (int r = 0; r < kStRows / kWave; ++r) {
                    const int row = r * kWave + (int)lane;
                    if (row < nrows) len = max(len, sxhi[row] - sxlo[row] + 1);
                }
                P = __builtin_amdgcn_readfirstlane(st_wave_max(len)) | 1;  // odd pitch: the rows of a steep ROI fall on different banks
                total = nrows * P;
                // span mode pays total / 64 loads per channel, direct staging 4 per slot
                if (total > kStCap || total > 512 * SPS) mode = kStFail;
            }
            if (mode == kStFail && allow_direct) mode = kStDirect;
            if (mode == kStFail) {
                st_barrier();  // the tables have been read: the next attempt may clear them
                return false;
            }

            if (!loader) {
                // ---- sampler: LDS byte offsets of the four taps, weights, output offsets
                unsigned ta[SPS][4];
                float tw[SPS][4];
                unsigned bo[SPS];
#pragma unroll
                for (int jj = 0; jj < SPS; ++jj) {
                    const uint2 rec = srec[(js + jj) * kWave + lane];
                    const unsigned f = rec.y;
                    const int x0 = (int)(rec.x & 0x7fffu), y0 = (int)((rec.x >> 15) & 0x7fffu);
                    const int dx = (f & kRdx) ? 1 : 0, dy = (f & kRdy) ? 1 : 0;
                    if (mode == kStSpan) {
                        const int r0 = y0 - ymin, r1 = r0 + dy;
                        const bool u0 = f & (kRv00 | kRv01), u1 = f & (kRv10 | kRv11);
                        const int b0 = u0 ? r0 * P + x0 - sxlo[u0 ? r0 : 0] : 0;
                        const int b1 = u1 ? r1 * P + x0 - sxlo[u1 ? r1 : 0] : 0;
                        ta[jj][0] = (f & kRv00) ? (unsigned)b0 * 8u : kStZeroOff;
                        ta[jj][1] = (f & kRv01) ? (unsigned)(b0 + dx) * 8u : kStZeroOff;
                        ta[jj][2] = (f & kRv11) ? (unsigned)(b1 + dx) * 8u : kStZeroOff;
                        ta[jj][3] = (f & kRv10) ? (unsigned)b1 * 8u : kStZeroOff;
                    } else {
                        // direct staging: load ((slot of the pass) * 4 + tap) holds this lane's own tap (0.0 when invalid)
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            ta[jj][t] = (((wave * SPS + jj) * 4u + t) * kWave + lane) * 8u;
                    }
                    const bool act = f & kRactive;
                    const float rx = st_wvalue(rec.x >> 30), ry = st_wvalue((f >> kRryShift) & 3u);
                    const float ux = 1.0f - rx, uy = 1.0f - ry;   // kernel.cu:131-134
                    tw[jj][0] = act ? ux * uy : 0.0f;
                    tw[jj][1] = act ? rx * uy : 0.0f;
                    tw[jj][2] = act ? rx * ry : 0.0f;
                    tw[jj][3] = act ? ux * ry : 0.0f;
                    const unsigned bin = bin_base + (js + jj) * kWave + lane;
                    bo[jj] = (bin < (unsigned)NB && !(dbg & 1)) ? bin * 4u : kOOB;
                }
                st_barrier();  // A: pair 0 is in buffer 0
                auto sample = [&](auto btag, unsigned cp) {
                    constexpr int B = decltype(btag)::value;
                    const char* base = reinterpret_cast<const char*>(sbuf[B]);
                    const unsigned co = 2u * cp * nb4;
#pragma unroll
                    for (int jj = 0; jj < SPS; ++jj) {
                        const v2f lt = *reinterpret_cast<const v2f*>(base + ta[jj][0]);
                        const v2f rt = *reinterpret_cast<const v2f*>(base + ta[jj][1]);
                        const v2f rb = *reinterpret_cast<const v2f*>(base + ta[jj][2]);
                        const v2f lb = *reinterpret_cast<const v2f*>(base + ta[jj][3]);
                        v2f v = {0.0f, 0.0f};  // kernel.cu:136-141, two channels at a time
                        v += lt * tw[jj][0];
                        v += rt * tw[jj][1];
                        v += rb * tw[jj][2];
                        v += lb * tw[jj][3];
                        // 64 consecutive bins of one channel: 256 contiguous bytes; a channel beyond C is
                        // beyond the descriptor's range and dropped
                        buf_store1<AUX>(rs_out, bo[jj] + co, v.x);
                        buf_store1<AUX>(rs_out, bo[jj] + co + nb4, v.y);
                        // two slots' taps in flight at a time (16 registers), not all of them
                        if (jj & 1) __builtin_amdgcn_sched_barrier(0);
                    }
                };
                for (unsigned cp = 0; cp < npairs; cp += 2) {
                    sample(std::integral_constant<int, 0>{}, cp);
                    st_barrier();
                    if (cp + 1 < npairs) {
                        sample(std::integral_constant<int, 1>{}, cp + 1);
                        st_barrier();
                    }
                }
            } else {
                // ---- loader: byte offsets of this lane's pixels inside a channel plane
                unsigned go[kStMaxLoads];
                int nloads;
                if (mode == kStSpan) {
                    nloads = (total + kWave - 1) / kWave;
                    const unsigned M = (1u << 21) / (unsigned)P + 1u;  // i / P for i * P < 2^21
#pragma unroll
                    for (int m = 0; m < kStMaxLoads; ++m) {
                        const unsigned i = (unsigned)m * kWave + lane;
                        const unsigned r = (i * M) >> 21;
                        const unsigned col = i - r * (unsigned)P;
                        const bool in = (int)i < total;
                        const int lo = sxlo[in ? r : 0], hi = sxhi[in ? r : 0];
                        const bool ok = in && (int)col <= hi - lo;
                        go[m] = ok ? (((unsigned)ymin + r) * (unsigned)width + (unsigned)lo + col) * 4u : kOOB;
                    }
                } else {
                    nloads = 8 * SPS;
#pragma unroll
                    for (int m = 0; m < kStMaxLoads; ++m) {
                        unsigned o = kOOB;
                        if (m < 8 * SPS) {
                            const uint2 rec = srec[(j0 + (m >> 2)) * kWave + lane];
                            const unsigned f = rec.y;
                            const unsigned x = (rec.x & 0x7fffu) + (((m & 3) == 1 || (m & 3) == 2) && (f & kRdx) ? 1u : 0u);
                            const unsigned y = ((rec.x >> 15) & 0x7fffu) + ((m & 3) >= 2 && (f & kRdy) ? 1u : 0u);
                            const unsigned vbit = (m & 3) == 0 ? kRv00 : (m & 3) == 1 ? kRv01 : (m & 3) == 2 ? kRv11 : kRv10;
                            if (f & vbit) o = (y * (unsigned)width + x) * 4u;
                        }
                        go[m] = o;
                    }
                }
                if (dbg & 2) nloads = 0;
                unsigned d0[kStMaxLoads], d1[kStMaxLoads];
                // loads go out in groups of four (one branch per group; a load beyond the image reads nothing)
                const int ngroups = (nloads + 3) >> 2;
                auto fetch = [&](unsigned cp) __attribute__((always_inline)) {
                    // a channel beyond C: its offsets are beyond the descriptor's range and read 0
                    const unsigned c0 = 2u * cp * HW * 4u, c1 = c0 + HW * 4u;
#pragma unroll
                    for (int g = 0; g < kStMaxLoads / 4; ++g)
                        if (g < ngroups) {
#pragma unroll
                            for (int m = 4 * g; m < 4 * g + 4; ++m) {
                                d0[m] = __builtin_amdgcn_raw_buffer_load_b32(rs_map, go[m] + c0, 0, 0);
                                d1[m] = __builtin_amdgcn_raw_buffer_load_b32(rs_map, go[m] + c1, 0, 0);
                            }
                        }
                };
                auto publish = [&](auto btag) __attribute__((always_inline)) {
                    constexpr int B = decltype(btag)::value;
#pragma unroll
                    for (int g = 0; g < kStMaxLoads / 4; ++g)
                        if (g < ngroups) {
#pragma unroll
                            for (int m = 4 * g; m < 4 * g + 4; ++m)
                                sbuf[B][m * kWave + lane] = v2f{as_f(d0[m]), as_f(d1[m])};
                        }
                };
                fetch(0);
                publish(std::integral_constant<int, 0>{});
                if (npairs > 1) fetch(1);
                st_barrier();  // A
                for (unsigned cp = 0; cp < npairs; cp += 2) {
                    // the samplers read pair cp from buffer 0
                    if (cp + 1 < npairs) publish(std::integral_constant<int, 1>{});
                    if (cp + 2 < npairs) fetch(cp + 2);
                    st_barrier();
                    if (cp + 1 < npairs) {
                        // the samplers read pair cp + 1 from buffer 1
                        if (cp + 2 < npairs) publish(std::integral_constant<int, 0>{});
                        if (cp + 3 < npairs) fetch(cp + 3);
                        st_barrier();
                    }
                }
            }
            return true;
        };

        // 8 slots as one pass; where that does not fit, 4 + 4; where a half does not fit either, the
        // half is staged directly
        unsigned j0 = 0;
        bool wide = nslots > 4, direct = false;
        while (j0 < nslots) {
            const bool ok = wide ? pass(std::integral_constant<int, 4>{}, j0, false)
                                 : pass(std::integral_constant<int, 2>{}, j0, direct);
            if (ok) {
                j0 += wide ? 8u : 4u;
                direct = false;
            } else if (wide) {
                wide = false;
            } else {
                direct = true;
            }
        }
    }
}

template <int AUX>
__global__ __launch_bounds__(192, 4) void rroi_fwd_staged_kernel(
    const float* __restrict__ feat, const float* __restrict__ rois, float* __restrict__ out, int num_rois, int C,
    int height, int width, int pooled_height, int pooled_width, int NB, float spatial_scale, int batch_size,
    int nchunks, int nblk, FastDiv div_blk, FastDiv div_pw, int dbg)
{
    __shared__ __attribute__((aligned(16))) v2f sbuf[2][kStCap + 1];
    __shared__ uint2 srec[kStSlots * kWave];
    __shared__ int sxlo[kStRows], sxhi[kStRows];
    __shared__ int syr[2];
    const StShared sh = {sbuf, srec, sxlo, sxhi, syr};
    if (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 2)
        st_run<AUX, true>(sh, feat, rois, out, num_rois, C, height, width, pooled_height, pooled_width, NB,
                          spatial_scale, batch_size, nchunks, nblk, div_blk, div_pw, dbg);
    else
        st_run<AUX, false>(sh, feat, rois, out, num_rois, C, height, width, pooled_height, pooled_width, NB,
                           spatial_scale, batch_size, nchunks, nblk, div_blk, div_pw, dbg);
}
