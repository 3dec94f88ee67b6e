// rroi_backward_tile_kernels.h -- backward gather with the pixel lists built INSIDE the kernel (K3t: the tiled backward for C <= 64, and for C <= 128 while the lists are short)
// Part of the single translation unit rroi_align_hip.hip (included inside its anonymous
// namespace after rroi_backward_kernels.h); not a standalone header.
#pragma once

// ------------------------------------------------------------------------------------
// K3t: "tile gather".  K3g (rroi_backward_kernels.h) inverts the (bin, tap) -> pixel relation with
// three launches of device-scope integer atomics (count, scan, fill: 27 us of the 164 us call at
// cfg3, NOT hidden behind the relayout they share launches with -- measured: the call without any
// relayout takes 98 us, with it 164) and then walks per-pixel lists that live in HBM, a chain of
// dependent round trips (offsets -> records -> data).  Here a workgroup owns one 8 x 4-pixel map
// tile and finds the bins that reach it ANALYTICALLY:
//   1. lane = ROI: which ROIs can touch the tile at all (bounding interval of the affine image of
//      its bin grid against the tile box) -> compacted candidate list in LDS;
//   2. lane = (candidate ROI, pooled row ph): the pw interval of that row whose bins can reach the
//      tile, from the inverse of the affine along the row -- a conservative SUPERSET (margins
//      below) -> row segments + exclusive scan of their lengths;
//   3. lane = candidate bin: the reference's exact recipe (bin_pairs, the same code the list
//      passes of K3g run) decides which of the bin's taps fall on which pixel of the tile;
//      the (pixel, line of the bin in the relaid-out top_diff, weight) pairs of a round (256
//      candidate bins, <= 1024 pairs) are bucketed by pixel with LDS integer atomics;
//   4. every wave owns the 8 pixels of one tile row: lane = (pixel, channel quad), the pixels'
//      lists advance in lock-step, one 16-byte load per lane, chunk and list entry, all channels
//      of the pixel accumulate in registers over all rounds, one store at the end.
// No device-scope atomics, no lists in HBM, no scan launches; the source lines that neighbouring
// pixels share (the 2 x 2 footprint of a bin) are re-read by the same CU.
//
// Why the superset of step 2 is one.  A bin's taps are floor/ceil of the centre of the rounded,
// clamped bounding box of its four transformed corners (kernel.cu:86-105).  Whatever the clamps do,
// a tap that passes the backward's bounds (0 < x < W-1, kernel.cu:267-274) lies inside
// [round(min corner x), round(max corner x)]: both bounds clamp TOWARDS the map, and when the box
// lies wholly outside the map the centre does too and every bound fails.  So bin (ph, pw) can
// reach columns [tx0, tx1] only if   tx0 - 0.5 - e_hi <= X(pw, ph) <= tx1 + 0.5 - e_lo,   X = the
// image of the bin's first corner, e_lo / e_hi the extent of the other corners; the same in y.
// Along a pooled row X and Y are affine in pw, so the condition is an interval of pw.  `slop`
// covers the fp32 rounding of X, Y and of the quotients (a few ulps of the largest term); ROIs whose
// affine is not finite or is huge are not analysed at all: every bin of theirs is a candidate
// (NaN geometry samples the map centre, kernel.cu:97-105 with the NaN-dropping min/max).
// ------------------------------------------------------------------------------------
constexpr int kTgThreads = 256;
constexpr unsigned kTgTilePx = 32;       // 8 x 4 pixels: the key tile of KeyLayout
constexpr unsigned kTgRound = kTgThreads;  // candidate bins per round
constexpr unsigned kTgList = 4 * kTgRound;  // pairs per round
constexpr unsigned kTgSegs = 2 * kTgThreads;  // row segments collected before the rounds run

__device__ __forceinline__ bool tg_regular(const Affine& A)
{
    const float big = 1.6e7f;  // also false for NaN / inf
    return fabsf(A.m00) < big && fabsf(A.m01) < big && fabsf(A.m02) < big && fabsf(A.m10) < big &&
           fabsf(A.m11) < big && fabsf(A.m12) < big;
}

// [lo, hi] (floats, possibly empty or infinite) of the t with  a <= c + m*t <= b
__device__ __forceinline__ void tg_interval(float a, float b, float c, float m, float& lo, float& hi)
{
    if (m == 0.0f) {
        const bool in = c >= a && c <= b;
        lo = in ? -3.0e38f : 3.0e38f;
        hi = in ? 3.0e38f : -3.0e38f;
        return;
    }
    const float p1 = (a - c) / m, p2 = (b - c) / m;
    lo = fminf(p1, p2);
    hi = fmaxf(p1, p2);
}

struct TgBox {  // the tile, widened by rounding (0.5): X(pw, ph) of a reaching bin lies in [ax - e_hi, bx - e_lo]
    float ax, bx, ay, by;
};

// exclusive scan over the 256 threads of the block; `total` = sum.  wsum: 4 unsigned in LDS.
__device__ __forceinline__ unsigned tg_block_scan(unsigned mine, unsigned* wsum, unsigned& total)
{
    const unsigned lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    unsigned incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned o = __shfl_up(incl, d, 64);
        if (lane >= (unsigned)d) incl += o;
    }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    const unsigned s0 = wsum[0], s1 = wsum[1], s2 = wsum[2], s3 = wsum[3];
    total = s0 + s1 + s2 + s3;
    const unsigned wbase = (wv > 0 ? s0 : 0u) + (wv > 1 ? s1 : 0u) + (wv > 2 ? s2 : 0u);
    __syncthreads();  // wsum may be rewritten
    return wbase + incl - mine;
}

// The accumulate phase of one round.  Wave wv owns tile row wv; lane = (pixel s of the row, channel
// quad q); the 8 pixels advance through their lists in lock-step, one 16-byte
// load per lane, entry and chunk (uniform 64-bit base of the chunk + 32-bit lane offset).
// EXACT == false: acc = fma(g, w, acc) and nothing else.  The reference also adds 0 * g for a tap that
// aliases another (kernel.cu:260-274 with rx == 0 or ry == 0), which changes nothing unless g is not
// finite -- and then acc is not finite either (every weight is positive), which the caller detects once
// per tile and repeats the tile with EXACT == true: the 0 * g terms are added as the reference does.
template <int NK, bool EXACT>
__device__ __forceinline__ void tg_accumulate(v4f (&acc)[NK], const bool (&cok)[NK], const char* const (&cbase)[NK],
                                              const uint2* __restrict__ list, unsigned beg, unsigned mylen,
                                              unsigned line_bytes, unsigned q)
{
    const v4f z4 = {0.f, 0.f, 0.f, 0.f};
    unsigned maxlen = mylen;
    maxlen = max(maxlen, (unsigned)__shfl_xor((int)maxlen, 8, 64));
    maxlen = max(maxlen, (unsigned)__shfl_xor((int)maxlen, 16, 64));
    maxlen = max(maxlen, (unsigned)__shfl_xor((int)maxlen, 32, 64));
    for (unsigned t = 0; t < maxlen; ++t) {
        const bool v0 = t < mylen;
        const uint2 e0 = v0 ? list[beg + t] : make_uint2(0u, 0u);
        const unsigned o0 = e0.x * line_bytes + q * 16u;
        v4f g0[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k)
            g0[k] = (v0 && cok[k]) ? *reinterpret_cast<const v4f*>(cbase[k] + o0) : z4;
        const float w0 = as_f(e0.y & 0x7fffffffu);
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            acc[k].x = __builtin_fmaf(g0[k].x, w0, acc[k].x);
            acc[k].y = __builtin_fmaf(g0[k].y, w0, acc[k].y);
            acc[k].z = __builtin_fmaf(g0[k].z, w0, acc[k].z);
            acc[k].w = __builtin_fmaf(g0[k].w, w0, acc[k].w);
            if (EXACT && (e0.y & 0x80000000u)) acc[k] += g0[k] * 0.0f;
        }
    }
}

// NK = channel chunks a lane accumulates per pass (1, 2, 4, 8): a pass covers NK * 32 channels
template <int NK, bool DST_NHWC>
__global__ __launch_bounds__(kTgThreads, 4) void rroi_bwd_tile_gather_kernel(
    const float* __restrict__ tdT, const Affine* __restrict__ aff, float* __restrict__ gcm, int num_rois, int C,
    int height, int width, int pitch, int pooled_height, int pooled_width, int batch_size, int nchunks,
    unsigned chunk_stride, unsigned line_stride, unsigned lines_per_roi, KeyLayout L, unsigned ntiles,
    unsigned per_xcd, FastDiv div_bt, FastDiv div_wt, FastDiv div_ph)
{
    __shared__ unsigned cand[kTgThreads];                 // candidate ROIs of one batch of 256 ROIs
    __shared__ unsigned seg_n[kTgSegs], seg_ph[kTgSegs], seg_lo[kTgSegs], seg_len[kTgSegs];
    __shared__ unsigned segoff[kTgSegs + 1];              // exclusive scan of the segment lengths
    __shared__ unsigned cnt[kTgTilePx], base[kTgTilePx + 1];
    __shared__ uint2 list[kTgList];                       // (line, weight | alias flag), bucketed by pixel
    __shared__ unsigned wsum[4];

    const unsigned tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    // block -> tile: the tiles are dealt to the 8 XCDs (block % 8) in 8 contiguous raster bands, so that
    // neighbouring tiles -- which share source lines -- meet in one L2
    const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const unsigned tile = xcd * per_xcd + slot;
    if (slot >= per_xcd || tile >= ntiles) return;
    const unsigned bimg = fdiv(tile, div_bt);
    const unsigned trem = tile - bimg * (L.Ht * L.Wt);
    const unsigned tby = fdiv(trem, div_wt), tbx = trem - tby * L.Wt;
    const float tx0 = (float)(tbx * 8u), ty0 = (float)(tby * 4u);
    const TgBox box = {tx0 - 0.5f, tx0 + 7.5f, ty0 - 0.5f, ty0 + 3.5f};
    const float fPW = (float)pooled_width, fPH = (float)pooled_height;
    const unsigned long long below = (1ull << lane) - 1ull;
    if (tid < kTgTilePx) cnt[tid] = 0u;

    // accumulate phase mapping: wave wv owns tile row wv; lane = (pixel s of the row, channel quad q)
    const unsigned s = lane >> 3, q = lane & 7u, mypix = wv * 8u + s;
    const v4f z4 = {0.f, 0.f, 0.f, 0.f};
    const unsigned line_bytes = line_stride * 4u;

    for (unsigned k0 = 0; k0 < (unsigned)nchunks; k0 += NK) {   // channel passes (one when C <= NK * 32)
        v4f acc[NK];
        bool cok[NK];
        const char* cbase[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            cok[k] = k0 + k < (unsigned)nchunks && (k0 + k) * kChunk + q * 4u < (unsigned)C;
            cbase[k] = reinterpret_cast<const char*>(tdT + (size_t)(k0 + k) * chunk_stride);
        }
        for (int exact = 0; exact < 2; ++exact) {   // see tg_accumulate
#pragma unroll
            for (int k = 0; k < NK; ++k) acc[k] = z4;

            // rounds over the collected segments: scan their lengths, then 256 candidate bins at a time
            auto flush = [&](unsigned nseg) {
                if (nseg == 0) return;   // uniform
                __syncthreads();
                {
                    const unsigned i0 = 2u * tid, i1 = i0 + 1u;
                    const unsigned l0 = i0 < nseg ? seg_len[i0] : 0u, l1 = i1 < nseg ? seg_len[i1] : 0u;
                    unsigned nb;
                    const unsigned ex = tg_block_scan(l0 + l1, wsum, nb);
                    segoff[i0] = ex;
                    segoff[i1] = ex + l0;
                    if (tid == 0) segoff[kTgSegs] = nb;
                }
                __syncthreads();
                const unsigned nbins = segoff[kTgSegs];
                for (unsigned cb = 0; cb < nbins; cb += kTgRound) {
                    unsigned keys[4];
                    float wts[4];
                    unsigned ok = 0, line = 0;
                    const unsigned g = cb + tid;
                    if (g < nbins) {
                        // the segment that holds candidate bin g: last i with segoff[i] <= g
                        unsigned i = 0;
#pragma unroll
                        for (unsigned st = kTgSegs / 2; st > 0; st >>= 1)
                            if (segoff[i + st] <= g) i += st;
                        const unsigned bn = seg_n[i], bph = seg_ph[i], bpw = seg_lo[i] + (g - segoff[i]);
                        const Affine A = aff[bn];
                        unsigned np = 0;
                        bin_pairs(A, bph, bpw, height, width, batch_size, L, [&](unsigned key, float w) {
                            // at most four calls, in a fixed order
                            const bool mine = (key >> 5) == tile;
                            if (np == 0) { keys[0] = key; wts[0] = w; if (mine) ok |= 1u; }
                            else if (np == 1) { keys[1] = key; wts[1] = w; if (mine) ok |= 2u; }
                            else if (np == 2) { keys[2] = key; wts[2] = w; if (mine) ok |= 4u; }
                            else { keys[3] = key; wts[3] = w; if (mine) ok |= 8u; }
                            ++np;
                        });
                        line = bn * lines_per_roi + bph * (unsigned)pooled_width + bpw;
                    }
                    unsigned rank[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        rank[t] = (ok >> t) & 1u ? atomicAdd(&cnt[keys[t] & 31u], 1u) : 0u;
                    __syncthreads();
                    if (tid < 64) {
                        const unsigned v = tid < kTgTilePx ? cnt[tid] : 0u;
                        unsigned incl = v;
#pragma unroll
                        for (int d = 1; d < 32; d <<= 1) {
                            const unsigned o = __shfl_up(incl, d, 64);
                            if (lane >= (unsigned)d) incl += o;
                        }
                        if (tid < kTgTilePx) {
                            base[tid] = incl - v;
                            cnt[tid] = 0u;
                        }
                        if (tid == kTgTilePx - 1) base[kTgTilePx] = incl;
                    }
                    __syncthreads();
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if ((ok >> t) & 1u) list[base[keys[t] & 31u] + rank[t]] = make_uint2(line, as_u(wts[t]));
                    __syncthreads();
                    {
                        const unsigned beg = base[mypix], mylen = base[mypix + 1] - beg;
                        if (exact) tg_accumulate<NK, true>(acc, cok, cbase, list, beg, mylen, line_bytes, q);
                        else tg_accumulate<NK, false>(acc, cok, cbase, list, beg, mylen, line_bytes, q);
                    }
                    __syncthreads();  // list, base are rewritten by the next round
                }
            };

            unsigned nseg = 0;
            for (unsigned rb = 0; rb < (unsigned)num_rois; rb += kTgThreads) {
                // ---- 1. candidate ROIs of this batch of 256
                bool is_cand = false;
                {
                    const unsigned n = rb + tid;
                    if (n < (unsigned)num_rois) {
                        const Affine A = aff[n];
                        // bin_centre: active iff (float)pw <= rpw, so a negative or NaN rpw has no bin at all
                        if (A.batch == (int)bimg && A.rpw >= 0.0f) {
                            if (!tg_regular(A)) {
                                is_cand = true;
                            } else {
                                const float pwmax = fminf(floorf(A.rpw), fPW - 1.0f), phmax = fPH - 1.0f;
                                const float sx = 0.01f + 2e-6f * (fabsf(A.m00) * fPW + fabsf(A.m01) * fPH + fabsf(A.m02));
                                const float sy = 0.01f + 2e-6f * (fabsf(A.m10) * fPW + fabsf(A.m11) * fPH + fabsf(A.m12));
                                const float exl = fminf(0.f, A.m00) + fminf(0.f, A.m01), exh = fmaxf(0.f, A.m00) + fmaxf(0.f, A.m01);
                                const float eyl = fminf(0.f, A.m10) + fminf(0.f, A.m11), eyh = fmaxf(0.f, A.m10) + fmaxf(0.f, A.m11);
                                const float xlo = A.m02 + fminf(0.f, A.m00 * pwmax) + fminf(0.f, A.m01 * phmax);
                                const float xhi = A.m02 + fmaxf(0.f, A.m00 * pwmax) + fmaxf(0.f, A.m01 * phmax);
                                const float ylo = A.m12 + fminf(0.f, A.m10 * pwmax) + fminf(0.f, A.m11 * phmax);
                                const float yhi = A.m12 + fmaxf(0.f, A.m10 * pwmax) + fmaxf(0.f, A.m11 * phmax);
                                is_cand = xhi >= box.ax - exh - sx && xlo <= box.bx - exl + sx &&
                                          yhi >= box.ay - eyh - sy && ylo <= box.by - eyl + sy;
                            }
                        }
                    }
                }
                const unsigned long long cm = __ballot(is_cand);
                if (lane == 0) wsum[wv] = (unsigned)__popcll(cm);
                __syncthreads();
                const unsigned c0 = wsum[0], c1 = wsum[1], c2 = wsum[2], c3 = wsum[3];
                const unsigned ncand = c0 + c1 + c2 + c3;
                if (is_cand) cand[(wv > 0 ? c0 : 0u) + (wv > 1 ? c1 : 0u) + (wv > 2 ? c2 : 0u) + (unsigned)__popcll(cm & below)] = rb + tid;
                __syncthreads();

                // ---- 2. row segments: lane = (candidate, ph); the non-empty ones are appended to seg_*
                const unsigned nrows = ncand * (unsigned)pooled_height;
                for (unsigned rowb = 0; rowb < nrows; rowb += kTgThreads) {
                    if (nseg + kTgThreads > kTgSegs) {   // uniform: no room for another batch of rows
                        flush(nseg);
                        nseg = 0;
                    }
                    unsigned len = 0, lo_i = 0, n = 0, ph = 0;
                    const unsigned r = rowb + tid;
                    if (r < nrows) {
                        const unsigned ci = fdiv(r, div_ph);
                        ph = r - ci * (unsigned)pooled_height;
                        n = cand[ci];
                        const Affine A = aff[n];
                        if (!tg_regular(A)) {
                            lo_i = 0;
                            len = (unsigned)pooled_width;   // every bin; bin_pairs applies the mask
                        } else {
                            const float pwmax = fminf(floorf(A.rpw), fPW - 1.0f);
                            const float sx = 0.01f + 2e-6f * (fabsf(A.m00) * fPW + fabsf(A.m01) * fPH + fabsf(A.m02));
                            const float sy = 0.01f + 2e-6f * (fabsf(A.m10) * fPW + fabsf(A.m11) * fPH + fabsf(A.m12));
                            const float exl = fminf(0.f, A.m00) + fminf(0.f, A.m01), exh = fmaxf(0.f, A.m00) + fmaxf(0.f, A.m01);
                            const float eyl = fminf(0.f, A.m10) + fminf(0.f, A.m11), eyh = fmaxf(0.f, A.m10) + fmaxf(0.f, A.m11);
                            const float cX = A.m01 * (float)ph + A.m02, cY = A.m11 * (float)ph + A.m12;
                            float lx, hx, ly, hy;
                            tg_interval(box.ax - exh - sx, box.bx - exl + sx, cX, A.m00, lx, hx);
                            tg_interval(box.ay - eyh - sy, box.by - eyl + sy, cY, A.m10, ly, hy);
                            // one more bin each side for the rounding of the quotients; clamp before converting
                            const float lo_f = fminf(fmaxf(ceilf(fmaxf(lx, ly) - 0.05f), 0.0f), fPW);
                            const float hi_f = fminf(fmaxf(floorf(fminf(hx, hy) + 0.05f), -1.0f), pwmax);
                            if (hi_f >= lo_f) {
                                lo_i = (unsigned)lo_f;
                                len = (unsigned)(hi_f - lo_f) + 1u;
                            }
                        }
                    }
                    unsigned nnew;
                    const unsigned pos = nseg + tg_block_scan(len ? 1u : 0u, wsum, nnew);
                    if (len) {
                        seg_n[pos] = n;
                        seg_ph[pos] = ph;
                        seg_lo[pos] = lo_i;
                        seg_len[pos] = len;
                    }
                    nseg += nnew;
                }
                __syncthreads();  // cand is rewritten by the next ROI batch
            }
            flush(nseg);

            if (exact) break;
            // any contribution that is not finite?  (rare: then the tile is redone the exact way)
            int bad = 0;
#pragma unroll
            for (int k = 0; k < NK; ++k)
                bad |= (int)!(fabsf(acc[k].x) <= 3.0e38f) | (int)!(fabsf(acc[k].y) <= 3.0e38f) | (int)!(fabsf(acc[k].z) <= 3.0e38f) |
                       (int)!(fabsf(acc[k].w) <= 3.0e38f);
            if (!__syncthreads_or(bad)) break;
        }

        // one store per pixel and chunk: chunk-major gradient (relaid out to NCHW afterwards) or the
        // caller's channels-last gradient (B, H, W, C) in place
        const unsigned y = tby * 4u + wv, x = tbx * 8u + s;
        if (y < (unsigned)height && x < (unsigned)width) {
            const unsigned slice_px = (unsigned)height * (unsigned)pitch;
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                if (!cok[k]) continue;
                float* dst = DST_NHWC
                                 ? gcm + (((size_t)bimg * height + y) * width + x) * (size_t)C + (k0 + k) * kChunk + q * 4u
                                 : gcm + (((size_t)bimg * nchunks + (k0 + k)) * slice_px + (size_t)y * pitch + x) * kChunk + q * 4u;
                *reinterpret_cast<v4f*>(dst) = acc[k];
            }
        }
    }
}
