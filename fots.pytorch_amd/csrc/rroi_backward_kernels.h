// rroi_backward_kernels.h -- backward: gather formulation (K3g), atomic scatter (K3), relayout back to NCHW, direct and literal kernels
// Part of the single translation unit rroi_align_hip.hip (included inside its anonymous
// namespace, in this order: rroi_device_common.h, rroi_forward_kernels.h,
// rroi_backward_kernels.h, rroi_callers_kernels.h); not a standalone header.
#pragma once

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

// list offset of key i after the two-level scan: off[] is local to the key's scan block, `bs` the
// exclusive prefix of the scan blocks' totals
__device__ __forceinline__ unsigned list_offset(const unsigned* __restrict__ off, const unsigned* __restrict__ bs, unsigned i)
{
    return off[i] + bs[i / kScanBlock];
}

// Up to kInlineScanBlocks block totals (256 K keys: any map up to 512 x 512 pixels per image) are
// prefix-summed by every consumer block itself, in LDS, instead of by a launch of their own
// (rroi_scan2_kernel, which stays for larger maps): returns the array list_offset() reads.
constexpr unsigned kInlineScanBlocks = 64;
__device__ __forceinline__ const unsigned* block_prefix(const unsigned* __restrict__ bsum, unsigned nblocks, bool raw,
                                                        unsigned* lds)
{
    if (!raw) return bsum;   // scanned in place by rroi_scan2_kernel
    if (threadIdx.x < kWave) {
        const unsigned v = threadIdx.x < nblocks ? bsum[threadIdx.x] : 0u;
        unsigned incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned o = __shfl_up(incl, d, 64);
            if ((int)threadIdx.x >= d) incl += o;
        }
        lds[threadIdx.x] = incl - v;
    }
    __syncthreads();
    return lds;
}

// Thread -> bin of the pair passes: the 64 lanes of a wave take a PATCH of the bin grid (8 pooled rows x
// 8 columns when PH >= 8), not 64 consecutive bins of one row.  The atomics of the passes are bound by the
// number of 128-byte counter lines an instruction touches; a square patch of bins lands in half as many
// 8 x 4-pixel key tiles as a 64-bin row segment.
struct PatchMap {
    FastDiv div_roi;        // / lanes per ROI
    FastDiv div_npx;        // / patches per pooled row band
    unsigned lanes_per_roi; // npy * npx * 64
    unsigned npx;
    unsigned pc_shift;      // log2(columns of a patch); rows of a patch = 64 >> pc_shift
};

// The overflow side of the bucket lists (MODE 2): a pair that finds its pixel's bucket full goes into one
// global array, chained per pixel (head[key] -> entry -> entry.next ...).
struct BucketLists {
    unsigned kshift;              // bucket of key k: pairs[k << kshift .. ), capacity 1 << kshift
    int* head;                    // per key: index of the newest overflow entry, -1 = none
    unsigned* ovcnt;              // entries handed out so far
    uint4* ov;                    // {bin line, weight bits, next, -}; room for every pair of the problem
};

// MODE 2 with the slot reservations AGGREGATED PER WAVE (round 5).  Device-scope atomics are bound by the number of
// 128-byte line REQUESTS, and a patch of 64 bins issued four atomic instructions (one per tap) that each touched the same
// half dozen 8 x 4-pixel counter tiles: with the reference's C = 64 -- the same bins, a quarter of the relayout's bytes --
// the pair pass set the launch (R = 512, 11 x 96: 58 us where the relayout alone takes 30).  Now a wave first counts its
// <= 256 pairs in a table in LDS -- up to kAggTiles counter tiles x 32 pixels, found with a small open-addressing probe;
// LDS atomics hand out the rank inside the wave --, then reserves the slots with ONE global atomic instruction per tile
// (lane = pixel of the tile: one line request carries all its counts), and every pair's slot is the tile's returned base
// plus its rank.  A pair whose tile finds no room in the table takes its slot alone, as before.  All 64 lanes of the wave
// call this together (lanes without a bin pass npairs = 0).
constexpr unsigned kAggTiles = 16;
constexpr unsigned kAggWords = kAggTiles + 2u * kAggTiles * 32u;   // ids | counts | bases: 1040 words per wave
constexpr unsigned kAggEmpty = 0xffffffffu;
__device__ __forceinline__ void pairs_reserve_wave(unsigned* __restrict__ tab, const unsigned (&keys)[4], unsigned npairs,
                                                   int* __restrict__ cnt, unsigned (&slots)[4])
{
    unsigned* const ids = tab;
    unsigned* const counts = tab + kAggTiles;
    unsigned* const bases = counts + kAggTiles * 32u;
    const unsigned lane = threadIdx.x & 63u;
    // clear: 16 ids + 512 counts (the bases are written before they are read)
    if (lane < kAggTiles) ids[lane] = kAggEmpty;
#pragma unroll
    for (unsigned e = 0; e < kAggTiles * 32u / kWave; ++e) counts[e * kWave + lane] = 0u;
    lds_wave_sync();
    // slots[p] doubles as (table entry | rank << 16) until the bases are known: registers are what keeps this kernel at
    // seven waves per SIMD
    constexpr unsigned kNoEntry = 0xffffu;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        slots[p] = kNoEntry;
        if ((unsigned)p < npairs) {
            const unsigned tile = keys[p] >> 5, in = keys[p] & 31u;
            unsigned h = (tile ^ (tile >> 4)) & (kAggTiles - 1u);
            for (unsigned probe = 0; probe < kAggTiles; ++probe, h = (h + 1u) & (kAggTiles - 1u)) {
                const unsigned old = atomicCAS(ids + h, kAggEmpty, tile);
                if (old == kAggEmpty || old == tile) {
                    slots[p] = h * 32u + in;
                    break;
                }
            }
            if (slots[p] != kNoEntry) slots[p] |= atomicAdd(counts + slots[p], 1u) << 16;   // (a wave has at most 256 pairs)
        }
    }
    lds_wave_sync();
    // one global atomic instruction per pair of table rows: lane = (row of two, pixel); empty rows cost nothing
#pragma unroll
    for (unsigned r2 = 0; r2 < kAggTiles / 2u; ++r2) {
        const unsigned row = r2 * 2u + (lane >> 5), in = lane & 31u;
        const unsigned tile = ids[row];
        const unsigned c = tile != kAggEmpty ? counts[row * 32u + in] : 0u;
        if (c) bases[row * 32u + in] = (unsigned)atomicAdd(cnt + (tile << 5) + in, (int)c);
    }
    lds_wave_sync();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if ((unsigned)p >= npairs) continue;
        slots[p] = (slots[p] & 0xffffu) != kNoEntry ? bases[slots[p] & 0xffffu] + (slots[p] >> 16)
                                                    : (unsigned)atomicAdd(cnt + keys[p], 1);
    }
    lds_wave_sync();   // (the table is cleared again by this wave's next patch)
}

// MODE 0: cnt[key] += 1 per pair (count pass).  MODE 1: cnt counts back down, handing out the slots of
// the key's segment (fill pass, after the scan).  MODE 2 (round 3): ONE pass -- cnt[key]++ hands out the
// slots of a fixed-capacity bucket per pixel (a few times the average list), the rare pair beyond it is
// chained into the overflow array: no count pass, no scan, and the relayout of top_diff is one launch.
template <int MODE>
__device__ __forceinline__ void pairs_body(unsigned idx, const Affine* __restrict__ aff, int num_rois,
                                           int height, int width, int pooled_height, int pooled_width, int batch_size,
                                           unsigned lines_per_roi, const PatchMap& pm,
                                           const KeyLayout& L, int* __restrict__ cnt,
                                           const unsigned* __restrict__ off, const unsigned* __restrict__ bsum,
                                           uint2* __restrict__ pairs, const BucketLists& bl, unsigned* __restrict__ agg = nullptr)
{
    constexpr bool FILL = MODE == 1;
    // (the lanes of a wave share the ROI and the patch: idx is a multiple of 64 plus the lane, the total a multiple of 64)
    const unsigned n = fdiv(idx, pm.div_roi);
    const unsigned rem = idx - n * pm.lanes_per_roi;
    const unsigned patch = rem >> 6, l = rem & 63u;
    const unsigned py = fdiv(patch, pm.div_npx), px = patch - py * pm.npx;
    const unsigned ph = py * (64u >> pm.pc_shift) + (l >> pm.pc_shift);
    const unsigned pw = (px << pm.pc_shift) + (l & ((1u << pm.pc_shift) - 1u));
    const bool has_bin = n < (unsigned)num_rois && ph < (unsigned)pooled_height && pw < (unsigned)pooled_width;
    const unsigned j = ph * (unsigned)pooled_width + pw;
    if (MODE == 2 && !agg) {
        // few patches per wave: one returning atomic per pair as it is found, no table to set up
        if (!has_bin) return;
        const Affine A = aff[n];
        bin_pairs(A, ph, pw, height, width, batch_size, L, [&](unsigned key, float w) {
            const unsigned slot = (unsigned)atomicAdd(cnt + key, 1);
            const uint2 rec = make_uint2(n * lines_per_roi + j, as_u(w));
            if (slot < (1u << bl.kshift)) {
                pairs[((size_t)key << bl.kshift) + slot] = rec;
            } else {
                const unsigned e = atomicAdd(bl.ovcnt, 1u);
                const int prev = atomicExch(bl.head + key, (int)e);
                bl.ov[e] = make_uint4(rec.x, rec.y, (unsigned)prev, 0u);
            }
        });
        return;
    }
    if (MODE == 2) {
        // every lane of the wave gets here: the slots are reserved by the wave as a whole (pairs_reserve_wave)
        unsigned keys[4] = {0u, 0u, 0u, 0u}, np = 0u, slots[4] = {0u, 0u, 0u, 0u};
        float wts[4] = {0.f, 0.f, 0.f, 0.f};
        if (has_bin) {
            const Affine A = aff[n];
            bin_pairs(A, ph, pw, height, width, batch_size, L, [&](unsigned key, float w) {   // at most four calls
                if (np == 0) { keys[0] = key; wts[0] = w; }
                else if (np == 1) { keys[1] = key; wts[1] = w; }
                else if (np == 2) { keys[2] = key; wts[2] = w; }
                else { keys[3] = key; wts[3] = w; }
                ++np;
            });
        }
        pairs_reserve_wave(agg, keys, np, cnt, slots);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if ((unsigned)p >= np) continue;
            const uint2 rec = make_uint2(n * lines_per_roi + j, as_u(wts[p]));
            if (slots[p] < (1u << bl.kshift)) {
                pairs[((size_t)keys[p] << bl.kshift) + slots[p]] = rec;
            } else {
                const unsigned e = atomicAdd(bl.ovcnt, 1u);
                const int prev = atomicExch(bl.head + keys[p], (int)e);
                bl.ov[e] = make_uint4(rec.x, rec.y, (unsigned)prev, 0u);
            }
        }
        return;
    }
    if (!has_bin) return;
    const Affine A = aff[n];
    bin_pairs(A, ph, pw, height, width, batch_size, L, [&](unsigned key, float w) {
        if (!FILL) {
            atomicAdd(cnt + key, 1);
        } else {
            const int slot = atomicAdd(cnt + key, -1) - 1;
            // line index of (roi n, bin j) in the relaid-out top_diff
            pairs[list_offset(off, bsum, key) + (unsigned)slot] = make_uint2(n * lines_per_roi + j, as_u(w));
        }
    });
}

// One launch, two kinds of blocks: [0, pair_blocks) count (FILL = false) or write (FILL = true)
// the pair lists -- bound by the atomic request rate -- and the rest relay out tiles
// [tile_begin, tile_end) of top_diff -- bound by HBM.  They share the chip instead of running
// one after the other; the host gives each of the two launches half of the tiles.
template <int MODE, int SAUX>
__global__ __launch_bounds__(256, 7) void rroi_bwd_pairs_relayout_kernel(   // (seven workgroups per CU: <= 72 VGPRs, as in rounds 3-4)
    const Affine* __restrict__ aff, int num_rois, int height, int width, int pooled_width, int NB,
    int batch_size, unsigned lines_per_roi, PatchMap pm, FastDiv div_pw, KeyLayout L,
    int* __restrict__ cnt, const unsigned* __restrict__ off, const unsigned* __restrict__ bsum,
    uint2* __restrict__ pairs, int pair_blocks, const float* __restrict__ top_diff,
    float* __restrict__ tdT, int C, int nchunks, int ptiles, int relayout_blocks, int tile_begin,
    int tile_end, unsigned scan_blocks, int raw_bsum, BucketLists bl = BucketLists{0u, nullptr, nullptr, nullptr},
    int skip_dead_bins = 1)   // flags: bit 0 dead bins are not copied, bit 1 the pair blocks aggregate their reservations per wave
{
    __shared__ __attribute__((aligned(16))) float T[kChunk * kTP];
    static_assert(4 * kAggWords <= kChunk * kTP, "the four waves' aggregation tables live in the relayout tile");
    if ((int)blockIdx.x < pair_blocks) {
        if (MODE == 1) bsum = block_prefix(bsum, scan_blocks, raw_bsum != 0, reinterpret_cast<unsigned*>(T));
        const unsigned total = (unsigned)num_rois * pm.lanes_per_roi;
        for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += (unsigned)pair_blocks * 256u)
            pairs_body<MODE>(idx, aff, num_rois, height, width, NB / pooled_width, pooled_width, batch_size,
                             lines_per_roi, pm, L, cnt, off, bsum, pairs, bl,
                             (skip_dead_bins & 2) ? reinterpret_cast<unsigned*>(T) + (threadIdx.x >> 6) * kAggWords : nullptr);
        return;
    }
    // block j takes the pixel ranges j, j + blocks, ... of [tile_begin, tile_end), all chunks of each
    // (round 4) a bin that enters no list is not copied: the list builder's own verdict, bin by bin
    auto live_bin = [&](int n, unsigned ph, unsigned pw) {
        if (!(skip_dead_bins & 1)) return true;   // (the exploration build's A/B arm)
        bool any = false;
        bin_pairs(aff[n], ph, pw, height, width, batch_size, L, [&](unsigned, float) { any = true; });
        return any;
    };
    relayout_run<SAUX, true, true>(T, top_diff, tdT, C, NB, pooled_width, pooled_width, div_pw, nchunks, ptiles,
                          tile_begin + ((int)blockIdx.x - pair_blocks) * nchunks, relayout_blocks * nchunks, tile_end,
                          aff, batch_size, live_bin);
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
// BUCKET: the lists are the fixed-capacity buckets of the one-pass build (`off` = the per-key counters, `bsum`
// unused) plus the per-key overflow chains.
// DST: where the gradient goes.  kDstChunkMajor: the chunk-major scratch (relaid out to NCHW by rroi_cm_to_nchw_kernel);
// kDstNhwc: the caller's channels-last bottom_diff in place; kDstNchw / kDstNchwAdd (round 4): the caller's NCHW
// bottom_diff in place (= / +=) -- with sub <= 32 the 256 / sub pixels of a workgroup are whole rows of an 8 x 4 key
// tile (eight consecutive x), their sums change hands through LDS and leave as whole 32-byte sectors of a map row per
// channel: the chunk-major round trip (26 MB written, read, written again at cfg3) and its launch are gone.  The
// host deals the channel passes to blockIdx.y then (a pass = sub / 8 chunks), so that a pixel still takes one round.
enum GatherDst { kDstChunkMajor = 0, kDstNhwc = 1, kDstNchw = 2, kDstNchwAdd = 3 };
template <int DST, bool BUCKET = false>
__global__ __launch_bounds__(256) void rroi_bwd_gather_kernel(
    const float* __restrict__ tdT, const unsigned* __restrict__ off, const unsigned* __restrict__ bsum,
    const uint2* __restrict__ pairs, float* __restrict__ gcm, int C, int height, int width, int pitch,
    int nchunks, unsigned chunk_stride, unsigned line_stride, unsigned sub_shift, KeyLayout L,
    FastDiv div_bt, FastDiv div_wt, unsigned scan_blocks, int raw_bsum,
    BucketLists bl = BucketLists{0u, nullptr, nullptr, nullptr}, unsigned tile_run = 0u)
{
    constexpr bool DST_NHWC = DST == kDstNhwc, TO_NCHW = DST >= kDstNchw;
    constexpr unsigned THREADS = 256u, kLogThreads = 8u;
    __shared__ unsigned bs_lds[kInlineScanBlocks];
    __shared__ float xpose[TO_NCHW ? THREADS * 4 : 4];
    if (!BUCKET) bsum = block_prefix(bsum, scan_blocks, raw_bsum != 0, bs_lds);  // before any thread leaves
    // Workgroup -> keys: the G = 2^gshift workgroups that cover one 8 x 4 key tile get
    // block indices that are equal modulo 8, i.e. run on ONE XCD: neighbouring pixels share
    // source lines (the 2 x 2 footprint of a bin), and only an XCD's own L2 can serve them twice.
    const unsigned gshift = sub_shift + 5u - kLogThreads;   // log2(workgroups per key tile)
    const unsigned bq = blockIdx.x >> 3, xcd = blockIdx.x & 7u;
    // 8 runs of 2^tile_run consecutive tiles (neighbours in x) in flight, one run per XCD.  TO_NCHW stores 32-byte
    // sectors; the four sectors of a 128-byte line belong to four neighbouring tiles, and only if those go through
    // ONE L2 within a few microseconds of each other does the line leave it as one write request instead of four
    const unsigned tq = bq >> gshift;
    const unsigned tile = ((tq >> tile_run) << (3u + tile_run)) + (xcd << tile_run) + (tq & ((1u << tile_run) - 1u));
    const unsigned wg = (tile << gshift) + (bq & ((1u << gshift) - 1u));
    const unsigned tid = wg * THREADS + threadIdx.x;
    const unsigned sub = 1u << sub_shift;              // lanes per pixel (8..64)
    const unsigned sl = tid & (sub - 1u);              // lane within the pixel's group
    const unsigned key = tid >> sub_shift;
    // key -> (b, y, x); the padding of the key space has no list (and, TO_NCHW, stays for the barrier)
    bool live = key < L.keys;
    const unsigned blk = (live ? key : 0u) >> 5, in = key & 31u;
    const unsigned b = fdiv(blk, div_bt);              // / (Ht*Wt)
    const unsigned r = blk - b * (L.Ht * L.Wt);
    const unsigned by = fdiv(r, div_wt);
    const unsigned y = by * 4u + (in >> 3), x = (r - by * L.Wt) * 8u + (in & 7u);
    live = live && y < (unsigned)height && x < (unsigned)width;
    if (!TO_NCHW && !live) return;
    unsigned beg = 0u, end = 0u;
    int chain = -1;   // BUCKET: newest overflow entry of this pixel
    uint2 rec0 = make_uint2(0u, 0u);   // BUCKET: the lane's record of the bucket's first group, fetched ahead
    if (!live) {
    } else if (BUCKET) {
        // (round 5) the walk is a chain of dependent round trips: counter -> records -> data.  A bucket's place does not
        // depend on its counter, so the first group of records is fetched TOGETHER with the counter -- whatever lies
        // beyond the count is never used -- and the chain is one round trip shorter for every pixel.
        if (sl < (1u << bl.kshift)) rec0 = pairs[((size_t)key << bl.kshift) + sl];
        const unsigned have = off[key];
        beg = key << bl.kshift;
        end = beg + min(have, 1u << bl.kshift);
        if (have > (1u << bl.kshift)) chain = bl.head[key];
    } else {
        beg = list_offset(off, bsum, key);
        end = list_offset(off, bsum, key + 1u);
    }
    const unsigned quad = sl & 7u;
    const unsigned slice_px = (unsigned)height * (unsigned)pitch;
    const v4f z4 = {0.f, 0.f, 0.f, 0.f};
#ifndef RROI_GATHER_DEPTH
#define RROI_GATHER_DEPTH 8
#endif
    constexpr int kDepth = RROI_GATHER_DEPTH;
    // channel passes of `sub / 8` chunks each (one pass when C <= 256)
    // (gridDim.y > 1: the channel passes are dealt to blockIdx.y instead of run one after the other)
    const unsigned kstep = gridDim.y * (sub >> 3);
    for (unsigned k0 = blockIdx.y * (sub >> 3); k0 < (unsigned)nchunks; k0 += kstep) {
        const unsigned k = k0 + (sl >> 3);
        const bool c_ok = k < (unsigned)nchunks && k * kChunk + quad * 4u < (unsigned)C;
        // where the 32 channels of chunk k of list entry `line` live: the relaid-out top_diff
        // (chunk_stride = 32, line_stride = nchunks * 32) or a channels-last top_diff consumed in
        // place (chunk_stride = 32, line_stride = C), both in floats
        const float* src = tdT + (size_t)k * chunk_stride + quad * 4u;
        // The walk is a chain of dependent memory round trips (offsets -> records -> data), and
        // the kernel is bound by that chain, not by bytes.  So the records are fetched `sub` at a
        // time -- lane j of the group loads record j, one coalesced access -- and handed round
        // with shuffles; only the data loads remain in the loop.
        // EXACT == false: acc = fma(g, w, acc) and nothing else.  The reference's extra 0 * g of an aliased
        // tap (flag in the weight's sign) changes nothing unless g is not finite -- and then acc is not
        // finite either (every weight is positive): the wave notices at the end of the walk and walks the
        // list again the exact way.  (Round 2: the kernel issues 11.4 M VALU instructions at cfg3, 445 per
        // pixel, half of its time on every SIMD -- the flag's branch and its 8 operations per entry were a
        // third of them; profiles/r02_pmc_bwd_kernels.md.)
        const unsigned group_base = (threadIdx.x & 63u) & ~(sub - 1u);
        auto walk = [&](auto exact_tag) {
            constexpr bool EXACT = decltype(exact_tag)::value;
            v4f acc = z4;
            for (unsigned base = beg; base < end; base += sub) {
                const unsigned m = min(sub, end - base);
                const uint2 rec = (BUCKET && base == beg) ? rec0 : (sl < m ? pairs[base + sl] : make_uint2(0u, 0u));
                for (unsigned j = 0; j < m; j += kDepth) {
                    v4f g[kDepth];
                    unsigned wb[kDepth];
#pragma unroll
                    for (int d = 0; d < kDepth; ++d) {
                        const int from = (int)(group_base + j + d);
                        const unsigned line = (unsigned)__shfl((int)rec.x, from, kWave);
                        wb[d] = (unsigned)__shfl((int)rec.y, from, kWave);
                        g[d] = (c_ok && j + d < m) ? *reinterpret_cast<const v4f*>(src + (size_t)line * line_stride) : z4;
                    }
#pragma unroll
                    for (int d = 0; d < kDepth; ++d) {
                        if (j + d < m) {
                            // kernel.cu:260-263: v_k = w_k * top_diff, then one add per tap
                            const float w = as_f(wb[d] & 0x7fffffffu);
                            if (EXACT) {
                                acc += g[d] * w;
                                if (wb[d] & 0x80000000u) acc += g[d] * 0.0f;
                            } else {
                                acc.x = __builtin_fmaf(g[d].x, w, acc.x);
                                acc.y = __builtin_fmaf(g[d].y, w, acc.y);
                                acc.z = __builtin_fmaf(g[d].z, w, acc.z);
                                acc.w = __builtin_fmaf(g[d].w, w, acc.w);
                            }
                        }
                    }
                }
            }
            if (BUCKET) {
                // the pixel's overflow chain (rare: a list longer than the bucket), one entry at a time
                for (int e = chain; e >= 0;) {
                    const uint4 rec = bl.ov[e];
                    const v4f g = c_ok ? *reinterpret_cast<const v4f*>(src + (size_t)rec.x * line_stride) : z4;
                    const float w = as_f(rec.y & 0x7fffffffu);
                    if (EXACT) {
                        acc += g * w;
                        if (rec.y & 0x80000000u) acc += g * 0.0f;
                    } else {
                        acc.x = __builtin_fmaf(g.x, w, acc.x);
                        acc.y = __builtin_fmaf(g.y, w, acc.y);
                        acc.z = __builtin_fmaf(g.z, w, acc.z);
                        acc.w = __builtin_fmaf(g.w, w, acc.w);
                    }
                    e = (int)rec.z;
                }
            }
            return acc;
        };
        v4f acc = walk(std::false_type{});
        {
            const bool bad = !(fabsf(acc.x) <= 3.0e38f) || !(fabsf(acc.y) <= 3.0e38f) || !(fabsf(acc.z) <= 3.0e38f) ||
                             !(fabsf(acc.w) <= 3.0e38f);
            if (__ballot(bad)) acc = walk(std::true_type{});   // rare: a gradient that is not finite
        }
        if (TO_NCHW) {
            // (pixel, 4 channels) per thread in, (channel, 4 consecutive x) per thread out: lanes 2i and 2i + 1 store
            // the two halves of one 32-byte sector of channel i's map row
            *reinterpret_cast<v4f*>(xpose + threadIdx.x * 4u) = acc;   // [pixel of the workgroup][channel of the pass]
            __syncthreads();
            const unsigned t = threadIdx.x, cp = sub * 4u;             // channels per pass
            const unsigned c = (t >> 1) & (cp - 1u), row = t >> (sub_shift + 3u), pw0 = row * 8u + (t & 1u) * 4u;
            const float* tr = xpose + pw0 * cp + c;
            v4f v = {tr[0], tr[cp], tr[2u * cp], tr[3u * cp]};
            // the workgroup's first key is the first of a tile row: the block is the same for all its pixels
            const unsigned key0 = (wg * THREADS) >> sub_shift;
            const unsigned in0 = (key0 & 31u) + pw0;
            const unsigned blk0 = (key0 < L.keys ? key0 : 0u) >> 5;
            const unsigned b0 = fdiv(blk0, div_bt), r0 = blk0 - b0 * (L.Ht * L.Wt), by0 = fdiv(r0, div_wt);
            const unsigned y0 = by0 * 4u + (in0 >> 3), x0 = (r0 - by0 * L.Wt) * 8u + (in0 & 7u);
            const unsigned cg = k0 * kChunk + c;
            if (key0 < L.keys && cg < (unsigned)C && y0 < (unsigned)height && x0 < (unsigned)width) {
                float* o = gcm + (((size_t)b0 * C + cg) * height + y0) * (size_t)width + x0;
                if ((width & 3) == 0 && (reinterpret_cast<uintptr_t>(gcm) & 15) == 0) {
                    v4f* o4 = reinterpret_cast<v4f*>(o);   // x0 % 4 == 0 and W % 4 == 0: inside together
                    if (DST == kDstNchwAdd) v += *o4;
                    *o4 = v;
                } else {
                    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (x0 + j < (unsigned)width) o[j] = DST == kDstNchwAdd ? o[j] + e[j] : e[j];
                }
            }
            if (k0 + kstep < (unsigned)nchunks) __syncthreads();   // the next pass reuses the tile
        } else if (c_ok) {
            // chunk-major gradient (relaid out to NCHW afterwards), or the caller's channels-last
            // gradient (B, H, W, C) written directly: `gcm` is then bottom_diff itself
            float* dst = DST_NHWC
                             ? gcm + (((size_t)b * height + y) * width + x) * (size_t)C + k * kChunk + quad * 4u
                             : gcm + (((size_t)b * nchunks + k) * slice_px + (size_t)y * pitch + x) * kChunk + quad * 4u;
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
// ACCUM (the reference-ABI launcher): bottom_diff += the gradient, as the reference's atomicAdds onto its
// zeroed buffer do (kernel.cu:260-274) -- a caller that accumulates over several calls gets the sum
template <bool ACCUM = false>
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
    // rows of 16-byte aligned float4 (p0 is a multiple of 128): four 16-byte stores per thread instead of sixteen
    // dwords -- a wave instruction writes two 512-byte runs (the prologue's read mapping, mirrored)
    if ((HW & 3) == 0 && (reinterpret_cast<uintptr_t>(nchw) & 15) == 0) {
        const int x4 = lane & 31, csub = lane >> 5;
        const int p = 4 * x4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = w * 8 + i * 2 + csub;
            if (c0 + c < C && p0 + p < HW) {   // HW % 4 == 0: the four pixels are inside together
                const float* tr = T + c * (kRelayoutPx + 1) + p;
                v4f v = {tr[0], tr[1], tr[2], tr[3]};
                v4f* o = reinterpret_cast<v4f*>(dst + (size_t)c * HW + p);
                if (ACCUM) v += *o;
                *o = v;
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = w * 8 + i;
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf) {
            const int p = hlf * 64 + lane;
            if (c0 + c < C && p0 + p < HW) {
                float* o = dst + (size_t)c * HW + p;
                const float v = T[c * (kRelayoutPx + 1) + p];
                *o = ACCUM ? *o + v : v;
            }
        }
    }
}

// Backward, direct NCHW (small R): thread = (roi, bin), loops a channel slab.
__global__ __launch_bounds__(256) void rroi_bwd_direct_kernel(
    const float* __restrict__ top_diff, const float* __restrict__ rois,
    float* __restrict__ bottom_diff, int num_rois, int C, int height, int width,
    int pooled_height, int pooled_width, float spatial_scale, int trig, int batch_size, int cslab)
{
    const int NB = pooled_height * pooled_width;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)num_rois * NB) return;
    const int n = (int)(gid / NB);
    const int bin = (int)(gid - (long)n * NB);
    const int ph = bin / pooled_width, pw = bin - ph * pooled_width;
    const Affine A = make_affine(rois + (size_t)n * 6, pooled_height, spatial_scale, trig);
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

