// rroi_forward_kernels.h -- forward: prologue relayout (K0), tiled gather (K1), direct small-R kernel (K2)
// Part of the single translation unit rroi_align_hip.hip (included inside its anonymous
// namespace, in this order: rroi_device_common.h, rroi_forward_kernels.h,
// rroi_backward_kernels.h, rroi_callers_kernels.h); not a standalone header.
#pragma once

// ------------------------------------------------------------------------------------
// K0: forward prologue, one launch:
//   blocks [0, relayout_blocks)      NCHW -> chunk-major: a [32 ch] x [128 px] tile goes
//                                    through LDS; reads are 512 B runs of a channel row,
//                                    writes are one contiguous 16 KiB run of the slice;
//   then the rest                    per-ROI affine table (R x 32 B).
// (Every slice of the copy still ENDS in one spare pixel -- rounds 1-3 kept a zero pixel there for invalid taps;
// they have pointed at out-of-range descriptor offsets since round 1 and nothing reads it, so nothing writes it.)
// ------------------------------------------------------------------------------------
constexpr int kRelayoutPx = 128;

// The relayout proper, shared by the forward prologue (feature map) and the backward (top_diff,
// R "images" of PH x PW "pixels").  MASK: image b is ROI b and pixels (ph, pw) with
// pw > roi_pooled_width (or every pixel of a ROI with an invalid batch index) are bins the
// forward masks -- nothing reads them again, so they are neither loaded nor written.
// LIVE (a predicate (image, y, x) -> bool, or NoLive): the backward's sharper form of the same -- a bin none of
// whose taps passes the backward's bounds (kernel.cu:267-274: the parts of a ROI that hang over the map's edge)
// enters no pixel's list, so its gradient is neither loaded nor written either.  The predicate must be the list
// builder's own (bin_pairs): a bin it calls dead is a line of the copy that holds whatever the workspace held.
struct NoLive {
    __device__ __forceinline__ bool operator()(int, unsigned, unsigned) const { return true; }
};
constexpr int kTP = kRelayoutPx + 4;

// PIXMAJOR: the copy is (image, pixel, nchunks * 32) -- all chunks of a pixel in one nchunks * 128-byte
// run -- instead of (image, chunk, pixel, 32).  The backward's copy of top_diff uses it: its gather
// reads the 8 chunks of a bin together, and eight lines of one DRAM page cost less than eight lines 65 KB
// apart (cfg3: gather 52 -> 39 us).
template <int AUX, bool MASK, bool PIXMAJOR = false, class Live = NoLive>
__device__ __forceinline__ void relayout_run(float* __restrict__ T, const float* __restrict__ nchw,
                                               float* __restrict__ cm, int C, int HW, int width, int pitch,
                                               FastDiv div_w, int nchunks, int ptiles, int first_tile,
                                               int tile_stride, int relayout_tiles,
                                               const Affine* __restrict__ mask_aff, int mask_batches,
                                               Live is_live = Live{})
{
    constexpr bool LIVE = !std::is_same<Live, NoLive>::value;
    static_assert(!LIVE || PIXMAJOR, "the predicate is evaluated once per pixel range, on its first chunk");
    // (LIVE) one bit per pixel of a 128-pixel range, evaluated by every wave for itself on the range's first chunk
    // (two pixels per lane, two ballots): the range being loaded, and the range being stored (loaded a chunk earlier)
    unsigned long long next_lo = ~0ull, next_hi = ~0ull, cur_lo = ~0ull, cur_hi = ~0ull;
    // [32 ch][128 px] tile, 132-float pitch (16-byte aligned rows for the b128 writes); the
    // pixel index of rows 8m..8m+7 is XORed with 4m so that the transposed ds_read_b32 of
    // phase 2 (8 channel quads x 4 pixels per 32-lane group) hits 32 different banks.
    const int tid = threadIdx.x;
    const size_t zp_index = (size_t)(HW / width) * pitch;  // pixel index of the slice's spare last pixel
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
            if (LIVE) {
                if (k == 0) {   // the chunks of a pixel range follow one another: the verdict holds for all of them
                    const unsigned g0 = (unsigned)(p0 + lane), g1 = g0 + 64u;
                    const unsigned y0 = fdiv(g0, div_w), y1 = fdiv(g1, div_w);
                    next_lo = __ballot(g0 < (unsigned)HW && is_live(b, y0, g0 - y0 * (unsigned)width));
                    next_hi = __ballot(g1 < (unsigned)HW && is_live(b, y1, g1 - y1 * (unsigned)width));
                }
                const unsigned long long half = x4 < 16 ? next_lo : next_hi;
                live = live && ((half >> (4 * (x4 & 15))) & 0xFull) != 0ull;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = w * 8 + i * 2 + csub;
            v4f v = {0.f, 0.f, 0.f, 0.f};
            if (c0 + c < C && live) {
                const float* sp = src + (size_t)c * HW + p;
                if (vec_ok && p0 + p + 3 < HW) {
                    // PIXMAJOR (the backward's top_diff): read once, streaming -- the copy, not the source,
                    // should be what the memory-side cache holds when the gather starts
                    v = PIXMAJOR ? __builtin_nontemporal_load(reinterpret_cast<const v4f*>(sp))
                                 : *reinterpret_cast<const v4f*>(sp);
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
    // PIXMAJOR: a block takes the nchunks tiles of one pixel range one after the other (first_tile and
    // tile_stride are multiples of nchunks), so that what it writes is one contiguous run of the copy
    int tile = first_tile;
    auto advance = [&](int t) {
        if (!PIXMAJOR) return t + tile_stride;
        return (t + 1) % nchunks ? t + 1 : t + 1 - nchunks + tile_stride;
    };
    // gfx950 counts loads and stores with ONE in-order vmcnt.  Inside the loop the tile in `r` was loaded
    // BEFORE the previous tile's four stores were issued, so waiting for it needs vmcnt(4), not vmcnt(0) --
    // but the compiler takes the most conservative count over the paths that reach the loop head, and on
    // the way in there are no stores yet: it emitted vmcnt(0), and every tile waited for the previous
    // tile's stores to be acknowledged.  Four stores that go nowhere make the two paths alike.
    if (tile < relayout_tiles) {
        load_tile(tile);
        const __amdgpu_buffer_rsrc_t nowhere = make_rsrc(cm, 0u);
#pragma unroll
        for (int j = 0; j < 4; ++j) buf_store<AUX>(nowhere, kOOB + 16u * j, v4f{0.f, 0.f, 0.f, 0.f});  // four distinct stores
    }
    while (tile < relayout_tiles) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = w * 8 + i * 2 + csub;
            *reinterpret_cast<v4f*>(T + c * kTP + ((4 * x4) ^ ((c >> 3) * 4))) = r[i];
        }
        __syncthreads();
        const int cur = tile;
        if (LIVE && cur % nchunks == 0) {   // the range now stored is the one whose first chunk was loaded last
            cur_lo = next_lo;
            cur_hi = next_hi;
        }
        tile = advance(tile);
        if (tile < relayout_tiles) load_tile(tile);
        {
            const int k = cur % nchunks;
            const int pt = (cur / nchunks) % ptiles;
            const int b = cur / (ptiles * nchunks);
            const int p0 = pt * kRelayoutPx;
            const size_t img_bytes = (size_t)HW * nchunks * kLineBytes;   // PIXMAJOR: one image of the copy
            float* dst = PIXMAJOR ? cm + (size_t)b * (img_bytes / 4) + (size_t)k * kChunk
                                  : cm + ((size_t)b * nchunks + k) * slice_stride;
            const unsigned px_floats = PIXMAJOR ? (unsigned)nchunks * kChunk : (unsigned)kChunk;
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
                // every wave issues its four stores on every path (a pixel that is not written is an
                // out-of-range offset, dropped by the descriptor check): see the note on s_waitcnt below
                const bool ok = p0 + p < HW && !(MASK && (float)x > lim) && (!LIVE || (((p < 64 ? cur_lo : cur_hi) >> (p & 63)) & 1ull) != 0ull);
                const __amdgpu_buffer_rsrc_t ws =
                    make_rsrc(dst, PIXMAJOR ? (unsigned)(img_bytes - (size_t)k * kLineBytes) : (unsigned)(slice_stride * 4));
                buf_store<AUX>(ws, ok ? (unsigned)((pix * px_floats + cq * 4) * 4) : kOOB, v);
            }
        }
        __syncthreads();
    }
}

// one (roi, bin) over the channels [c_begin, c_end): geometry once, then the reference's four taps per channel
__device__ __forceinline__ void direct_bin(const float* __restrict__ feat, const Affine& A, float* __restrict__ out,
                                           float* __restrict__ idx_x, float* __restrict__ idx_y, int n, int bin, int C,
                                           int height, int width, int pooled_width, int NB, int batch_size,
                                           int c_begin, int c_end)
{
    const int ph = bin / pooled_width, pw = bin - ph * pooled_width;
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

// XCD GROUPS (round 5; C < 256).  Workgroup b runs on XCD b % 8 (observed; it only ever changes speed).  With nchunks = 8
// chunk k lives on XCD k: the prologue's blocks of XCD k write slice k through that XCD's L2 and the gather's blocks of
// XCD k read it there.  With FEWER chunks (the reference's own call: C = 64, two chunks) rounds 1-4 dealt chunk k to
// the XCDs x % nchunks == k and every item to every one of them: a chunk's slices were written through 8 / nchunks L2s
// and gathered from all of them -- each XCD's working set was the whole chunk of every image (4.9 MB at C = 64, two
// 120 x 160 maps: more than an L2), fetched from the memory-side cache first.  Now the G = 8 / nchunks XCDs of a chunk
// split the WORK spatially: group g = x / nchunks relays out the g-th G-quantile of the (image, map row) space and
// gathers the g-th G-quantile of the ROIs sorted by (image, centre row) -- `order`, built by one block of the
// prologue launch with a counting sort (rroi_sort_rois).  For ROIs spread evenly over the images the two quantiles
// coincide: an XCD gathers what it relaid out, and its working set is 1 / G of a chunk.  For any other spread the
// ROI quantiles keep the load balanced and only the first touch comes from farther away.
struct XcdGroups {
    int G;                        // groups per chunk: 8 / nchunks (the host: for one or two chunks and enough ROIs), else 1 (= rounds 1-4)
    const int* order;             // G > 1: ROI index by sorted position
};

// Counting sort of the ROIs by key = (image, band of the centre's map row): one workgroup, LDS histogram (`hist`:
// kSortBuckets + 1 words), ranks by LDS atomics -- the order INSIDE a bucket depends on the atomics' order, which only
// ever changes which workgroup processes a ROI.  ROIs whose image index is not in [0, batch_size) sort last.
// The block must not outlast the relayout blocks of its launch (the prologue at C = 64 is 5 us): a thread keeps the
// keys and ranks of its first kSortRegs ROIs in registers (R <= 1024: no round trip through `rank`), the histogram has
// 1024 buckets (4 per thread in the scan).
constexpr int kSortBuckets = 1024;
constexpr int kSortRegs = 4;
__device__ __forceinline__ void rroi_sort_rois(unsigned* hist, const float* __restrict__ rois, int num_rois, int batch_size,
                                               int height, float spatial_scale, int* __restrict__ rank, int* __restrict__ order)
{
    const int tid = threadIdx.x;
    int bands = kSortBuckets / batch_size;           // row bands per image (>= 1 while batch_size <= kSortBuckets)
    if (bands > height) bands = height;
    if (bands < 1) bands = 1;
    const bool by_image_only = bands * batch_size > kSortBuckets;   // more images than buckets
    const int nb = (by_image_only ? kSortBuckets : bands * batch_size) + 1;   // + the invalid bucket
    constexpr int kPer = (kSortBuckets + 1 + 255) / 256;   // buckets per thread in the scan
    for (int i = tid; i < nb; i += 256) hist[i] = 0u;
    auto key_of = [&](int n) -> int {
        const float* r = rois + (size_t)n * 6;
        const int b = f2i_sat(r[0]);
        if (b < 0 || b >= batch_size) return nb - 1;
        if (by_image_only) return b % (nb - 1);
        const float y = r[2] * spatial_scale;                       // the centre's map row (NaN -> band 0, +-inf -> an end)
        int band = f2i_sat((y / (float)height) * (float)bands);
        band = band < 0 ? 0 : (band >= bands ? bands - 1 : band);
        return b * bands + band;
    };
    int key[kSortRegs], rk[kSortRegs];
#pragma unroll
    for (int e = 0; e < kSortRegs; ++e) {
        const int n = tid + e * 256;
        key[e] = n < num_rois ? key_of(n) : 0;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < kSortRegs; ++e)
        if (tid + e * 256 < num_rois) rk[e] = (int)atomicAdd(&hist[key[e]], 1u);
    for (int n = tid + kSortRegs * 256; n < num_rois; n += 256) rank[n] = (int)atomicAdd(&hist[key_of(n)], 1u);
    __syncthreads();
    // exclusive scan of the histogram: 256 threads x kPer consecutive buckets + a block scan of the partial sums
    {
        unsigned mine[kPer], sum = 0;
#pragma unroll
        for (int e = 0; e < kPer; ++e) {
            const int i = tid * kPer + e;
            mine[e] = i < nb ? hist[i] : 0u;
            sum += mine[e];
        }
        unsigned incl = sum;
        const unsigned lane = tid & 63u, wv = tid >> 6;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned o = __shfl_up(incl, d, 64);
            if (lane >= (unsigned)d) incl += o;
        }
        __shared__ unsigned wsum[4];
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();     // (every hist[] value has been read into registers)
        unsigned run = incl - sum + (wv > 0 ? wsum[0] : 0u) + (wv > 1 ? wsum[1] : 0u) + (wv > 2 ? wsum[2] : 0u);
#pragma unroll
        for (int e = 0; e < kPer; ++e) {
            const int i = tid * kPer + e;
            if (i < nb) hist[i] = run;
            run += mine[e];
        }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < kSortRegs; ++e)
        if (tid + e * 256 < num_rois) order[hist[key[e]] + (unsigned)rk[e]] = tid + e * 256;
    for (int n = tid + kSortRegs * 256; n < num_rois; n += 256) order[hist[key_of(n)] + (unsigned)rank[n]] = n;
}

template <int AUX>
__global__ __launch_bounds__(256) void rroi_prologue_kernel(
    const float* __restrict__ nchw, float* __restrict__ cm, int C, int HW, int width, int pitch,
    FastDiv div_w, int nchunks, int ptiles, int relayout_blocks, int relayout_tiles,
    int batch_size, const float* __restrict__ rois, int num_rois, int pooled_height,
    float spatial_scale, int trig, Affine* __restrict__ aff, int aff_blocks = 0, float* __restrict__ rest_out = nullptr,
    int pooled_width = 0, int groups = 1, int* __restrict__ sort_rank = nullptr, int* __restrict__ sort_order = nullptr)
{
    __shared__ __attribute__((aligned(16))) float T[kChunk * kTP];
    static_assert(sizeof(T) >= (kSortBuckets + 1) * sizeof(unsigned), "the sort's histogram lives in the relayout tile");
    const int tid = threadIdx.x;
    // block order: [relayout][affine table][ROI sort (groups > 1)][the launcher's rest-of-the-images blocks]
    const int sort_blocks = groups > 1 ? 1 : 0;
    if (sort_blocks && (int)blockIdx.x == relayout_blocks + aff_blocks) {
        rroi_sort_rois(reinterpret_cast<unsigned*>(T), rois, num_rois, batch_size, HW / width, spatial_scale, sort_rank,
                       sort_order);
        return;
    }
    if (rest_out && (int)blockIdx.x >= relayout_blocks + aff_blocks + sort_blocks) {
        // The reference-ABI launcher (one more block per ROI).  Its signature does not say how many images
        // `nchw` holds, so the copy and the tiled gather serve image 0; the ROIs of images >= 1 -- none, as a
        // rule: the block reads the index and leaves -- are sampled here from the NCHW tensor, trusting the
        // index as the reference does, and the gather leaves their crops alone.
        const int n = (int)blockIdx.x - relayout_blocks - aff_blocks - sort_blocks;
        if (f2i_sat(rois[(size_t)n * 6]) < batch_size) return;
        const Affine A = make_affine(rois + (size_t)n * 6, pooled_height, spatial_scale, trig);
        const int NB = pooled_height * pooled_width;
        for (int bin = tid; bin < NB; bin += 256)
            direct_bin(nchw, A, rest_out, nullptr, nullptr, n, bin, C, HW / width, width, pooled_width, NB, /*trust*/ -1, 0, C);
        return;
    }
    if ((int)blockIdx.x >= relayout_blocks) {
        const int n = ((int)blockIdx.x - relayout_blocks) * 256 + tid;
        if (n < num_rois) aff[n] = make_affine(rois + (size_t)n * 6, pooled_height, spatial_scale, trig);
        return;
    }
    if (groups > 1) {
        // XCD x = block % 8 serves chunk x % nchunks and the g-th quantile (g = x / nchunks) of the spatial tiles
        // sp = image * ptiles + pixel tile; tile index = sp * nchunks + chunk (relayout_run decodes it)
        const int x = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3, per_xcd = relayout_blocks >> 3;
        const int k = x % nchunks, g = x / nchunks;
        const int SP = relayout_tiles / nchunks;
        const int lo = (int)(((long)g * SP) / groups), hi = (int)(((long)(g + 1) * SP) / groups);
        relayout_run<AUX, false>(T, nchw, cm, C, HW, width, pitch, div_w, nchunks, ptiles, (lo + j) * nchunks + k,
                                   per_xcd * nchunks, hi * nchunks, nullptr, 0);
        return;
    }
    relayout_run<AUX, false>(T, nchw, cm, C, HW, width, pitch, div_w, nchunks, ptiles, (int)blockIdx.x,
                               relayout_blocks, relayout_tiles, nullptr, 0);
}

// per-ROI affine table; the backward also has it clear its pixel counters (`zero`, `nzero` ints)
// instead of paying a memset launch of their own
// (bucket lists, round 3: also the overflow chains' heads = -1 and the overflow counter = 0)
__global__ void rroi_affine_kernel(const float* __restrict__ rois, int num_rois, int pooled_height,
                                   float spatial_scale, int trig, Affine* __restrict__ aff, int* __restrict__ zero = nullptr,
                                   unsigned nzero = 0, int* __restrict__ heads = nullptr,
                                   unsigned* __restrict__ counter = nullptr)
{
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    for (unsigned j = gid; j < nzero; j += gridDim.x * blockDim.x) {
        zero[j] = 0;
        if (heads) heads[j] = -1;
    }
    if (counter && gid == 0) *counter = 0u;
    if (gid < (unsigned)num_rois) aff[gid] = make_affine(rois + (size_t)gid * 6, pooled_height, spatial_scale, trig);
}

// ------------------------------------------------------------------------------------
// K1: the hot kernel, rroi_fwd_split_kernel.  A workgroup is two waves on one LDS tile; block -> channel chunk
// k = blockIdx % nchunks (XCD affinity) and a walk over (roi, 64-bin tile) items:
//   phase A  (storer wave) lane = bin: geometry -> one 16-byte tap record per bin in LDS, sorted by class (bins
//            with <= 2 distinct taps, bins with 4); an invalid tap is the out-of-range offset kOOB, which the
//            buffer descriptor turns into 0.0.  No memory access: it fills the time the storer would idle.
//   phase B  (gatherer wave) lane = (bin b of 8, channel quad q of 8): 2 or 4 buffer loads per group of 8 bins,
//            blend, transpose through the LDS tile; the loads of group g + 1 are in flight while group g is blended.
//   phase C  (storer wave) the [32 ch][64 bin] tile leaves LDS as 16-byte stores, 256 B per row (or, ONHWC, 8 bins x
//            128 B of channels-last crops), seven of eight streaming, one write-through.
// gfx950 counts loads and stores with ONE in-order vmcnt: a wave that does both cannot consume a load issued behind
// its stores before those stores are acknowledged (microseconds while 256 MiB stream out).  Rounds 1-2 ordered the
// phases of a one-wave kernel around that (rroi_fwd_tiled_kernel: retired in round 4, see profiles/NOTEBOOK.md);
// here no wave does both.
// Template parameters: VEC_STORE 16-byte stores (PH * PW % 4 == 0) | EARLY LO groups issued before barrier 2 | OCC
// waves per SIMD (__launch_bounds__) | HID HI groups double-buffered unrolled (2) or rolled (3) | ONHWC channels-last
// crops | SHIFT crops whose rows are not whole 64-byte sectors: overlapped tiles and sector-aligned (1) or, for crops beyond
// the memory-side cache, line-aligned (2) store windows, any row offset (see drain_shift, drain_shift2) | NCHW_SRC the one-launch form over the caller's NCHW map (below) | WAUX >= 0 the MERGING form for
// such crops under XCD groups: strided tiles, every store with that one cache policy (0 = plain write-back: the XCD's L2 puts a
// row's partial sectors together), 16-byte stores at dword alignment.  dbg: bit 0 drops the stores, bit 1
// the tap loads (ablations), bit 5 the reference-ABI launcher's mode.  (The per-workgroup time stamps, the
// contiguous-items knob and the free-first-item ablation of round 4 are tools/experiments/r04_wg_trace_instrumentation.patch.)
// Shipped instantiations: the switch in forward_impl (rroi_align_hip.hip).
// ------------------------------------------------------------------------------------
constexpr int kStoreAux = 2;      // output stores stream (nt) ...
constexpr int kMinorStores = 1;   // ... except this many of a tile's eight, which go out write-through (kMinorAux)
// The SHIFT form's windows are HALF lines (16 rows x 64 B per instruction): streamed, each half is a write request of
// its own (4.36 TB/s for the pattern, store-only); write-through (sc1) lets the L2 put a line's halves together:
// 6.5 TB/s (tools/kbench win48, profiles/r04_win48_store_policy.txt; plain write-back 5.9-6.1)
#ifndef RROI_SHIFT_AUX
#define RROI_SHIFT_AUX 16
#endif
constexpr int kShiftAux = RROI_SHIFT_AUX;
#ifndef RROI_SHIFT2_AUX
#define RROI_SHIFT2_AUX 2
#endif
constexpr int kShift2Aux = RROI_SHIFT2_AUX;   // the line-aligned windows (SHIFT == 2) stream: whole lines (tools/kbench bigout: nt 4.3-5.0 TB/s, sc1 4.0-4.8)
// Workgroup barrier that orders LDS traffic only: s_barrier does not wait for vector memory, and unlike
// __syncthreads() nothing here makes the compiler emit s_waitcnt vmcnt(0).
__device__ __forceinline__ void wg_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }


// NCHW_SRC (round 5, RROI_PATH_FUSED): the ONE-LAUNCH form for few ROIs -- the reference's own call shapes: at most 32 ROIs
// per training step (src/ocr_process.py:253-255), 1 ... 24 per image in inference.  No prologue launch at all: `map` is the
// caller's NCHW tensor, a "slice" is the 32 channel PLANES of chunk k, a lane's four channels of a pixel are four dword
// loads a plane apart instead of one 16-byte load (four times the load instructions per item -- the price of not relaying
// out a whole map for a handful of crops; it is the address rate that bounds this form, so it serves few ROIs only), and the
// storer wave evaluates the affine of its item's ROI itself (`roi_src`).  Everything else -- records, class-sorted groups,
// blend, tile, stores -- is the two-launch form's, bit for bit.
struct RoiSource {   // NCHW_SRC: where the storer takes its affines from
    const float* rois;
    int pooled_height;
    float spatial_scale;
    int trig;
};
template <bool VEC_STORE, int EARLY, int OCC, int HID, bool ONHWC, int SHIFT, bool NCHW_SRC = false, int WAUX = -1>
__global__ __launch_bounds__(2 * kWave, OCC) void rroi_fwd_split_kernel(
    const float* __restrict__ map, const Affine* __restrict__ aff, float* __restrict__ out,
    int num_rois, int C, int height, int width, int pooled_width, int NB, int batch_size,
    int nchunks, int ntiles, SliceLayout lay, FastDiv div_tiles, FastDiv div_pw, int dbg, XcdGroups xg,
    RoiSource roi_src = RoiSource{nullptr, 0, 0.0f, 0})
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
    // two classes, each padded to a multiple of 8: ceil(a/8) + ceil(b/8) <= 9 for a + b <= 64 (and one
    // forced LO group + 8 HI groups when a = 0)
    constexpr int kMaxGroups = kIters + 1;
    // LO groups whose loads are issued before barrier 2, besides the first one (0 in the NCHW forms: no difference
    // measured with the loads in a wave of their own; 2 in the channels-last form, round 3's first shape)
    constexpr int kEarly = EARLY;
    constexpr unsigned kPadPos = kTileBins;  // "bin position" of a padding record
    // T: rows 0..31 are the tile; the tail absorbs the writes of padding records (4 rows at the
    // tile's pitch, 32 columns: 32 different banks).  LDS is granted in 1280-byte granules on gfx950;
    // the block (T 9648 + records 2448 = 12096 B) stays within the 10 granules that 12 waves per CU
    // allow (a smaller block that admits 14 was measured in round 2 and is no faster: the kernel is bound by the
    // write path, not by latency; profiles/NOTEBOOK.md 5.2 -- and again in round 5 at the 64-channel shapes, without the
    // pad area, 11,192 B, seven waves per SIMD: R = 512, 11 x 96 33.2 against 32.6-32.9 us, 11 x 83 32.0 against 30.2-31.0).
    // SHIFT: the tile is BIN-MAJOR instead -- T[lane's bin * 32 + channel], 64 gathered bins of which the LAST 48 are the
    // tile's own (the first 16 are the tile before's last: every item is self-contained, see drain_shift); row 64 takes
    // the padding records, rows 65..79 are never written (a window position beyond the gathered bins reads them and
    // is never stored).  10240 B; with the records 12720 B = 10 granules, like the channel-major tile
    // SHIFT == 2 (round 5, crops beyond the memory-side cache): LINE-aligned windows instead -- a tile advances by 32 bins and
    // gathers 64 (its own 32 and the 32 in front: all a window shifted by h <= 31 can reach), every store instruction is
    // eight rows x one whole 128-byte line (see drain_shift2); rows up to 95 can be read (and are never stored): 12288 B
    constexpr int kFront = SHIFT == 2 ? 32 : SHIFT ? 16 : 0;       // gathered bins in front of a tile's own
    constexpr unsigned kBmRows = kTileBins + kFront;
    constexpr int kOwnBins = kTileBins - kFront;                   // bins a tile stores / advances by
    static_assert(SHIFT >= 0 && SHIFT <= 2, "SHIFT: 0 none, 1 sector-aligned windows, 2 line-aligned windows");
    __shared__ __attribute__((aligned(16))) float T[SHIFT ? kBmRows * kChunk : kChunk * kTStride + 3 * kTStride + 32];
    // tap records of two items: item i+1 is sampled out of one set while the other is being
    // built for item i+2
    constexpr int kRecs = kMaxGroups * kBinsPerIter;
    __shared__ __attribute__((aligned(16))) uint4 Gbuf[2 * kRecs];
    __shared__ unsigned char HPbuf[2 * kRecs];
    __shared__ uint4 shead[2];  // per record set: LO groups, HI groups, mask of the bins that are in a group
    __shared__ int sbatch[2];   // ... and the item's image index (the gatherer takes it from here, not from memory)

    const unsigned lane = threadIdx.x & 63u;
    // wave 0 gathers (loads only), wave 1 streams the finished tiles out (stores only)
    const bool storer = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 1;
    // block -> (chunk, slot) and the ROIs the slot's items come from.  One group: chunk = block % nchunks, every
    // (roi, tile) item of the call, dealt every nslots-th.  XCD groups (see XcdGroups): XCD x = block % 8 -> chunk
    // x % nchunks, group g = x / nchunks, the items of the sorted ROIs [r0, r1) dealt to the XCD's gridDim / 8 blocks.
    unsigned k, slot, nslots, r0 = 0u, r1 = (unsigned)num_rois;
    if (xg.G > 1) {
        const unsigned x = blockIdx.x & 7u, g = x / (unsigned)nchunks;
        k = x % (unsigned)nchunks;
        slot = blockIdx.x >> 3;
        nslots = gridDim.x >> 3;
        r0 = (unsigned)(((unsigned long long)g * (unsigned)num_rois) / (unsigned)xg.G);
        r1 = (unsigned)(((unsigned long long)(g + 1u) * (unsigned)num_rois) / (unsigned)xg.G);
    } else {
        k = blockIdx.x % (unsigned)nchunks;
        slot = blockIdx.x / (unsigned)nchunks;
        nslots = gridDim.x / (unsigned)nchunks;
    }
    const unsigned items = (r1 - r0) * (unsigned)ntiles;
    // item c of the slot's sequence -> ROI (through the sort's order where there is one) and tile
    auto roi_of = [&](unsigned c, unsigned& t) -> unsigned {
        const unsigned pos = fdiv(c, div_tiles);
        t = c - pos * (unsigned)ntiles;
        return xg.G > 1 ? (unsigned)xg.order[r0 + pos] : pos;
    };
    const unsigned px_bytes = lay.px_bytes;
    const unsigned row_bytes = lay.row_bytes;

    // lane = q + 8*b: the 8 lanes that fetch the 8 channel quads of ONE pixel (one 128-byte
    // line) are consecutive, so the texture addresser merges them into two 64-byte
    // accesses.  (With the quads strided over the wave every lane costs its own access:
    // measured 43 vs 16 TCP accesses per load instruction.)
    const unsigned q = lane & (kQuads - 1), b = lane >> 3;
    // a channel quad wholly beyond C never loads (its rows are not stored either)
    // NCHW_SRC: lay.slice_bytes is ONE channel plane (H * W * 4 bytes); the quad's first channel is 4 q planes into the chunk
    const unsigned plane_bytes = lay.slice_bytes;
    const unsigned q_bytes = ((dbg & 2) || k * kChunk + q * 4 >= (unsigned)C) ? kQuadOOB
                             : (NCHW_SRC ? q * 4u * plane_bytes : q * 16u);
    // LDS tile: row r = channel, 68-dword pitch; the column of rows 8m..8m+7 is XORed with
    // 4*m so that the 32 lanes of a store group (8 quads x 4 bins) spread over the banks
    // while rows stay 16-byte aligned for the ds_read_b128 of phase C.
    const unsigned wswz = (q >> 1) * 4u;  // rows 4q..4q+3 -> m = q >> 1
    const unsigned col = (lane & 15) * 4, row0 = lane >> 4;
    const unsigned chans_here = min((unsigned)kChunk, (unsigned)C - k * kChunk);  // rows of this chunk < C
    const v4f z4 = {0.f, 0.f, 0.f, 0.f};

    unsigned g_lo = 0, g_hi = 0;          // groups of the current item (wave-uniform)

    // phase A of one item: lane = bin, geometry -> sorted 16-byte tap records in LDS:
    //   LO: {off_lt, off_2nd, w_lt, bin position}      (w_2nd = 1 - w_lt, see blend_lo)
    //   HI: {off_lt, off_rt, off_lb, off_rb}, bin position in HP[]   (all four weights are 1/4)
    // Offsets are byte offsets into the slice; kOOB reads as 0.0, which is what
    // kernel.cu:116-126 substitutes for a tap outside the map.
    auto geometry = [&](const Affine& A, unsigned t, unsigned p, unsigned& n_lo_groups,
                        unsigned& n_hi_groups, unsigned long long& amask) {
        uint4* const G = Gbuf + p * kRecs;
        unsigned char* const HP = HPbuf + p * kRecs;
        // where phase B puts the bin: the lane's index -- its column of the channel-major tile, or (SHIFT) its row of the
        // bin-major tile; kPadPos = 64: nowhere / row 64
        auto tile_pos = [](unsigned lane_of_bin) -> unsigned { return lane_of_bin; };
        const bool batch_ok = A.batch >= 0 && A.batch < batch_size;
        // SHIFT: lanes 0..15 are the 16 bins in front of the tile's own 48 (negative for the first tile of a row)
        const int sbin = (int)(t * (unsigned)kOwnBins + lane) - kFront;
        const unsigned bin = (unsigned)max(sbin, 0);
        const unsigned ph = fdiv(bin, div_pw);
        const unsigned pw = bin - ph * (unsigned)pooled_width;
        float bcx, bcy;
        bool active = bin_centre(A, (int)ph, (int)pw, height, width, bcx, bcy);
        active = active && sbin >= 0 && bin < (unsigned)NB && batch_ok;
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
            G[idx] = make_uint4(o_lt, dx ? o_rt : o_lb, hi ? o_lb : as_u(wlt), hi ? o_rb : tile_pos(lane));
            HP[idx] = (unsigned char)tile_pos(lane);
        }
        // pad both classes to whole groups with records that load nothing and store nowhere
        const unsigned pad_lo = hi_base - n_lo, pad_hi = n_hi_groups * kBinsPerIter - n_hi;
        if (lane < pad_lo + pad_hi) {
            const bool plo = lane < pad_lo;
            const unsigned pidx = plo ? n_lo + lane : hi_base + n_hi + (lane - pad_lo);
            G[pidx] = make_uint4(kOOB, kOOB, plo ? 0u : kOOB, plo ? tile_pos(kPadPos) : kOOB);
            HP[pidx] = (unsigned char)tile_pos(kPadPos);
        }
    };
    // register sets 0 / 1: the depth-2 pipeline of phase B; sets 2 .. 2 + kEarly - 1: the first LO groups of
    // an item, whose loads go out BEFORE the previous tile's stores (see the loop at the end)
    uint4 ra[2 + kEarly];
    unsigned hpos[2];
    v4f lt[2 + kEarly], rt[2 + kEarly], lb[2], rbv[2];
    auto fetch_lo = [&](unsigned p, unsigned grp, int s) { ra[s] = Gbuf[p * kRecs + grp * kBinsPerIter + b]; };
    auto fetch_hi = [&](unsigned p, unsigned grp, int s) {
        ra[s] = Gbuf[p * kRecs + grp * kBinsPerIter + b];
        hpos[s] = HPbuf[p * kRecs + grp * kBinsPerIter + b];
    };
    // one tap of the lane's four channels: a 16-byte load of the chunk-major / channels-last pixel, or (NCHW_SRC) four
    // dwords a plane apart -- a channel beyond C is beyond the descriptor's range (chans_here planes) and reads 0.0
    auto tap4 = [&](__amdgpu_buffer_rsrc_t rs, unsigned off) -> v4f {
        if (!NCHW_SRC) return buf_load(rs, off);
        v4f v;
        v.x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
        v.y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off + plane_bytes, 0, 0));
        v.z = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off + 2u * plane_bytes, 0, 0));
        v.w = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off + 3u * plane_bytes, 0, 0));
        return v;
    };
    auto issue_lo = [&](__amdgpu_buffer_rsrc_t rs, int s) {
        // offsets are < 2^30 or kOOB = 2^31, q_bytes is < 128 (NCHW_SRC: < 2^30 less four planes) or kQuadOOB = 2^30: every
        // sum with an out-of-range term lies in [2^30, 2^32) -- beyond any slice (shape_ok), and it cannot wrap
        lt[s] = tap4(rs, ra[s].x + q_bytes);
        rt[s] = tap4(rs, ra[s].y + q_bytes);  // the bin's one other distinct tap, if any
    };
    auto issue_hi = [&](__amdgpu_buffer_rsrc_t rs, int s) {
        lt[s] = tap4(rs, ra[s].x + q_bytes);
        rt[s] = tap4(rs, ra[s].y + q_bytes);
        lb[s] = tap4(rs, ra[s].z + q_bytes);
        rbv[s] = tap4(rs, ra[s].w + q_bytes);
    };
    float* const t_row = T + (q * 4) * kTStride;
    float* const t_pad = T + kChunk * kTStride + (lane & 31u);
    auto put = [&](unsigned pos, v4f v) {
        if (SHIFT) {   // bin-major tile: the lane's four channels of the bin are one 16-byte piece of the bin's row
            *reinterpret_cast<v4f*>(reinterpret_cast<char*>(T) + ((pos << 7) + (q << 4))) = v;
            return;
        }
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

    // phase C, the storer's: the [rows < C] x [64 bins] tile leaves T for its registers between the two
    // barriers, then goes out as 256-byte row segments (dbg & 1, the ablation knob, drops the stores)
    auto drain_tile = [&](unsigned n, unsigned t, unsigned long long cur_mask, bool skip) {
        if (ONHWC) {
            // channels-last crops (R, PH*PW, C): lane = (channel quad q, bin b) as in phase B; the tile is read back
            // along its columns (the mapping put() wrote it with), a store covers 8 bins x 128 B
            v4f c[kIters];
#pragma unroll
            for (int it8 = 0; it8 < kIters; ++it8) {
                const unsigned bl = (unsigned)it8 * kBinsPerIter + b;   // bin within the tile
                const float* tr = t_row + (bl ^ wswz);
                c[it8] = v4f{tr[0], tr[kTStride], tr[2 * kTStride], tr[3 * kTStride]};
            }
            wg_lds_barrier();  // T has been read
            const bool live = !(dbg & 1) && !skip;
            float* obase = out + (size_t)n * NB * C;
            const __amdgpu_buffer_rsrc_t ws = make_rsrc(obase, (unsigned)NB * (unsigned)C * 4u);
            const bool q_ok = k * kChunk + q * 4 < (unsigned)C;
#pragma unroll
            for (int it8 = 0; it8 < kIters; ++it8) {
                const unsigned bl = (unsigned)it8 * kBinsPerIter + b;
                const bool on = (cur_mask >> bl) & 1ull;
                const v4f o = {on ? c[it8].x : 0.f, on ? c[it8].y : 0.f, on ? c[it8].z : 0.f, on ? c[it8].w : 0.f};
                const unsigned bin = t * kTileBins + bl;
                const unsigned off = (bin * (unsigned)C + k * kChunk + q * 4u) * 4u;
                const unsigned o_off = (live && q_ok && bin < (unsigned)NB) ? off : kOOB;
                if (it8 < kMinorStores) buf_store<kMinorAux>(ws, o_off, o);
                else buf_store<kStoreAux>(ws, o_off, o);
            }
            return;
        }
        v4f v[kChunk / 4];
#pragma unroll
        for (int s4 = 0; s4 < kChunk / 4; ++s4) {
            const unsigned r = s4 * 4 + row0;
            v[s4] = *reinterpret_cast<const v4f*>(T + r * kTStride + (col ^ ((r >> 3) * 4u)));
        }
        wg_lds_barrier();  // T has been read: the gatherer may blend the next tile into it
        const bool live = !(dbg & 1) && !skip;
        // descriptor over this (roi, chunk) block of the output: rows >= C fall out of range
        float* obase = out + ((size_t)n * C + k * kChunk) * NB;
        const __amdgpu_buffer_rsrc_t ws = make_rsrc(obase, chans_here * (unsigned)NB * 4u);
        const unsigned bin0 = t * kTileBins + col;
        // bins that were in no group (masked by pw > roi_pooled_width) are zero
        const unsigned nib = (unsigned)(cur_mask >> col) & 15u;
        const bool a0 = nib & 1u, a1 = nib & 2u, a2 = nib & 4u, a3 = nib & 8u;
#pragma unroll
        for (int s4 = 0; s4 < kChunk / 4; ++s4) {
            const unsigned r = s4 * 4 + row0;
            const unsigned off = (r * (unsigned)NB + bin0) * 4u;
            const v4f o = {a0 ? v[s4].x : 0.f, a1 ? v[s4].y : 0.f, a2 ? v[s4].z : 0.f, a3 ? v[s4].w : 0.f};
            if (VEC_STORE) {  // NB % 4 == 0: the 4 bins are all inside or all outside the row
                // WAUX >= 0: ONE policy for all eight stores (the merging form for rows that are not whole sectors, see the host)
                if (WAUX >= 0) {
                    // any NB: a quad wholly inside the row is ONE 16-byte store (dword-aligned: the rows of such crops start
                    // anywhere); the row's last quad, when NB % 4 != 0, leaves as its <= 3 valid dwords -- in a row's last tile only
                    constexpr int A = WAUX >= 0 ? WAUX : 0;
                    buf_store<A>(ws, (live && bin0 + 4u <= (unsigned)NB) ? off : kOOB, o);
                    if (t == (unsigned)ntiles - 1u && ((unsigned)NB & 3u)) {   // (wave-uniform)
                        const bool part = live && bin0 < (unsigned)NB && bin0 + 4u > (unsigned)NB;
                        buf_store1<A>(ws, part ? off + 0 : kOOB, o.x);
                        buf_store1<A>(ws, (part && bin0 + 1 < (unsigned)NB) ? off + 4 : kOOB, o.y);
                        buf_store1<A>(ws, (part && bin0 + 2 < (unsigned)NB) ? off + 8 : kOOB, o.z);
                    }
                } else if (s4 < kMinorStores)
                    buf_store<kMinorAux>(ws, (live && bin0 < (unsigned)NB) ? off : kOOB, o);
                else
                    buf_store<kStoreAux>(ws, (live && bin0 < (unsigned)NB) ? off : kOOB, o);
            } else {
                buf_store1<kStoreAux>(ws, (live && bin0 + 0 < (unsigned)NB) ? off + 0 : kOOB, o.x);
                buf_store1<kStoreAux>(ws, (live && bin0 + 1 < (unsigned)NB) ? off + 4 : kOOB, o.y);
                buf_store1<kStoreAux>(ws, (live && bin0 + 2 < (unsigned)NB) ? off + 8 : kOOB, o.z);
                buf_store1<kStoreAux>(ws, (live && bin0 + 3 < (unsigned)NB) ? off + 12 : kOOB, o.w);
            }
        }
    };

    // SHIFT (crops whose rows are not multiples of 64 bytes: PH * PW % 16 != 0).  A 256-byte row segment that
    // starts 16, 32 or 48 bytes into a 64-byte sector ends in one too: two partial sectors per row and tile, each
    // completed later by ANOTHER workgroup -- and HBM pays for a partial sector with a read-modify-write
    // (tools/align_probe.py: 11 x 100 crops 58 us against 28 us for 11 x 96; PH * PW % 4 != 0, where the stores
    // were dwords: 215 us).  So every channel row's store WINDOW is shifted left by h = (the row's float offset in
    // memory) mod 16 -- the crops' own base address included: window position p of tile t is bin 48 t + p - h of the
    // row, every window starts on a sector and every 16-byte piece a lane stores is aligned.  Partial sectors are
    // left at the two ends of a ROW only.
    // Round 4: OVERLAPPED tiles.  A tile advances by 48 bins and gathers 64: its own 48 and the 16 in front of them,
    // which is all a window shifted by h <= 15 can reach -- so an item needs nothing from its neighbour, items are
    // dealt to the workgroups every n-th like the strided kernel's, and neighbouring tiles of a row are written at
    // the same time by neighbouring workgroups.  (Round 3 CARRIED the h bins from tile to tile in LDS: a workgroup
    // walked a run of consecutive tiles, so every 256 bytes of a row were written microseconds after their
    // neighbours -- isolated DRAM accesses; with the store pattern alone the kernel took 131 of 150 us, and runs, pre
    // items, cuts and flushes were half of its code.)  The price is 64 / 48 of the gather work per stored byte.
    // The tile is bin-major for this -- T[lane's bin * 32 + channel], a bin's 32 channels one 128-byte row -- and the
    // storer reads it with lane = (channel, piece):
    //   * the gatherer writes a lane's four channels of a bin as ONE ds_write_b128 (the channel-major tile takes four
    //     ds_write_b32); the eight lanes of a bin fill its row: no bank conflict, no swizzle;
    //   * rows r and r + 16 of a chunk have the same h (16 NB is a multiple of 16), so a storer lane that serves the
    //     channels ch16 and ch16 + 16 has ONE h, and every element it reads -- window positions 16 s + 4 pcl + e of
    //     those two channels -- sits at a compile-time offset from ONE address: 24 ds_read_b32 (32 in a row's last tile) with immediates, no
    //     address arithmetic, whatever h is (round 3 rebuilt every window of a channel-major tile with a funnel of
    //     selects from two ds_read_b128 per row set: two kernels, 96 / 117 VGPRs, 10 / 8 workgroups per CU);
    //   * a wave store instruction covers 16 channel rows x one whole 64-byte sector (4 NEIGHBOURING lanes x 16 B); a tile is three
    //     sectors per row -- four in the last tile of a row, whose window runs on to the row's end.  (The strided
    //     kernel's shape, 4 rows x 256 B per instruction, was built too -- lane = (channel of four, piece of sixteen) on
    //     a quad-swizzled tile, two addresses and three selects per piece: 145 against 147 us at C = 256, 11 x 100, with
    //     a quarter more instructions in the storer; not kept.  profiles/r04_shift_forms.md)
    auto drain_shift = [&](unsigned n, unsigned t, unsigned long long cur_mask, bool skip) {
        // lane = (row of sixteen, 16-byte piece of the sector) with the PIECE fastest: the four lanes of a 64-byte sector
        // are neighbours, which is what the TCP coalesces -- 15 tag accesses per store instruction; with the row
        // fastest (rounds 3-4: an LDS read without bank conflicts) every lane was an access of its own, 61 per
        // instruction, and the stores took the TCP from the gatherer's tap loads (tools/kbench desync, PMC)
        const unsigned ch16 = lane >> 2, pcl = lane & 3u;
        const unsigned nb15 = (unsigned)NB & 15u;
        // float index of (roi n, first channel of the chunk, bin 0), modulo a sector -- out's own alignment included
        const unsigned h0 = (((unsigned)(reinterpret_cast<size_t>(out) >> 2) & 15u) +
                             (((n & 15u) * ((unsigned)C & 15u) + ((k * kChunk) & 15u)) & 15u) * nb15) & 15u;
        const unsigned h = (h0 + ch16 * nb15) & 15u;           // of both channels of this lane
        const int left = NB - (int)(t * (unsigned)kOwnBins);   // bins of the row from this tile's own first on (>= 1)
        const bool first = t == 0, last = left <= kOwnBins;    // the row's first / last tile: partial sectors possible
        // instruction i = (u = i & 1: channel ch16 + 16 u, s = i >> 1: sector of the window); element e: window position
        // 16 s + 4 pcl + e = lane (= row of the tile) 16 + 16 s + 4 pcl + e - h
        const float* wb = T + (16u + 4u * pcl - h) * kChunk + ch16;
        v4f o[kChunk / 4];
#pragma unroll
        for (int i = 0; i < kChunk / 4; ++i) {
            if ((i >> 1) == 3 && !last) continue;   // the fourth sector exists in a row's last tile only
            const float* wp = wb + (i & 1) * 16 + (i >> 1) * 16 * kChunk;
            o[i] = v4f{wp[0], wp[kChunk], wp[2 * kChunk], wp[3 * kChunk]};
        }
        // the <= 3 floats in front of / behind the whole 16-byte pieces of a row's valid positions (the row's first
        // and last tile only): lane = (row, i), two passes -> fix[0..1] head, fix[2..3] tail
        float fix[4] = {0.f, 0.f, 0.f, 0.f};
        const unsigned fr = lane & 31u, fi = lane >> 5;
        const unsigned fh = (h0 + fr * nb15) & 15u;
        const int pl = first ? (int)fh : 0;                          // first valid position of row fr
        const int ph = last ? left + (int)fh : kOwnBins;             // one past its last (<= 63)
        auto win_at = [&](unsigned rr, int pos, unsigned hrow) -> float {   // one window position, masked
            const unsigned ln = (unsigned)(pos - (int)hrow + 16);            // the lane that gathered it (< 64 where valid)
            const float x = T[ln * kChunk + rr];
            return ((cur_mask >> (ln & 63u)) & 1ull) ? x : 0.0f;
        };
        if (first || last) {
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const int pa = pl + (int)fi + 2 * ps, pb = (ph & ~3) + (int)fi + 2 * ps;
                if (pa < ((pl + 3) & ~3) && pa < ph) fix[ps] = win_at(fr, pa, fh);
                if (pb < ph && pb >= pl && (ph & ~3) >= ((pl + 3) & ~3)) fix[2 + ps] = win_at(fr, pb, fh);
            }
        }
        wg_lds_barrier();  // T has been read: the gatherer may blend the next tile into it
        const bool live = !(dbg & 1) && !skip;
        float* obase = out + ((size_t)n * C + k * kChunk) * NB;
        const __amdgpu_buffer_rsrc_t ws = make_rsrc(obase, chans_here * (unsigned)NB * 4u);
        const int rl = first ? (int)h : 0, rh = last ? left + (int)h : kOwnBins;
        const unsigned msh = 16u + 4u * pcl - h;                // the lane that gathered this lane's first element of sector 0
#pragma unroll
        for (int i = 0; i < kChunk / 4; ++i) {
            if ((i >> 1) == 3 && !last) continue;
            const unsigned r = ch16 + 16u * (i & 1);
            const int p0 = 16 * (i >> 1) + 4 * (int)pcl;
            // bins in no group (masked by pw > roi_pooled_width, or outside the row) are zero
            const unsigned nib = (unsigned)((cur_mask >> (16 * (i >> 1))) >> msh);
            const v4f v = {(nib & 1u) ? o[i].x : 0.f, (nib & 2u) ? o[i].y : 0.f, (nib & 4u) ? o[i].z : 0.f,
                           (nib & 8u) ? o[i].w : 0.f};
            const bool whole = p0 >= rl && p0 + 4 <= rh;
            // float offset of position p0 within the (roi, chunk) block; never negative where `whole`
            const unsigned off = (r * (unsigned)NB + t * (unsigned)kOwnBins + (unsigned)p0 - h) * 4u;
            const unsigned o_off = (live && whole && r < chans_here) ? off : kOOB;
            buf_store<kShiftAux>(ws, o_off, v);
        }
        if (first || last) {
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const int pa = pl + (int)fi + 2 * ps, pb = (ph & ~3) + (int)fi + 2 * ps;
                const bool oka = live && fr < chans_here && pa < ((pl + 3) & ~3) && pa < ph;
                const bool okb = live && fr < chans_here && pb < ph && pb >= pl && (ph & ~3) >= ((pl + 3) & ~3);
                buf_store1<kShiftAux>(ws, oka ? (fr * (unsigned)NB + t * (unsigned)kOwnBins + (unsigned)pa - fh) * 4u : kOOB, fix[ps]);
                buf_store1<kShiftAux>(ws, okb ? (fr * (unsigned)NB + t * (unsigned)kOwnBins + (unsigned)pb - fh) * 4u : kOOB, fix[2 + ps]);
            }
        }
    };

    // SHIFT == 2: the same idea with LINE-aligned windows, for crops larger than the 256 MB memory-side cache -- there a half
    // line reaches HBM as a half line, and the store stream of the 48-bin windows alone runs at 3.1 TB/s (profiles/
    // r05_big_crops.txt; whole lines: 4.0-5.0 in tools/kbench bigout).  h = (row's float offset in memory) mod 32; window
    // position p of tile t is bin 32 t + p - h of the row; a tile stores ONE line per row (two in a row's last tile, whose
    // window runs on to the row's end) and gathers 64 bins for it: 2 x the gather work per byte (SHIFT == 1: 4 / 3).
    // lane = (row of eight, 16-byte piece of the line): the eight lanes of a line are neighbours (what the TCP coalesces),
    // instruction i = (u = i & 3: row row8 + 8 u, s = i >> 2: line of the window).  Rows 8 apart do not share h when NB is
    // odd, so every instruction has its own h and its own LDS address (16 ds_read_b32 per line set).
    auto drain_shift2 = [&](unsigned n, unsigned t, unsigned long long cur_mask, bool skip) {
        const unsigned row8 = lane >> 3, pc = lane & 7u;
        const unsigned nb31 = (unsigned)NB & 31u;
        const unsigned h0 = (((unsigned)(reinterpret_cast<size_t>(out) >> 2) & 31u) +
                             (((n & 31u) * ((unsigned)C & 31u)) & 31u) * nb31) & 31u;   // (k * 32 channels: whole lines)
        const int left = NB - (int)(t * (unsigned)kOwnBins);
        const bool first = t == 0, last = left <= kOwnBins;
        v4f o[8];
        unsigned hh[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) hh[u] = (h0 + (row8 + 8u * u) * nb31) & 31u;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if ((i >> 2) == 1 && !last) continue;   // the second line exists in a row's last tile only
            const unsigned u = i & 3, r = row8 + 8u * u;
            const float* wp = T + ((unsigned)kFront + 32u * (i >> 2) + 4u * pc - hh[u]) * kChunk + r;
            o[i] = v4f{wp[0], wp[kChunk], wp[2 * kChunk], wp[3 * kChunk]};
        }
        // the <= 3 floats in front of / behind the whole 16-byte pieces of a row's valid positions (first / last tile)
        float fix[4] = {0.f, 0.f, 0.f, 0.f};
        const unsigned fr = lane & 31u, fi = lane >> 5;
        const unsigned fh = (h0 + fr * nb31) & 31u;
        const int pl = first ? (int)fh : 0;
        const int ph = last ? left + (int)fh : kOwnBins;   // (<= 63)
        auto win_at = [&](unsigned rr, int pos, unsigned hrow) -> float {
            const unsigned ln = (unsigned)(pos - (int)hrow + kFront);   // the lane that gathered it (< 64 where valid)
            const float x = T[ln * kChunk + rr];
            return ((cur_mask >> (ln & 63u)) & 1ull) ? x : 0.0f;
        };
        if (first || last) {
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const int pa = pl + (int)fi + 2 * ps, pb = (ph & ~3) + (int)fi + 2 * ps;
                if (pa < ((pl + 3) & ~3) && pa < ph) fix[ps] = win_at(fr, pa, fh);
                if (pb < ph && pb >= pl && (ph & ~3) >= ((pl + 3) & ~3)) fix[2 + ps] = win_at(fr, pb, fh);
            }
        }
        wg_lds_barrier();  // T has been read: the gatherer may blend the next tile into it
        const bool live = !(dbg & 1) && !skip;
        float* obase = out + ((size_t)n * C + k * kChunk) * NB;
        const __amdgpu_buffer_rsrc_t ws = make_rsrc(obase, chans_here * (unsigned)NB * 4u);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if ((i >> 2) == 1 && !last) continue;
            const unsigned u = i & 3, r = row8 + 8u * u, h = hh[u];
            const int p0 = 32 * (i >> 2) + 4 * (int)pc;
            const int rl = first ? (int)h : 0, rh = last ? left + (int)h : kOwnBins;
            // the lanes that gathered the piece: kFront + p0 - h .. + 3 (< 64 where the piece is whole)
            const unsigned msh = (unsigned)kFront + 4u * pc - h;   // <= 60
            const unsigned nib = (unsigned)((cur_mask >> (32 * (i >> 2))) >> msh);
            const v4f v = {(nib & 1u) ? o[i].x : 0.f, (nib & 2u) ? o[i].y : 0.f, (nib & 4u) ? o[i].z : 0.f,
                           (nib & 8u) ? o[i].w : 0.f};
            const bool whole = p0 >= rl && p0 + 4 <= rh;
            const unsigned off = (r * (unsigned)NB + t * (unsigned)kOwnBins + (unsigned)p0 - h) * 4u;
            buf_store<kShift2Aux>(ws, (live && whole && r < chans_here) ? off : kOOB, v);
        }
        if (first || last) {
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const int pa = pl + (int)fi + 2 * ps, pb = (ph & ~3) + (int)fi + 2 * ps;
                const bool oka = live && fr < chans_here && pa < ((pl + 3) & ~3) && pa < ph;
                const bool okb = live && fr < chans_here && pb < ph && pb >= pl && (ph & ~3) >= ((pl + 3) & ~3);
                buf_store1<kShift2Aux>(ws, oka ? (fr * (unsigned)NB + t * (unsigned)kOwnBins + (unsigned)pa - fh) * 4u : kOOB, fix[ps]);
                buf_store1<kShift2Aux>(ws, okb ? (fr * (unsigned)NB + t * (unsigned)kOwnBins + (unsigned)pb - fh) * 4u : kOOB, fix[2 + ps]);
            }
        }
    };

    // The two waves walk the same items.  gfx950 counts loads and stores with ONE in-order counter, so in
    // a wave that does both a load issued after a tile's stores cannot be consumed before those stores are
    // acknowledged (microseconds, with 256 MiB streaming out).  Here no wave does both: the gatherer's vmcnt only ever sees loads, the storer's only
    // stores, and they meet at two s_barriers per tile (which do not drain vmcnt):
    //   gatherer, item i:  barrier 1: the records of item i are in set p, tile i-1 is complete in T |
    //                      first loads of item i | barrier 2: T has been read | phase B -> T
    //   storer:            barrier 1 | tile i-1: T -> registers | barrier 2 | its eight 1 KiB stores, never
    //                      waited for | geometry of item i+1 -> record set p^1
    // Items of this workgroup: every nslots-th one (SHIFT: a row has ceil(NB / 48) tiles -- the host passes that as
    // ntiles).  Both waves walk the same sequence with next(); kEnd ends it.
    constexpr unsigned kEnd = 0xffffffffu;
    auto next = [&](unsigned c) -> unsigned { return c + nslots < items ? c + nslots : kEnd; };
    unsigned cur = slot < items ? slot : kEnd;
    if (cur == kEnd) return;
    if (storer) {
        unsigned t;
        unsigned n = roi_of(cur, t);
        unsigned p = 0;
        unsigned gl, gh;
        unsigned long long mask_cur = 0, mask_prev = 0;
        unsigned n_prev = 0, t_prev = 0;
        // dbg & 32 (the reference-ABI launcher): the crops of ROIs whose image index is >= batch_size have been
        // written by the prologue launch -- they are not zero-filled here
        bool skip_cur = false, skip_prev = false;
        auto plan = [&](unsigned pn, unsigned pt, unsigned pp, unsigned long long& m) {
            const Affine A = NCHW_SRC ? make_affine(roi_src.rois + (size_t)pn * 6, roi_src.pooled_height, roi_src.spatial_scale, roi_src.trig)
                                      : aff[pn];
            skip_cur = (dbg & 32) && A.batch >= batch_size;  // (a negative index still yields zeros)
            geometry(A, pt, pp, gl, gh, m);
            if (lane == 0) {
                shead[pp] = make_uint4(gl, gh, (unsigned)m, (unsigned)(m >> 32));
                sbatch[pp] = A.batch;
            }
        };
        plan(n, t, 0, mask_cur);
        for (bool have_prev = false;; have_prev = true) {
            wg_lds_barrier();  // 1: records of item `cur` are in set p; the tile of the previous item is in T
            if (have_prev) {
                // T -> registers | barrier 2 | stores
                if (SHIFT == 2) drain_shift2(n_prev, t_prev, mask_prev, skip_prev);
                else if (SHIFT) drain_shift(n_prev, t_prev, mask_prev, skip_prev);
                else drain_tile(n_prev, t_prev, mask_prev, skip_prev);
            } else {
                wg_lds_barrier();
            }
            if (cur == kEnd) break;
            n_prev = n;
            t_prev = t;
            mask_prev = mask_cur;
            skip_prev = skip_cur;
            cur = next(cur);
            if (cur != kEnd) {
                n = roi_of(cur, t);
                p ^= 1u;
                plan(n, t, p, mask_cur);  // while the gatherer blends the item before
            }
        }
        return;
    }
    // the gatherer's chain (records -> loads -> blend) is the latency of a tile; the storer's geometry is
    // not urgent: with the gatherer ahead in the issue arbitration the call is 0.7 us shorter (any level > 0)
    __builtin_amdgcn_s_setprio(2);
    unsigned p = 0;
    for (;; cur = next(cur), p ^= 1u) {
        wg_lds_barrier();  // 1: the storer has put this item's records into set p (and our previous tile is complete)
        if (cur == kEnd) {
            wg_lds_barrier();  // 2
            break;
        }
        const int batch = __builtin_amdgcn_readfirstlane(sbatch[p]);
        const uint4 hd = shead[p];
        g_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)hd.x);
        g_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)hd.y);
        const bool batch_ok = batch >= 0 && batch < batch_size;
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(
            map + (size_t)(batch_ok ? batch : 0) * lay.img_stride + (size_t)k * lay.chunk_stride,
            NCHW_SRC ? chans_here * plane_bytes : lay.slice_bytes);
        // the first kEarly LO groups go out while the storer takes the previous tile out of T,
        // unconditionally (a record beyond the item's LO groups is a HI record, padding or stale: its
        // offsets are in range or kOOB, the data is never used) -- straight-line code keeps the s_waitcnt
        // counts exact
#pragma unroll
        for (int e = 0; e < kEarly; ++e) {
            fetch_lo(p, e, 2 + e);
            issue_lo(rs, 2 + e);
        }
        fetch_lo(p, kEarly, 0);
        issue_lo(rs, 0);
        wg_lds_barrier();  // 2: the storer holds the previous tile in registers: T is free

        // ---- phase B: the loads of group g+1 are issued before group g is blended.  The loops are
        // unrolled with an early exit, and the two exit paths end in different (empty) asm statements so
        // that the compiler cannot merge their tails: each blend then has ONE predecessor and its
        // s_waitcnt knows exactly how many younger loads are in flight.
#pragma unroll
        for (int e = 0; e < kEarly; ++e) {
            if ((unsigned)e < g_lo) {
                pin_lo(2 + e);
                blend_lo(2 + e);
            }
        }
        if (g_lo > (unsigned)kEarly) {
#pragma unroll
            for (int it = kEarly; it < kMaxGroups; ++it) {
                const int s = (it - kEarly) & 1;
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
        }
        if (HID == 3) {
            // two register sets, a ROLLED loop over pairs of groups (the unrolled form costs 28 VGPRs more)
            if (g_hi > 0) {
                fetch_hi(p, g_lo, 0);
                issue_hi(rs, 0);
#pragma unroll 1
                for (unsigned it = 0;; it += 2) {
                    if (it + 1 < g_hi) {
                        fetch_hi(p, g_lo + it + 1, 1);
                        issue_hi(rs, 1);
                        pin_hi(0);
                    }
                    blend_hi(0);
                    if (it + 1 >= g_hi) break;
                    if (it + 2 < g_hi) {
                        fetch_hi(p, g_lo + it + 2, 0);
                        issue_hi(rs, 0);
                        pin_hi(1);
                    }
                    blend_hi(1);
                    if (it + 2 >= g_hi) break;
                }
            }
        } else
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
    int width, int pooled_height, int pooled_width, float spatial_scale, int trig, int batch_size,
    int cslab)
{
    const int NB = pooled_height * pooled_width;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)num_rois * NB) return;
    const int n = (int)(gid / NB);
    const int bin = (int)(gid - (long)n * NB);
    const Affine A = make_affine(rois + (size_t)n * 6, pooled_height, spatial_scale, trig);
    const int c_begin = blockIdx.y * cslab;
    direct_bin(feat, A, out, idx_x, idx_y, n, bin, C, height, width, pooled_width, NB, batch_size, c_begin,
               min(C, c_begin + cslab));
}

// ------------------------------------------------------------------------------------
// K2p (round 5): the lean ONE-LAUNCH forward for few ROIs at few channels -- the reference's own call: <= 32 ROIs per
// training step on its 64-channel map (src/ocr_process.py:253-267), 1 ... 24 per image in inference.  K2 above pays the
// double-precision affine and the whole bin geometry per thread and 4-16 channels, and five vector-memory instructions
// (four taps, one store) plus ~45 VALU per 64 bin-channels: it runs at the VALU and address rates
// (profiles/r05_small_r_forward.md).  Here:
//   * workgroup = (ROI, one PATCH of up to 64 bins -- 4 x 16, or whatever shape fills the pooled size best --, four channel
//     slabs); the first wave computes what does not depend on the
//     channel ONCE -- affine, bin centres, tap offsets, validity -- and leaves a 32-byte record per bin in LDS;
//   * the taps of a bin are whole pixels at (y0 | y1, x0 | x0 + 1): the two pixels of a map ROW come with ONE 8-byte load
//     (dword-aligned: gfx950 runs with unaligned access enabled), so a bin costs two loads per channel instead of four.
//     A pair starts at x0 -- or at x0 - 1 when x0 is the row's last pixel, so that it never leaves the plane; a row none
//     of whose two pixels is a valid tap of its own (kernel.cu:116-126) is the out-of-range offset, which costs no access;
//   * lane = bin: the stores are rows of a patch's width in consecutive floats of the crop (the crops of few ROIs stay in the L2s until the
//     launch ends, which merges rows that are not whole sectors).
// Same arithmetic as every other forward path: blend1 on the reference's four taps in its order.
// ------------------------------------------------------------------------------------
struct PatchRec {     // per bin, written by the first wave
    unsigned o_top, o_bot;   // byte offsets of the row pairs inside a channel plane, or kOOB
    unsigned flags;          // kV00.. kV11 | kDx | kDy | kActive | kPairShift (the pair starts at x0 - 1)
    float rx, ry;
    float cx, cy;            // WITH_IDX: the bin centre the reference ABI's con_idx_x / con_idx_y hold (0 where masked)
};
constexpr unsigned kPairShift = 1u << 20;

// WITH_IDX: also the reference ABI's con_idx_x / con_idx_y (kernel.cu:144-145); batch_size < 0 = unknown (that ABI): the
// image index is trusted as the reference trusts it.
template <int U, bool WITH_IDX = false>
__global__ __launch_bounds__(256) void rroi_fwd_patch_kernel(
    const float* __restrict__ feat, const float* __restrict__ rois, float* __restrict__ out, int num_rois, int C, int height,
    int width, int pooled_height, int pooled_width, float spatial_scale, int trig, int batch_size, int cwave, int npx,
    int npatches, int prows, int pcols, float* __restrict__ idx_x = nullptr, float* __restrict__ idx_y = nullptr)
{
    __shared__ PatchRec rec[kWave];
    __shared__ int s_batch;
    const int n = (int)blockIdx.x / npatches, patch = (int)blockIdx.x - n * npatches;
    // (the wave index as a SCALAR: the channel planes' buffer descriptors derive from it -- left in a VGPR the compiler
    // wraps every load in a waterfall loop)
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63u);
    const int py = patch / npx, px = patch - py * npx;
    // a patch is prows x pcols bins (prows * pcols <= 64; the host picks the shape that wastes the fewest lanes on this
    // pooled size: 4 x 16 for 11 x 96, 3 x 21 for 11 x 83), lane = row-major position in it
    const int lrow = lane / pcols, lcol = lane - lrow * pcols;
    const int ph = py * prows + lrow, pw = px * pcols + lcol;
    const bool inside = lrow < prows && ph < pooled_height && pw < pooled_width;
    if (wv == 0) {
        const Affine A = make_affine(rois + (size_t)n * 6, pooled_height, spatial_scale, trig);
        float bcx, bcy;
        const bool in_rroi = bin_centre(A, ph, pw, height, width, bcx, bcy);
        const bool batch_ok = batch_size < 0 || (A.batch >= 0 && A.batch < batch_size);
        const bool active = in_rroi && batch_ok && inside;
        const Taps tp = make_taps(bcx, bcy, active, height, width, 4u);   // byte offsets inside a channel plane
        const unsigned f = tp.flags;
        const bool dx = f & kDx, dy = f & kDy;
        // a row's pair is fetched when one of its two pixels is a valid tap OF ITS OWN (an aliased tap copies)
        const bool top = (f & kV00) || (dx && (f & kV01));
        const bool bot = dy && ((f & kV10) || (dx && (f & kV11)));
        // x0 is the row's last pixel: the pair starts one pixel earlier (x1 = W is never a valid tap)
        const bool shift = (f & kActive) && f2i_sat(floorf(bcx)) == width - 1 && width >= 2;
        const unsigned o0 = tp.o_lt - (shift ? 4u : 0u);
        PatchRec r;
        r.o_top = top ? o0 : kOOB;
        r.o_bot = bot ? o0 + (unsigned)width * 4u : kOOB;
        r.flags = f | (shift ? kPairShift : 0u);
        r.rx = tp.rx;
        r.ry = tp.ry;
        r.cx = (in_rroi && batch_ok) ? bcx : 0.0f;
        r.cy = (in_rroi && batch_ok) ? bcy : 0.0f;
        rec[lane] = r;
        if (lane == 0) s_batch = batch_ok ? A.batch : 0;
    }
    __syncthreads();
    const int c_begin = ((int)blockIdx.y * 4 + wv) * cwave;
    if (c_begin >= C) return;
    const int c_end = min(C, c_begin + cwave);
    const PatchRec r = rec[lane];
    const int batch = __builtin_amdgcn_readfirstlane(s_batch);
    const unsigned f = r.flags;
    const bool act = f & kActive, dx = f & kDx, dy = f & kDy, sh = f & kPairShift;
    const bool v00 = f & kV00, v01 = f & kV01, v10 = f & kV10, v11 = f & kV11;
    float wlt, wrt, wrb, wlb;
    tap_weights(r.rx, r.ry, wlt, wrt, wrb, wlb);
    const int NB = pooled_height * pooled_width;
    const size_t HW = (size_t)height * width;
    const unsigned plane_bytes = (unsigned)(HW * 4u);
    const float* plane = feat + ((size_t)batch * C + c_begin) * HW;
    const size_t o0 = ((size_t)n * C + c_begin) * NB + (size_t)ph * pooled_width + pw;
    float* op = out + o0;
    typedef float v2f __attribute__((ext_vector_type(2)));
    auto pair = [&](const float* pl, unsigned o) -> v2f {
        return __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(make_rsrc(pl, plane_bytes), o, 0, 0));
    };
    auto sample = [&](v2f t, v2f b) -> float {
        // the reference's four taps (kernel.cu:110-126) out of the two pairs: element 0 is x0 (element 1 when the pair was
        // shifted), the other one x0 + 1; an invalid tap is 0.0, an aliased tap IS the tap it aliases
        const float lt = v00 ? (sh ? t.y : t.x) : 0.0f;
        const float rt = dx ? (v01 ? t.y : 0.0f) : lt;
        const float lb = dy ? (v10 ? (sh ? b.y : b.x) : 0.0f) : lt;
        const float rb = dx ? (dy ? (v11 ? b.y : 0.0f) : rt) : lb;
        return act ? blend1(lt, rt, rb, lb, wlt, wrt, wrb, wlb) : 0.0f;
    };
    int c = c_begin;
    for (; c + U <= c_end; c += U, plane += U * HW, op += U * (size_t)NB) {
        v2f t[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            t[u] = pair(plane + (size_t)u * HW, r.o_top);
            b[u] = pair(plane + (size_t)u * HW, r.o_bot);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float v = sample(t[u], b[u]);
            if (inside) op[(size_t)u * NB] = v;
        }
    }
    for (; c < c_end; ++c, plane += HW, op += NB) {
        const float v = sample(pair(plane, r.o_top), pair(plane, r.o_bot));
        if (inside) *op = v;
    }
    if (WITH_IDX && inside) {
        for (int cc = 0; cc < c_end - c_begin; ++cc) {
            __builtin_nontemporal_store(r.cx, idx_x + o0 + (size_t)cc * NB);
            __builtin_nontemporal_store(r.cy, idx_y + o0 + (size_t)cc * NB);
        }
    }
}

// con_idx_x / con_idx_y of the reference ABI (kernel.cu:144-145): the bin centre of (roi, ph, pw)
// replicated over the C channels, 0 where the bin is masked.  thread = (roi, bin), loops a channel
// slab; consecutive lanes write consecutive bins.
__global__ __launch_bounds__(256) void rroi_con_idx_kernel(
    const float* __restrict__ rois, float* __restrict__ idx_x, float* __restrict__ idx_y, int num_rois, int C,
    int height, int width, int pooled_height, int pooled_width, float spatial_scale, int trig, int cslab)
{
    const int NB = pooled_height * pooled_width;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)num_rois * NB) return;
    const int n = (int)(gid / NB);
    const int bin = (int)(gid - (long)n * NB);
    const int ph = bin / pooled_width, pw = bin - ph * pooled_width;
    const Affine A = make_affine(rois + (size_t)n * 6, pooled_height, spatial_scale, trig);
    float bcx, bcy;
    const bool in_rroi = bin_centre(A, ph, pw, height, width, bcx, bcy);
    const float vx = in_rroi ? bcx : 0.0f, vy = in_rroi ? bcy : 0.0f;
    const int c_begin = blockIdx.y * cslab;
    const int c_end = min(C, c_begin + cslab);
    size_t o = ((size_t)n * C + c_begin) * NB + bin;
    for (int c = c_begin; c < c_end; ++c, o += NB) {
        __builtin_nontemporal_store(vx, idx_x + o);
        __builtin_nontemporal_store(vy, idx_y + o);
    }
}

