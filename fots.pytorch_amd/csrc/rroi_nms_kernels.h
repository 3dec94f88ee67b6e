// rroi_nms_kernels.h -- detection post-processing, device part: RBOX decode + threshold + ordered compaction
// Part of the single translation unit rroi_align_hip.hip (included inside its anonymous namespace
// after rroi_callers_kernels.h); not a standalone header.
#pragma once

// ------------------------------------------------------------------------------------
// the step right before ROI construction.  The reference copies three
// full-resolution maps to the host (test.py:86-93), transposes them with numpy and walks every
// pixel in C++ (nms/adaptor.cpp:76-117) before the locality-aware merge (nms/nms.h:149-213).
// Here the per-pixel part runs where the maps are: every pixel whose score passes the threshold
// becomes one 64-byte candidate record -- the quad in 1/10000 px integers, the score, the four
// corner confidences, the pixel -- written in RASTER ORDER (the merge depends on it), and only
// those records cross to the host.  Channels-first inputs, as the network emits them: no transposes.
//
// Ordered compaction over many workgroups (round 3; round 2 walked the map with ONE workgroup): workgroup
// i owns pixels [1024 i, 1024 i + 1024) and needs the number of passing pixels before them.  It counts
// them itself -- a 16-byte load per thread covers 4096 pixels of the score map per step, the whole
// 176 x 320 map of a 1280 x 704 image in 14 steps, out of L2 -- so there is no second launch, no flag
// chain between workgroups and no counter that somebody would have to clear.  (Maps beyond 256 K pixels,
// where the redundant counting would cost more than a launch, take per-slab counts from a first launch.)
// fp32 arithmetic exactly as adaptor.cpp writes it (-ffp-contract=off).  The corner confidences
// (adaptor.cpp:97-100, 107: products of expf(-r / 9)) are NOT formed here: the record carries the four raw
// RBOX distances and the host merge, which consumes them, calls the same C library expf the reference calls
// (round 4; rounds 2-3 evaluated exp in double on the device, <= 2 ulp off the library's expf).
// ------------------------------------------------------------------------------------
struct NmsCandidate {  // 64 bytes
    int quad[8];       // x0,y0 .. x3,y3 in 1/10000 px (adaptor.cpp:101-104)
    float score;
    float rdist[4];    // r[0..3] = distances to the top, bottom, left, right edge (:79); the confidences
                       // p_left*p_bt, p_left*p_top, p_right*p_top, p_right*p_bt (:97-100, 107) are formed on the host
    int x, y;
    int pad;
};
static_assert(sizeof(NmsCandidate) == 64, "one candidate = one 64-byte record");

// number of pixels of segm[lo, hi) above the threshold, counted by the whole workgroup (1024 threads)
__device__ __forceinline__ unsigned nms_count_passing(const float* __restrict__ segm, int lo, int hi, float thr,
                                                      unsigned* wave_total)
{
    const unsigned tid = threadIdx.x;
    unsigned n = 0;
    const bool vec = (reinterpret_cast<uintptr_t>(segm) & 15) == 0;
    int p = lo;
    if (vec) {
        for (; p + 4096 <= hi; p += 4096) {
            const v4f v = *reinterpret_cast<const v4f*>(segm + p + 4 * (int)tid);
            n += (v.x > thr) + (v.y > thr) + (v.z > thr) + (v.w > thr);
        }
    }
    for (; p < hi; p += 1024) n += (p + (int)tid < hi && segm[p + (int)tid] > thr) ? 1u : 0u;
#pragma unroll
    for (int o = 32; o; o >>= 1) n += __shfl_xor(n, o);
    if ((tid & 63u) == 0) wave_total[tid >> 6] = n;
    __syncthreads();
    unsigned total = 0;
    for (unsigned k = 0; k < 16; ++k) total += wave_total[k];
    __syncthreads();
    return total;
}

// slab_counts == nullptr: count the pixels before this slab here; else slab_counts[j] = passing pixels of slab j
__global__ __launch_bounds__(1024) void rroi_rbox_decode_kernel(
    const float* __restrict__ segm, const float* __restrict__ rbox, const float* __restrict__ angle, int h, int w,
    float segm_thresh, NmsCandidate* __restrict__ out, int capacity, int* __restrict__ count,
    const unsigned* __restrict__ slab_counts)
{
    __shared__ unsigned wave_total[16];
    const unsigned tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const int hw = h * w;
    const int p0 = (int)blockIdx.x * 1024;
    unsigned base;
    if (slab_counts) {
        unsigned n = 0;
        for (unsigned j = tid; j < blockIdx.x; j += 1024) n += slab_counts[j];
#pragma unroll
        for (int o = 32; o; o >>= 1) n += __shfl_xor(n, o);
        if (lane == 0) wave_total[wv] = n;
        __syncthreads();
        base = 0;
        for (unsigned k = 0; k < 16; ++k) base += wave_total[k];
        __syncthreads();
    } else {
        base = nms_count_passing(segm, 0, p0, segm_thresh, wave_total);
    }
    {
        const int p = p0 + (int)tid;
        const bool pass = p < hw && segm[p] > segm_thresh;
        const unsigned long long m = __ballot(pass);
        if (lane == 0) wave_total[wv] = (unsigned)__popcll(m);
        __syncthreads();
        unsigned before = 0, total = 0;
        for (unsigned k = 0; k < 16; ++k) {
            const unsigned v = wave_total[k];
            if (k < wv) before += v;
            total += v;
        }
        const unsigned slot = base + before + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
        if (pass && slot < (unsigned)capacity) {
            const int y = p / w, x = p - y * w;
            const size_t q = (size_t)p, n = (size_t)hw;
            const float r0 = rbox[q], r1 = rbox[n + q], r2 = rbox[2 * n + q], r3 = rbox[3 * n + q];
            const float angle_sin = angle[q], angle_cos = angle[n + q];   // a[0], a[1] (:84-85)
            const float scale_factor = 4.0f, precision = 10000.0f;
            const float xp = (float)x + 0.25f, yp = (float)y + 0.25f;
            const float pos_r_x = (xp - r2 * angle_cos) * scale_factor;
            const float pos_r_y = (yp - r2 * angle_sin) * scale_factor;
            const float pos_r2_x = (xp + r3 * angle_cos) * scale_factor;
            const float pos_r2_y = (yp + r3 * angle_sin) * scale_factor;
            NmsCandidate c;
            c.quad[0] = (int)roundf(precision * (pos_r_x - r1 * angle_sin * scale_factor));
            c.quad[1] = (int)roundf(precision * (pos_r_y + r1 * angle_cos * scale_factor));
            c.quad[2] = (int)roundf(precision * (pos_r_x + r0 * angle_sin * scale_factor));
            c.quad[3] = (int)roundf(precision * (pos_r_y - r0 * angle_cos * scale_factor));
            c.quad[4] = (int)roundf(precision * (pos_r2_x + r0 * angle_sin * scale_factor));
            c.quad[5] = (int)roundf(precision * (pos_r2_y - r0 * angle_cos * scale_factor));
            c.quad[6] = (int)roundf(precision * (pos_r2_x - r1 * angle_sin * scale_factor));
            c.quad[7] = (int)roundf(precision * (pos_r2_y + r1 * angle_cos * scale_factor));
            c.score = segm[p];
            c.rdist[0] = r0;
            c.rdist[1] = r1;
            c.rdist[2] = r2;
            c.rdist[3] = r3;
            c.x = x;
            c.y = y;
            c.pad = 0;
            out[slot] = c;
        }
        // the workgroup of the last slab knows the total; it may exceed `capacity`: the caller sees how many there were
        if (blockIdx.x == gridDim.x - 1 && tid == 0) *count = (int)(base + total);
    }
}

// maps beyond 256 K pixels: passing pixels per 1024-pixel slab (the decode launch then sums the slabs before its own)
__global__ __launch_bounds__(1024) void rroi_rbox_count_kernel(const float* __restrict__ segm, int hw, float segm_thresh,
                                                              unsigned* __restrict__ slab_counts)
{
    __shared__ unsigned wave_total[16];
    const int p0 = (int)blockIdx.x * 1024;
    const unsigned total = nms_count_passing(segm, p0, min(hw, p0 + 1024), segm_thresh, wave_total);
    if (threadIdx.x == 0) slab_counts[blockIdx.x] = total;
}
