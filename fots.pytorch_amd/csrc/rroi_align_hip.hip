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
// (image, chunk) slice is one buffer descriptor, whose range check turns invalid
// taps into 0.0.  A workgroup (a gatherer and a storer wave) then owns a
// [32 channel] x [64 bin] output tile at a time: lanes = 8 bins x 8
// channel-quads fetch taps as 16-byte loads, blend in the reference's order,
// transpose through an LDS tile, and the tile streams out as full
// 256-byte row segments of the (R,C,PH,PW) tensor.  Channel chunk k is handled
// by blocks with blockIdx % nchunks == k, i.e. (8 chunks at C=256) by one XCD,
// whose 4 MiB L2 then holds exactly its 3.2 MB slice of the map.
//
// Files: this one holds the host side and the C-ABI; the device code is in parts that are
// included below, inside the anonymous namespace: rroi_device_common.h (constants, geometry
// recipe, descriptor helpers), rroi_forward_kernels.h, rroi_backward_kernels.h,
// rroi_callers_kernels.h, rroi_nms_kernels.h (+ rroi_nms_host.h, host C++).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <vector>

#include "rroi_align_hip.h"

#pragma clang fp contract(off)

namespace {

#include "rroi_device_common.h"
#include "rroi_forward_kernels.h"
#include "rroi_backward_kernels.h"
#include "rroi_backward_tile_kernels.h"
#include "rroi_callers_kernels.h"
#include "rroi_nms_kernels.h"
#include "rroi_nms_host.h"

// ------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------
inline int status_of(hipError_t e) { return e == hipSuccess ? 1 : -(int)e; }
inline int launch_status() { return status_of(hipGetLastError()); }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// CU count per device ordinal (a process may drive several GPUs, one per thread or per call):
// grids are sized for the device that is current at the call.
constexpr int kMaxDevices = 64;
std::atomic<int> g_cus[kMaxDevices];

int num_cus()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
    int cus = g_cus[dev].load(std::memory_order_relaxed);
    if (cus > 0) return cus;
    cus = 256;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) cus = p.multiProcessorCount;
    g_cus[dev].store(cus, std::memory_order_relaxed);
    return cus;
}

// Launch-shape constants of the shipped configuration.  The product reads them as compile-time constants.  The tools'
// exploration build (tools/rroi_align_hip_explore.hip: kbench, build_explore.sh) includes this file with
// RROI_TUNING_QUALIFIER defined empty -- a mutable struct -- and adds its rroi_align_debug_set_* setters BEHIND the
// include: nothing of it is in this file or in the product library.
struct Tuning {
    int row_pad = -1;             // chunk-major row pitch: -1 = W | 1 (see row_pitch)
    int waves_per_cu = 12;        // the backward's one-wave atomic scatter (rroi_bwd_tiled_kernel): 12.4 KB of LDS each
    // rroi_fwd_split_kernel: 62-64 VGPRs under __launch_bounds__(128, 6) and 12.1 KB of LDS (10 granules of 1280 B)
    // -> 12 workgroups per CU; a larger grid would run its surplus as a second round
    int split_wgs_per_cu = 12;
    int fwd_shift = 1;            // 1: crops with PH * PW % 16 != 0 take the SHIFT forms where they pay; 2: always; 0: never
    int shift_wgs_per_cu = 0;     // > 0 overrides the SHIFT kernels' workgroups per CU
    int fwd_dbg = 0;              // ablations: 1 = skip output stores, 2 = all taps out of range, 256 = free first item
    int prologue_blocks_per_cu = 3;
    int fwd_patch = 1;            // the direct path = K2p (round 5: shared geometry, row pairs as 8-byte loads); 0: rounds 1-4's thread-per-bin kernel
    int fwd_patch_waves = 4096;   // ... channel slabs sized for about this many waves ...
    int fwd_patch_cwave = 8;      // ... of at most this many channels each
    int fwd_fused = 1;            // AUTO may take the one-launch gather from the NCHW map for few ROIs (round 5)
    double fwd_fused_min_elems = 1.5e6;   // ... from this many output elements up (below: the direct kernel) ...
    int fwd_fused_min_channels = 128;     // ... and this many channels
    double fwd_tiled_min_elems = 3.8e6;   // ... and the two-launch path from this many up
    int fwd_shift_lines = 1;      // 1: line-aligned windows for rows that are not whole sectors beyond 320 MB of crops (round 5); 0: never; 2: always
    int shift_lines_wgs_per_cu = 10;
    int fwd_merge = 1;            // rows that are not whole sectors under XCD groups: strided tiles + plain stores where it pays (round 5); 0: SHIFT always; 2: always
    int fwd_groups = 1;           // XCD groups for nchunks in {1, 2} (round 5; XcdGroups in rroi_forward_kernels.h); 0: off
    int fwd_groups_min_rois = 64; // ... from this many ROIs up
    int bwd_buckets = 1;          // one-pass pixel lists (round 3); 0: count / scan / fill as in rounds 1-2
    int bwd_pair_aggregate = 1;       // bucket slots reserved per wave through LDS (round 5); 0: one atomic per pair
    int bwd_pair_blocks_per_cu = 0;   // 0: 256 / C, clamped to 2 .. 4 (see backward_impl)
    int bwd_tile_run = 2;         // the in-place NCHW gather: 2^v neighbouring key tiles per XCD turn
    int bwd_skip_dead = 1;        // the relayout of top_diff leaves out the bins that enter no list (round 4)
    int bwd_nchw_direct = 16;     // the list gather stores an NCHW bottom_diff itself (round 4): always for C <= 128, up to this
                                  // many bins per map pixel beyond; 0: never
};
#ifndef RROI_TUNING_QUALIFIER
#define RROI_TUNING_QUALIFIER constexpr
#endif
RROI_TUNING_QUALIFIER Tuning g_tune{};
// store policy of the backward's top_diff relayout (round 2 sweep, four runs: write-through (sc1) 163.4-165.9 us per
// call, streaming (nt) 166.8-168.8, plain 166.2-169.2): write-through leaves no dirty lines for the end of the launch
// to flush.  Non-temporal LOADS in the gather: +16 us.
constexpr int kBwdRelayoutAux = 16;

// Row pitch (pixels = 128-byte lines) of the chunk-major copy: W plus a pad that makes the
// pitch odd, so that the lines of vertically adjacent pixels differ in their low address
// bits and spread over the L2 channels.
int row_pitch(int width)
{
    if (g_tune.row_pad >= 0) return width + g_tune.row_pad;
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
    // tap offsets: < kQuadOOB = 2^30, so that (offset or kOOB) + (quad offset or kQuadOOB) never wraps
    if (slice_px * kLineBytes >= (1L << 30)) return false;                          // chunk-major
    if ((long)height * width * channels * 4 >= (1L << 30)) return false;            // channels-last
    if ((long)kChunk * pooled_height * pooled_width * 4 >= (1L << 31)) return false; // one output block
    if (slice_px * kChunk * nchunks >= (1L << 32)) return false;                    // img_stride
    if ((long)pooled_height * pooled_width >= (1L << 31)) return false;
    return true;
}

// grid for the tiled kernels: `per_cu` workgroups per CU (default: the 12 that 12.4 KB of LDS per workgroup admit; LDS is granted in 1280-byte granules, and a grid larger than the resident set would run its surplus
// blocks as a second, mostly empty round) and a multiple of lcm(nchunks, 8) so that blockIdx % nchunks is also
// stable per XCD.
int tiled_grid(long items, int nchunks, int per_cu = 0)
{
    long want = items * nchunks;
    const long cap = (long)num_cus() * (per_cu > 0 ? per_cu : g_tune.waves_per_cu);
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

// K2p, the direct path (rroi_fwd_patch_kernel): workgroup = (ROI, patch of up to 64 bins, four channel slabs of `cw` channels).
// Returns false where the form does not apply (the caller falls back to rounds 1-4's thread-per-bin kernel).
bool launch_patch_forward(const float* features, const float* rois, float* top_data, float* idx_x, float* idx_y, int num_rois,
                          int channels, int height, int width, int pooled_height, int pooled_width, float spatial_scale, int trig,
                          int batch_size, hipStream_t stream)
{
    const long NB = (long)pooled_height * pooled_width;
    if (!g_tune.fwd_patch || width < 2 || (long)num_rois * NB >= (1L << 30)) return false;
    // patch shape: rows x columns <= 64 with the fewest patches for this pooled size (ties: the wider rows -- longer store
    // runs); 11 x 96 -> 4 x 16 (18 patches), 11 x 83 -> 3 x 21 (16 instead of 18), 8 x 64 -> 4 x 16
    int prows = 4, pcols = 16;
    {
        long best = -1;
        // (at least three rows where the pooled height has them: a 1 x 64 or 2 x 32 patch is a long thin line of the map --
        // 11 x 128 with 1 x 64 patches: 11.0 us where 4 x 16 takes 8.6, although it needs two patches fewer)
        for (int r = std::min(3, pooled_height); r <= 8 && r <= pooled_height; ++r) {
            const int c = std::min(pooled_width, 64 / r);
            const long np = (long)ceil_div(pooled_height, r) * ceil_div(pooled_width, c);
            if (best < 0 || np < best || (np == best && c > pcols)) {
                best = np;
                prows = r;
                pcols = c;
            }
        }
    }
    const int npx = ceil_div(pooled_width, pcols), npatches = ceil_div(pooled_height, prows) * npx;
    long slabs = ceil_div((long)g_tune.fwd_patch_waves, (long)num_rois * npatches);
    if (slabs < 1) slabs = 1;
    int cw = (ceil_div(channels, slabs) + 3) / 4 * 4;
    if (cw > g_tune.fwd_patch_cwave) cw = g_tune.fwd_patch_cwave;
    const dim3 pgrid((unsigned)((long)num_rois * npatches), (unsigned)ceil_div(channels, 4 * cw), 1);
    if (pgrid.y > 65535u) return false;
    // (tools/patch_sweep.py: 2-8 K waves, 8-32 channels per wave, 4 or 8 in flight -- all within 0.3 us)
    if (idx_x)
        hipLaunchKernelGGL((rroi_fwd_patch_kernel<4, true>), pgrid, dim3(256), 0, stream, features, rois, top_data, num_rois, channels,
                           height, width, pooled_height, pooled_width, spatial_scale, trig, batch_size, cw, npx, npatches, prows, pcols,
                           idx_x, idx_y);
    else
        hipLaunchKernelGGL((rroi_fwd_patch_kernel<4, false>), pgrid, dim3(256), 0, stream, features, rois, top_data, num_rois, channels,
                           height, width, pooled_height, pooled_width, spatial_scale, trig, batch_size, cw, npx, npatches, prows, pcols,
                           (float*)nullptr, (float*)nullptr);
    return true;
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

PatchMap make_patch_map(int pooled_height, int pooled_width)
{
    unsigned pr = 1;   // rows of a patch: the largest power of two <= min(PH, 8)
    while (pr * 2 <= (unsigned)pooled_height && pr * 2 <= 8u) pr *= 2;
    PatchMap pm;
    pm.pc_shift = 0;
    while ((64u >> pm.pc_shift) > pr) ++pm.pc_shift;   // 64 / pr columns
    const unsigned pc = 1u << pm.pc_shift;
    const unsigned npy = ((unsigned)pooled_height + pr - 1) / pr;
    pm.npx = ((unsigned)pooled_width + pc - 1) / pc;
    pm.lanes_per_roi = npy * pm.npx * 64u;
    pm.div_roi = make_fastdiv(pm.lanes_per_roi);
    pm.div_npx = make_fastdiv(pm.npx);
    return pm;
}

struct Workspace {
    Affine* aff;
    int* sort_rank;    // XCD groups: the counting sort's scratch and ...
    int* sort_order;   // ... its result, ROI index by sorted position (R ints each)
    float* cm;
    size_t cm_bytes;
    size_t bytes;
};

// XCD groups (XcdGroups, rroi_forward_kernels.h): G = 8 / nchunks groups per chunk for one or two chunks (C <= 64)
// and enough ROIs; else one group, the mapping of rounds 1-4
int forward_groups(int num_rois, int nchunks)
{
    // measured (tools/groups_ab.py, profiles/r05_groups_ab.txt; us per call, one group / G groups): C = 64, two
    // 120 x 160 maps, 11 x 96: R = 512 38.3 / 33.3, R = 128 19.0 / 15.2, R = 32 12.0 / 12.4 (the sort's block is the last of
    // its launch to finish: few ROIs keep one group); C = 128 (G = 2, a 3.3 MB slice per XCD already): 33.4 / 33.5
    // the same with the cheap sort (r05_groups_ab2.txt): R = 512 38.2 / 32.1 (11 x 83: 39.3 / 32.9, 11 x 100: 42.0 / 35.3), R = 64
    // 14.3 / 12.7, R = 32 12.0 / 12.1; eight 160 x 160 maps, R = 512, 11 x 100: 57.3 / 47.3.  ROIs bunched in a third of
    // one image: +0.5 us.  Crops beyond the 256 MB memory-side cache: with the HALF-line windows of the SHIFT form the sorted
    // order scattered a moment's half lines over the whole tensor (R = 2048, 11 x 100, 577 MB: 260 / 289) and the first
    // version kept one group there; stores of whole lines gain more beyond the cache than anywhere (tools/big_crops_probe.py,
    // profiles/r05_big_crops.txt: rows of whole sectors, eight 160 x 160 maps, R = 2048, 11 x 96, 528 MB: 152.7 / 117.9; the
    // line-aligned windows that such sizes now take, 11 x 100: 192-208 / 167, 11 x 83: 173-185 / 145-151) -- a (roi, chunk)
    // block's 135 KB leave one XCD together -- and between 256 and 320 MB the groups bring the merging form with them
    // (tools/shift_lines_threshold.py: 267 MB 80-86 / 66-67, 275 MB 73-85 / 73-75, eight maps 118-123 / 112): no size limit.
    if (!g_tune.fwd_groups || (nchunks != 1 && nchunks != 2)) return 1;
    const int G = 8 / nchunks;
    return num_rois >= g_tune.fwd_groups_min_rois ? G : 1;
}

// [affine table | sort rank | sort order | chunk-major copy (B, nchunks, HW+1, 32)]; the copy is absent when
// channels-last features with C % 4 == 0 are consumed in place.
Workspace carve(void* ws, int batch_size, int channels, int height, int width, int num_rois,
                int layout)
{
    Workspace w;
    const size_t aff_bytes = align_up((size_t)(num_rois > 0 ? num_rois : 1) * sizeof(Affine), 256);
    const size_t sort_bytes = align_up((size_t)(num_rois > 0 ? num_rois : 1) * sizeof(int), 256);   // x 2: rank, order
    const size_t nchunks = (channels + kChunk - 1) / kChunk;
    w.cm_bytes = layout == RROI_LAYOUT_NHWC
                     ? 0
                     : align_up((size_t)batch_size * nchunks * ((size_t)height * row_pitch(width) + 1) * kLineBytes, 256);
    w.aff = reinterpret_cast<Affine*>(ws);
    w.sort_rank = reinterpret_cast<int*>(reinterpret_cast<char*>(ws) + aff_bytes);
    w.sort_order = reinterpret_cast<int*>(reinterpret_cast<char*>(ws) + aff_bytes + sort_bytes);
    w.cm = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + aff_bytes + 2 * sort_bytes);
    w.bytes = aff_bytes + 2 * sort_bytes + w.cm_bytes;
    return w;
}

// backward: [affine | chunk-major gradient | pixel counters | pixel offsets | pairs |
//            top_diff relaid out to (R, NB, nchunks * 32)]
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
    // one-pass lists (round 3): per-pixel buckets of 1 << kshift pairs in `pairs`, the overflow array behind
    // them, the chains' heads in `off`, the overflow counter in `bsum`
    bool bucket_ok, bucket_pref;
    unsigned kshift;
    uint4* ov;
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
    // bucket capacity: ~3x the average list (about 2 pairs per bin over the map's pixels), a power of two from 16
    // up -- as far as the buckets stay within max(64 MB, twice the exact lists' bytes): only the entries in use are
    // ever read, a large bucket costs address space, not bandwidth.  Where that cap leaves the bucket below the
    // AVERAGE list, most of a list would live in the (slow) chains and AUTO / TILED keep count / scan / fill.
    const double avg = 2.0 * (double)R * NB / ((double)batch_size * height * width);
    const size_t bucket_cap = std::max<size_t>((size_t)64 << 20, 2 * 4 * R * NB * sizeof(uint2));
    w.kshift = 4;
    while ((1u << w.kshift) < 3.0 * avg && w.kshift < 12 && (nkeys << (w.kshift + 1)) * sizeof(uint2) <= bucket_cap)
        ++w.kshift;
    w.bucket_pref = (double)(1u << w.kshift) >= avg;
    w.bucket_ok = (nkeys << w.kshift) < (1ull << 32) && 4 * R * NB < (1ull << 31);
    const size_t bucket_bytes = w.bucket_ok ? align_up((nkeys << w.kshift) * sizeof(uint2), 256) : 0;
    const size_t ov_bytes = w.bucket_ok ? align_up(4 * R * NB * sizeof(uint4), 256) : 0;
    const size_t pair_bytes = std::max(align_up(4 * R * NB * sizeof(uint2), 256), bucket_bytes + ov_bytes);
    const size_t td_bytes = align_up(R * (size_t)NB * nchunks * kLineBytes, 256);   // (R, NB, nchunks * 32)
    // pair slots, top_diff line indices and keys are 32-bit; the gather launches one thread group per key
    // ... and its thread index (key * lanes-per-pixel, at most 64) and block index are 32-bit too
    w.gather_ok = 4 * R * NB < (1ull << 32) && R * (size_t)NB < (1ull << 32) &&
                  (size_t)NB * nchunks * kLineBytes < (1ull << 32) && nkeys * 64 < (1ull << 32) && nkeys < (1ull << 31);
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
    // the big arrays start on 4 KiB boundaries of the ADDRESS: a bin's run of the pixel-major copy (nchunks x 128 B) then
    // never straddles a page, whatever the small arrays in front add up to (round 5, tools/experiments/
    // r05_clean_workspace_backward: with the copy 256 B further on configs[2]'s list launch took 84 us instead of 74;
    // page-aligned the call is 2 us faster than round 4's layout)
    auto page_up = [](char* p) { return reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(p) + 4095u) & ~(uintptr_t)4095u); };
    b = page_up(b);
    w.pairs = reinterpret_cast<uint2*>(b);
    w.ov = reinterpret_cast<uint4*>(b + bucket_bytes);
    b = page_up(b + pair_bytes);
    w.tdT = reinterpret_cast<float*>(b);
    b += td_bytes;
    // (+ 2 x 4 KiB: the two roundings, whatever the caller's base address is)
    w.bytes = aff_bytes + w.gcm_bytes +
              (w.gather_ok ? w.cnt_bytes + off_bytes + bsum_bytes + pair_bytes + td_bytes + 8192 : 0);
    return w;
}

// AUTO.  The tiled paths have a fixed cost (forward: two launches; backward: three) plus a term in the size of the
// whole map (relayout; the backward also visits every pixel); the direct paths cost per output element, the
// backward's with four float atomics.
// Measured crossovers (MI355X; forward: round 5, tools/fused_probe.py, us per call between events, direct = K2p):
//   forward   C=64 two 120x160 maps 11x96:  R=32 direct 8.8 / tiled 12.1,  R=48 12.7 / 13.4,  R=64 13.5 / 13.0,  R=128 21.9 / 14.9
//             C=256 160x160 8x64:  R=8 direct 6.1 / fused 7.3 / tiled 13.9,  R=16 11.5 / 9.7 / 14.4,  R=32 22.1 / 16.0 / 15.5
//   (rounds 1-4, tools/crossover.py, wall time per call:)
//   backward  C=256:  R=4 direct 59 / tiled 43,   R=32 448 / 50,   R=512 7145 / 195
//             C=64:   R=4 direct 15 / tiled 34,   R=16 105 / 39,   R=512 1745 / 113
bool pick_tiled_fwd(int batch_size, int channels, int height, int width, int num_rois, int NB)
{
    const double out_elems = (double)num_rois * channels * NB;
    const double map_elems = (double)batch_size * channels * height * width;
    return out_elems >= g_tune.fwd_tiled_min_elems && out_elems >= map_elems / 4;
}
// Below the two-launch path's crossover: the one-launch gather from the NCHW map (RROI_PATH_FUSED) or the direct path's
// patch kernel (K2p)?  Measured (tools/fused_probe.py, profiles/r05_fused_probe.txt; us per call, K2p / fused / two-launch):
//   C = 64, two 120 x 160 maps, 11 x 96:  R = 8  4.9 / 6.4 / 10.9,  16  7.3 / 7.9 / 11.0,  32  8.9 / 11.7 / 12.1,  64  12.7 / 17.9 / 13.0
//   C = 128, 160 x 160, 8 x 64:           R = 32  10.9 / 8.2 / 11.5
//   C = 256, 160 x 160, 8 x 64:           R = 8  6.5 / 7.4 / 13.9,  16  11.6 / 9.7 / 14.4,  32  21.0 / 16.1 / 15.5
// Both one-launch forms pay per output element (K2p three vector-memory instructions and ~16 VALU per 64 bin-channels, the
// fused form four dword loads per tap but one 16-byte store and one blend per FOUR channels) and meet at ~1.5 M elements;
// the fused form wins from there up to the two-launch crossover WHEN THERE ARE CHANNELS TO AMORTISE ITS ITEMS OVER: C >= 128.
// At the reference's own C = 64 K2p stays ahead up to the two-launch path (profiles/r05_small_r_forward.md).
bool pick_fused_fwd(int batch_size, int channels, int height, int width, int num_rois, int NB)
{
    (void)batch_size; (void)height; (void)width;
    const double out_elems = (double)num_rois * channels * NB;
    return g_tune.fwd_fused && channels >= g_tune.fwd_fused_min_channels && out_elems >= g_tune.fwd_fused_min_elems;
}
bool pick_tiled_bwd(int batch_size, int channels, int height, int width, int num_rois, int NB)
{
    const double out_elems = (double)num_rois * channels * NB;
    const double map_elems = (double)batch_size * channels * height * width;
    // the scatter's cost is the memset of the map plus its atomics: with the one-pass lists (round 3) the gather
    // wins from ~0.2 M gradient elements up (one 64 x 176 x 320 map, R = 4, 0.27 M: 26-27 us whatever the draw, the
    // scatter 18.6 or 41.4 depending on how the four ROIs overlap), and below that wherever the map alone is 8 M
    // elements (eight 64 x 160 x 160 maps, R = 4: 41.7 against 48.0 us); tools/crossover.py, tools/bucket_ab.py
    // ... unless the map dwarfs the gradient: the gather path moves the whole map three times (chunk-major gradient
    // written, read back, written as NCHW), the scatter once (its memset), and the scatter's atomics cost ~110 us
    // per million gradient elements (R = 16, C = 256: 250 us for 2.1 M)
    if (2.0 * map_elems * 4.0 / 7.0e6 > 110.0 * out_elems / 1.0e6 - 21.0 && map_elems >= 32.0e6) return false;
    return out_elems >= 0.2e6 || map_elems >= 8.0e6;
}


// ------------------------------------------------------------------------------------
// Which instantiation of rroi_fwd_split_kernel gathers a problem, with what grid and flags.
// ------------------------------------------------------------------------------------
enum class FwdKernel {
    kStrided,        // every n-th (roi, tile) item per workgroup, 16-byte stores: crops whose rows are whole sectors
    kChannelsLast,   // channels-last crops (R, PH, PW, C)
    kShift,          // SHIFT: overlapped tiles, sector-aligned store windows -- crops whose rows are not whole sectors
    kStridedMerge,   // strided tiles on rows that are not whole sectors, plain stores: the XCD's L2 merges the partial sectors (XCD groups only)
    kShiftLines,     // SHIFT == 2: line-aligned windows (32 own bins of 64 gathered) -- such crops beyond the memory-side cache
};
struct ForwardPlan {
    FwdKernel kernel;
    int grid;
    int ntiles;   // tiles per (roi, chunk) block: ceil(NB / 64), SHIFT: ceil(NB / 48)
    int dbg;      // the kernel's flag word (see its header comment)
};

constexpr int kShiftWgsPerCu = 12;   // the SHIFT instantiation: <= 80 VGPRs under __launch_bounds__(128, 6), 12.4 KB of LDS
constexpr int kShiftOwnBins = kTileBins - 16;   // bins a SHIFT tile advances by (it gathers 64: kOwnBins in the kernel)

// allow_lines = false: the one-launch NCHW_SRC form has no SHIFT == 2 instantiation -- rows that are not whole sectors
// take kShift there at any size (the plan used to return kShiftLines beyond 320 MB, which the fused launch
// then ran as SHIFT = 0 on a grid sized for 32-bin tiles)
ForwardPlan plan_forward_gather(int num_rois, int channels, int NB, int nchunks, bool out_nhwc, bool launcher_rest, int groups = 1,
                                size_t map_bytes_per_xcd = 0, bool allow_lines = true)
{
    const int base_dbg = (g_tune.fwd_dbg & ~0xe0) | (launcher_rest ? 32 : 0);   // bits 5-7 are the host's
    const int ntiles = ceil_div(NB, kTileBins);
    if (out_nhwc)   // (91 VGPRs -> five waves per SIMD: 10 workgroups per CU are resident, and no more are launched)
        return {FwdKernel::kChannelsLast, tiled_grid((long)num_rois * ntiles, nchunks, 10), ntiles, g_tune.fwd_dbg & ~0xe0};
    // Crops whose rows are not whole 64-byte sectors (PH * PW % 16 != 0) take the SHIFT form: sector-aligned store windows
    // over OVERLAPPED tiles (a tile advances by 48 bins and gathers 64, rroi_forward_kernels.h), items dealt every n-th
    // like the strided form's.  It costs 4 / 3 of the gather work per byte and was never slower than the strided items on
    // such crops, from R = 8 to R = 2048 (tools/align_probe.py, profiles/r04_align_probe.txt: R = 32, C = 64, 11 x 83: 6.6
    // against 9.4 us; R = 128, 11 x 100: 13.7 against 16.5; R = 512, 11 x 83: 40 against 209)
    // Round 5: beyond the 256 MB memory-side cache a half line reaches HBM as a half line -- the SHIFT
    // form's store stream ALONE runs at 3.1 TB/s there (tools/big_crops_ablate.py).  SHIFT == 2 stores whole LINES: a tile
    // advances by 32 bins and gathers 64 (twice the gather work per byte instead of 4 / 3).  tools/shift_lines_ab.py,
    // profiles/r05_shift_lines.txt (us per call, SHIFT / lines): eight 160 x 160 maps, C = 64, R = 2048, 11 x 100 (550 MB) 255 / 191;
    // two 120 x 160 maps, 11 x 83: R = 2048 (456 MB) 172 / 141, R = 4096 (913 MB) 358 / 272; C = 256, 11 x 50, R = 1024 (550 MB)
    // 251 / 176; 11 x 100, R = 600 (645 MB) 215 / 183.  Around the cache's size it is a draw or worse (267-275 MB: 80 / 87,
    // 73 / 87, 104 / 93, 76 / 83; 322-334 MB: 105 / 97, 112 / 108): from 320 MB up.
    if (allow_lines && NB % 16 != 0 && (g_tune.fwd_shift_lines == 2 ||
                         (g_tune.fwd_shift_lines == 1 && (size_t)num_rois * channels * NB * sizeof(float) > ((size_t)320 << 20)))) {
        const int nt = ceil_div(NB, kTileBins - 32);
        return {FwdKernel::kShiftLines, tiled_grid((long)num_rois * nt, nchunks, g_tune.shift_lines_wgs_per_cu), nt, base_dbg};
    }
    // Round 5, with XCD groups: ALL tiles of a (roi, chunk) block go through ONE XCD at about the same time, so the partial
    // sectors that strided tiles leave at their ends on rows that are not whole sectors meet in that XCD's L2 -- if the
    // stores are plain (write-back) instead of streamed.  Then such crops need no SHIFT form (4 / 3 of the items): strided
    // tiles, 16-byte stores at dword alignment, a row's last <= 3 dwords in its last tile (WAUX = 0 in the kernel).
    // tools/merge_ab.py, profiles/r05_merge_ab.txt (us per call, SHIFT / merging): C = 64, two 120 x 160 maps, R = 512: 11 x 83
    // 32.9 / 30.1, 11 x 84 32.2 / 29.6, 11 x 91 34.5 / 33.0, 11 x 100 35.4 / 34.8; C = 32 20.9 / 19.9.  It needs the L2 room:
    // R = 128 15.7 / 15.6, R = 64 12.5 / 13.1, eight 160 x 160 maps (6.6 MB of map per XCD) 47.1 / 49.8 -- so: XCD groups, at
    // least 48 MB of crops, at most 2 MB of the chunk-major copy per XCD.
    if (g_tune.fwd_merge && groups > 1 && NB % 16 != 0 && (g_tune.fwd_merge > 1 ||
        ((size_t)num_rois * channels * NB * sizeof(float) >= ((size_t)48 << 20) && map_bytes_per_xcd <= ((size_t)2 << 20))))
        return {FwdKernel::kStridedMerge, tiled_grid((long)num_rois * ntiles, nchunks, g_tune.split_wgs_per_cu), ntiles, base_dbg};
    if (NB % 4 != 0 || (g_tune.fwd_shift && (NB % 16 != 0 || g_tune.fwd_shift == 2))) {   // (rows of dwords: always)
        const int wpc = g_tune.shift_wgs_per_cu > 0 ? g_tune.shift_wgs_per_cu : kShiftWgsPerCu;
        const int nt = ceil_div(NB, kShiftOwnBins);
        return {FwdKernel::kShift, tiled_grid((long)num_rois * nt, nchunks, wpc), nt, base_dbg};
    }
    return {FwdKernel::kStrided, tiled_grid((long)num_rois * ntiles, nchunks, g_tune.split_wgs_per_cu), ntiles, base_dbg};
}

// ------------------------------------------------------------------------------------
// Scratch of the reference-ABI launchers.  Their signatures carry no workspace, so the library keeps
// one buffer per (device, stream), grown on demand and reused: calls on one stream are ordered, so the
// next call may overwrite what the previous one left.  No allocator round trip per call, and a call
// whose buffer exists enqueues kernels only -- it can be captured into a HIP graph.
// Lifetime and locking rules:
//   * a buffer that has been handed out WHILE ITS STREAM WAS CAPTURING is baked into a graph: it is PINNED and from
//     then on GRAPH-EXCLUSIVE -- never freed (not by a later, larger call, not by the least-recently-used eviction,
//     not by rroi_align_release_launcher_scratch()) and never handed out again: later eager calls on that stream get
//     a buffer of their own (a second table entry), later captures take stream-ordered memory that their graph owns.
//     So a replay of the graph -- on whatever stream -- shares its scratch with nothing but itself;
//   * a call made while capturing that finds no unpinned buffer of its stream (or one that is too small) takes
//     stream-ordered memory for that call alone: the graph owns it;
//   * locks: the TABLE lock covers look-up, creation and eviction only; every entry has its OWN lock, taken before
//     the table lock is dropped and held until the caller has ENQUEUED its launches (ScratchLease) -- so only callers
//     of the same (device, stream) entry serialise, and a second thread sharing the stream cannot free, in stream
//     order and ahead of those launches, a buffer that the first thread is about to launch on.  Lock order: table,
//     then entry; a lease holder never takes the table lock.
// ------------------------------------------------------------------------------------
struct LauncherArena {
    bool used = false;
    int device = 0;
    hipStream_t stream = nullptr;
    void* ptr = nullptr;
    size_t bytes = 0;
    unsigned long long last_use = 0;
    bool pinned = false;   // handed out during a stream capture: a graph replays with this address
    unsigned long long capture_id = 0;   // ... the capture that pinned it: later calls of the SAME capture may reuse it
    std::mutex in_use;     // held by the lease of the call that is enqueuing on this buffer
};
constexpr int kMaxArenas = 64;   // pinned entries keep their slot for the life of the process (documented in the header)
std::mutex g_table_mutex;
LauncherArena g_arenas[kMaxArenas];
unsigned long long g_arena_clock = 0;
std::atomic<unsigned long long> g_transient_calls{0};

// A buffer of at least `bytes` for launches on `stream`, held under its entry's lock until give_back().
struct ScratchLease {
    std::unique_lock<std::mutex> lock;   // the entry's (empty for a transient buffer)
    void* ptr = nullptr;
    bool transient = false;   // the buffer belongs to this call alone: give_back() returns it in stream order
    hipError_t err = hipSuccess;
    hipError_t give_back(hipStream_t stream)
    {
        hipError_t e = hipSuccess;
        if (transient && ptr) e = hipFreeAsync(ptr, stream);
        ptr = nullptr;
        if (lock.owns_lock()) lock.unlock();
        return e;
    }
};

ScratchLease launcher_scratch(hipStream_t stream, size_t bytes)
{
    ScratchLease L;
    int dev = 0;
    if ((L.err = hipGetDevice(&dev)) != hipSuccess) return L;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    unsigned long long cap_id = 0;
    const bool capturing = hipStreamGetCaptureInfo(stream, &cap, &cap_id) == hipSuccess && cap != hipStreamCaptureStatusNone;
    auto take_transient = [&]() {
        L.err = hipMallocAsync(&L.ptr, bytes, stream);
        L.transient = L.err == hipSuccess;
        if (!L.transient) L.ptr = nullptr;
        g_transient_calls.fetch_add(1, std::memory_order_relaxed);
    };
    std::unique_lock<std::mutex> table(g_table_mutex);
    LauncherArena* a = nullptr;
    std::unique_lock<std::mutex> mine;
    for (;;) {
        // the stream's own entry: the one THIS capture pinned earlier (a second launcher call inside one capture reuses
        // it instead of taking mem-alloc nodes), else its unpinned one (a pinned one belongs to its graph)
        a = nullptr;
        for (LauncherArena& e : g_arenas)
            if (e.used && e.device == dev && e.stream == stream && capturing && e.pinned && e.capture_id == cap_id && e.bytes >= bytes)
                a = &e;
        if (!a)
            for (LauncherArena& e : g_arenas)
                if (e.used && !e.pinned && e.device == dev && e.stream == stream) a = &e;
        if (!a) break;
        mine = std::unique_lock<std::mutex>(a->in_use, std::try_to_lock);
        if (mine.owns_lock()) break;
        // another thread is enqueuing on this entry: wait for it WITHOUT the table lock (holding it stalled
        // every other stream's look-up), then look the entry up again -- it may have grown, been pinned or evicted
        table.unlock();
        { std::lock_guard<std::mutex> wait(a->in_use); }   // (the entries are static: the mutex outlives any eviction)
        table.lock();
    }
    if (a) {
        if (a->bytes >= bytes) {
            a->last_use = ++g_arena_clock;
            if (capturing && !a->pinned) {             // this graph's from now on
                a->pinned = true;
                a->capture_id = cap_id;
            }
            L.ptr = a->ptr;
            L.lock = std::move(mine);
            return L;
        }
        if (capturing) {   // too small, and the graph owns what it allocates: the entry stays as it is
            mine.unlock();
            table.unlock();
            take_transient();
            return L;
        }
        // grow: the old buffer goes back in stream order, behind the launches that still use it (every user held
        // this entry's lock until its launches were enqueued, so none of them can come after this free)
        if ((L.err = hipFreeAsync(a->ptr, stream)) != hipSuccess) return L;
        a->ptr = nullptr;
        a->bytes = 0;
        void* p = nullptr;
        if ((L.err = hipMallocAsync(&p, bytes, stream)) != hipSuccess) {
            a->used = false;
            return L;
        }
        a->ptr = p;
        a->bytes = bytes;
        a->last_use = ++g_arena_clock;
        L.ptr = p;
        L.lock = std::move(mine);
        return L;
    }
    if (capturing) {   // nothing cached for this stream: the graph owns what it allocates
        table.unlock();
        take_transient();
        return L;
    }
    // a new entry: a free slot, or the least recently used buffer that no graph holds and nobody is enqueuing on
    LauncherArena* slot = nullptr;
    for (LauncherArena& e : g_arenas)
        if (!e.used && !slot) slot = &e;
    if (slot) {
        mine = std::unique_lock<std::mutex>(slot->in_use);
    } else {
        LauncherArena* lru = nullptr;
        for (LauncherArena& e : g_arenas)
            if (!e.pinned && (!lru || e.last_use < lru->last_use)) lru = &e;
        if (lru) mine = std::unique_lock<std::mutex>(lru->in_use, std::try_to_lock);
        if (!lru || !mine.owns_lock()) {   // every entry is pinned (or the candidate is busy): nothing to cache in
            table.unlock();
            take_transient();
            return L;
        }
        int cur = dev;   // (its stream may be gone: a synchronous free)
        (void)hipSetDevice(lru->device);
        (void)hipFree(lru->ptr);
        (void)hipSetDevice(cur);
        lru->used = false;
        slot = lru;
    }
    void* p = nullptr;
    if ((L.err = hipMallocAsync(&p, bytes, stream)) != hipSuccess) return L;
    slot->used = true;
    slot->device = dev;
    slot->stream = stream;
    slot->ptr = p;
    slot->bytes = bytes;
    slot->pinned = false;
    slot->last_use = ++g_arena_clock;
    L.ptr = p;
    L.lock = std::move(mine);
    return L;
}

}  // namespace

// ====================================================================================
extern "C" {

const char* rroi_align_hip_version(void) { return "rroi_align_hip 0.8.0 gfx950"; }

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

static int forward_impl(const float* features, int feature_layout, int top_layout, float spatial_scale,
                        int batch_size, int num_rois, int height, int width, int channels,
                        int pooled_height, int pooled_width, const float* rois, float* top_data,
                        void* workspace, size_t workspace_bytes, int path, int stages, void* stream_,
                        bool launcher_rest = false);

int rroi_align_forward_stages_hip(const float* features, int feature_layout, float spatial_scale,
                                  int batch_size, int num_rois, int height, int width,
                                  int channels, int pooled_height, int pooled_width,
                                  const float* rois, float* top_data, void* workspace,
                                  size_t workspace_bytes, int path, int stages, void* stream_)
{
    return forward_impl(features, feature_layout, RROI_LAYOUT_NCHW, spatial_scale, batch_size, num_rois,
                        height, width, channels, pooled_height, pooled_width, rois, top_data, workspace,
                        workspace_bytes, path, stages, stream_);
}

int rroi_align_forward_layout_hip(const float* features, int feature_layout, int top_layout,
                                  float spatial_scale, int batch_size, int num_rois, int height,
                                  int width, int channels, int pooled_height, int pooled_width,
                                  const float* rois, float* top_data, void* workspace,
                                  size_t workspace_bytes, int path, void* stream_)
{
    return forward_impl(features, feature_layout, top_layout, spatial_scale, batch_size, num_rois, height,
                        width, channels, pooled_height, pooled_width, rois, top_data, workspace,
                        workspace_bytes, path, RROI_STAGE_ALL, stream_);
}

// launcher_rest (the reference-ABI launcher; tiled NCHW path): ROIs whose image index is >= batch_size are
// sampled from the NCHW tensor by extra blocks of the prologue launch and left alone by the gather
static int forward_impl(const float* features, int feature_layout, int top_layout, float spatial_scale,
                        int batch_size, int num_rois, int height, int width, int channels,
                        int pooled_height, int pooled_width, const float* rois, float* top_data,
                        void* workspace, size_t workspace_bytes, int path, int stages, void* stream_,
                        bool launcher_rest)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (path & ~(0xff | RROI_PATH_TRIG_FP32)) return 0;   // unknown flag bits
    const int trig = (path & RROI_PATH_TRIG_FP32) ? RROI_TRIG_FP32 : RROI_TRIG_DOUBLE;
    path &= 0xff;
    if ((stages & ~RROI_STAGE_ALL) || stages == 0) return 0;
    if (top_layout != RROI_LAYOUT_NCHW && top_layout != RROI_LAYOUT_NHWC) return 0;
    const bool out_nhwc = top_layout == RROI_LAYOUT_NHWC;
    // channels-last crops are written by the tiled kernel only: 16-byte channel quads
    if (out_nhwc && (channels % 4 != 0 || path == RROI_PATH_DIRECT ||
                     (long)pooled_height * pooled_width * channels * 4 >= (1L << 31)))
        return 0;
    if (!shape_ok(batch_size, num_rois, height, width, channels, pooled_height, pooled_width))
        return 0;
    if (feature_layout != RROI_LAYOUT_NCHW && feature_layout != RROI_LAYOUT_NHWC) return 0;
    if (path != RROI_PATH_AUTO && path != RROI_PATH_DIRECT && path != RROI_PATH_TILED && path != RROI_PATH_FUSED) return 0;
    if (num_rois == 0) return 1;
    if (!features || !rois || !top_data) return 0;
    const int NB = pooled_height * pooled_width;

    bool tiled;
    if (out_nhwc)
        tiled = true;
    else if (path == RROI_PATH_AUTO)
        tiled = feature_layout == RROI_LAYOUT_NHWC ||
                pick_tiled_fwd(batch_size, channels, height, width, num_rois, NB);
    else
        tiled = path == RROI_PATH_TILED;
    // the one-launch form for few ROIs (the gather reading the NCHW map itself): NCHW in, NCHW out, not for the launcher
    const bool fused = !launcher_rest && !out_nhwc && feature_layout == RROI_LAYOUT_NCHW &&
                       (path == RROI_PATH_FUSED ||
                        (path == RROI_PATH_AUTO && !tiled && pick_fused_fwd(batch_size, channels, height, width, num_rois, NB)));
    if (path == RROI_PATH_FUSED && !fused) return 0;
    if (fused) {
        if (!(stages & RROI_STAGE_GATHER)) return 1;  // one launch, run under the gather stage
        const int nchunks = ceil_div(channels, kChunk);
        const ForwardPlan plan = plan_forward_gather(num_rois, channels, NB, nchunks, false, false, 1, 0, /*allow_lines*/ false);
        if (plan.kernel != FwdKernel::kShift && plan.kernel != FwdKernel::kStrided) return 0;   // the two forms NCHW_SRC has
        if ((long)num_rois * plan.ntiles >= (1L << 31)) return 0;
        const unsigned HWu = (unsigned)height * (unsigned)width;
        SliceLayout lay;
        lay.px_bytes = 4u;                       // a "pixel" of a channel plane
        lay.row_bytes = (unsigned)width * 4u;
        lay.slice_bytes = HWu * 4u;              // ONE plane: the kernel's descriptor covers the chunk's planes < C
        lay.chunk_stride = kChunk * HWu;         // floats (shape_ok: C * H * W * 4 < 2^30)
        lay.img_stride = (unsigned)channels * HWu;
        const FastDiv dt = make_fastdiv((unsigned)plan.ntiles), dp = make_fastdiv((unsigned)pooled_width);
        const RoiSource rsrc = {rois, pooled_height, spatial_scale, trig};
#define RROI_FUSED(...)                                                                                                   \
    hipLaunchKernelGGL((rroi_fwd_split_kernel<__VA_ARGS__>), dim3(plan.grid), dim3(2 * kWave), 0, stream, features,          \
                       (const Affine*)nullptr, top_data, num_rois, channels, height, width, pooled_width, NB, batch_size,    \
                       nchunks, plan.ntiles, lay, dt, dp, plan.dbg, XcdGroups{1, nullptr}, rsrc)
        if (plan.kernel == FwdKernel::kShift) RROI_FUSED(true, 0, 4, 3, false, 1, true);
        else RROI_FUSED(true, 0, 4, 3, false, 0, true);
#undef RROI_FUSED
        return launch_status();
    }
    if (!tiled && feature_layout != RROI_LAYOUT_NCHW) return 0;  // direct path reads NCHW only

    if (!tiled) {
        if (!(stages & RROI_STAGE_GATHER)) return 1;  // the direct path has no prologue
        if (launch_patch_forward(features, rois, top_data, nullptr, nullptr, num_rois, channels, height, width, pooled_height,
                                 pooled_width, spatial_scale, trig, batch_size, stream))
            return launch_status();
        dim3 grid;
        int cslab;
        direct_grid(num_rois, NB, channels, grid, cslab);
        hipLaunchKernelGGL(rroi_fwd_direct_kernel, grid, dim3(256), 0, stream, features, rois,
                           top_data, (float*)nullptr, (float*)nullptr, num_rois, channels, height,
                           width, pooled_height, pooled_width, spatial_scale, trig, batch_size, cslab);
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
    const int groups = launcher_rest ? 1 : forward_groups(num_rois, nchunks);

    // prologue: relayout + affine table in one launch
    if (stages & RROI_STAGE_PROLOGUE) {
        const int ptiles = ceil_div(HW, kRelayoutPx);
        const int relayout_tiles = zero_copy ? 0 : ptiles * nchunks * batch_size;
        // ~3 resident blocks per CU, each streaming several tiles with the next tile prefetched
        int relayout_blocks = relayout_tiles;
        if (groups > 1) relayout_blocks = (relayout_blocks + 7) / 8 * 8;   // whole XCD rounds (a block without a tile leaves)
        if (relayout_blocks > num_cus() * g_tune.prologue_blocks_per_cu) {
            relayout_blocks = num_cus() * g_tune.prologue_blocks_per_cu;
            long unit = nchunks;
            while (unit % 8) unit += nchunks;  // lcm(nchunks, 8): keeps block -> chunk -> XCD stable
            if (relayout_blocks >= unit) relayout_blocks = (int)(relayout_blocks / unit * unit);
        }
        const int aff_blocks = ceil_div(num_rois, 256);
        const int rest_blocks = launcher_rest ? num_rois : 0;
#define RROI_LAUNCH_PRO(AUX)                                                                        \
    hipLaunchKernelGGL(rroi_prologue_kernel<AUX>,                                                     \
                       dim3(relayout_blocks + aff_blocks + (groups > 1 ? 1 : 0) + rest_blocks), dim3(256), 0, \
                       stream, features, ws.cm, channels, HW, width, pitch,                           \
                       make_fastdiv((unsigned)width), nchunks, ptiles, relayout_blocks,               \
                       relayout_tiles, batch_size, rois, num_rois, pooled_height,                     \
                       spatial_scale, trig, ws.aff, aff_blocks, launcher_rest ? top_data : (float*)nullptr, \
                       pooled_width, groups, ws.sort_rank, ws.sort_order)
        RROI_LAUNCH_PRO(0);   // plain stores: the copy stays in the L2s that wrote it (write-through: 1.8 us faster alone, the step is not)
#undef RROI_LAUNCH_PRO
        const int st = launch_status();
        if (st != 1) return st;
    }
    if (stages & RROI_STAGE_GATHER) {
        const ForwardPlan plan = plan_forward_gather(num_rois, channels, NB, nchunks, out_nhwc, launcher_rest, groups,
                                                     zero_copy ? (size_t)batch_size * HW * channels * 4 / 8 : ws.cm_bytes / 8);
        const int ntiles = plan.ntiles;
        if ((long)num_rois * ntiles >= (1L << 31)) return 0;
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
        // the shipped instantiations of rroi_fwd_split_kernel<VEC_STORE, EARLY, OCC, HID, ONHWC, SHIFT>, one per FwdKernel
#define RROI_GATHER(...)                                                                                              \
    hipLaunchKernelGGL((rroi_fwd_split_kernel<__VA_ARGS__>), dim3(plan.grid), dim3(2 * kWave), 0, stream, map, ws.aff, \
                       top_data, num_rois, channels, height, width, pooled_width, NB, batch_size, nchunks, ntiles, lay, \
                       dt, dp, plan.dbg, XcdGroups{groups, ws.sort_order})
        switch (plan.kernel) {
        case FwdKernel::kStrided:       RROI_GATHER(true, 0, 6, 3, false, 0); break;   // 62-64 VGPRs, 12.1 KB of LDS: 12 per CU
        case FwdKernel::kChannelsLast:  RROI_GATHER(true, 2, 5, 2, true, 0); break;    // 91 VGPRs: 10 per CU
        case FwdKernel::kShift:         RROI_GATHER(true, 0, 6, 3, false, 1); break;    // 79 VGPRs, 12.4 KB of LDS: 12 per CU
        case FwdKernel::kStridedMerge:  RROI_GATHER(true, 0, 6, 3, false, 0, false, 0); break;   // plain stores (write-through: 32.1 against 30.1 us)
        case FwdKernel::kShiftLines:    RROI_GATHER(true, 0, 5, 3, false, 2); break;    // 84 VGPRs, 14.8 KB of LDS: 10 per CU
        }
#undef RROI_GATHER
    }
    return launch_status();
}

int rroi_align_backward_hip(const float* top_diff, float spatial_scale, int batch_size,
                            int num_rois, int height, int width, int channels,
                            int pooled_height, int pooled_width, const float* rois,
                            float* bottom_diff, void* workspace, size_t workspace_bytes, int path,
                            void* stream_)
{
    return rroi_align_backward_layout_hip(top_diff, RROI_LAYOUT_NCHW, RROI_LAYOUT_NCHW, spatial_scale,
                                          batch_size, num_rois, height, width, channels, pooled_height,
                                          pooled_width, rois, bottom_diff, workspace, workspace_bytes, path,
                                          stream_);
}

static int backward_impl(const float* top_diff, int top_diff_layout, int bottom_diff_layout,
                         float spatial_scale, int batch_size, int num_rois, int height, int width, int channels,
                         int pooled_height, int pooled_width, const float* rois, float* bottom_diff,
                         void* workspace, size_t workspace_bytes, int path, void* stream_, bool accumulate);

int rroi_align_backward_layout_hip(const float* top_diff, int top_diff_layout, int bottom_diff_layout,
                                   float spatial_scale, int batch_size, int num_rois, int height,
                                   int width, int channels, int pooled_height, int pooled_width,
                                   const float* rois, float* bottom_diff, void* workspace,
                                   size_t workspace_bytes, int path, void* stream_)
{
    return backward_impl(top_diff, top_diff_layout, bottom_diff_layout, spatial_scale, batch_size, num_rois, height,
                         width, channels, pooled_height, pooled_width, rois, bottom_diff, workspace, workspace_bytes,
                         path, stream_, false);
}

// accumulate (the reference-ABI launcher, tiled NCHW paths only): bottom_diff += gradient instead of = gradient
// AUTO / TILED choose among the gathers; LISTS and INKERNEL name one
static inline bool gather_choice_is_open(int path)
{
    return path != RROI_PATH_TILED_LISTS && path != RROI_PATH_TILED_INKERNEL;
}

static int backward_impl(const float* top_diff, int top_diff_layout, int bottom_diff_layout,
                         float spatial_scale, int batch_size, int num_rois, int height, int width, int channels,
                         int pooled_height, int pooled_width, const float* rois, float* bottom_diff,
                         void* workspace, size_t workspace_bytes, int path, void* stream_, bool accumulate)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (path & ~(0xff | RROI_PATH_TRIG_FP32)) return 0;   // unknown flag bits
    const int trig = (path & RROI_PATH_TRIG_FP32) ? RROI_TRIG_FP32 : RROI_TRIG_DOUBLE;
    path &= 0xff;
    if (top_diff_layout != RROI_LAYOUT_NCHW && top_diff_layout != RROI_LAYOUT_NHWC) return 0;
    if (bottom_diff_layout != RROI_LAYOUT_NCHW && bottom_diff_layout != RROI_LAYOUT_NHWC) return 0;
    const bool td_nhwc = top_diff_layout == RROI_LAYOUT_NHWC, bd_nhwc = bottom_diff_layout == RROI_LAYOUT_NHWC;
    // channels-last tensors are read / written in place by the gather formulation only
    if ((td_nhwc || bd_nhwc) &&
        (channels % 4 != 0 || path == RROI_PATH_DIRECT || path == RROI_PATH_TILED_ATOMIC))
        return 0;
    if (!shape_ok(batch_size, num_rois, height, width, channels, pooled_height, pooled_width))
        return 0;
    if (path != RROI_PATH_AUTO && path != RROI_PATH_DIRECT && path != RROI_PATH_TILED &&
        path != RROI_PATH_TILED_ATOMIC && path != RROI_PATH_TILED_LISTS && path != RROI_PATH_TILED_INKERNEL &&
        path != RROI_PATH_TILED_BUCKETS)
        return 0;
    if (!bottom_diff) return 0;
    const int NB = pooled_height * pooled_width;
    const size_t HW = (size_t)height * width;
    const size_t in_bytes = (size_t)batch_size * channels * HW * sizeof(float);
    if (num_rois == 0) return status_of(hipMemsetAsync(bottom_diff, 0, in_bytes, stream));
    if (!top_diff || !rois) return 0;

    const bool tiled = td_nhwc || bd_nhwc || (path == RROI_PATH_AUTO
                                       ? pick_tiled_bwd(batch_size, channels, height, width, num_rois, NB)
                                       : path != RROI_PATH_DIRECT);
    if (accumulate && (!tiled || bd_nhwc)) return 0;
    if (!tiled) {
        hipError_t e = hipMemsetAsync(bottom_diff, 0, in_bytes, stream);
        if (e != hipSuccess) return status_of(e);
        dim3 grid;
        int cslab;
        direct_grid(num_rois, NB, channels, grid, cslab);
        hipLaunchKernelGGL(rroi_bwd_direct_kernel, grid, dim3(256), 0, stream, top_diff, rois,
                           bottom_diff, num_rois, channels, height, width, pooled_height,
                           pooled_width, spatial_scale, trig, batch_size, cslab);
        return launch_status();
    }

    const BwdWorkspace ws = carve_bwd(workspace, batch_size, channels, height, width, num_rois, NB);
    if (!workspace || workspace_bytes < ws.bytes) return 0;
    if ((td_nhwc || bd_nhwc) && (!ws.gather_ok || (size_t)num_rois * NB >= (1ull << 32))) return 0;
    const int nchunks = ceil_div(channels, kChunk);
    const int pitch = row_pitch(width);
    const int ptiles = ceil_div((long)HW, kRelayoutPx);
    const bool gather = path != RROI_PATH_TILED_ATOMIC && ws.gather_ok;
    // Two gathers.  K3t builds the pixel lists inside the gather kernel (rroi_backward_tile_kernels.h), K3g
    // with count / scan / fill launches in HBM.  Measured (tools/crossover.py, MI355X, us per call,
    // K3t / K3g): C = 64, 176x320 map, 11x96: R = 4 26 / 33, 32 31 / 37, 128 41 / 49, 512 88 / 98;  C = 64,
    // 8 images of 160x160, 11x100: 45 / 51, 54 / 54, 68 / 72, 133 / 148;  C = 256, 160x160, 8x64: 39 / 39,
    // 43 / 40, 77 / 62, 201 / 168 -- K3t's serial work per tile is hidden when a lane carries two
    // channel chunks, not when it carries eight.  K3t addresses its source with 32-bit byte offsets.
    const size_t src_bytes = td_nhwc ? (size_t)num_rois * NB * channels * 4
                                     : (size_t)num_rois * NB * nchunks * kLineBytes;
    const bool inkernel_ok = src_bytes < (1ull << 32);
    if (path == RROI_PATH_TILED_INKERNEL && !(gather && inkernel_ok)) return 0;
    // ... with four chunks per lane (C <= 128) it still wins where the lists are short: C = 128, 160x160,
    // 8x64: R = 32 33 / 42, 128 48 / 48, 512 101 / 93;  C = 96, 176x320, 11x96: 38 / 45, 56 / 61, 114 / 122
    const bool short_lists = (double)num_rois * NB <= 8.0 * (double)batch_size * HW;   // bins per map pixel
    // every map tile's workgroup scans ALL the ROIs (in batches of 256): O(tiles x R), measured up to R = 512
    // and 8 images -- beyond that the lists in HBM, whose cost does not grow that way, are the safe choice
    const bool scan_ok = (double)num_rois * batch_size <= 8192.0;
    // round 3 (profiles/r03_crossover.txt): up to 256 channels it also wins while there is at most one bin per map
    // pixel -- C = 256, 160 x 160, 8 x 64: R = 4 45.9 / 52.8, 16 49.5 / 52.5, 32 49.7 / 53.7, 64 56.9 / 57.4, 128 78.3 / 65.5
    const bool very_short = (double)num_rois * NB <= 1.0 * (double)batch_size * HW;
    const bool prefer_inkernel = scan_ok && (nchunks <= 2 || (nchunks <= 4 && short_lists) || (nchunks <= 8 && very_short));
    if (path == RROI_PATH_TILED_BUCKETS && !(gather && ws.bucket_ok)) return 0;
    // Round 3: the lists in HBM built in ONE pass over the bins (fixed buckets of 2^kshift entries per pixel plus
    // overflow chains, rroi_backward_kernels.h) instead of count / scan / fill.  Measured (tools/crossover.py,
    // tools/bucket_ab.py, profiles/r03_crossover_buckets.txt; us per call, buckets / exact lists / K3t): cfg3 129 /
    // 144 / 179;  C = 256 R = 16 37 / 53 / 48;  C = 64 176x320 R = 128 40 / 48 / 43;  8 images of 160x160 R = 64
    // 49 / 61 / 61 -- they win wherever the bucket holds at least the average list (128 entries per pixel in buckets
    // of 128: 74 / 89 / 164) and lose where most of a list lives in the chains (512 per pixel in buckets of 128:
    // 200 / 128 / 318): the bucket grows with the density as far as carve_bwd's cap lets it, bucket_pref says if
    // that was far enough.
    const bool buckets = gather && ws.bucket_ok && gather_choice_is_open(path) &&
                         (path == RROI_PATH_TILED_BUCKETS || (ws.bucket_pref && g_tune.bwd_buckets));
    const bool lists = buckets || path == RROI_PATH_TILED_LISTS || !inkernel_ok ||
                       (path != RROI_PATH_TILED_INKERNEL && !prefer_inkernel);
    const BucketLists BL = {ws.kshift, reinterpret_cast<int*>(ws.off), ws.bsum, ws.ov};
    {
        // affine table; the list passes' pixel counters (K3g) are cleared by the same launch
        const unsigned nzero = gather && lists ? ws.keys.keys : 0u;
        int ablocks = ceil_div(num_rois, 256);
        const int zblocks = nzero ? (int)std::min<long>(ceil_div((long)nzero, 1024), 2L * num_cus()) : 0;
        if (zblocks > ablocks) ablocks = zblocks;
        hipLaunchKernelGGL(rroi_affine_kernel, dim3(ablocks), dim3(256), 0, stream, rois, num_rois, pooled_height,
                           spatial_scale, trig, ws.aff, ws.cnt, nzero, buckets ? BL.head : (int*)nullptr,
                           buckets ? BL.ovcnt : (unsigned*)nullptr);
    }
    int st = launch_status();
    if (st != 1) return st;

    if (gather && !lists) {
        // K3t: relayout of top_diff (one launch, masked bins skipped), then the tile gather
        const KeyLayout KL = ws.keys;
        // a list entry names a bin; its chunk k is the 128-byte line at entry * line_stride + k * 32 floats:
        // in the relaid-out copy (R, NB, nchunks * 32) or in a channels-last top_diff (R, NB, C) consumed in place
        const unsigned lines_per_roi = (unsigned)NB;
        const unsigned chunk_stride = (unsigned)kChunk;
        const unsigned line_stride = td_nhwc ? (unsigned)channels : (unsigned)nchunks * (unsigned)kChunk;
        const FastDiv dpw = make_fastdiv((unsigned)pooled_width);
        const PatchMap dnb = make_patch_map(pooled_height, pooled_width);
        if (!td_nhwc) {
            const int tt = ceil_div(NB, kRelayoutPx);
            const long tiles = (long)tt * nchunks * num_rois;
            if (tiles >= (1L << 31)) return 0;
            // one block per pixel range (all its chunks), at most 8 resident blocks per CU
            long blocks = tiles / nchunks;
            const long cap = (long)num_cus() * 8;
            if (blocks > cap) blocks = cap;
#define RROI_LAUNCH_R(SAUX)                                                                              \
    hipLaunchKernelGGL((rroi_bwd_pairs_relayout_kernel<0, SAUX>), dim3((unsigned)blocks), dim3(256), 0,       \
                       stream, ws.aff, num_rois, height, width, pooled_width, NB, batch_size, lines_per_roi, \
                       dnb, dpw, KL, ws.cnt, ws.off, ws.bsum, ws.pairs, 0, top_diff, ws.tdT, channels,       \
                       nchunks, tt, (int)blocks, 0, (int)tiles, ws.scan_blocks, 0,                          \
                       BucketLists{0u, nullptr, nullptr, nullptr}, g_tune.bwd_skip_dead)
            RROI_LAUNCH_R(kBwdRelayoutAux);
#undef RROI_LAUNCH_R
            st = launch_status();
            if (st != 1) return st;
        }
        const unsigned ntiles = KL.keys / 32u;
        const unsigned per_xcd = (ntiles + 7u) / 8u;
        const FastDiv dbt = make_fastdiv(KL.Ht * KL.Wt), dwt = make_fastdiv(KL.Wt), dph = make_fastdiv((unsigned)pooled_height);
        float* dst = bd_nhwc ? bottom_diff : ws.gcm;
        const float* srcT = td_nhwc ? top_diff : ws.tdT;
#define RROI_LAUNCH_TG(NK, NHWC)                                                                          \
    hipLaunchKernelGGL((rroi_bwd_tile_gather_kernel<NK, NHWC>), dim3(per_xcd * 8u), dim3(kTgThreads), 0, stream, \
                       srcT, ws.aff, dst, num_rois, channels, height, width, pitch, pooled_height,           \
                       pooled_width, batch_size, nchunks, chunk_stride, line_stride, lines_per_roi, KL,      \
                       ntiles, per_xcd, dbt, dwt, dph)
#define RROI_LAUNCH_TG_NK(NHWC)                          \
    do {                                                 \
        if (nchunks > 4) RROI_LAUNCH_TG(8, NHWC);        \
        else if (nchunks > 2) RROI_LAUNCH_TG(4, NHWC);   \
        else if (nchunks > 1) RROI_LAUNCH_TG(2, NHWC);   \
        else RROI_LAUNCH_TG(1, NHWC);                    \
    } while (0)
        if (bd_nhwc) {
            RROI_LAUNCH_TG_NK(true);
            return launch_status();  // written in place: no relayout back
        }
        RROI_LAUNCH_TG_NK(false);
#undef RROI_LAUNCH_TG_NK
#undef RROI_LAUNCH_TG
        st = launch_status();
        if (st != 1) return st;
    } else
    if (gather) {
        // (1) pixel -> (bin, weight) lists: count, scan, fill
        const KeyLayout KL = ws.keys;
        // few scan blocks: every consumer block prefix-sums their totals itself (no second scan launch)
        const int raw_bsum = ws.scan_blocks <= kInlineScanBlocks ? 1 : 0;
        // a list entry names a bin; its chunk k is the 128-byte line at entry * line_stride + k * 32 floats:
        // in the relaid-out copy (R, NB, nchunks * 32) or in a channels-last top_diff (R, NB, C) consumed in place
        const unsigned lines_per_roi = (unsigned)NB;
        const unsigned chunk_stride = (unsigned)kChunk;
        const unsigned line_stride = td_nhwc ? (unsigned)channels : (unsigned)nchunks * (unsigned)kChunk;
        // one pair block per CU, looping over the bins: the pair passes need outstanding atomics,
        // not CU slots -- more blocks only take residency from the relayout (207 -> 197 us per call)
        const FastDiv dpw = make_fastdiv((unsigned)pooled_width);
        const PatchMap dnb = make_patch_map(pooled_height, pooled_width);
        if ((long)num_rois * dnb.lanes_per_roi >= (1L << 32)) return 0;
        int pblocks = ceil_div((long)num_rois * dnb.lanes_per_roi, 256);
        {
            // pair blocks per CU.  A pair wave walks a chain of returning atomics (~2.5 us per patch of 64 bins under
            // load), so its launch time is patches per wave x that; the relayout it shares the launch with takes
            // bytes / bandwidth.  One block per CU hides the pairs behind the relayout of 256 channels (round 2); with
            // FEWER channels the same bins bring a quarter of the bytes and the pair pass set the launch (R = 512, C = 64,
            // 11 x 96: 58 us where the relayout alone takes 30): blocks per CU ~ 256 / C.
            // (tools/pair_blocks_ab.py, profiles/r05_pair_blocks_ab2.txt: with the wave-aggregated reservations, us per call at 1 / 2
            // / 4 / 8 blocks per CU -- C = 64, R = 512, 11 x 96: 90.6 / 73.0 / 67.0 / 68.8 (round 4: 84.3); C = 128: 66.5 / 58.9 /
            // 59.7 / 59.7; configs[2]: 104.4 / 103.3 / 103.6 / 103.5)
            int per_cu = g_tune.bwd_pair_blocks_per_cu;
            if (per_cu <= 0) per_cu = std::min(4, std::max(2, 256 / std::max(channels, 1)));
            // channels-last gradients need no relayout: the pair blocks have the launch to themselves (CL=1 tools/pair_blocks_ab.py,
            // profiles/r05_pair_blocks_ab_cl.txt: C = 64, R = 512 45.3 / 41.4 us at 4 / 8 per CU, round 4: 64.0; configs[2] 52.8 / 49.0, 51.3)
            if (g_tune.bwd_pair_blocks_per_cu <= 0 && td_nhwc) per_cu = 8;
            const long cap = (long)num_cus() * per_cu;
            if (pblocks > cap) pblocks = (int)cap;
        }
        // the bucket slots reserved per WAVE through a table in LDS (pairs_reserve_wave) -- where a wave has several
        // patches to walk; with one patch per wave the table's set-up is pure latency (R = 32, C = 64: +0.8 us)
        const bool aggregate = g_tune.bwd_pair_aggregate &&
                               (long)num_rois * (dnb.lanes_per_roi / 64) > 8L * num_cus();
        // count || first half of the relayout;  scan;  fill || second half.  The relayout is the
        // forward's, with R "images" of PH x PW "pixels" and the masked bins skipped:
        // top_diff (R, C, NB) -> (R, NB, nchunks * 32)
        const int tt = ceil_div(NB, kRelayoutPx);
        const long tiles = td_nhwc ? 0 : (long)tt * nchunks * num_rois;  // nothing to relay out
        if (tiles >= (1L << 31)) return 0;
        const long unit = nchunks;  // a block takes all chunks of a pixel range: whole ranges per launch
        const long half = (tiles / 2 + unit - 1) / unit * unit < tiles ? (tiles / 2 + unit - 1) / unit * unit : tiles;
        auto relayout_grid = [&](long n) {   // n tiles -> blocks: one per pixel range, at most 8 resident per CU
            const long cap = (long)num_cus() * 8;
            return n / unit <= cap ? n / unit : cap;
        };
#define RROI_LAUNCH_PR(FILL, SAUX, BLOCKS, T0, T1)                                                   \
    hipLaunchKernelGGL((rroi_bwd_pairs_relayout_kernel<FILL, SAUX>), dim3((unsigned)(pblocks + (BLOCKS))), \
                       dim3(256), 0, stream, ws.aff, num_rois, height, width, pooled_width, NB,          \
                       batch_size, lines_per_roi, dnb, dpw, KL, ws.cnt, ws.off, ws.bsum, ws.pairs,       \
                       pblocks, top_diff, ws.tdT, channels, nchunks, tt, (int)(BLOCKS), (int)(T0), (int)(T1),            \
                       ws.scan_blocks, raw_bsum, BL, (g_tune.bwd_skip_dead ? 1 : 0) | (aggregate ? 2 : 0))
        if (buckets) {
            // ONE launch: every pair into its pixel's bucket (or overflow chain) || the whole relayout
            const long blocks = relayout_grid(tiles);
            RROI_LAUNCH_PR(2, kBwdRelayoutAux, blocks, 0, tiles);
        } else {
        {
            const long blocks = relayout_grid(half);
            RROI_LAUNCH_PR(0, kBwdRelayoutAux, blocks, 0, half);
        }
        hipLaunchKernelGGL(rroi_scan1_kernel, dim3(ws.scan_blocks), dim3(1024), 0, stream, ws.cnt, ws.off,
                           ws.bsum, KL.keys);
        if (!raw_bsum) hipLaunchKernelGGL(rroi_scan2_kernel, dim3(1), dim3(1024), 0, stream, ws.bsum, ws.scan_blocks);
        {
            const long blocks = relayout_grid(tiles - half);
            RROI_LAUNCH_PR(1, kBwdRelayoutAux, blocks, half, tiles);
        }
        }
#undef RROI_LAUNCH_PR
        st = launch_status();
        if (st != 1) return st;
        // (3) gather: one thread group per key, no grid-stride
        unsigned sub_shift = 3;  // 8 lanes = one chunk
        while ((1u << sub_shift) < 8u * (unsigned)nchunks && sub_shift < 6) ++sub_shift;
        // NCHW bottom_diff written in place (round 4; tools/bwd_nchw_ab.py, profiles/r04_bwd_nchw_ab.txt): a workgroup
        // needs whole rows (8 pixels) of a key tile, i.e. at most 32 lanes per pixel -- wider pixels (C > 128) deal
        // their passes of four chunks to blockIdx.y, which walks every list once per pass: that pays while the lists
        // are short (cfg3, 10 bins per map pixel: 125.6 -> 117.5 us; 20 per pixel: 274 -> 283), so C > 128 keeps the
        // scratch form beyond 16 bins per pixel.  C <= 128 gains at every density measured: R = 512, C = 64 / 128
        // 76.0 -> 71.9 / 76.5 -> 71.2, R = 16...32 36.8 -> 29.0 / 24.3 -> 17.9 (the relayout launch was a fifth of
        // those calls), 84 bins per pixel 159 -> 153.
        const bool nchw_direct = !bd_nhwc && g_tune.bwd_nchw_direct != 0 &&
                                 (nchunks <= 4 || (double)num_rois * NB <= (double)g_tune.bwd_nchw_direct * (double)batch_size * HW);
        unsigned gy = 1;
        if (nchw_direct && sub_shift == 6) {
            sub_shift = 5;
            gy = (unsigned)ceil_div(nchunks, 4);
        }
        const unsigned groups_per_block = 256u >> sub_shift;
        // whole groups of 8 key tiles (the kernel deals the tiles of a group to the 8 XCDs)
        const long wg_per_tile = 32 / groups_per_block;  // 1, 2, 4 or 8
        const unsigned tile_run = nchw_direct ? (unsigned)g_tune.bwd_tile_run : 0u;
        const long gblocks = ceil_div(ceil_div((long)KL.keys, 32L), 8L << tile_run) * (8L << tile_run) * wg_per_tile;
        // the lists: count / scan / fill segments (`off` = scanned offsets) or buckets (`off` = the counters)
        const unsigned* loff = buckets ? reinterpret_cast<const unsigned*>(ws.cnt) : ws.off;
#define RROI_LAUNCH_G(DSTK, BUCK, DST)                                                                        \
    hipLaunchKernelGGL((rroi_bwd_gather_kernel<DSTK, BUCK>), dim3((unsigned)gblocks, gy), dim3(256), 0, stream, \
                       td_nhwc ? top_diff : ws.tdT, loff, ws.bsum, ws.pairs, DST, channels, height, width,    \
                       pitch, nchunks, chunk_stride, line_stride, sub_shift, KL, make_fastdiv(KL.Ht * KL.Wt), \
                       make_fastdiv(KL.Wt), ws.scan_blocks, raw_bsum, BL, tile_run)
        if (bd_nhwc) {
            if (buckets) RROI_LAUNCH_G(kDstNhwc, true, bottom_diff);
            else RROI_LAUNCH_G(kDstNhwc, false, bottom_diff);
            return launch_status();  // written in place: no relayout back
        }
        if (nchw_direct) {
            if (accumulate) {
                if (buckets) RROI_LAUNCH_G(kDstNchwAdd, true, bottom_diff);
                else RROI_LAUNCH_G(kDstNchwAdd, false, bottom_diff);
            } else {
                if (buckets) RROI_LAUNCH_G(kDstNchw, true, bottom_diff);
                else RROI_LAUNCH_G(kDstNchw, false, bottom_diff);
            }
            return launch_status();  // written in place
        }
        if (buckets) RROI_LAUNCH_G(kDstChunkMajor, true, ws.gcm);
        else RROI_LAUNCH_G(kDstChunkMajor, false, ws.gcm);
#undef RROI_LAUNCH_G
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
    if (accumulate)
        hipLaunchKernelGGL(rroi_cm_to_nchw_kernel<true>, dim3(ptiles * nchunks * batch_size), dim3(256), 0,
                           stream, ws.gcm, bottom_diff, channels, (int)HW, width, pitch,
                           make_fastdiv((unsigned)width), nchunks, ptiles);
    else
        hipLaunchKernelGGL(rroi_cm_to_nchw_kernel<false>, dim3(ptiles * nchunks * batch_size), dim3(256), 0,
                           stream, ws.gcm, bottom_diff, channels, (int)HW, width, pitch,
                           make_fastdiv((unsigned)width), nchunks, ptiles);
    return launch_status();
}

int rroi_align_bin_centres_hip(float spatial_scale, int num_rois, int height, int width,
                               int pooled_height, int pooled_width, const float* rois,
                               float* geom, void* stream_)
{
    return rroi_align_bin_centres_trig_hip(spatial_scale, num_rois, height, width, pooled_height, pooled_width, rois,
                                           geom, RROI_TRIG_DOUBLE, stream_);
}

int rroi_align_bin_centres_trig_hip(float spatial_scale, int num_rois, int height, int width,
                                    int pooled_height, int pooled_width, const float* rois,
                                    float* geom, int trig_recipe, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (trig_recipe != RROI_TRIG_DOUBLE && trig_recipe != RROI_TRIG_FP32) return 0;
    if (num_rois < 0 || height <= 0 || width <= 0 || pooled_height <= 0 || pooled_width <= 0)
        return 0;
    if (num_rois == 0) return 1;
    if (!rois || !geom) return 0;
    const long threads = (long)num_rois * pooled_height * pooled_width;
    hipLaunchKernelGGL(rroi_bin_centres_kernel, dim3(ceil_div(threads, 256)), dim3(256), 0, stream,
                       rois, geom, num_rois, height, width, pooled_height, pooled_width,
                       spatial_scale, trig_recipe);
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

int rroi_align_gt_quads_to_rois_hip(const float* quads, const float* batch_index, const float* height_jitter,
                                    int n, float* rois, float* max_ratio, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (n < 0) return 0;
    if (!max_ratio || (n > 0 && (!quads || !rois))) return 0;
    hipLaunchKernelGGL(rroi_gt_quads_to_rois_kernel, dim3(1), dim3(256), 0, stream, quads, batch_index,
                       height_jitter, n, rois, max_ratio);
    return launch_status();
}

int rroi_rbox_decode_hip(const float* segm, const float* rbox, const float* angle, int height, int width,
                         float segm_thresh, void* candidates, int capacity, int* count, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (height <= 0 || width <= 0 || capacity < 0 || (long)height * width >= (1L << 30)) return 0;
    if (!segm || !rbox || !angle || !count || (capacity > 0 && !candidates)) return 0;
    const int hw = height * width;
    const int slabs = ceil_div(hw, 1024);
    unsigned* slab_counts = nullptr;
    // large map: per-slab counts from a first launch, kept behind the records the caller's buffer can ever
    // need (a map of hw pixels yields at most hw records) -- when the buffer has that room.  A buffer sized
    // for fewer records (a caller relying on *count to report the overflow, or one sized to exactly h * w)
    // keeps the one-launch form, in which every workgroup counts the pixels before its slab itself.
    if (slabs > 256 && (long)capacity >= (long)hw + ceil_div((long)slabs * 4, 64)) {
        slab_counts = reinterpret_cast<unsigned*>(static_cast<NmsCandidate*>(candidates) + hw);
        hipLaunchKernelGGL(rroi_rbox_count_kernel, dim3(slabs), dim3(1024), 0, stream, segm, hw, segm_thresh, slab_counts);
        const int st = launch_status();
        if (st != 1) return st;
    }
    hipLaunchKernelGGL(rroi_rbox_decode_kernel, dim3(slabs), dim3(1024), 0, stream, segm, rbox, angle, height, width,
                       segm_thresh, static_cast<NmsCandidate*>(candidates), slab_counts ? hw : capacity, count, slab_counts);
    return launch_status();
}

int rroi_nms_record_format(void) { return RROI_NMS_RECORD_FORMAT; }

int rroi_nms_merge_host(const void* candidates, int num_candidates, int width, int height, float iou_threshold,
                        float iou_threshold2, float* boxes, int max_boxes)
{
    if (num_candidates < 0 || width <= 0 || height <= 0 || max_boxes < 0) return -1;
    if (num_candidates > 0 && !candidates) return -1;
    const NmsCandidate* cand = static_cast<const NmsCandidate*>(candidates);
    for (int i = 0; i < num_candidates; ++i)
        if (cand[i].x < 0 || cand[i].x >= width || cand[i].y < 0 || cand[i].y >= height) return -1;
    const std::vector<NmsPoly> out = nms_merge(cand, num_candidates, width, height, iou_threshold, iou_threshold2);
    const int n = (int)out.size();
    for (int i = 0; i < n && i < max_boxes; ++i) {
        float* b = boxes + (size_t)i * 9;
        for (int v = 0; v < 4; ++v) {  // adaptor.cpp:14-31 (float of the integers), nms/__init__.py:15-16 (/ 10000)
            b[2 * v] = (float)out[(size_t)i].X[v] / 10000.0f;
            b[2 * v + 1] = (float)out[(size_t)i].Y[v] / 10000.0f;
        }
        b[8] = out[(size_t)i].score;
    }
    return n;
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

int rroi_align_write_probe_hip(float* out, size_t num_floats, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!out || num_floats % 4 != 0 || reinterpret_cast<size_t>(out) % 16 != 0) return 0;
    const size_t n4 = num_floats / 4;
    if (n4 == 0) return 1;
    if ((n4 + 255) / 256 >= (1ull << 31)) return 0;
    hipLaunchKernelGGL(rroi_write_probe_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, out, n4);
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

// The reference-ABI launchers have no `path` to carry the trig recipe: they use RROI_TRIG_DOUBLE, unless the process
// was started with RROI_ALIGN_LAUNCHER_TRIG=fp32 in its environment (read ONCE, at the first launcher call: a constant
// of the process, not a switch) -- for a drop-in user who wants every bin equal to the reference's own build for
// this GPU rather than to the correctly rounded recipe.
static int launcher_trig()
{
    static const int t = [] {
        const char* e = getenv("RROI_ALIGN_LAUNCHER_TRIG");
        return (e && (!strcmp(e, "fp32") || !strcmp(e, "1"))) ? RROI_TRIG_FP32 : RROI_TRIG_DOUBLE;
    }();
    return t;
}

// ---- the reference's launcher ABI (rroi_align_kernel.h:8-18) ------------------------
// The signatures carry no workspace (and the forward's no batch count), so the fast paths take
// their scratch from the library's per-(device, stream) buffers (launcher_scratch above: grown on
// demand, reused by later calls, pinned once a graph has captured them; no synchronisation).
// Small problems keep the one-kernel direct paths (no scratch).
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
    if ((con_idx_x == nullptr) != (con_idx_y == nullptr)) return 0;
    const int NB = pooled_height * pooled_width;
    dim3 grid;
    int cslab;
    direct_grid(num_rois, NB, channels, grid, cslab);
    if (!pick_tiled_fwd(1, channels, height, width, num_rois, NB)) {
        if (launch_patch_forward(bottom_data, bottom_rois, top_data, con_idx_x, con_idx_y, num_rois, channels, height, width,
                                 pooled_height, pooled_width, spatial_scale, launcher_trig(), /*batch_size unknown*/ -1, stream))
            return launch_status();
        hipLaunchKernelGGL(rroi_fwd_direct_kernel, grid, dim3(256), 0, stream, bottom_data, bottom_rois,
                           top_data, con_idx_x, con_idx_y, num_rois, channels, height, width,
                           pooled_height, pooled_width, spatial_scale, launcher_trig(), /*batch_size unknown*/ -1,
                           cslab);
        return launch_status();
    }
    // Tiled path for the ROIs of image 0 (every ROI, in inference and in the benchmark); the
    // signature does not say how many images `bottom_data` holds, so the ROIs of images >= 1 are
    // sampled from the NCHW tensor by one more block per ROI of the prologue launch (trusting the
    // index as the reference does; a block whose ROI is of image 0 reads the index and leaves) and
    // skipped by the gather: the same two launches as the native call.
    const size_t bytes = carve(nullptr, 1, channels, height, width, num_rois, RROI_LAYOUT_NCHW).bytes;
    ScratchLease lease = launcher_scratch(stream, bytes);
    void* const ws = lease.ptr;
    if (!ws) return status_of(lease.err);
    int st = forward_impl(bottom_data, RROI_LAYOUT_NCHW, RROI_LAYOUT_NCHW, spatial_scale, 1, num_rois, height,
                          width, channels, pooled_height, pooled_width, bottom_rois, top_data, ws, bytes,
                          RROI_PATH_TILED | (launcher_trig() ? RROI_PATH_TRIG_FP32 : 0), RROI_STAGE_ALL, stream_,
                          /*launcher_rest*/ true);
    if (st == 1 && con_idx_x) {
        hipLaunchKernelGGL(rroi_con_idx_kernel, grid, dim3(256), 0, stream, bottom_rois, con_idx_x, con_idx_y,
                           num_rois, channels, height, width, pooled_height, pooled_width, spatial_scale, launcher_trig(),
                           cslab);
        st = launch_status();
    }
    const hipError_t e = lease.give_back(stream);   // (everything that uses the buffer is enqueued)
    return st != 1 ? st : status_of(e);
}

// con_idx_x / con_idx_y must be the tensors the forward wrote for the same rois (the reference
// re-reads the bin centres from them, kernel.cu:232-233): the fast path recomputes the centres from
// the rois instead of reading 2 x (R, C, PH, PW) floats back.  bottom_diff: the gradient is ADDED to it
// on every path, as the reference's atomicAdds do (kernel.cu:260-274; functions/rroi_align.py:35 hands
// over zeros) -- the result does not depend on which path the problem size selects.
int RROIAlignBackwardLaucher(const float* top_diff, const float spatial_scale,
                             const int batch_size, const int num_rois, const int height,
                             const int width, const int channels, const int pooled_height,
                             const int pooled_width, const float* bottom_rois,
                             float* bottom_diff, const float* con_idx_x, const float* con_idx_y,
                             void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!shape_ok(batch_size, num_rois, height, width, channels, pooled_height, pooled_width))
        return 0;
    if (num_rois == 0) return 1;
    if (!top_diff || !bottom_rois || !bottom_diff || !con_idx_x || !con_idx_y) return 0;
    const int NB = pooled_height * pooled_width;
    if (pick_tiled_bwd(batch_size, channels, height, width, num_rois, NB)) {
        const size_t bytes = carve_bwd(nullptr, batch_size, channels, height, width, num_rois, NB).bytes;
        ScratchLease lease = launcher_scratch(stream, bytes);
        void* const ws = lease.ptr;
        if (!ws) return status_of(lease.err);
        const int st = backward_impl(top_diff, RROI_LAYOUT_NCHW, RROI_LAYOUT_NCHW, spatial_scale, batch_size, num_rois,
                                     height, width, channels, pooled_height, pooled_width, bottom_rois, bottom_diff,
                                     ws, bytes, RROI_PATH_TILED | (launcher_trig() ? RROI_PATH_TRIG_FP32 : 0), stream_,
                                     /*accumulate*/ true);
        const hipError_t e = lease.give_back(stream);
        return st != 1 ? st : status_of(e);
    }
    const long nthreads = (long)num_rois * pooled_height * pooled_width * channels;
    long blocks = (nthreads + 255) / 256;
    const long cap = (long)num_cus() * 32;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(rroi_bwd_literal_kernel, dim3((unsigned)blocks), dim3(256), 0, stream,
                       top_diff, con_idx_x, con_idx_y, bottom_rois, bottom_diff, nthreads, channels,
                       height, width, pooled_height, pooled_width);
    return launch_status();
}

int rroi_align_set_trig_recipe_hip(int recipe) { return recipe == RROI_TRIG_DOUBLE ? 1 : 0; }   // deprecated shim, see the header
int rroi_align_get_trig_recipe_hip(void) { return RROI_TRIG_DOUBLE; }

int rroi_align_launcher_scratch_stats(int* in_use, int* pinned, int* capacity, unsigned long long* transient_calls)
{
    std::lock_guard<std::mutex> table(g_table_mutex);
    int u = 0, p = 0;
    for (const LauncherArena& a : g_arenas) {
        u += a.used ? 1 : 0;
        p += a.used && a.pinned ? 1 : 0;
    }
    if (in_use) *in_use = u;
    if (pinned) *pinned = p;
    if (capacity) *capacity = kMaxArenas;
    if (transient_calls) *transient_calls = g_transient_calls.load(std::memory_order_relaxed);
    return 1;
}

int rroi_align_release_launcher_scratch(void)
{
    std::unique_lock<std::mutex> table(g_table_mutex);
    int cur = 0;
    (void)hipGetDevice(&cur);
    hipError_t e = hipSuccess;
    for (int i = 0; i < kMaxArenas; ++i) {
        LauncherArena& a = g_arenas[i];
        if (!a.used || a.pinned) continue;   // pinned: a graph replays with this address, it lives as long as the process
        std::unique_lock<std::mutex> mine(a.in_use, std::try_to_lock);
        if (!mine.owns_lock()) {   // a call is enqueuing on it: wait without the table lock, then look at the entry again
            table.unlock();
            { std::lock_guard<std::mutex> wait(a.in_use); }
            table.lock();
            --i;
            continue;
        }
        (void)hipSetDevice(a.device);
        const hipError_t ei = hipFree(a.ptr);  // synchronous: the buffers' streams may be gone
        if (ei != hipSuccess) e = ei;
        a.used = false;
        a.ptr = nullptr;
        a.bytes = 0;
    }
    (void)hipSetDevice(cur);
    return status_of(e);
}

}  // extern "C"

