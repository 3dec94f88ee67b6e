// tools/kbench.hip -- standalone micro-benchmarks for the RoIRotate forward on MI355X.
// Measurement tooling (not product, not shipped): includes the product translation unit to
// reuse its device code, adds ablation kernels, times everything with hipEvents.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -o tools/kbench tools/kbench.hip
#include "rroi_align_hip_explore.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <time.h>
#include <unistd.h>
#include <random>
#include <string>
#include <vector>

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

namespace {

constexpr int R = 512, C = 256, H = 160, W = 160, PH = 8, PW = 64, NB = PH * PW;

// ---- store-only: the product's tile pattern (256 B row segments, 2 KiB apart) -------------
template <int MODE>  // 0 = nontemporal, 1 = plain, 2 = sc1-ish (agent-scope relaxed atomic store path n/a)
__global__ __launch_bounds__(64) void k_store_tile(float* __restrict__ out, int nchunks, int ntiles)
{
    const unsigned lane = threadIdx.x;
    const unsigned k = blockIdx.x % nchunks, slot = blockIdx.x / nchunks, nslots = gridDim.x / nchunks;
    const unsigned items = R * ntiles;
    const unsigned col = (lane & 15) * 4, row0 = lane >> 4;
    const v4f v = {1.f, 2.f, 3.f, (float)lane};
    for (unsigned item = slot; item < items; item += nslots) {
        const unsigned n = item / ntiles, t = item % ntiles;
        float* obase = out + ((size_t)n * C + k * 32) * NB + (size_t)t * 64;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const unsigned r = s * 4 + row0;
            float* op = obase + (size_t)(r * NB + col);
            if (MODE == 0) __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(op));
            else *reinterpret_cast<v4f*>(op) = v;
        }
    }
}

// ---- store-only: fully linear, every wave writes contiguous 8 KiB pieces -------------------
template <int MODE>
__global__ __launch_bounds__(64) void k_store_linear(float* __restrict__ out, unsigned pieces)
{
    const unsigned lane = threadIdx.x;
    const v4f v = {1.f, 2.f, 3.f, (float)lane};
    for (unsigned p = blockIdx.x; p < pieces; p += gridDim.x) {
        float* base = out + (size_t)p * 2048 + lane * 4;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (MODE == 0) __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(base + s * 256));
            else *reinterpret_cast<v4f*>(base + s * 256) = v;
        }
    }
}

// ---- store-only with 256-thread blocks, linear (what a tuned fill looks like) --------------
template <int MODE>
__global__ __launch_bounds__(256) void k_store_linear256(float* __restrict__ out, size_t n4)
{
    const v4f v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        if (MODE == 0) __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(out) + i);
        else reinterpret_cast<v4f*>(out)[i] = v;
    }
}

// ---- write ceiling (round 4): ONE float4 per thread, torch's elementwise launch shape; DATA 0 zeros, 1 a non-zero
// constant, 2 values that depend on the index; AUX cache policy of the store (0 plain, 2 nt, 16 sc1, 17 sc0 sc1)
template <int DATA, int AUX>
__global__ __launch_bounds__(256) void k_store_once(float* __restrict__ out, unsigned n4)
{
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n4) return;
    const float f = DATA == 2 ? (float)(i * 2654435761u >> 8) * 1.1920929e-7f : DATA == 1 ? 1.0f : 0.0f;
    const v4f v = {f, DATA == 2 ? f + 1.0f : f, DATA == 2 ? f * 3.0f : f, DATA == 2 ? -f : f};
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(out, 0xffffffffu);
    if (AUX == 0) reinterpret_cast<v4f*>(out)[i] = v;
    else if ((size_t)i * 16 < 0xffffffffull) buf_store<AUX>(rs, i * 16u, v);
}

// ---- raw L2->CU gather bandwidth: every lane group of LPG lanes reads one contiguous run of
// LPG*16 bytes at a pseudo-random offset inside `region_bytes`; DEPTH independent loads in
// flight per wave.  LPG = 64: 1 KiB contiguous per instruction; 8: eight 128 B lines; 1: 64 x 16 B.
template <int LPG, int DEPTH>
__global__ __launch_bounds__(64) void k_l2_gather(const char* __restrict__ src0, unsigned region_bytes,
                                                 int iters, float* __restrict__ sink, int slice_mode = 0)
{
    const unsigned lane = threadIdx.x;
    const unsigned slice = slice_mode == 1 ? blockIdx.x % 8 : slice_mode == 2 ? (blockIdx.x / 8) % 8 : 0;
    const char* src = src0 + (size_t)slice * region_bytes;
    const unsigned grp = lane / LPG, sub = lane % LPG;
    const unsigned run = LPG * 16u;
    const unsigned nruns = region_bytes / run;
    unsigned state = (blockIdx.x * 64u + grp) * 2654435761u + 12345u;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < iters; ++i) {
        v4f v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            state = state * 1664525u + 1013904223u;
            const unsigned r = (unsigned)(((unsigned long long)(state >> 8) * nruns) >> 24);
            v[d] = *reinterpret_cast<const v4f*>(src + (size_t)r * run + sub * 16u);
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc += v[d];
    }
    if (acc.x == 12345.678f) sink[blockIdx.x * 64 + lane] = acc.x + acc.y + acc.z + acc.w;
}

// ---- store-only, product tile pattern, buffer_store with cache-policy bits AUX (gfx940 family:
// 1 = sc0, 2 = nt, 16 = sc1) --------------------------------------------------------------
template <int AUX>
__global__ __launch_bounds__(64) void k_store_aux(float* __restrict__ out, int nchunks, int ntiles)
{
    const unsigned lane = threadIdx.x;
    const unsigned k = blockIdx.x % nchunks, slot = blockIdx.x / nchunks, nslots = gridDim.x / nchunks;
    const unsigned items = R * ntiles;
    const unsigned col = (lane & 15) * 4, row0 = lane >> 4;
    const v4u v = {1u, 2u, 3u, lane};
    for (unsigned item = slot; item < items; item += nslots) {
        const unsigned n = item / ntiles, t = item % ntiles;
        float* obase = out + ((size_t)n * C + k * 32) * NB;
        const __amdgpu_buffer_rsrc_t ws = make_rsrc(obase, 32u * NB * 4u);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const unsigned r = s * 4 + row0;
            __builtin_amdgcn_raw_buffer_store_b128(v, ws, (r * NB + t * 64 + col) * 4u, 0, AUX);
        }
    }
}

// ---- store-only, the shipped policy mix (one store of eight sc0 sc1, seven nt), two address patterns:
// PATTERN 0 = the product's tile (a wave writes 32 row segments of 256 B, 2 KiB apart);
// PATTERN 1 = what 8 waves that share a (roi, chunk) block in LDS could write: a wave writes 4 whole
//             rows = 8 KiB contiguous, every instruction 1 KiB contiguous.
template <int PATTERN, int MINOR>
__global__ __launch_bounds__(64) void k_store_mix(float* __restrict__ out, int nchunks, int ntiles)
{
    const unsigned lane = threadIdx.x;
    const unsigned k = blockIdx.x % nchunks, slot = blockIdx.x / nchunks, nslots = gridDim.x / nchunks;
    const unsigned items = R * ntiles;
    const unsigned col = (lane & 15) * 4, row0 = lane >> 4;
    const v4u v = {1u, 2u, 3u, lane};
    for (unsigned item = slot; item < items; item += nslots) {
        const unsigned n = item / ntiles, t = item % ntiles;
        float* obase = out + ((size_t)n * C + k * 32) * NB;
        const __amdgpu_buffer_rsrc_t ws = make_rsrc(obase, 32u * NB * 4u);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            unsigned off;
            if (PATTERN == 0) off = ((s * 4 + row0) * NB + t * 64 + col) * 4u;
            else off = t * 8192u + s * 1024u + lane * 16u;
            if (s < MINOR) __builtin_amdgcn_raw_buffer_store_b128(v, ws, off, 0, 17);
            else __builtin_amdgcn_raw_buffer_store_b128(v, ws, off, 0, 2);
        }
    }
}

// ---- store shapes (round 4): the same bytes -- every (roi, chunk) block of 32 rows x 512 floats -- written in tiles of
// 64 floats per row with different PER-INSTRUCTION shapes and alignments (policy: 1 of 8 write-through, the rest nt):
//   SHAPE 0  4 rows x 256 B per instruction (the strided kernel), SHAPE 1  16 rows x 64 B (round 4's first SHIFT
//   mapping), SHAPE 2  8 rows x 128 B;   `skew` floats are added to every row's start: 0 = rows on 128-byte lines,
//   16 = on 64-byte sectors only, 4 = on 16-byte pieces only.  The buffer is allocated with room for the skew.
// the SHIFT form's own pattern: tiles of 48 floats per row (three sectors), 16 rows x 64 B per instruction, six
// instructions per tile; rows of NBX floats (NBX % 16 != 0 gives every row its own phase h, windows start at -h)
// LANES 0: lane = (row of 16, 16-byte piece of 4) with the ROW fastest (what the SHIFT storer did until round 4: an LDS
// read without bank conflicts); 1: the PIECE fastest -- four neighbouring lanes store one 64-byte sector
template <int NBX, int WIN = 48, int ORDER = 0, int AUXA = 17, int AUXB = 2, int JIT = 0, int LANES = 0>
__global__ __launch_bounds__(64) void k_store_win48(float* __restrict__ out, int nchunks, unsigned rois)
{
    const unsigned lane = threadIdx.x, ch16 = LANES ? lane >> 2 : lane & 15u, pcl = LANES ? lane & 3u : lane >> 4;
    const unsigned k = blockIdx.x % nchunks, slot = blockIdx.x / nchunks, nslots = gridDim.x / nchunks;
    constexpr unsigned ntiles = (NBX + WIN - 1) / WIN;
    const unsigned items = rois * ntiles;
    const v4u v = {1u, 2u, 3u, lane};
    for (unsigned item = slot; item < items; item += nslots) {
        const unsigned n = item / ntiles, t = item % ntiles;
        if (JIT) {   // desynchronise the waves: 0 ... ~0.8 us of idling, different for every (wave, item)
            const unsigned hsh = (item * 2654435761u + blockIdx.x * 40503u) >> 28;   // 0..15
            for (unsigned w = 0; w < hsh; ++w) __builtin_amdgcn_s_sleep(2);          // 64 x 2 clocks at a time
        }
        const size_t blk = ((size_t)n * 256 + k * 32) * NBX;
        float* obase = out + blk;
        const __amdgpu_buffer_rsrc_t ws = make_rsrc(obase, 32u * NBX * 4u);
        const unsigned h = (unsigned)((blk + (size_t)ch16 * NBX) & 15u);
#pragma unroll
        for (int i = 0; i < WIN / 8; ++i) {
            // ORDER 0: the two channel halves alternate (a row's next sector two instructions later); 1: a row's sectors
            // in consecutive instructions
            const unsigned uu = ORDER ? i / (WIN / 16) : (i & 1), ss = ORDER ? i % (WIN / 16) : (i >> 1);
            const unsigned r = ch16 + 16u * uu, p0 = 16u * ss + 4u * pcl;
            const int j = (int)(t * WIN + p0) - (int)h;
            const bool ok = j >= 0 && j + 4 <= NBX;
            const unsigned off = ok ? (r * NBX + (unsigned)j) * 4u : 0x80000000u;
            if (i < 1) __builtin_amdgcn_raw_buffer_store_b128(v, ws, off, 0, AUXA);
            else __builtin_amdgcn_raw_buffer_store_b128(v, ws, off, 0, AUXB);
        }
    }
}

// what a line-aligned SHIFT form would store: items of 96 floats per row (three 128-byte lines), 8 rows x 128 B per
// instruction, twelve instructions per item; rows of NBX floats, windows start at -(row offset mod 32)
template <int NBX, int AUX, int JIT>
__global__ __launch_bounds__(64) void k_store_win96(float* __restrict__ out, int nchunks, unsigned rois)
{
    const unsigned lane = threadIdx.x, ch8 = lane & 7u, pcl = lane >> 3;
    const unsigned k = blockIdx.x % nchunks, slot = blockIdx.x / nchunks, nslots = gridDim.x / nchunks;
    constexpr unsigned ntiles = (NBX + 31 + 95) / 96;
    const unsigned items = rois * ntiles;
    const v4u v = {1u, 2u, 3u, lane};
    for (unsigned item = slot; item < items; item += nslots) {
        const unsigned n = item / ntiles, t = item % ntiles;
        if (JIT) {
            const unsigned hsh = (item * 2654435761u + blockIdx.x * 40503u) >> 28;
            for (unsigned w = 0; w < hsh; ++w) __builtin_amdgcn_s_sleep(2);
        }
        const size_t blk = ((size_t)n * 256 + k * 32) * NBX;
        float* obase = out + blk;
        const __amdgpu_buffer_rsrc_t ws = make_rsrc(obase, 32u * NBX * 4u);
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const unsigned r = ch8 + 8u * (i & 3), p0 = 32u * (i >> 2) + 4u * pcl;
            const unsigned h = (unsigned)((blk + (size_t)r * NBX) & 31u);
            const int j = (int)(t * 96 + p0) - (int)h;
            const bool ok = j >= 0 && j + 4 <= NBX;
            const unsigned off = ok ? (r * NBX + (unsigned)j) * 4u : 0x80000000u;
            __builtin_amdgcn_raw_buffer_store_b128(v, ws, off, 0, AUX);
        }
    }
}

template <int SHAPE>
__global__ __launch_bounds__(64) void k_store_shape(float* __restrict__ out, int nchunks, int ntiles, unsigned skew)
{
    const unsigned lane = threadIdx.x;
    const unsigned k = blockIdx.x % nchunks, slot = blockIdx.x / nchunks, nslots = gridDim.x / nchunks;
    const unsigned items = R * ntiles;
    const v4u v = {1u, 2u, 3u, lane};
    for (unsigned item = slot; item < items; item += nslots) {
        const unsigned n = item / ntiles, t = item % ntiles;
        float* obase = out + ((size_t)n * C + k * 32) * NB;
        const __amdgpu_buffer_rsrc_t ws = make_rsrc(obase, 32u * NB * 4u + 256u);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            unsigned row, col;
            if (SHAPE == 0) { row = s * 4 + (lane >> 4); col = (lane & 15) * 4; }
            else if (SHAPE == 1) { row = (lane & 15) + 16 * (s & 1); col = (s >> 1) * 16 + (lane >> 4) * 4; }
            else { row = (lane & 7) + 8 * (s & 3); col = (s >> 2) * 32 + (lane >> 3) * 4; }
            const unsigned off = (row * NB + t * 64 + col + skew) * 4u;
            if (s < 1) __builtin_amdgcn_raw_buffer_store_b128(v, ws, off, 0, 17);
            else __builtin_amdgcn_raw_buffer_store_b128(v, ws, off, 0, 2);
        }
    }
}

// ---- TA instruction rate: buffer_load_dwordx4 in a 16 KiB (L1-resident) window; a fraction of
// the 8-lane groups is out of range (MODE 0: none, 1: half, 2: 7/8, 3: all) or exec-masked
// (MODE 4: half masked by a branch). 8 independent loads in flight.
template <int MODE>
__global__ __launch_bounds__(64) void k_ta_rate(const float* __restrict__ src, int iters, float* __restrict__ sink)
{
    const unsigned lane = threadIdx.x, grp = lane >> 3;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(src, 16384u);
    bool oob = false;
    if (MODE == 1) oob = grp & 1;
    if (MODE == 2) oob = grp != 0;
    if (MODE == 3) oob = true;
    const unsigned off0 = oob ? kOOB : lane * 16u;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < iters; ++i) {
        v4f v[8];
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            const unsigned off = off0 + (unsigned)(((i * 8 + d) * 1024) & 15360);
            if (MODE == 4) {
                v[d] = acc;
                if (grp & 1) v[d] = buf_load(rs, off);
            } else {
                v[d] = buf_load(rs, off);
            }
        }
#pragma unroll
        for (int d = 0; d < 8; ++d) acc += v[d];
    }
    if (acc.x == 12345.678f) sink[blockIdx.x * 64 + lane] = acc.x + acc.y + acc.z + acc.w;
}

// ---- fp32 atomic-add throughput into an L2-resident slice (3 MiB per XCD, block %% 8 affinity).
// MODE 0: dense   -- 32 consecutive floats of 2 random pixels per instruction (2 full lines)
// MODE 1: strided -- 8 random pixels per instruction, lane q of a pixel adds to float 4q+j
// MODE 2: dense, all waves hammer 64 hot pixels (contention)
template <int MODE>
__global__ __launch_bounds__(64) void k_atomic(float* __restrict__ dst0, int iters)
{
    const unsigned lane = threadIdx.x;
    float* dst = dst0 + (size_t)(blockIdx.x % 8) * (3u << 18);  // 3 MiB slices
    const unsigned npx = (3u << 20) / 128;
    unsigned state = (blockIdx.x * 64u + (MODE == 1 ? (lane >> 3) : (lane >> 5))) * 2654435761u + 12345u;
    for (int i = 0; i < iters; ++i) {
        state = state * 1664525u + 1013904223u;
        unsigned p = (unsigned)(((unsigned long long)(state >> 8) * npx) >> 24);
        if (MODE == 2) p &= 63u;
        if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) unsafeAtomicAdd(dst + (size_t)p * 32 + (lane & 7) * 4 + j, 1.0f);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsafeAtomicAdd(dst + (size_t)((p + j * 977u) % npx) * 32 + (lane & 31), 1.0f);
            }
        }
    }
}

// ---- XCD-local hand-off primitives (probed for the one-launch forward experiment of round 2,
// tools/experiments/r02_fused_forward.patch; profiles/r02_fused_experiment.md) ---------------
// rec[b] = raw XCC id register of block b; cnt[x*32] counts arrivals on XCD x with an L2-scope
// atomic; every block then polls its XCD's counter (MODE 0: sc1 load, MODE 1: L2 atomic add 0,
// MODE 2: plain load) until it reaches `expect` or `cap` polls; res[b] = polls used (cap = gave up).
template <int MODE>
__global__ __launch_bounds__(64) void k_xcd_probe(unsigned* rec, unsigned* cnt, unsigned* res, unsigned expect, unsigned cap)
{
    unsigned raw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(raw));
    const unsigned x = raw & 7u;
    if (threadIdx.x == 0) {
        rec[blockIdx.x] = raw;
        __hip_atomic_fetch_add(cnt + x * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        unsigned spins = 0;
        for (;;) {
            unsigned v;
            if (MODE == 0) v = __hip_atomic_load(cnt + x * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (MODE == 1) v = __hip_atomic_fetch_add(cnt + x * 32, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else v = *(volatile unsigned*)(cnt + x * 32);
            if (v >= expect || ++spins >= cap) break;
            __builtin_amdgcn_s_sleep(2);
        }
        res[blockIdx.x] = spins;
    }
}

// ---- returning counter atomics (round 4): does it matter whether the counters a workgroup hits are
// also hit from other XCDs?  Every wave instruction adds 1 to the 32 counters of `lines` random 128-byte lines
// (lines = 1: all 64 lanes on one line, two lanes per counter; 8: eight lanes per line -- the pair pass sees 2-6).
// LOCAL = 0: one counter array for the chip; 1: an array per block % 8 (= per XCD); NORET: fire and forget.
template <int LOCAL, int NORET>
__global__ __launch_bounds__(256) void k_counter_atomics(unsigned* cnt, unsigned* sink, unsigned nlines, unsigned lines, unsigned iters)
{
    const unsigned lane = threadIdx.x & 63u, wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
    unsigned* base = cnt + (LOCAL ? (size_t)(blockIdx.x % 8u) * nlines * 32u : 0u);
    unsigned h = wave * 2654435761u + 12345u, acc = 0;
    const unsigned per = 64u / lines;           // lanes per line
    for (unsigned i = 0; i < iters; ++i) {
        h = h * 1664525u + 1013904223u;
        const unsigned line = ((h >> 8) + (lane / per) * 7919u) % nlines;
        unsigned* p = base + (size_t)line * 32u + (lane % per) % 32u;
        if (NORET) atomicAdd(p, 1u);
        else acc += atomicAdd(p, 1u);
    }
    if (!NORET && acc == 0xdeadbeefu) sink[0] = acc;
}

struct Timer {
    hipEvent_t a, b;
    Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
    template <class F>
    double us(F&& f, int iters = 50, int warm = 5)
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

}  // namespace

int main(int argc, char** argv)
{
    const size_t out_elems = (size_t)R * C * NB;
    float *feat, *out, *rois_d, *sink;
    void* ws;
    const size_t wsb = rroi_align_forward_workspace_bytes(1, C, H, W, R, 0) + (8u << 20);
    CK(hipMalloc(&feat, (size_t)C * H * W * 4));
    CK(hipMalloc(&out, out_elems * 4 + 4096));   // (+ room for the skewed rows of `kbench shape`)
    CK(hipMalloc(&rois_d, R * 24));
    CK(hipMalloc(&ws, wsb));
    CK(hipMalloc(&sink, 1 << 22));
    std::mt19937 rng(0);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    std::vector<float> hf((size_t)C * H * W), hr(R * 6);
    for (auto& x : hf) x = U(rng) - 0.5f;
    for (int i = 0; i < R; ++i) {
        const float h = 16 + 48 * U(rng);
        hr[i * 6 + 0] = 0;
        hr[i * 6 + 1] = 640 * U(rng);
        hr[i * 6 + 2] = 640 * U(rng);
        hr[i * 6 + 3] = h;
        hr[i * 6 + 4] = h * (4 + 4 * U(rng));
        hr[i * 6 + 5] = -90 + 180 * U(rng);
        // SURVEY 8(d) sensitivity points: every bin active / axis-aligned ROIs
        if (getenv("RROI_KB_ALL_ACTIVE")) hr[i * 6 + 4] = h * 8;
        if (getenv("RROI_KB_AXIS")) hr[i * 6 + 5] = 0;
    }
    CK(hipMemcpy(feat, hf.data(), hf.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(rois_d, hr.data(), hr.size() * 4, hipMemcpyHostToDevice));
    Timer T;
    const double MB = out_elems * 4 / 1e6;
    auto report = [&](const char* name, double us, double mb) {
        printf("%-44s %9.2f us  %8.1f GB/s\n", name, us, mb / us * 1e3);
        fflush(stdout);
    };

    if (argc > 1 && std::string(argv[1]) == "pmc") {
        // short list for counter collection: each kernel a few times, no timing loops
        auto stage = [&](int s) {
            int rc = rroi_align_forward_stages_hip(feat, 0, 0.25f, 1, R, H, W, C, PH, PW, rois_d, out, ws, wsb, 2, s, 0);
            if (rc != 1) { fprintf(stderr, "stage rc=%d\n", rc); exit(1); }
        };
        for (int rep = 0; rep < 3; ++rep) {
            stage(1);
            stage(2);
            if (argc > 2) continue;  // "pmc product": the product kernels only
            hipLaunchKernelGGL((k_l2_gather<8, 4>), dim3(4096), dim3(64), 0, 0, (const char*)feat, 3u << 20, 64, sink, 1);
            hipLaunchKernelGGL(k_store_tile<1>, dim3(4096), dim3(64), 0, 0, out, 8, 8);
            CK(hipMemsetAsync(out, 0, out_elems * 4, 0));
            CK(hipDeviceSynchronize());
        }
        return 0;
    }
    if (argc > 1 && std::string(argv[1]) == "ablstep") {
        // the ablations of the gather kernel INSIDE the step (prologue + gather, 20 back-to-back): dbg bit 0 =
        // output stores dropped by the descriptor check, bit 1 = every tap out of range (no map reads)
        auto stage = [&](int s) {
            int rc = rroi_align_forward_stages_hip(feat, 0, 0.25f, 1, R, H, W, C, PH, PW, rois_d, out, ws, wsb, 2, s, 0);
            if (rc != 1) { fprintf(stderr, "stage rc=%d\n", rc); exit(1); }
        };
        for (int i = 0; i < 300; ++i) stage(3);
        CK(hipDeviceSynchronize());
        for (int rep = 0; rep < 2; ++rep)
            for (int dbg : {0, 1, 2, 3}) {
                rroi_align_debug_set_fwd_dbg(dbg);
                char nm[96];
                snprintf(nm, 96, "ablation=%d: gather alone", dbg);
                report(nm, T.us([&] { stage(2); }, 200, 20), MB);
                snprintf(nm, 96, "ablation=%d: whole step", dbg);
                report(nm, T.us([&] { for (int i = 0; i < 20; ++i) stage(3); }, 50, 5) / 20, MB);
            }
        rroi_align_debug_set_fwd_dbg(0);
        report("prologue alone, 20 back-to-back", T.us([&] { for (int i = 0; i < 20; ++i) stage(1); }, 50, 5) / 20, 52.4);
        report("  step: prologue + store-only tile pattern 1/8", T.us([&] { for (int i = 0; i < 20; ++i) { stage(1); hipLaunchKernelGGL((k_store_mix<0, 1>), dim3(3072), dim3(64), 0, 0, out, 8, 8); } }, 50, 5) / 20, MB);
        return 0;
    }
    if (argc > 1 && std::string(argv[1]) == "xatom") {
        const unsigned nlines = 3200;   // 160 x 160 map in 8 x 4 key tiles = 800 lines per image; x4 for spread
        unsigned *cnt, *sink;
        CK(hipMalloc(&cnt, (size_t)8 * nlines * 128));
        CK(hipMalloc(&sink, 64));
        CK(hipMemset(cnt, 0, (size_t)8 * nlines * 128));
        const unsigned iters = 64, blocks = 256 * 4;
        const double atom = (double)blocks * 256 * iters;
        for (unsigned lines : {1u, 2u, 4u, 8u, 16u, 64u}) {
            for (int rep = 0; rep < 2; ++rep) {
                const double a = T.us([&] { hipLaunchKernelGGL((k_counter_atomics<0, 0>), dim3(blocks), dim3(256), 0, 0, cnt, sink, nlines, lines, iters); }, 20, 3);
                const double b = T.us([&] { hipLaunchKernelGGL((k_counter_atomics<1, 0>), dim3(blocks), dim3(256), 0, 0, cnt, sink, nlines, lines, iters); }, 20, 3);
                const double c = T.us([&] { hipLaunchKernelGGL((k_counter_atomics<0, 1>), dim3(blocks), dim3(256), 0, 0, cnt, sink, nlines, lines, iters); }, 20, 3);
                const double d = T.us([&] { hipLaunchKernelGGL((k_counter_atomics<1, 1>), dim3(blocks), dim3(256), 0, 0, cnt, sink, nlines, lines, iters); }, 20, 3);
                printf("lines/instr %2u: returning shared %8.1f us (%6.2f G lane-atomics/s, %6.2f G line-requests/s) | per-XCD arrays %8.1f us (%6.2f, %6.2f) | no-return shared %8.1f us, per-XCD %8.1f us\n",
                       lines, a, atom / a / 1e3, atom / 64 * lines / a / 1e3, b, atom / b / 1e3, atom / 64 * lines / b / 1e3, c, d);
                fflush(stdout);
            }
        }
        return 0;
    }
    if (argc > 1 && std::string(argv[1]) == "bigout") {
        // does the rate of the half-line windows hold when the output is larger than the 256 MB memory-side cache?
        float* big;
        CK(hipMalloc(&big, (size_t)900 * 256 * 1100 * 4));
        for (int rep = 0; rep < 2; ++rep)
            for (unsigned rois : {238u, 476u, 900u}) {
                char nm[128];
                const double mb = (double)rois * 256 * 1100 * 4 / 1e6;
                snprintf(nm, 128, "%u ROIs x 256 x 1100 (%.0f MB): 48-float windows, sc1, piece-fastest", rois, mb);
                report(nm, T.us([&] { hipLaunchKernelGGL((k_store_win48<1100, 48, 0, 16, 16, 1, 1>), dim3(3072), dim3(64), 0, 0, big, 8, rois); }, 30), mb);
                snprintf(nm, 128, "%u ROIs x 256 x 1100 (%.0f MB): 48-float windows, plain", rois, mb);
                report(nm, T.us([&] { hipLaunchKernelGGL((k_store_win48<1100, 48, 0, 0, 0, 1, 1>), dim3(3072), dim3(64), 0, 0, big, 8, rois); }, 30), mb);
                snprintf(nm, 128, "%u ROIs x 256 x 1100 (%.0f MB): 96-float line-aligned windows, sc1", rois, mb);
                report(nm, T.us([&] { hipLaunchKernelGGL((k_store_win96<1100, 16, 1>), dim3(3072), dim3(64), 0, 0, big, 8, rois); }, 30), mb);
                snprintf(nm, 128, "%u ROIs x 256 x 1100 (%.0f MB): 96-float line-aligned windows, nt", rois, mb);
                report(nm, T.us([&] { hipLaunchKernelGGL((k_store_win96<1100, 2, 1>), dim3(3072), dim3(64), 0, 0, big, 8, rois); }, 30), mb);
            }
        return 0;
    }
    if (argc > 1 && std::string(argv[1]) == "desync") {
        // does the write path's rate for HALF-line windows depend on the two halves of a line arriving together?
        // the same stores with every wave idling a random 0 ... 0.8 us before each item (the idle time itself is hidden:
        // 12 waves per CU), against line-aligned windows of 96 floats
        for (int rep = 0; rep < 2; ++rep) {
            const double mb913 = 287.0 * 256 * 913 * 4 / 1e6, mb1100 = 238.0 * 256 * 1100 * 4 / 1e6;
            report("48-float windows, 913, sc1, piece-fastest lanes, in step", T.us([&] { hipLaunchKernelGGL((k_store_win48<913, 48, 0, 16, 16, 0, 1>), dim3(3072), dim3(64), 0, 0, out, 8, 287u); }, 50), mb913);
            report("48-float windows, 913, sc1, piece-fastest lanes, desynchronised", T.us([&] { hipLaunchKernelGGL((k_store_win48<913, 48, 0, 16, 16, 1, 1>), dim3(3072), dim3(64), 0, 0, out, 8, 287u); }, 50), mb913);
            report("48-float windows, 913, nt, piece-fastest lanes, desynchronised", T.us([&] { hipLaunchKernelGGL((k_store_win48<913, 48, 0, 2, 2, 1, 1>), dim3(3072), dim3(64), 0, 0, out, 8, 287u); }, 50), mb913);
            report("48-float windows, 913, 1/6 sc0sc1 + nt, piece-fastest lanes, desynchronised", T.us([&] { hipLaunchKernelGGL((k_store_win48<913, 48, 0, 17, 2, 1, 1>), dim3(3072), dim3(64), 0, 0, out, 8, 287u); }, 50), mb913);
            report("48-float windows, 913, plain, piece-fastest lanes, desynchronised", T.us([&] { hipLaunchKernelGGL((k_store_win48<913, 48, 0, 0, 0, 1, 1>), dim3(3072), dim3(64), 0, 0, out, 8, 287u); }, 50), mb913);
            report("48-float windows, 913, sc1, waves in step", T.us([&] { hipLaunchKernelGGL((k_store_win48<913, 48, 0, 16, 16, 0>), dim3(3072), dim3(64), 0, 0, out, 8, 287u); }, 50), mb913);
            report("48-float windows, 913, sc1, waves desynchronised", T.us([&] { hipLaunchKernelGGL((k_store_win48<913, 48, 0, 16, 16, 1>), dim3(3072), dim3(64), 0, 0, out, 8, 287u); }, 50), mb913);
            report("48-float windows, 913, plain, waves desynchronised", T.us([&] { hipLaunchKernelGGL((k_store_win48<913, 48, 0, 0, 0, 1>), dim3(3072), dim3(64), 0, 0, out, 8, 287u); }, 50), mb913);
            report("48-float windows, 913, nt, waves desynchronised", T.us([&] { hipLaunchKernelGGL((k_store_win48<913, 48, 0, 2, 2, 1>), dim3(3072), dim3(64), 0, 0, out, 8, 287u); }, 50), mb913);
            report("48-float windows, 1100, sc1, waves desynchronised", T.us([&] { hipLaunchKernelGGL((k_store_win48<1100, 48, 0, 16, 16, 1>), dim3(3072), dim3(64), 0, 0, out, 8, 238u); }, 50), mb1100);
            report("96-float line-aligned windows, 913, sc1, in step", T.us([&] { hipLaunchKernelGGL((k_store_win96<913, 16, 0>), dim3(3072), dim3(64), 0, 0, out, 8, 287u); }, 50), mb913);
            report("96-float line-aligned windows, 913, sc1, desynchronised", T.us([&] { hipLaunchKernelGGL((k_store_win96<913, 16, 1>), dim3(3072), dim3(64), 0, 0, out, 8, 287u); }, 50), mb913);
            report("96-float line-aligned windows, 913, nt, desynchronised", T.us([&] { hipLaunchKernelGGL((k_store_win96<913, 2, 1>), dim3(3072), dim3(64), 0, 0, out, 8, 287u); }, 50), mb913);
            report("96-float line-aligned windows, 913, plain, desynchronised", T.us([&] { hipLaunchKernelGGL((k_store_win96<913, 0, 1>), dim3(3072), dim3(64), 0, 0, out, 8, 287u); }, 50), mb913);
            report("96-float line-aligned windows, 1100, sc1, desynchronised", T.us([&] { hipLaunchKernelGGL((k_store_win96<1100, 16, 1>), dim3(3072), dim3(64), 0, 0, out, 8, 238u); }, 50), mb1100);
            report("96-float line-aligned windows, 1100, nt, desynchronised", T.us([&] { hipLaunchKernelGGL((k_store_win96<1100, 2, 1>), dim3(3072), dim3(64), 0, 0, out, 8, 238u); }, 50), mb1100);
        }
        return 0;
    }
    if (argc > 1 && std::string(argv[1]) == "win48") {
        // store-only ceiling of the SHIFT form's windows (48 floats per tile and row, sector-aligned), by row length
        for (int rep = 0; rep < 2; ++rep) {
            report("48-float windows, rows of 1100 floats (11 x 100), 238 ROIs", T.us([&] { hipLaunchKernelGGL(k_store_win48<1100>, dim3(3072), dim3(64), 0, 0, out, 8, 238u); }, 100), 238.0 * 256 * 1100 * 4 / 1e6);
            report("48-float windows, rows of 913 floats (11 x 83), 287 ROIs", T.us([&] { hipLaunchKernelGGL(k_store_win48<913>, dim3(3072), dim3(64), 0, 0, out, 8, 287u); }, 100), 287.0 * 256 * 913 * 4 / 1e6);
            report("48-float windows, rows of 1056 floats (11 x 96), 248 ROIs", T.us([&] { hipLaunchKernelGGL(k_store_win48<1056>, dim3(3072), dim3(64), 0, 0, out, 8, 248u); }, 100), 248.0 * 256 * 1056 * 4 / 1e6);
            report("48-float windows, 1100, plain stores (write-back: the L2 may merge a line's halves)", T.us([&] { hipLaunchKernelGGL((k_store_win48<1100, 48, 0, 0, 0>), dim3(3072), dim3(64), 0, 0, out, 8, 238u); }, 100), 238.0 * 256 * 1100 * 4 / 1e6);
            report("48-float windows, 913, plain stores", T.us([&] { hipLaunchKernelGGL((k_store_win48<913, 48, 0, 0, 0>), dim3(3072), dim3(64), 0, 0, out, 8, 287u); }, 100), 287.0 * 256 * 913 * 4 / 1e6);
            report("48-float windows, 913, plain stores, a row's sectors back to back", T.us([&] { hipLaunchKernelGGL((k_store_win48<913, 48, 1, 0, 0>), dim3(3072), dim3(64), 0, 0, out, 8, 287u); }, 100), 287.0 * 256 * 913 * 4 / 1e6);
            report("48-float windows, 913, 1/6 write-through + plain", T.us([&] { hipLaunchKernelGGL((k_store_win48<913, 48, 0, 17, 0>), dim3(3072), dim3(64), 0, 0, out, 8, 287u); }, 100), 287.0 * 256 * 913 * 4 / 1e6);
            report("48-float windows, 913, all write-through (sc1)", T.us([&] { hipLaunchKernelGGL((k_store_win48<913, 48, 0, 16, 16>), dim3(3072), dim3(64), 0, 0, out, 8, 287u); }, 100), 287.0 * 256 * 913 * 4 / 1e6);
            report("48-float windows, 1100, a row's sectors back to back", T.us([&] { hipLaunchKernelGGL((k_store_win48<1100, 48, 1>), dim3(3072), dim3(64), 0, 0, out, 8, 238u); }, 100), 238.0 * 256 * 1100 * 4 / 1e6);
            report("48-float windows, 913, a row's sectors back to back", T.us([&] { hipLaunchKernelGGL((k_store_win48<913, 48, 1>), dim3(3072), dim3(64), 0, 0, out, 8, 287u); }, 100), 287.0 * 256 * 913 * 4 / 1e6);
            report("64-float windows, 913, a row's sectors back to back", T.us([&] { hipLaunchKernelGGL((k_store_win48<913, 64, 1>), dim3(3072), dim3(64), 0, 0, out, 8, 287u); }, 100), 287.0 * 256 * 913 * 4 / 1e6);
            report("64-float windows, rows of 1100 floats, 238 ROIs", T.us([&] { hipLaunchKernelGGL((k_store_win48<1100, 64>), dim3(3072), dim3(64), 0, 0, out, 8, 238u); }, 100), 238.0 * 256 * 1100 * 4 / 1e6);
            report("64-float windows, rows of 913 floats, 287 ROIs", T.us([&] { hipLaunchKernelGGL((k_store_win48<913, 64>), dim3(3072), dim3(64), 0, 0, out, 8, 287u); }, 100), 287.0 * 256 * 913 * 4 / 1e6);
            report("32-float windows, rows of 913 floats, 287 ROIs", T.us([&] { hipLaunchKernelGGL((k_store_win48<913, 32>), dim3(3072), dim3(64), 0, 0, out, 8, 287u); }, 100), 287.0 * 256 * 913 * 4 / 1e6);
            report("128-float windows, rows of 913 floats, 287 ROIs", T.us([&] { hipLaunchKernelGGL((k_store_win48<913, 128>), dim3(3072), dim3(64), 0, 0, out, 8, 287u); }, 100), 287.0 * 256 * 913 * 4 / 1e6);
            report("112-float windows, rows of 913 floats, 287 ROIs", T.us([&] { hipLaunchKernelGGL((k_store_win48<913, 112>), dim3(3072), dim3(64), 0, 0, out, 8, 287u); }, 100), 287.0 * 256 * 913 * 4 / 1e6);
        }
        return 0;
    }
    if (argc > 1 && std::string(argv[1]) == "shape") {
        // what the write path makes of a tile's stores by the shape of one wave instruction and the rows' alignment
        for (int rep = 0; rep < 2; ++rep)
            for (unsigned skew : {0u, 16u, 4u, 1u}) {
                char nm[96];
                snprintf(nm, 96, "4 rows x 256 B per instruction, rows + %u floats", skew);
                report(nm, T.us([&] { hipLaunchKernelGGL(k_store_shape<0>, dim3(3072), dim3(64), 0, 0, out, 8, 8, skew); }, 100), MB);
                snprintf(nm, 96, "16 rows x 64 B per instruction, rows + %u floats", skew);
                report(nm, T.us([&] { hipLaunchKernelGGL(k_store_shape<1>, dim3(3072), dim3(64), 0, 0, out, 8, 8, skew); }, 100), MB);
                snprintf(nm, 96, "8 rows x 128 B per instruction, rows + %u floats", skew);
                report(nm, T.us([&] { hipLaunchKernelGGL(k_store_shape<2>, dim3(3072), dim3(64), 0, 0, out, 8, 8, skew); }, 100), MB);
            }
        return 0;
    }
    if (argc > 1 && std::string(argv[1]) == "wceil") {
        // what a plain write of 256 MiB reaches on this box, by data and by store policy (profiles/r04_write_ceiling.txt)
        const unsigned n4 = (unsigned)((size_t)R * C * NB / 4);
        const unsigned blocks = (n4 + 255) / 256;
        for (int rep = 0; rep < 3; ++rep) {
            report("hipMemsetAsync 0x00", T.us([&] { CK(hipMemsetAsync(out, 0, (size_t)n4 * 16, 0)); }, 100), MB);
            report("hipMemsetAsync 0x3f", T.us([&] { CK(hipMemsetAsync(out, 0x3f, (size_t)n4 * 16, 0)); }, 100), MB);
            report("one float4 per thread, zeros, plain", T.us([&] { hipLaunchKernelGGL((k_store_once<0, 0>), dim3(blocks), dim3(256), 0, 0, out, n4); }, 100), MB);
            report("one float4 per thread, constant 1.0, plain", T.us([&] { hipLaunchKernelGGL((k_store_once<1, 0>), dim3(blocks), dim3(256), 0, 0, out, n4); }, 100), MB);
            report("one float4 per thread, index-dependent values, plain", T.us([&] { hipLaunchKernelGGL((k_store_once<2, 0>), dim3(blocks), dim3(256), 0, 0, out, n4); }, 100), MB);
            report("one float4 per thread, index-dependent values, nt", T.us([&] { hipLaunchKernelGGL((k_store_once<2, 2>), dim3(blocks), dim3(256), 0, 0, out, n4); }, 100), MB);
            report("one float4 per thread, index-dependent values, sc1", T.us([&] { hipLaunchKernelGGL((k_store_once<2, 16>), dim3(blocks), dim3(256), 0, 0, out, n4); }, 100), MB);
            report("one float4 per thread, index-dependent values, sc0 sc1", T.us([&] { hipLaunchKernelGGL((k_store_once<2, 17>), dim3(blocks), dim3(256), 0, 0, out, n4); }, 100), MB);
            report("grid-stride 256-thread blocks, constant, plain", T.us([&] { hipLaunchKernelGGL((k_store_linear256<1>), dim3(256 * 8), dim3(256), 0, 0, out, (size_t)n4); }, 100), MB);
            report("the gather's tile pattern, 1/8 sc0sc1 + nt (what ships)", T.us([&] { hipLaunchKernelGGL((k_store_mix<0, 1>), dim3(3072), dim3(64), 0, 0, out, 8, 8); }, 100), MB);
        }
        return 0;
    }
    if (argc > 1 && std::string(argv[1]) == "stmix") {
        // store-only kernels with the shipped policy mix, two address patterns; alone and followed by the
        // product's prologue (what the write stream does to the next call's relayout)
        auto stage = [&](int s) {
            int rc = rroi_align_forward_stages_hip(feat, 0, 0.25f, 1, R, H, W, C, PH, PW, rois_d, out, ws, wsb, 2, s, 0);
            if (rc != 1) { fprintf(stderr, "stage rc=%d\n", rc); exit(1); }
        };
        for (int i = 0; i < 300; ++i) stage(3);
        CK(hipDeviceSynchronize());
        for (int rep = 0; rep < 2; ++rep) {
            report("tile pattern, 1/8 sc0sc1 + nt", T.us([&] { hipLaunchKernelGGL((k_store_mix<0, 1>), dim3(3072), dim3(64), 0, 0, out, 8, 8); }, 100), MB);
            report("8 KiB contiguous per wave, 1/8 sc0sc1 + nt", T.us([&] { hipLaunchKernelGGL((k_store_mix<1, 1>), dim3(3072), dim3(64), 0, 0, out, 8, 8); }, 100), MB);
            report("tile pattern, pure nt", T.us([&] { hipLaunchKernelGGL((k_store_mix<0, 0>), dim3(3072), dim3(64), 0, 0, out, 8, 8); }, 100), MB);
            report("8 KiB contiguous per wave, pure nt", T.us([&] { hipLaunchKernelGGL((k_store_mix<1, 0>), dim3(3072), dim3(64), 0, 0, out, 8, 8); }, 100), MB);
            report("tile pattern, 2/8 sc0sc1 + nt", T.us([&] { hipLaunchKernelGGL((k_store_mix<0, 2>), dim3(3072), dim3(64), 0, 0, out, 8, 8); }, 100), MB);
            report("8 KiB contiguous per wave, 2/8 sc0sc1 + nt", T.us([&] { hipLaunchKernelGGL((k_store_mix<1, 2>), dim3(3072), dim3(64), 0, 0, out, 8, 8); }, 100), MB);
            report("  step: prologue + tile pattern 1/8", T.us([&] { for (int i = 0; i < 20; ++i) { stage(1); hipLaunchKernelGGL((k_store_mix<0, 1>), dim3(3072), dim3(64), 0, 0, out, 8, 8); } }, 20, 3) / 20, MB);
            report("  step: prologue + 8 KiB contiguous 1/8", T.us([&] { for (int i = 0; i < 20; ++i) { stage(1); hipLaunchKernelGGL((k_store_mix<1, 1>), dim3(3072), dim3(64), 0, 0, out, 8, 8); } }, 20, 3) / 20, MB);
            report("  step: prologue + 8 KiB contiguous 2/8", T.us([&] { for (int i = 0; i < 20; ++i) { stage(1); hipLaunchKernelGGL((k_store_mix<1, 2>), dim3(3072), dim3(64), 0, 0, out, 8, 8); } }, 20, 3) / 20, MB);
            report("  step: product (prologue + gather)", T.us([&] { for (int i = 0; i < 20; ++i) stage(3); }, 20, 3) / 20, MB);
        }
        return 0;
    }
    if (argc > 1 && std::string(argv[1]) == "xcd") {
        const int grid = 3072;
        unsigned *rec, *cnt, *res;
        CK(hipMalloc(&rec, grid * 4)); CK(hipMalloc(&cnt, 8 * 128)); CK(hipMalloc(&res, grid * 4));
        std::vector<unsigned> hrec(grid), hres(grid), hcnt(256);
        for (int mode = 0; mode < 3; ++mode) {
            CK(hipMemset(cnt, 0, 8 * 128));
            if (mode == 0) hipLaunchKernelGGL(k_xcd_probe<0>, dim3(grid), dim3(64), 0, 0, rec, cnt, res, grid / 8, 20000u);
            if (mode == 1) hipLaunchKernelGGL(k_xcd_probe<1>, dim3(grid), dim3(64), 0, 0, rec, cnt, res, grid / 8, 20000u);
            if (mode == 2) hipLaunchKernelGGL(k_xcd_probe<2>, dim3(grid), dim3(64), 0, 0, rec, cnt, res, grid / 8, 20000u);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(hrec.data(), rec, grid * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hres.data(), res, grid * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hcnt.data(), cnt, 8 * 128, hipMemcpyDeviceToHost));
            unsigned mism = 0, gaveup = 0, maxs = 0, ormask = 0, andmask = ~0u;
            for (int b = 0; b < grid; ++b) {
                mism += (hrec[b] & 7u) != (unsigned)(b % 8);
                gaveup += hres[b] >= 20000u;
                maxs = std::max(maxs, hres[b]);
                ormask |= hrec[b]; andmask &= hrec[b];
            }
            printf("mode %d: xcc != block%%8 in %u of %d blocks; raw id or-mask 0x%x and-mask 0x%x; gave up %u; max polls %u; counts",
                   mode, mism, grid, ormask, andmask, gaveup, maxs);
            for (int x = 0; x < 8; ++x) printf(" %u", hcnt[x * 32]);
            printf("\n");
            fflush(stdout);
        }
        return 0;
    }
    if (argc > 1 && std::string(argv[1]) == "order") {
        // does the ORDER in which the ROIs are processed matter?  (spatially sorted ROIs keep a smaller
        // part of the map slice hot in the L2 at any time)
        auto stage = [&](int st) {
            int rc = rroi_align_forward_stages_hip(feat, 0, 0.25f, 1, R, H, W, C, PH, PW, rois_d, out, ws, wsb, 2, st, 0);
            if (rc != 1) { fprintf(stderr, "stage rc=%d\n", rc); exit(1); }
        };
        for (int i = 0; i < 300; ++i) stage(3);
        CK(hipDeviceSynchronize());
        std::vector<int> idx(R);
        for (int mode = 0; mode < 4; ++mode) {
            for (int i = 0; i < R; ++i) idx[i] = i;
            auto morton = [&](int i) {
                unsigned x = (unsigned)(hr[i * 6 + 1] * 0.25f) >> 3, y = (unsigned)(hr[i * 6 + 2] * 0.25f) >> 3, m = 0;
                for (int b = 0; b < 6; ++b) m |= ((x >> b) & 1u) << (2 * b) | ((y >> b) & 1u) << (2 * b + 1);
                return m;
            };
            if (mode == 1) std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return hr[a * 6 + 2] < hr[b * 6 + 2]; });
            if (mode == 2) std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return morton(a) < morton(b); });
            if (mode == 3) std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) {   // bands of 32 map rows, x inside
                const int ba = (int)(hr[a * 6 + 2] * 0.25f) / 32, bb = (int)(hr[b * 6 + 2] * 0.25f) / 32;
                return ba != bb ? ba < bb : hr[a * 6 + 1] < hr[b * 6 + 1]; });
            std::vector<float> pr(R * 6);
            for (int i = 0; i < R; ++i) for (int k = 0; k < 6; ++k) pr[i * 6 + k] = hr[idx[i] * 6 + k];
            CK(hipMemcpy(rois_d, pr.data(), pr.size() * 4, hipMemcpyHostToDevice));
            const char* nm4[4] = {"as generated", "sorted by centre y", "sorted by Morton code of the centre (8 px cells)", "32-row bands, x inside"};
            char nm[128];
            // (round 2 also swept the write-through stores per tile here; that template parameter is a constant now)
            stage(3);
            snprintf(nm, 128, "gather, ROIs %s", nm4[mode]);
            report(nm, T.us([&] { stage(2); }, 200, 20), MB);
            snprintf(nm, 128, "  whole step, ROIs %s", nm4[mode]);
            report(nm, T.us([&] { for (int i = 0; i < 20; ++i) stage(3); }, 50, 5) / 20, MB);
        }
        return 0;
    }
    if (argc > 1 && std::string(argv[1]) == "pro") {
        // prologue shape sweep: resident blocks per CU x store policy, as the whole step sees it
        auto stage = [&](int st) {
            int rc = rroi_align_forward_stages_hip(feat, 0, 0.25f, 1, R, H, W, C, PH, PW, rois_d, out, ws, wsb, 2, st, 0);
            if (rc != 1) { fprintf(stderr, "stage rc=%d\n", rc); exit(1); }
        };
        for (int i = 0; i < 300; ++i) stage(3);
        CK(hipDeviceSynchronize());
        rroi_align_debug_set_prologue_blocks(3);
        for (int rep = 0; rep < 5; ++rep)
            for (int paux : {0, 16}) {   // A/B/A/B: is the write-through prologue better over the whole step?
    // (the prologue store-policy knob went with round 5: plain stores ship)
                char nm[96];
                snprintf(nm, 96, "A/B %d: whole step, prologue aux=%d, 50 x 20 steps", rep, paux);
                report(nm, T.us([&] { for (int i = 0; i < 20; ++i) stage(3); }, 50, 5) / 20, MB);
            }
        for (int paux : {0, 16}) {
    // (the prologue store-policy knob went with round 5: plain stores ship)
            for (int bpc : {3}) {
                rroi_align_debug_set_prologue_blocks(bpc);
                char nm[96];
                snprintf(nm, 96, "prologue aux=%d blocks/CU=%d: 20 back-to-back", paux, bpc);
                report(nm, T.us([&] { for (int i = 0; i < 20; ++i) stage(1); }, 20, 3) / 20, 52.4);
                snprintf(nm, 96, "  whole step, prologue aux=%d blocks/CU=%d", paux, bpc);
                report(nm, T.us([&] { for (int i = 0; i < 20; ++i) stage(3); }, 20, 3) / 20, MB);
            }
        }
        return 0;
    }
    if (argc > 1 && std::string(argv[1]) == "abl") {
        // gather-kernel ablations only: dbg bit 0 = no output stores, bit 1 = all taps out of range
        auto stage = [&](int s) {
            int rc = rroi_align_forward_stages_hip(feat, 0, 0.25f, 1, R, H, W, C, PH, PW, rois_d, out, ws, wsb, 2, s, 0);
            if (rc != 1) { fprintf(stderr, "stage rc=%d\n", rc); exit(1); }
        };
        for (int i = 0; i < 300; ++i) stage(3);  // clocks
        CK(hipDeviceSynchronize());
        for (int wpc : {10, 12}) {
            rroi_align_debug_set_split_wgs_per_cu(wpc);
            for (int dbg : {0, 1}) {
                rroi_align_debug_set_fwd_dbg(dbg);
                char nm[96];
                snprintf(nm, 96, "gather %d workgroups/CU ablation=%d", wpc, dbg);
                report(nm, T.us([&] { stage(2); }, 100), MB);
            }
            rroi_align_debug_set_fwd_dbg(0);
            const double pipe = T.us([&] { for (int i = 0; i < 20; ++i) stage(3); }, 10, 2) / 20;
            char nm[96];
            snprintf(nm, 96, "pipeline step workgroups/CU=%d", wpc);
            report(nm, pipe, MB);
        }
        return 0;
    }
    report("hipMemsetAsync 256MiB", T.us([&] { CK(hipMemsetAsync(out, 0, out_elems * 4, 0)); }), MB);
    for (int g : {2048, 4096, 8192, 16384})
    {
        char nm[96];
        snprintf(nm, 96, "store_linear256 nt grid=%d", g);
        report(nm, T.us([&] { hipLaunchKernelGGL(k_store_linear256<0>, dim3(g), dim3(256), 0, 0, out, out_elems / 4); }), MB);
        snprintf(nm, 96, "store_linear256 plain grid=%d", g);
        report(nm, T.us([&] { hipLaunchKernelGGL(k_store_linear256<1>, dim3(g), dim3(256), 0, 0, out, out_elems / 4); }), MB);
    }
    for (int g : {4096, 8192, 32768}) {
        char nm[96];
        snprintf(nm, 96, "store_linear(wave 8KiB pieces) nt grid=%d", g);
        report(nm, T.us([&] { hipLaunchKernelGGL(k_store_linear<0>, dim3(g), dim3(64), 0, 0, out, (unsigned)(out_elems / 2048)); }), MB);
        snprintf(nm, 96, "store_linear(wave 8KiB pieces) plain grid=%d", g);
        report(nm, T.us([&] { hipLaunchKernelGGL(k_store_linear<1>, dim3(g), dim3(64), 0, 0, out, (unsigned)(out_elems / 2048)); }), MB);
    }
    for (int g : {4096, 8192, 32768}) {
        char nm[96];
        snprintf(nm, 96, "store_tile(product pattern) nt grid=%d", g);
        report(nm, T.us([&] { hipLaunchKernelGGL(k_store_tile<0>, dim3(g), dim3(64), 0, 0, out, 8, 8); }), MB);
        snprintf(nm, 96, "store_tile(product pattern) plain grid=%d", g);
        report(nm, T.us([&] { hipLaunchKernelGGL(k_store_tile<1>, dim3(g), dim3(64), 0, 0, out, 8, 8); }), MB);
    }

    report("store_aux plain(0)", T.us([&] { hipLaunchKernelGGL(k_store_aux<0>, dim3(4096), dim3(64), 0, 0, out, 8, 8); }), MB);
    report("store_aux sc0(1)", T.us([&] { hipLaunchKernelGGL(k_store_aux<1>, dim3(4096), dim3(64), 0, 0, out, 8, 8); }), MB);
    report("store_aux nt(2)", T.us([&] { hipLaunchKernelGGL(k_store_aux<2>, dim3(4096), dim3(64), 0, 0, out, 8, 8); }), MB);
    report("store_aux sc0+nt(3)", T.us([&] { hipLaunchKernelGGL(k_store_aux<3>, dim3(4096), dim3(64), 0, 0, out, 8, 8); }), MB);
    report("store_aux sc1(16)", T.us([&] { hipLaunchKernelGGL(k_store_aux<16>, dim3(4096), dim3(64), 0, 0, out, 8, 8); }), MB);
    report("store_aux sc0+sc1(17)", T.us([&] { hipLaunchKernelGGL(k_store_aux<17>, dim3(4096), dim3(64), 0, 0, out, 8, 8); }), MB);
    report("store_aux sc1+nt(18)", T.us([&] { hipLaunchKernelGGL(k_store_aux<18>, dim3(4096), dim3(64), 0, 0, out, 8, 8); }), MB);
    report("store_aux sc0+sc1+nt(19)", T.us([&] { hipLaunchKernelGGL(k_store_aux<19>, dim3(4096), dim3(64), 0, 0, out, 8, 8); }), MB);
    {
        auto ta = [&](const char* name, auto kern) {
            const int iters = 64, grid = 4096;
            const double us = T.us([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, feat, iters, sink); }, 20, 3);
            const double instr_per_cu = (double)grid * iters * 8 / 256;
            printf("%-44s %9.2f us  %6.1f clk/instr/CU @2.1GHz\n", name, us, us * 2100.0 / instr_per_cu);
        };
        ta("ta_rate all lanes in range", k_ta_rate<0>);
        ta("ta_rate half the groups OOB", k_ta_rate<1>);
        ta("ta_rate 7/8 groups OOB", k_ta_rate<2>);
        ta("ta_rate all OOB", k_ta_rate<3>);
        ta("ta_rate half the groups exec-masked", k_ta_rate<4>);
    }
    {
        auto at = [&](const char* name, auto kern) {
            const int iters = 64, grid = 4096;
            const double us = T.us([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, feat, iters); }, 10, 2);
            const double instr = (double)grid * iters * 4;
            printf("%-44s %9.2f us  %7.2f G lane-atomics/s  %6.1f clk/instr/CU @2.1GHz\n", name, us,
                   instr * 64 / us / 1e3, us * 2100.0 / (instr / 256));
        };
        at("atomic f32 dense (2 lines/instr)", k_atomic<0>);
        at("atomic f32 strided (8 lines/instr)", k_atomic<1>);
        at("atomic f32 dense, 64 hot pixels", k_atomic<2>);
    }
    // product stages
    auto stage = [&](int s) {
        int rc = rroi_align_forward_stages_hip(feat, 0, 0.25f, 1, R, H, W, C, PH, PW, rois_d, out, ws, wsb, 2, s, 0);
        if (rc != 1) { fprintf(stderr, "stage rc=%d\n", rc); exit(1); }
    };
    stage(3);
    CK(hipDeviceSynchronize());
    for (int paux : {0, 2, 16}) {
    // (the prologue store-policy knob went with round 5: plain stores ship)
        char nm[96];
        snprintf(nm, 96, "product prologue aux=%d warm", paux);
        report(nm, T.us([&] { stage(1); }), 52.4);
        snprintf(nm, 96, "product prologue aux=%d after 256MiB fill", paux);
        const double both = T.us([&] { CK(hipMemsetAsync(out, 0, out_elems * 4, 0)); stage(1); });
        const double fill = T.us([&] { CK(hipMemsetAsync(out, 0, out_elems * 4, 0)); });
        report(nm, both - fill, 52.4);
        snprintf(nm, 96, "product all, prologue aux=%d", paux);
        report(nm, T.us([&] { stage(3); }, 100), MB);
        // steady-state pipeline: 20 back-to-back steps between two events
        const double pipe = T.us([&] { for (int i = 0; i < 20; ++i) stage(3); }, 10, 2) / 20;
        snprintf(nm, 96, "pipeline step (20 back-to-back), prologue aux=%d", paux);
        report(nm, pipe, MB);
    }
    // (the prologue store-policy knob went with round 5: plain stores ship)
    rroi_align_debug_set_prologue_blocks(3);
    // (the prologue store-policy knob went with round 5: plain stores ship)
    for (int wpc : {10, 12}) {
        rroi_align_debug_set_split_wgs_per_cu(wpc);
        char nm[96];
        for (int dbg : {0, 1}) {
            rroi_align_debug_set_fwd_dbg(dbg);
            snprintf(nm, 96, "gather %d workgroups/CU ablation=%d", wpc, dbg);
            report(nm, T.us([&] { stage(2); }, 100), MB);
        }
        rroi_align_debug_set_fwd_dbg(0);
        const double pipe = T.us([&] { for (int i = 0; i < 20; ++i) stage(3); }, 10, 2) / 20;
        snprintf(nm, 96, "pipeline step workgroups/CU=%d", wpc);
        report(nm, pipe, MB);
    }
    rroi_align_debug_set_split_wgs_per_cu(12);
    report("product all", T.us([&] { stage(3); }, 100), MB);

    // raw gather bandwidth out of L2 / L1 (per-XCD slice sized regions)
    {
        const char* src = reinterpret_cast<const char*>(feat);
        auto run = [&](const char* name, auto kern, unsigned region, int depth, int grid) {
            const int iters = 256 / depth;
            const double bytes = (double)grid * 64 * 16 * iters * depth;
            const double us = T.us([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, src, region, iters, sink, 0); }, 20, 3);
            printf("%-52s %9.2f us  %8.1f GB/s  %6.1f B/clk/CU@2.1GHz\n", name, us, bytes / us / 1e3, bytes / us / 1e3 / 256 / 2.1);
            fflush(stdout);
        };
        auto run2 = [&](const char* name, auto kern, unsigned region, int depth, int grid, int mode) {
            const int iters = 256 / depth;
            const double bytes = (double)grid * 64 * 16 * iters * depth;
            const double us = T.us([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, src, region, iters, sink, mode); }, 20, 3);
            printf("%-52s %9.2f us  %8.1f GB/s  %6.1f B/clk/CU@2.1GHz\n", name, us, bytes / us / 1e3, bytes / us / 1e3 / 256 / 2.1);
            fflush(stdout);
        };
        run2("l2 8x128B region=3MB shared by all   depth4", k_l2_gather<8, 4>, 3u << 20, 4, 4096, 0);
        run2("l2 8x128B region=1MB shared by all   depth4", k_l2_gather<8, 4>, 1u << 20, 4, 4096, 0);
        run2("l2 8x128B 3MB slice = block%8        depth4", k_l2_gather<8, 4>, 3u << 20, 4, 4096, 1);
        run2("l2 8x128B 3MB slice = (block/8)%8    depth4", k_l2_gather<8, 4>, 3u << 20, 4, 4096, 2);
        run2("l2 1KiB   3MB slice = block%8        depth4", k_l2_gather<64, 4>, 3u << 20, 4, 4096, 1);
        run2("l2 8x128B 3MB slice = block%8        depth8", k_l2_gather<8, 8>, 3u << 20, 8, 4096, 1);
        run2("l2 8x128B 3MB slice = block%8 grid8192 d4", k_l2_gather<8, 4>, 3u << 20, 4, 8192, 1);
        const unsigned L2R = 24u << 20, L1R = 16u << 10;
        for (int grid : {4096, 8192}) {
            char nm[96];
            snprintf(nm, 96, "l2 contiguous 1KiB depth1 grid=%d", grid); run(nm, k_l2_gather<64, 1>, L2R, 1, grid);
            snprintf(nm, 96, "l2 contiguous 1KiB depth4 grid=%d", grid); run(nm, k_l2_gather<64, 4>, L2R, 4, grid);
            snprintf(nm, 96, "l2 contiguous 1KiB depth8 grid=%d", grid); run(nm, k_l2_gather<64, 8>, L2R, 8, grid);
            snprintf(nm, 96, "l2 8x128B lines   depth1 grid=%d", grid); run(nm, k_l2_gather<8, 1>, L2R, 1, grid);
            snprintf(nm, 96, "l2 8x128B lines   depth4 grid=%d", grid); run(nm, k_l2_gather<8, 4>, L2R, 4, grid);
            snprintf(nm, 96, "l2 8x128B lines   depth8 grid=%d", grid); run(nm, k_l2_gather<8, 8>, L2R, 8, grid);
            snprintf(nm, 96, "l2 16x64B         depth4 grid=%d", grid); run(nm, k_l2_gather<4, 4>, L2R, 4, grid);
            snprintf(nm, 96, "l2 64x16B         depth4 grid=%d", grid); run(nm, k_l2_gather<1, 4>, L2R, 4, grid);
            snprintf(nm, 96, "L1 8x128B lines   depth4 grid=%d", grid); run(nm, k_l2_gather<8, 4>, L1R, 4, grid);
            snprintf(nm, 96, "L1 contiguous     depth4 grid=%d", grid); run(nm, k_l2_gather<64, 4>, L1R, 4, grid);
        }
    }
    return 0;
}
