// tools/anyorder_probe.hip -- does a kernel launched with hipExtAnyOrderLaunch (AQL packet without the barrier bit)
// start while its predecessor on the SAME stream is still running on gfx950, and are the workgroups of the two
// dispatches handed out in queue order?  Measurement tooling (round 4), not product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/anyorder_probe tools/anyorder_probe.hip
// Cases (timestamps are s_memrealtime, 100 MHz):
//   1  A (spins 30 us, 256 small blocks) then B, normal launch          -> B starts after A ends (the boundary cost)
//   2  A then B with hipExtAnyOrderLaunch                                -> does B start before A ends?
//   3  A oversubscribed (more blocks than the chip holds, each spins 10 us) then B any-order:
//      B's first start against the start of A's LAST block             -> is the grid walk of A finished before B's begins?
//   4  hand-off: A = producer blocks that publish a flag at their end, B (any order) polls the flags with a bounded
//      spin and records when it saw them                                -> the latency a device-side dependency costs
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <time.h>

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

__device__ __forceinline__ unsigned long long now() { return wall_clock64(); }   // 100 MHz

struct Stamps {
    unsigned long long a_first_start, a_last_start, a_last_end, b_first_start, b_last_start, b_seen_last, b_timeouts;
};

__global__ void k_reset(Stamps* s)
{
    s->a_first_start = ~0ull;
    s->a_last_start = 0;
    s->a_last_end = 0;
    s->b_first_start = ~0ull;
    s->b_last_start = 0;
    s->b_seen_last = 0;
    s->b_timeouts = 0;
}

// A: every block spins `ticks` of the 100 MHz clock, optionally with LDS so that only `per_cu` blocks fit a CU
template <int LDS_BYTES>
__global__ __launch_bounds__(256) void k_a(Stamps* s, unsigned ticks, unsigned* flags, unsigned tag, unsigned stagger = 0)
{
    __shared__ char pad[LDS_BYTES];
    ticks += stagger * blockIdx.x;
    if (threadIdx.x == 0) {
        const unsigned long long t0 = now();
        if (stagger) atomicMin(&s->b_seen_last, t0 + ticks);   // (staggered runs: earliest planned END of an A block)
        atomicMin(&s->a_first_start, t0);
        atomicMax(&s->a_last_start, t0);
        pad[0] = (char)t0;
        while (now() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
        atomicMax(&s->a_last_end, now());
        if (flags) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(flags + blockIdx.x, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (pad[0] == 1 && ticks == 0xffffffffu) s->b_timeouts = pad[LDS_BYTES - 1];
    }
}

// B: records its start; with flags, waits (bounded) until all nflags carry the tag
__global__ __launch_bounds__(128) void k_b(Stamps* s, const unsigned* flags, unsigned nflags, unsigned tag, unsigned max_ticks)
{
    if (threadIdx.x >= 64) return;
    const unsigned long long t0 = now();
    if (threadIdx.x == 0) {
        atomicMin(&s->b_first_start, t0);
        atomicMax(&s->b_last_start, t0);
    }
    if (!flags) return;
    bool ok = false;
    while (now() - t0 < max_ticks) {
        bool mine = true;
        for (unsigned j = threadIdx.x; j < nflags; j += 64)
            mine = mine && __hip_atomic_load(flags + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == tag;
        if (__ballot(mine) == ~0ull) {
            ok = true;
            break;
        }
        __builtin_amdgcn_s_sleep(4);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (threadIdx.x == 0) {
        if (ok) atomicMax(&s->b_seen_last, now());
        else atomicAdd(&s->b_timeouts, 1ull);
    }
}

static void report(const char* what, const Stamps& h)
{
    const double t0 = (double)h.a_first_start;
    auto us = [&](unsigned long long t) { return ((double)t - t0) / 100.0; };
    printf("%-58s A last start %7.2f  A last end %7.2f | B first start %7.2f  B last start %7.2f", what, us(h.a_last_start),
           us(h.a_last_end), us(h.b_first_start), us(h.b_last_start));
    if (h.b_seen_last) printf("  B saw flags %7.2f", us(h.b_seen_last));
    if (h.b_timeouts) printf("  TIMEOUTS %llu", h.b_timeouts);
    printf("\n");
}

int main()
{
    hipStream_t st;
    CK(hipStreamCreate(&st));
    Stamps* d;
    CK(hipMalloc(&d, sizeof(Stamps)));
    unsigned* flags;
    CK(hipMalloc(&flags, 1 << 20));
    CK(hipMemset(flags, 0, 1 << 20));
    Stamps h;
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("device %s, %d CUs\n", p.gcnArchName, p.multiProcessorCount);
    const int cus = p.multiProcessorCount;

    for (int rep = 0; rep < 3; ++rep) {
        for (int any = 0; any < 2; ++any) {
            // cases 1 / 2
            hipLaunchKernelGGL(k_reset, dim3(1), dim3(1), 0, st, d);
            hipLaunchKernelGGL(k_a<16>, dim3(cus), dim3(256), 0, st, d, 3000u, (unsigned*)nullptr, 0u);
            hipExtLaunchKernelGGL(k_b, dim3(3072), dim3(128), 0, st, nullptr, nullptr, any ? hipExtAnyOrderLaunch : 0, d,
                                  (const unsigned*)nullptr, 0u, 0u, 0u);
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost));
            report(any ? "2 A(30us, 1 block/CU) -> B any-order" : "1 A(30us, 1 block/CU) -> B ordered", h);
        }
        {
            // case 3: A oversubscribed: 64 KB of LDS per block -> 2 blocks per CU resident, 6 per CU launched
            hipLaunchKernelGGL(k_reset, dim3(1), dim3(1), 0, st, d);
            hipLaunchKernelGGL(k_a<65536>, dim3(cus * 6), dim3(256), 0, st, d, 1000u, (unsigned*)nullptr, 0u);
            hipExtLaunchKernelGGL(k_b, dim3(3072), dim3(128), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, d,
                                  (const unsigned*)nullptr, 0u, 0u, 0u);
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost));
            report("3 A oversubscribed (6 x 10us rounds of 2/CU) -> B any-order", h);
        }
        for (int any = 0; any < 2; ++any) {
            // case 4: hand-off through flags; A = 768 blocks (3 per CU) of 8 us, B = 3072 workgroups polling 96 flags each
            const unsigned tag = 1000u + rep * 2 + any;
            hipLaunchKernelGGL(k_reset, dim3(1), dim3(1), 0, st, d);
            hipLaunchKernelGGL(k_a<16384>, dim3(768), dim3(256), 0, st, d, 800u, flags, tag);
            hipExtLaunchKernelGGL(k_b, dim3(3072), dim3(128), 0, st, nullptr, nullptr, any ? hipExtAnyOrderLaunch : 0, d,
                                  (const unsigned*)flags, 768u, tag, 200000u /* 2 ms */);
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost));
            report(any ? "4 A(768 x 8us, publishes flags) -> B any-order polls" : "4 A(768 x 8us, publishes flags) -> B ordered polls", h);
        }
    }
    // ---- round 2 of questions: what does B's start follow? ------------------------------------------------
    {
        hipStream_t st2;
        CK(hipStreamCreate(&st2));
        for (unsigned dur : {1500u, 3000u, 6000u}) {
            for (int mode = 0; mode < 3; ++mode) {   // 0 any-order same stream, 1 other stream, 2 ordered
                hipLaunchKernelGGL(k_reset, dim3(1), dim3(1), 0, st, d);
                CK(hipStreamSynchronize(st));
                timespec ta, tb, tc;
                clock_gettime(CLOCK_MONOTONIC, &ta);
                hipLaunchKernelGGL(k_a<16>, dim3(cus), dim3(256), 0, st, d, dur, (unsigned*)nullptr, 0u, 0u);
                clock_gettime(CLOCK_MONOTONIC, &tb);
                hipExtLaunchKernelGGL(k_b, dim3(3072), dim3(128), 0, mode == 1 ? st2 : st, nullptr, nullptr,
                                      mode == 0 ? hipExtAnyOrderLaunch : 0, d, (const unsigned*)nullptr, 0u, 0u, 0u);
                clock_gettime(CLOCK_MONOTONIC, &tc);
                CK(hipStreamSynchronize(st));
                CK(hipStreamSynchronize(st2));
                CK(hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost));
                char what[128];
                snprintf(what, sizeof what, "5 A(%u us) -> B %s [host: A %.1f us, B %.1f us]", dur / 100,
                         mode == 0 ? "any-order" : mode == 1 ? "other stream" : "ordered",
                         (tb.tv_sec - ta.tv_sec) * 1e6 + (tb.tv_nsec - ta.tv_nsec) * 1e-3,
                         (tc.tv_sec - tb.tv_sec) * 1e6 + (tc.tv_nsec - tb.tv_nsec) * 1e-3);
                report(what, h);
            }
        }
        // staggered A: block i ends at 10 us + 0.1 us * i (256 blocks: 10 ... 35.5 us)
        for (int mode = 0; mode < 2; ++mode) {
            hipLaunchKernelGGL(k_reset, dim3(1), dim3(1), 0, st, d);
            CK(hipStreamSynchronize(st));
            CK(hipMemset(&d->b_seen_last, 0xff, 8));
            hipLaunchKernelGGL(k_a<16>, dim3(cus), dim3(256), 0, st, d, 1000u, (unsigned*)nullptr, 0u, 10u);
            hipExtLaunchKernelGGL(k_b, dim3(3072), dim3(128), 0, mode == 1 ? st2 : st, nullptr, nullptr,
                                  mode == 0 ? hipExtAnyOrderLaunch : 0, d, (const unsigned*)nullptr, 0u, 0u, 0u);
            CK(hipStreamSynchronize(st));
            CK(hipStreamSynchronize(st2));
            CK(hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost));
            report(mode == 0 ? "6 A staggered 10..35.5us -> B any-order ('saw flags' = first A end)" : "6 A staggered -> B other stream", h);
        }
        // A with memory traffic instead of sleeping?  (the product's predecessor streams 52 MB)
    }
    // wall time of the pair, back to back, 200 times: ordered against any-order (events around the loop)
    for (int any = 0; any < 2; ++any) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        for (int w = 0; w < 2; ++w) {
            if (w == 1) CK(hipEventRecord(e0, st));
            for (int i = 0; i < 200; ++i) {
                const unsigned tag = 5000u + (unsigned)(any * 1000 + w * 300 + i);
                hipLaunchKernelGGL(k_a<16384>, dim3(768), dim3(256), 0, st, d, 800u, flags, tag);
                hipExtLaunchKernelGGL(k_b, dim3(3072), dim3(128), 0, st, nullptr, nullptr, any ? hipExtAnyOrderLaunch : 0, d,
                                      (const unsigned*)flags, 768u, tag, 200000u);
            }
        }
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("pair A(8us)+B, 200 back to back, %s: %.2f us per pair\n", any ? "any-order" : "ordered  ", ms * 1000.0 / 200);
    }
    CK(hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost));
    printf("timeouts over the loops: %llu\n", h.b_timeouts);
    return 0;
}
