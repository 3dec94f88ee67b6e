// tools/c_caller_probe.cpp -- round 6: the few-ROI forward called from C (no Python, no ctypes): is the ~4 us "launch floor" of
// the Python-side loops (R = 1: 3.8-4.0 us per call between events) the device's launch-to-launch rate or the caller's enqueue rate?
// Build: hipcc --offload-arch=gfx950 -O2 -std=c++17 -Iinclude -Wno-unused-result -o tools/c_caller_probe tools/c_caller_probe.cpp -ldl
// Run (from the repo root, on the GPU box): ./tools/c_caller_probe
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "rroi_align_hip.h"

__global__ void empty_kernel() {}
struct Fat { unsigned a[6]; };
__global__ void args_kernel(const float* p0, const float* p1, float* p2, int a0, int a1, int a2, int a3, int a4, int a5, float f0, int a6,
                            int a7, int a8, int a9, int a10, int a11, int a12, float* p3, float* p4, int a13) {}
__global__ void struct_kernel(const float* p0, const void* p1, float* p2, int a0, int a1, int a2, int a3, int a4, int a5, int a6, int a7, int a8,
                              Fat s0, Fat s1, Fat s2, int a9, Fat s3, Fat s4) {}

int main()
{
    void* lib = dlopen("fots.pytorch_amd/rroi_align/_ext/rroi_align/librroi_align_hip.so", RTLD_NOW);
    if (!lib) { fprintf(stderr, "%s\n", dlerror()); return 1; }
    auto fwd = reinterpret_cast<decltype(&rroi_align_forward_hip)>(dlsym(lib, "rroi_align_forward_hip"));
    auto wsb = reinterpret_cast<decltype(&rroi_align_forward_workspace_bytes)>(dlsym(lib, "rroi_align_forward_workspace_bytes"));
    if (!fwd || !wsb) { fprintf(stderr, "symbols missing\n"); return 1; }
    const int B = 2, C = 64, H = 120, W = 160, PH = 11, PW = 96;
    std::mt19937 rng(7);
    std::normal_distribution<float> nd;
    std::vector<float> hf((size_t)B * C * H * W);
    for (auto& v : hf) v = nd(rng);
    float* feats; hipMalloc(&feats, hf.size() * 4); hipMemcpy(feats, hf.data(), hf.size() * 4, hipMemcpyHostToDevice);
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto loop = [&](auto&& call, int warm, int n, const char* what) {
        for (int i = 0; i < warm; ++i) call();
        hipStreamSynchronize(st);
        auto t0 = std::chrono::steady_clock::now();
        hipEventRecord(e0, st);
        for (int i = 0; i < n; ++i) call();
        hipEventRecord(e1, st);
        auto t1 = std::chrono::steady_clock::now();
        hipStreamSynchronize(st);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipStreamSynchronize(st);
        auto b0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 200; ++i) call();      // a burst into an EMPTY queue: the caller's own cost per call
        auto b1 = std::chrono::steady_clock::now();
        hipStreamSynchronize(st);
        printf("%-46s %6.2f us per call between events, host enqueue %5.2f us per call (burst of 200 into an empty queue: %5.2f)\n", what, ms / n * 1e3,
               std::chrono::duration<double, std::micro>(t1 - t0).count() / n, std::chrono::duration<double, std::micro>(b1 - b0).count() / 200);
    };
    loop([&] { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st); }, 500, 5000, "empty kernel, 1 workgroup");
    loop([&] { hipLaunchKernelGGL(empty_kernel, dim3(2304), dim3(256), 0, st); }, 500, 5000, "empty kernel, 2304 workgroups of 256");
    loop([&] { hipLaunchKernelGGL(args_kernel, dim3(576), dim3(256), 0, st, feats, feats, feats, 1, 2, 3, 4, 5, 6, 0.25f, 7, 8, 9, 10, 11, 12, 13, (float*)nullptr, (float*)nullptr, 0); },
         500, 5000, "empty kernel with 20 scalar arguments");
    { Fat f{}; loop([&] { hipLaunchKernelGGL(struct_kernel, dim3(576), dim3(128), 0, st, feats, (const void*)feats, feats, 1, 2, 3, 4, 5, 6, 7, 8, 9, f, f, f, 10, f, f); },
         500, 5000, "empty kernel with 18 arguments, 5 structs"); }
    { int dev; auto t0 = std::chrono::steady_clock::now(); for (int i = 0; i < 100000; ++i) { hipGetDevice(&dev); (void)hipGetLastError(); }
      printf("hipGetDevice + hipGetLastError: %.3f us per pair\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 1e5); }
    { auto t0 = std::chrono::steady_clock::now(); int ok = 0; for (int i = 0; i < 100000; ++i) ok += fwd(feats, RROI_LAYOUT_NCHW, 0.25f, B, 0, H, W, C, PH, PW, feats, feats, nullptr, 0, RROI_PATH_AUTO, st);
      printf("forward with num_rois = 0 (validation only): %.3f us per call (%d)\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 1e5, ok); }
    {   // fork / join between two streams around one empty kernel each: what an in-call second stream would cost per call
        hipStream_t s2; (void)hipStreamCreate(&s2);
        hipEvent_t f, j; (void)hipEventCreateWithFlags(&f, hipEventDisableTiming); (void)hipEventCreateWithFlags(&j, hipEventDisableTiming);
        loop([&] { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st); hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st); },
             500, 5000, "two empty kernels on one stream");
        loop([&] { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st);
                   (void)hipEventRecord(f, st); (void)hipStreamWaitEvent(s2, f, 0);
                   hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s2);
                   (void)hipEventRecord(j, s2); (void)hipStreamWaitEvent(st, j, 0); },
             500, 5000, "the same with the second one forked to / joined from another stream");
    }
    for (int R : {1, 8, 16, 32, 64}) {
        std::uniform_real_distribution<float> u(0.f, 1.f);
        std::vector<float> hr((size_t)R * 6);
        for (int n = 0; n < R; ++n) {
            const float h = 16 + 48 * u(rng);
            hr[n * 6 + 0] = (float)(rng() % B); hr[n * 6 + 1] = 4 * W * u(rng); hr[n * 6 + 2] = 4 * H * u(rng);
            hr[n * 6 + 3] = h; hr[n * 6 + 4] = h * (2 + (PW / 11.0f - 2) * u(rng)); hr[n * 6 + 5] = -45 + 90 * u(rng);
        }
        float *rois, *out; void* ws;
        hipMalloc(&rois, hr.size() * 4); hipMemcpy(rois, hr.data(), hr.size() * 4, hipMemcpyHostToDevice);
        hipMalloc(&out, (size_t)R * C * PH * PW * 4);
        const size_t nb = wsb(B, C, H, W, R, RROI_LAYOUT_NCHW);
        hipMalloc(&ws, nb ? nb : 1);
        char what[96];
        for (int path : {RROI_PATH_AUTO, RROI_PATH_TILED}) {
            snprintf(what, sizeof what, "forward R = %2d, C = 64, 11 x 96, %s", R, path == RROI_PATH_AUTO ? "AUTO" : "two-launch path");
            loop([&] { if (fwd(feats, RROI_LAYOUT_NCHW, 0.25f, B, R, H, W, C, PH, PW, rois, out, ws, nb, path, st) != 1) abort(); }, 300, 3000, what);
        }
        hipFree(rois); hipFree(out); hipFree(ws);
    }
    return 0;
}
