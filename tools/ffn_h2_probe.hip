// Time budget of the split-f16 feed-forward kernel (not product code):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-inline-asm -I interdiff_amd/csrc tools/ffn_h2_probe.hip -o build_tools/ffn_h2_probe
// csrc/ffn_h2.h at M rows in its product form and its ablations (no MFMAs / no DMA after the prologue / no slab stores), launches that walk
// through 8 weight streams like a denoiser step, and the per-workgroup phase stamps (shader clock) of the stamped build.
#include "ffn_h2.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

bool g_idf_prof_on = false;
void idf_prof_mark_slow(int, hipStream_t) {}
using namespace idf_ffn_h2;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int TT, int S, int MODE, int LW = 0>
float run(const float *x2, int M, const float *pack, const float *b1, const float *b2, float *parts, int reps, int layers) {
    constexpr int BM = 16 * TT;
    const dim3 grid((unsigned)(idf_cdiv(M, BM) * NSL));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&ffn_h2_kernel<TT, S, MODE, LW>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_REQUEST));
    auto go = [&](int i) { hipLaunchKernelGGL((ffn_h2_kernel<TT, S, MODE, LW>), grid, dim3(NT + 64 * LW), LDS_REQUEST, 0, x2, M, (int)grid.x, pack + (size_t)(i % layers) * NSL * SLICE_FLOATS, b1, b2, parts, 0); };
    for (int i = 0; i < 10; ++i) go(i);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) go(i);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return 1e3f * ms / reps;
}

template <int TT, int S>
void probe(int M) {
    constexpr int BM = 16 * TT;
    const int nwg = (int)idf_cdiv(M, BM) * NSL, layers = 8;
    std::vector<float> h((size_t)layers * NSL * SLICE_FLOATS + (size_t)M * D + 2048);
    srand(1);
    for (auto &v : h) v = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;                 // (weights as random fp32 words: as halves they are arbitrary finite-or-not values; timing only)
    for (size_t i = 0; i < (size_t)layers * NSL * SLICE_FLOATS; ++i) { uint32_t u = 0x2c002c00u + (uint32_t)(rand() & 0x03ff03ff); std::memcpy(&h[i], &u, 4); }      // halves of magnitude ~0.06
    float *d, *parts;
    CK(hipMalloc(&d, h.size() * 4));
    CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&parts, (size_t)NSL * M * D * 4 + (size_t)nwg * 32 * 8));
    const float *pack = d, *x2 = d + (size_t)layers * NSL * SLICE_FLOATS, *b1 = x2 + (size_t)M * D, *b2 = b1 + 1200;
    printf("rows/tile %d, M=%d, %d workgroups\n", BM, M, nwg);
    printf("  %-40s %8.2f us\n", "product, 8 weight streams", run<TT, S, 0>(x2, M, pack, b1, b2, parts, 400, layers));
    printf("  %-40s %8.2f us\n", "product, ONE weight stream (L2-warm)", run<TT, S, 0>(x2, M, pack, b1, b2, parts, 400, 1));
    printf("  %-40s %8.2f us\n", "no MFMA", run<TT, S, 1>(x2, M, pack, b1, b2, parts, 400, layers));
    printf("  %-40s %8.2f us\n", "no DMA after the prologue", run<TT, S, 2>(x2, M, pack, b1, b2, parts, 400, layers));
    printf("  %-40s %8.2f us\n", "no slab stores", run<TT, S, 4>(x2, M, pack, b1, b2, parts, 400, layers));
    printf("  %-40s %8.2f us\n", "product again", run<TT, S, 0>(x2, M, pack, b1, b2, parts, 400, layers));
    {   // eight loader waves (16-wave workgroup): bits against the 8-wave kernel, time, stamps
        std::vector<float> a((size_t)NSL * M * D), b(a.size());
        CK(hipMemset(parts, 0xff, a.size() * 4));
        run<TT, S, 0>(x2, M, pack, b1, b2, parts, 1, 1);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(a.data(), parts, a.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemset(parts, 0xff, a.size() * 4));
        run<TT, S, 0, 8>(x2, M, pack, b1, b2, parts, 1, 1);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(b.data(), parts, b.size() * 4, hipMemcpyDeviceToHost));
        size_t diff = 0;
        for (size_t i = 0; i < a.size(); ++i) diff += std::memcmp(&a[i], &b[i], 4) ? 1 : 0;
        printf("  loader-wave form vs 8-wave form: %zu of %zu elements differ\n", diff, a.size());
        for (int rep = 0; rep < 3; ++rep) {
            printf("  %-40s %8.2f us\n", "8 waves, 8 weight streams", run<TT, S, 0>(x2, M, pack, b1, b2, parts, 400, layers));
            printf("  %-40s %8.2f us\n", "8 + 8 loader waves, 8 weight streams", run<TT, S, 0, 8>(x2, M, pack, b1, b2, parts, 400, layers));
        }
        run<TT, S, 3, 8>(x2, M, pack, b1, b2, parts, 20, layers);
        std::vector<long long> st8((size_t)nwg * 32);
        CK(hipMemcpy(st8.data(), reinterpret_cast<char *>(parts) + (size_t)NSL * M * D * 4, st8.size() * 8, hipMemcpyDeviceToHost));
        double acc8[32] = {0};
        for (int w = 0; w < nwg; ++w)
            for (int i = 1; i < 20; ++i) acc8[i] += (double)(st8[(size_t)w * 32 + i] - st8[(size_t)w * 32 + i - 1]);
        printf("  stamped run, loader-wave form:\n   ");
        for (int i = 1; i < 20; ++i) printf(" %d:%.0f", i, acc8[i] / nwg);
        printf("\n");
    }
    run<TT, S, 3>(x2, M, pack, b1, b2, parts, 20, layers);
    std::vector<long long> st((size_t)nwg * 32);
    CK(hipMemcpy(st.data(), reinterpret_cast<char *>(parts) + (size_t)NSL * M * D * 4, st.size() * 8, hipMemcpyDeviceToHost));
    // stamps: 0 entry, 1 rows split, 2..9 phase-1 steps published, 10 phase 1 done, 11 gelu done, 12..17 phase-2 steps, 18 staged, 19 stores issued
    long long t0 = st[0], t1 = 0;
    for (int w = 0; w < nwg; ++w) t0 = std::min(t0, st[(size_t)w * 32]);
    double acc[32] = {0};
    for (int w = 0; w < nwg; ++w) {
        for (int i = 1; i < 20; ++i) acc[i] += (double)(st[(size_t)w * 32 + i] - st[(size_t)w * 32 + i - 1]);
        t1 = std::max(t1, st[(size_t)w * 32 + 19]);
    }
    printf("  stamped run: first entry -> last exit %lld ticks; mean ticks per phase over workgroups (100 MHz ticks x ~21 = shader cycles):\n   ", t1 - t0);
    for (int i = 1; i < 20; ++i) printf(" %d:%.0f", i, acc[i] / nwg);
    printf("\n");
    CK(hipFree(d)); CK(hipFree(parts));
}

int main(int argc, char **argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 1600;
    probe<2, 4>(M);
    probe<1, 4>(M / 2);
    probe<4, 2>(2 * M);
    return 0;
}
