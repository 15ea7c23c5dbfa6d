// Time budget of the fused feed-forward kernel (not product code):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-inline-asm -I interdiff_amd/csrc tools/ffn_probe.hip -o build_tools/ffn_probe
// Runs csrc/ffn.h at M rows in its product form and in four ablations (no MFMAs / no DMA after the prologue / no LDS fragment
// reads in the loops) with back-to-back launches, and prints the per-workgroup phase stamps (shader clock) of the stamped build.
// Weights are random (the stream layout does not matter for timing).
#include "ffn.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

bool g_idf_prof_on = false;
void idf_prof_mark_slow(int, hipStream_t) {}
using namespace idf_ffn;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE>
float run(const float *x2, int M, const float *pack, const float *b1, const float *b2, float *parts, int reps) {
    const dim3 grid((unsigned)(idf_cdiv(M, BM) * NSL));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(ffn_fused_kernel<MODE>, grid, dim3(NT), 0, 0, x2, M, pack, b1, b2, parts);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(ffn_fused_kernel<MODE>, grid, dim3(NT), 0, 0, x2, M, pack, b1, b2, parts);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return 1e3f * ms / reps;
}

int main(int argc, char **argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 1600;
    const int nwg = (int)idf_cdiv(M, BM) * NSL;
    std::vector<float> h((size_t)2 * D * FF + (size_t)M * D + FF + D + 1024);
    srand(1);
    for (auto &v : h) v = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
    float *d, *parts;
    CK(hipMalloc(&d, h.size() * 4));
    CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&parts, (size_t)NSL * M * D * 4 + (size_t)nwg * 32 * 8));
    const float *pack = d, *x2 = d + (size_t)2 * D * FF, *b1 = x2 + (size_t)M * D, *b2 = b1 + FF;
    const double fl = 2.0 * 2.0 * M * D * FF;
    const char *names[6] = {"product", "no MFMA", "no DMA after prologue", "stamped", "no LDS fragment reads in the loops", "no barrier in the loops (wrong results)"};
    float us[6];
    us[0] = run<0>(x2, M, pack, b1, b2, parts, 200);
    us[1] = run<1>(x2, M, pack, b1, b2, parts, 200);
    us[2] = run<2>(x2, M, pack, b1, b2, parts, 200);
    us[4] = run<4>(x2, M, pack, b1, b2, parts, 200);
    us[5] = run<5>(x2, M, pack, b1, b2, parts, 200);
    us[3] = run<3>(x2, M, pack, b1, b2, parts, 50);
    for (int i = 0; i < 6; ++i) printf("M=%d  %-36s %8.2f us  (%.1f TFLOP/s equivalent)\n", M, names[i], us[i], fl / us[i] / 1e6);
    printf("M=%d  %-36s %8.2f us\n", M, "no slab stores", run<6>(x2, M, pack, b1, b2, parts, 200));
    printf("M=%d  %-36s %8.2f us\n", M, "plain (write-back) slab stores", run<7>(x2, M, pack, b1, b2, parts, 200));
    printf("M=%d  %-36s %8.2f us\n", M, "product again", run<0>(x2, M, pack, b1, b2, parts, 200));
    std::vector<long long> st((size_t)nwg * 32);
    CK(hipMemcpy(st.data(), reinterpret_cast<char *>(parts) + (size_t)NSL * M * D * 4, st.size() * 8, hipMemcpyDeviceToHost));
    // stamps: 0 entry, 1 after prologue barrier, 2..9 phase-1 pairs, 10 after the gelu epilogue, 11.. phase-2 pairs, last = exit
    long long t0 = st[0], t1 = 0;
    for (int w = 0; w < nwg; ++w) { t0 = std::min(t0, st[(size_t)w * 32]); }
    double acc[32] = {0};
    int cnt[32] = {0};
    for (int w = 0; w < nwg; ++w) {
        const int ns = (w % NSL) < NSL - 1 ? 2 + 8 + 1 + 7 + 1 : 2 + 8 + 1 + 6 + 1;
        for (int i = 1; i < ns; ++i) { acc[i] += (double)(st[(size_t)w * 32 + i] - st[(size_t)w * 32 + i - 1]); cnt[i]++; }
        t1 = std::max(t1, st[(size_t)w * 32 + ns - 1]);
    }
    printf("stamped run: first entry -> last exit %lld ticks; mean ticks per phase over workgroups:\n", t1 - t0);
    for (int i = 1; i < 20; ++i) if (cnt[i]) printf("  phase %2d: %9.0f  (n=%d)\n", i, acc[i] / cnt[i], cnt[i]);
    for (int w : {0, 1, 4, 124, 249}) {
        if (w >= nwg) continue;
        printf("  wg %3d entry +%lld:", w, st[(size_t)w * 32] - t0);
        for (int i = 1; i < 20; ++i) printf(" %lld", st[(size_t)w * 32 + i] - st[(size_t)w * 32 + i - 1]);
        printf("\n");
    }
    return 0;
}
