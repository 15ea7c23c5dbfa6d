// GEMM tile-configuration probe (not product code):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DIDF_GEMM_PROBE -I interdiff_amd/csrc tools/gemm_probe.hip -o build_tools/gemm_probe
// Times every tile configuration of gemm.h on the denoiser's shapes (M = 1600) with back-to-back launches and
// prints the per-workgroup phase breakdown (prologue / k-loop / epilogue, in shader-clock ticks) from the probe stamps.
#include "gemm.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <vector>

bool g_idf_prof_on = false;
void idf_prof_mark_slow(int, hipStream_t) {}
using namespace idf_gemm;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

std::vector<float> hA;
struct Shape { const char *name; int M, N, K, epi; };

template <int BM, int BN, int WM, int WN, int KS, int KC, int EPI, int NS>
void launch_any(const Args &g) {
    if constexpr (KS == 0) launch<BM, BN, WM, WN, KC, A_PLAIN, EPI>(0, g);
    else launch_glds<BM, BN, WM, WN, KS, KC, A_PLAIN, EPI, NS>(0, g);
}

template <int BM, int BN, int WM, int WN, int KS, int KC, int EPI, int NS>
void run(const char *cfg, const Shape &sh, Args g, long long *probe_d) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    g.probe = nullptr;
    for (int i = 0; i < 20; ++i) launch_any<BM, BN, WM, WN, KS, KC, EPI, NS>(g);
    CK(hipDeviceSynchronize());
    const int n = 200;
    CK(hipEventRecord(e0));
    for (int i = 0; i < n; ++i) launch_any<BM, BN, WM, WN, KS, KC, EPI, NS>(g);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const int nwg = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
    g.probe = probe_d;
    launch_any<BM, BN, WM, WN, KS, KC, EPI, NS>(g);
    CK(hipDeviceSynchronize());
    std::vector<long long> p(nwg * 4);
    CK(hipMemcpy(p.data(), probe_d, p.size() * 8, hipMemcpyDeviceToHost));
    double pro = 0, loop = 0, epi = 0, mx = 0;
    long long t0 = p[0], t1 = p[3];
    for (int w = 0; w < nwg; ++w) {
        pro += p[w * 4 + 1] - p[w * 4]; loop += p[w * 4 + 2] - p[w * 4 + 1]; epi += p[w * 4 + 3] - p[w * 4 + 2];
        mx = std::max(mx, (double)(p[w * 4 + 3] - p[w * 4]));
        t0 = std::min(t0, p[w * 4]); t1 = std::max(t1, p[w * 4 + 3]);
    }
    // spot check against an fp64 host dot product (transpose-detecting: A, W are not symmetric)
    extern std::vector<float> hA;
    std::vector<float> hc((size_t)g.M * g.N);
    CK(hipMemcpy(hc.data(), g.C, hc.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int smp = 0; smp < 400; ++smp) {
        const int r = (smp * 7919 + 13) % g.M, c = (smp * 104729 + 7) % g.N;
        double d = hA[c % 1024];                                    // bias
        for (int k = 0; k < g.K; ++k) d += (double)hA[(size_t)r * g.K + k] * hA[(size_t)c * g.K + k];
        if (EPI == E_GELU) d = 0.5 * d * (1.0 + erf(d / sqrt(2.0)));
        else d += hA[(size_t)r * g.N + c];
        worst = std::max(worst, fabs(d - hc[(size_t)r * g.N + c]));
    }
    const double fl = 2.0 * g.M * g.N * g.K;
    printf("%-6s %-22s %7.2f us  %6.1f TF/s | wgs %4d  ticks/wg: prologue %6.0f  loop %6.0f  epilogue %6.0f  max-wg %6.0f  err %.1e\n", sh.name, cfg,
           1e3 * ms / n, fl / (1e-3 * ms / n) / 1e12, nwg, pro / nwg, loop / nwg, epi / nwg, mx, worst);
}

int main() {
    const int M = 1600;
    float *A, *W, *bias, *C, *R; long long *probe;
    CK(hipMalloc(&A, (size_t)M * 1024 * 4)); CK(hipMalloc(&W, (size_t)1024 * 1024 * 4)); CK(hipMalloc(&bias, 1024 * 4));
    CK(hipMalloc(&C, (size_t)M * 1024 * 4)); CK(hipMalloc(&R, (size_t)M * 1024 * 4)); CK(hipMalloc(&probe, 8 * 4 * 8192));
    hA.resize((size_t)M * 1024);
    std::vector<float> &h = hA;
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.0f - 0.5f;
    CK(hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(R, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(W, h.data(), (size_t)1024 * 1024 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(bias, h.data(), 4096, hipMemcpyHostToDevice));
    const Shape shapes[] = {{"ffn1", M, 1024, 256, E_GELU}, {"ffn2", M, 256, 1024, E_RESID}, {"oproj", M, 256, 256, E_RESID}};
    for (const Shape &sh : shapes) {
        Args g{};
        g.A = A; g.lda = sh.K; g.K = sh.K; g.W = W; g.bias = bias; g.C = C; g.ldc = sh.N; g.M = sh.M; g.N = sh.N; g.resid = R; g.T = 100;
#define RUN(BM, BN, WM, WN, KS, KC, NS)                                                                     \
    if (sh.epi == E_GELU) run<BM, BN, WM, WN, KS, KC, E_GELU, NS>(#BM "x" #BN " ks" #KS " kc" #KC " ns" #NS, sh, g, probe); \
    else run<BM, BN, WM, WN, KS, KC, E_RESID, NS>(#BM "x" #BN " ks" #KS " kc" #KC " ns" #NS, sh, g, probe);
        // KS = 0: register-staged kernel; KS >= 1: LDS-DMA pipeline with KS k-slices per workgroup, NS stages
        RUN(32, 32, 2, 2, 2, 64, 3)
        RUN(32, 32, 2, 2, 4, 64, 3)
        RUN(32, 32, 2, 2, 4, 64, 4)
        RUN(64, 32, 2, 2, 1, 32, 3)
        RUN(64, 32, 2, 2, 2, 64, 3)
        RUN(64, 64, 2, 2, 4, 64, 3)
        RUN(64, 64, 2, 2, 2, 32, 3)
    }
    return 0;
}
