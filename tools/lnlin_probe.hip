// Correctness probe of the LayerNorm+linear (QKV) kernel of csrc/ffn.h (not product code):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-inline-asm -I interdiff_amd/csrc tools/lnlin_probe.hip -o build_tools/lnlin_probe
// One-hot weights (W[n][k] = [k == k0]) turn the output into a copy of column k0 of the input rows, so a wrong fragment address
// shows up as "column k' instead of k0".
#include "ffn.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

bool g_idf_prof_on = false;
void idf_prof_mark_slow(int, hipStream_t) {}
using namespace idf_ffn;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

static const int G4[4] = {0, 3, 2, 1};
// mdm.py pack_linear160
static void pack160(const std::vector<float> &w, int N, std::vector<float> &out) {
    const int ns = (N + LHS - 1) / LHS;
    out.assign((size_t)ns * LHS * D, 0.f);
    size_t o = 0;
    for (int n0 = 0; n0 < ns * LHS; n0 += LHS)
        for (int g = 0; g < 16; ++g)
            for (int r = 0; r < LHS; ++r)
                for (int p = 0; p < 4; ++p) {
                    const int cell = p ^ G4[(r >> 2) & 3];
                    for (int e = 0; e < 4; ++e) out[o++] = n0 + r < N ? w[(size_t)(n0 + r) * D + 16 * g + cell * 4 + e] : 0.f;
                }
}

int main(int argc, char **argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 64;
    const int N = 768;
    std::vector<float> A((size_t)M * D), W((size_t)N * D), bias(N, 0.f), P, Cg((size_t)M * N), xn((size_t)M * D);
    srand(3);
    for (auto &v : A) v = rand() / (float)RAND_MAX - 0.5f;
    float *dA, *dP, *dB, *dC, *dX;
    CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dP, (size_t)((N + LHS - 1) / LHS) * LHS * D * 4)); CK(hipMalloc(&dB, N * 4)); CK(hipMalloc(&dC, Cg.size() * 4));
    CK(hipMalloc(&dX, xn.size() * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, bias.data(), N * 4, hipMemcpyHostToDevice));
    for (int k0 : {0, 1, 4, 5, 16, 21, 37, 100, 255}) {
        for (size_t i = 0; i < W.size(); ++i) W[i] = ((int)(i % D) == k0) ? 1.f : 0.f;
        pack160(W, N, P);
        CK(hipMemcpy(dP, P.data(), P.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemset(dC, 0xff, Cg.size() * 4));
        launch_ln_linear<1>(0, dA, 0, nullptr, nullptr, M, N, dP, dB, dC, N, dX);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(Cg.data(), dC, Cg.size() * 4, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int m = 0; m < M && bad < 6; ++m)
            for (int n = 0; n < N && bad < 6; ++n) {
                const float got = Cg[(size_t)m * N + n], want = A[(size_t)m * D + k0];
                if (got != want) {
                    int src = -1, srow = -1;
                    for (int mm = 0; mm < M && src < 0; ++mm)
                        for (int k = 0; k < D; ++k) if (A[(size_t)mm * D + k] == got) { src = k; srow = mm; break; }
                    printf("k0=%d: C[%d][%d] = %g, want %g (that value is A[%d][%d])\n", k0, m, n, got, want, srow, src);
                    ++bad;
                }
            }
        long nb = 0;
        for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) nb += Cg[(size_t)m * N + n] != A[(size_t)m * D + k0];
        printf("k0=%d: %ld of %d outputs wrong\n", k0, nb, M * N);
    }
    // random weights
    for (auto &v : W) v = rand() / (float)RAND_MAX - 0.5f;
    pack160(W, N, P);
    CK(hipMemcpy(dP, P.data(), P.size() * 4, hipMemcpyHostToDevice));
    launch_ln_linear<1>(0, dA, 0, nullptr, nullptr, M, N, dP, dB, dC, N, dX);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(Cg.data(), dC, Cg.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(xn.data(), dX, xn.size() * 4, hipMemcpyDeviceToHost));
    double e = 0, ex = 0;
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            double s = 0;
            for (int k = 0; k < D; ++k) s += (double)A[(size_t)m * D + k] * W[(size_t)n * D + k];
            e = std::max(e, std::fabs(s - Cg[(size_t)m * N + n]));
        }
    for (size_t i = 0; i < xn.size(); ++i) ex = std::max(ex, (double)std::fabs(xn[i] - A[i]));
    printf("random W: max abs err %g, xn err %g\n", e, ex);
    return 0;
}
