// Phase stamps of the SMPL blend + skin kernel (not product code):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-inline-asm -I interdiff_amd/csrc tools/smpl_probe.hip -o build_tools/smpl_probe
// Builds csrc/smpl.hip as one translation unit with IDF_SMPL_STAMP defined; random model / inputs (timing only), N frames.
#include <hip/hip_runtime.h>
__device__ long long g_stamps[32768 * 16];
#define IDF_SMPL_STAMP(i) do { if (threadIdx.x == 0) g_stamps[blockIdx.x * 16 + (i)] = clock64(); } while (0)
#include "smpl.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
bool g_idf_prof_on = false;
void idf_prof_mark_slow(int, hipStream_t) {}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main(int argc, char **argv) {
    const int64_t N = argc > 1 ? atoi(argv[1]) : 1600;
    const int V = 6890, J = 52, nb = 10, KB = 480, S = 4;
    std::vector<float> h((size_t)3 * V * KB + 4096);
    srand(1);
    for (auto &v : h) v = (rand() / (float)RAND_MAX - 0.5f) * 0.01f;
    float *blend, *jt, *js, *sw, *pose, *betas, *trans, *verts, *jtr;
    int32_t *parents, *sidx;
    CK(hipMalloc(&blend, (size_t)((V + 63) / 64) * 192 * KB * 4)); CK(hipMemcpy(blend, h.data(), (size_t)3 * V * KB * 4, hipMemcpyHostToDevice));      // fragment-order basis: [ceil(V/64)] tiles of 192 rows (random content: timing only)
    CK(hipMalloc(&jt, J * 3 * 4)); CK(hipMalloc(&js, J * 3 * nb * 4)); CK(hipMalloc(&sw, (size_t)V * S * 4)); CK(hipMalloc(&sidx, (size_t)V * S * 4));
    CK(hipMalloc(&parents, J * 4));
    CK(hipMemcpy(jt, h.data(), J * 3 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(js, h.data(), J * 3 * nb * 4, hipMemcpyHostToDevice));
    std::vector<int32_t> par(J), si((size_t)V * S);
    for (int j = 0; j < J; ++j) par[j] = j ? (j - 1) / 2 : 0;
    // argv[2] = "coherent": the bones of a vertex follow its id (blocks of ~133 ids share a bone and its three successors), the way
    // neighbouring SMPL-H vertex ids do; default: four unrelated bones per vertex (worst case for the skinning phase's LDS reads)
    const bool coherent = argc > 2 && argv[2][0] == 'c';
    for (size_t i = 0; i < si.size(); ++i) si[i] = coherent ? (int32_t)((i / S) * J / V + i % S) % J : rand() % J;
    std::vector<float> w((size_t)V * S, 0.25f);
    CK(hipMemcpy(parents, par.data(), J * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(sidx, si.data(), si.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(sw, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&pose, N * 156 * 4)); CK(hipMalloc(&betas, N * 10 * 4)); CK(hipMalloc(&trans, N * 3 * 4));
    CK(hipMemcpy(pose, h.data(), N * 156 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(betas, h.data(), N * 10 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(trans, h.data(), N * 3 * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&verts, (size_t)N * V * 3 * 4)); CK(hipMalloc(&jtr, N * J * 3 * 4));
    idf_smpl_model m{V, J, nb, KB, S, blend, jt, js, parents, sidx, sw};
    const size_t wsb = interdiff_smpl_workspace_bytes(&m, N);
    void *ws; CK(hipMalloc(&ws, wsb));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) if (interdiff_smpl_forward(&m, pose, betas, trans, N, verts, jtr, nullptr, ws, wsb, nullptr)) { printf("forward failed\n"); return 1; }
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) interdiff_smpl_forward(&m, pose, betas, trans, N, verts, jtr, nullptr, ws, wsb, nullptr);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("N=%lld: %.1f us per SMPL forward (pose + blend/skin kernels)%s\n", (long long)N, 1e3 * ms / 10, coherent ? ", bones coherent with vertex ids" : "");
    const int nwg = (int)(((N + FT - 1) / FT + TBF - 1) / TBF * (((V + VT - 1) / VT + TBV - 1) / TBV) * TBF * TBV);
    std::vector<long long> st((size_t)nwg * 16);
    CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * 8));
    const int ns = 3 + 3 * (FT / SF);
    double acc[16] = {0};
    int live = 0;
    for (int wgi = 0; wgi < nwg; ++wgi) {
        if (!st[(size_t)wgi * 16 + ns - 1]) continue;                 // edge workgroups that exited at once
        ++live;
        for (int i = 1; i < ns; ++i) acc[i] += (double)(st[(size_t)wgi * 16 + i] - st[(size_t)wgi * 16 + i - 1]);
    }
    const char *nm[16] = {"", "feature tile -> LDS", "blend-shape GEMM (k-loop)", "stage + joint transforms staged (sub 0)", "skinning (sub 0)", "stores (sub 0)",
                          "joint transforms staged (sub 1)", "skinning (sub 1)", "stores (sub 1)", "sub2 stage", "sub2 skin", "sub2 store", "sub3 stage", "sub3 skin", "sub3 store", ""};
    double tot = 0;
    for (int i = 1; i < ns; ++i) { printf("  %-42s %9.0f\n", nm[i], acc[i] / live); tot += acc[i] / live; }
    printf("  total per workgroup %.0f cycles, %d workgroups\n", tot, live);
    return 0;
}
