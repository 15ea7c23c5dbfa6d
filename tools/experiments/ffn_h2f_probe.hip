// The chunk-pipelined split-f16 feed-forward kernel (csrc/ffn_h2f.h) against ffn_h2.h's kernel (not product code):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-inline-asm -mllvm -amdgpu-kernarg-preload-count=16 -I interdiff_amd/csrc -I tools/experiments tools/experiments/ffn_h2f_probe.hip -o build_tools/ffn_h2f_probe
// Same random weights in both stream orders (the new one is a permutation of the old one's 1-KiB fragments), bit-for-bit comparison of the five slabs,
// back-to-back launch times walking through 8 weight streams, and the half-tick stamps of the stamped build (P wave 0 / Q wave 4).
#include "ffn_h2f.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

bool g_idf_prof_on = false;
void idf_prof_mark_slow(int, hipStream_t) {}
using namespace idf_ffn_h2f;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

static void permute_slice(const char *old_s, char *new_s) {
    size_t o = 0;
    auto a = [&](int j) {
        const int c = j / 2, hh = j % 2;
        for (int s = 0; s < 4; ++s)
            for (int hp = 0; hp < (c < 6 ? 2 : 1); ++hp)
                for (int pl = 0; pl < 2; ++pl) { std::memcpy(new_s + o, old_s + (size_t)(4 * hh + s) * P1B + ((2 * c + hp) * 2 + pl) * 1024, 1024); o += 1024; }
    };
    auto b = [&](int j) {
        const int c = j / 2, hh = j % 2;
        for (int q = 0; q < 4; ++q)
            for (int i = 0; i < 2; ++i)
                for (int pl = 0; pl < 2; ++pl) { std::memcpy(new_s + o, old_s + (size_t)KS1 * P1B + (size_t)c * P2B + ((4 * q + 2 * hh + i) * 2 + pl) * 1024, 1024); o += 1024; }
    };
    for (int r = 0; r < NR; ++r) {
        bool done = false;
        for (int j = 0; j < 14 && !done; ++j) {
            if (a_ring(j) == r) { a(j); done = true; }
            else if (b_ring(j) == r) { b(j); done = true; }
        }
        if (!done) { printf("ring index %d unassigned\n", r); exit(1); }
    }
    if (o != (size_t)SLICE_BYTES) { printf("permutation size %zu\n", o); exit(1); }
}

template <class F>
float timeit(F go, int reps) {
    for (int i = 0; i < 10; ++i) go(i);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) go(i);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return 1e3f * ms / reps;
}

int main(int argc, char **argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 1600;
    const int layers = 8, nwg = (int)idf_cdiv(M, 32) * NSL;
    const size_t packf = (size_t)layers * NSL * SLICE_FLOATS;
    std::vector<float> hold(packf), hnew(packf), hx((size_t)M * D), hb(1040 + 256 + 256);
    srand(1);
    for (size_t i = 0; i < packf; ++i) { uint32_t u = 0x2c002c00u + (uint32_t)(rand() & 0x03ff03ff) + ((rand() & 1) ? 0x80000000u : 0u) + ((rand() & 1) ? 0x8000u : 0u); std::memcpy(&hold[i], &u, 4); }
    for (int l = 0; l < layers * NSL; ++l) permute_slice(reinterpret_cast<const char *>(hold.data() + (size_t)l * SLICE_FLOATS), reinterpret_cast<char *>(hnew.data() + (size_t)l * SLICE_FLOATS));
    for (auto &v : hx) v = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
    for (auto &v : hb) v = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
    float *dold, *dnew, *dx, *db, *p0, *p1;
    CK(hipMalloc(&dold, packf * 4)); CK(hipMalloc(&dnew, packf * 4)); CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&db, hb.size() * 4));
    const size_t pbytes = (size_t)NSL * M * D * 4 + (size_t)nwg * 96 * 8;
    CK(hipMalloc(&p0, pbytes)); CK(hipMalloc(&p1, pbytes));
    CK(hipMemcpy(dold, hold.data(), packf * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dnew, hnew.data(), packf * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    const float *b1 = db, *b2 = db + 1040 + 256;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&ffn_h2_kernel<2, 4, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_REQUEST));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&ffn_h2f_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_REQUEST));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&ffn_h2f_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_REQUEST));
    auto go_old = [&](int i) { hipLaunchKernelGGL((ffn_h2_kernel<2, 4, 0>), dim3(nwg), dim3(NT), LDS_REQUEST, 0, dx, M, nwg, dold + (size_t)(i % layers) * NSL * SLICE_FLOATS, b1, b2, p0, 0); };
    auto go_new = [&](int i) { hipLaunchKernelGGL((ffn_h2f_kernel<0>), dim3(nwg), dim3(FNT), LDS_REQUEST, 0, dx, M, nwg, dnew + (size_t)(i % layers) * NSL * SLICE_FLOATS, b1, b2, p1, 0); };
    // bit-for-bit, every layer's weights
    size_t bad = 0;
    std::vector<float> o0((size_t)NSL * M * D), o1(o0.size());
    for (int l = 0; l < layers; ++l) {
        CK(hipMemset(p0, 0xff, (size_t)NSL * M * D * 4)); CK(hipMemset(p1, 0xff, (size_t)NSL * M * D * 4));
        go_old(l); go_new(l);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(o0.data(), p0, o0.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(o1.data(), p1, o1.size() * 4, hipMemcpyDeviceToHost));
        size_t d = 0; double mx = 0;
        for (size_t i = 0; i < o0.size(); ++i) { if (std::memcmp(&o0[i], &o1[i], 4)) { if (!d) printf("  first difference: layer %d element %zu (slab %zu row %zu col %zu): %g vs %g\n", l, i, i / ((size_t)M * D), i / D % M, i % D, o0[i], o1[i]); ++d; } mx = std::max(mx, (double)std::fabs(o0[i])); }
        bad += d;
        printf("layer %d: %zu of %zu elements differ (max |out| %.3g)\n", l, d, o0.size(), mx);
    }
    printf("bit-for-bit: %s\n", bad ? "DIFFERENT" : "identical");
    for (int rep = 0; rep < 3; ++rep) {
        printf("  ffn_h2_kernel<32 rows, 4 slots>  %8.2f us\n", timeit(go_old, 400));
        printf("  ffn_h2f_kernel                   %8.2f us\n", timeit(go_new, 400));
    }
    auto go_st = [&](int i) { hipLaunchKernelGGL((ffn_h2f_kernel<3>), dim3(nwg), dim3(FNT), LDS_REQUEST, 0, dx, M, nwg, dnew + (size_t)(i % layers) * NSL * SLICE_FLOATS, b1, b2, p1, 0); };
    for (int i = 0; i < 20; ++i) go_st(i);
    CK(hipDeviceSynchronize());
    std::vector<long long> st((size_t)nwg * 96);
    CK(hipMemcpy(st.data(), reinterpret_cast<char *>(p1) + (size_t)NSL * M * D * 4, st.size() * 8, hipMemcpyDeviceToHost));
    // stamps: 0 entry, 1 rows split, 2..19 half-ticks 0..17 opened, 20 staged, 21 stores issued
    const char *role[3] = {"P (wave 0)", "Q (wave 8)", "L (wave 12)"};
    for (int ro = 0; ro < 3; ++ro) {
        double acc[32] = {0};
        for (int w = 0; w < nwg; ++w)
            for (int i = 1; i < 22; ++i) acc[i] += (double)(st[(size_t)w * 96 + ro * 32 + i] - st[(size_t)w * 96 + ro * 32 + i - 1]);
        double tot = 0, skew = 0, to_b0 = 0, to_end = 0;
        for (int w = 0; w < nwg; ++w) {
            skew += (double)(st[(size_t)w * 96 + ro * 32] - st[(size_t)w * 96]);
            to_b0 += (double)(st[(size_t)w * 96 + ro * 32 + 2] - st[(size_t)w * 96]);
            to_end += (double)(st[(size_t)w * 96 + ro * 32 + (ro == 1 ? 19 : 20)] - st[(size_t)w * 96]);
        }
        printf("  %s: entry %.0f cycles after wave 0's; barrier 0 passed at %.0f, last stamp at %.0f (from wave 0's entry)\n", role[ro], skew / nwg, to_b0 / nwg, to_end / nwg);
        printf("  %s, mean cycles per stamp interval over workgroups:\n   ", role[ro]);
        for (int i = 1; i < 22; ++i) { printf(" %d:%.0f", i, acc[i] / nwg); tot += acc[i] / nwg; }
        printf("\n    total %.0f\n", tot);
    }
    // MODE 4: per role (P wave 0 [even chunks], P wave 4 [odd chunks], Q wave 8, L wave 12), half-ticks 8..11: barrier passed -> work (L: issue) done -> next barrier passed ...
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&ffn_h2f_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_REQUEST));
    auto go_s4 = [&](int i) { hipLaunchKernelGGL((ffn_h2f_kernel<4>), dim3(nwg), dim3(FNT), LDS_REQUEST, 0, dx, M, nwg, dnew + (size_t)(i % layers) * NSL * SLICE_FLOATS, b1, b2, p1, 0); };
    for (int i = 0; i < 20; ++i) go_s4(i);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(st.data(), reinterpret_cast<char *>(p1) + (size_t)NSL * M * D * 4, st.size() * 8, hipMemcpyDeviceToHost));
    const char *r4[4] = {"P wave 0", "P wave 4", "Q wave 8", "L wave 12"};
    for (int ro = 0; ro < 4; ++ro) {
        double acc[8] = {0};
        for (int w = 0; w < nwg; ++w)
            for (int i = 1; i < 8; ++i) acc[i] += (double)(st[(size_t)w * 96 + ro * 24 + i] - st[(size_t)w * 96 + ro * 24 + i - 1]);
        printf("  %s sub-stamps, half-ticks 8..11 (work | wait for the barrier, x4):", r4[ro]);
        for (int i = 1; i < 8; ++i) printf(" %.0f", acc[i] / nwg);
        printf("\n");
    }
    return 0;
}
