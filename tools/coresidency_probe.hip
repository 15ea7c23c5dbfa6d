// Minimal reproducer hunt for the co-residency effect of DESIGN.md 4.2 ("exclusive CU"): does a trivial kernel that only issues MFMAs (f16 or fp32), with or
// without memory / LDS traffic, make the one-wave SMPL pose kernel (csrc/smpl.hip, compiled here as part of this translation unit) compute different bits
// when both run at the same time on two streams?  (not product code)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-inline-asm -I interdiff_amd/csrc tools/coresidency_probe.hip -o build_tools/coresidency_probe
#include <hip/hip_runtime.h>
#include "smpl.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
bool g_idf_prof_on = false;
void idf_prof_mark_slow(int, hipStream_t) {}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// The victim restated with switches (VAR): 0 = as csrc/smpl.hip; 1 = a real s_barrier where the compiler elides __syncthreads() for a one-wave workgroup;
// 2 = 4 KiB of unused LDS behind the arrays (is it the TAIL of the allocation?); 3 = the parent table copied to LDS first (is it the scalar loads?);
// 4 = debug: also writes the parent indices and the chain matrices of joints 49..51 it saw; 5 = all 64 lanes active in the Rodrigues part;
// 6 (round 5) = the Rodrigues block between s_setprio 3 / s_nop fences (does raising the victim's issue priority or draining the VALU pipe around the block matter?);
// 7 (round 5) = every product of the quaternion -> matrix block through an opaque register (asm volatile "+v"): no packed-fp32 instruction can be formed there
// whatever the vectoriser does (the whole-binary form of the same question is the -fno-slp-vectorize build: build_tools/coresidency_probe_noslp)
template <int VAR>
__global__ __launch_bounds__(64) void pose_victim(const idf_smpl_model m, const float *__restrict__ pose, const float *__restrict__ betas,
                                                  const float *__restrict__ trans, float *__restrict__ feat, float *__restrict__ A, float *__restrict__ jtr,
                                                  float *__restrict__ dbg) {
    __shared__ float Rs[MAXJ * 9], Js[MAXJ * 3], Gs[MAXJ * 12];
    __shared__ float pad[VAR == 2 ? 1024 : 1];
    __shared__ int par_s[MAXJ];
    const int64_t n = blockIdx.x;
    const int j = threadIdx.x, J = m.J, nb = m.n_betas, KB = m.KB;
    const float *beta = betas + n * nb;
    if (VAR == 2 && j == 63 && pose[0] == 1234.5f) pad[j] = 1.f, feat[0] = pad[(j * 7) & 1023];
    if (VAR == 3) par_s[j] = j < J ? m.parents[j] : 0;
    if (VAR == 5 || j < J) {           // VAR 5: all 64 lanes run the Rodrigues / joint code (lanes >= J on joint J - 1's inputs, into the spare rows): no partial exec mask
        const int jj = VAR == 5 ? min(j, J - 1) : j;
        if (VAR == 6) asm volatile("s_setprio 3\n s_nop 7\n s_nop 7" ::: "memory");
        if (VAR == 7) {
            const float *a = pose + n * 3 * J + 3 * jj;
            auto pin = [](float v) { asm volatile("" : "+v"(v)); return v; };
            const float ex = a[0] + 1e-8f, ey = a[1] + 1e-8f, ez = a[2] + 1e-8f;
            const float ang = sqrtf(pin(pin(ex * ex) + pin(ey * ey)) + pin(ez * ez));
            const float half = ang * 0.5f, sn = sinf(half);
            float w = cosf(half), x = pin(sn * pin(a[0] / ang)), y = pin(sn * pin(a[1] / ang)), z = pin(sn * pin(a[2] / ang));
            const float nq = sqrtf(pin(pin(pin(w * w) + pin(x * x)) + pin(y * y)) + pin(z * z));
            w = pin(w / nq); x = pin(x / nq); y = pin(y / nq); z = pin(z / nq);
            const float w2 = pin(w * w), x2 = pin(x * x), y2 = pin(y * y), z2 = pin(z * z);
            const float wx = pin(w * x), wy = pin(w * y), wz = pin(w * z), xy = pin(x * y), xz = pin(x * z), yz = pin(y * z);
            float *mm = Rs + j * 9;
            mm[0] = pin(pin(pin(w2 + x2) - y2) - z2); mm[1] = pin(pin(2 * xy) - pin(2 * wz)); mm[2] = pin(pin(2 * wy) + pin(2 * xz));
            mm[3] = pin(pin(2 * wz) + pin(2 * xy)); mm[4] = pin(pin(pin(w2 - x2) + y2) - z2); mm[5] = pin(pin(2 * yz) - pin(2 * wx));
            mm[6] = pin(pin(2 * xz) - pin(2 * wy)); mm[7] = pin(pin(2 * wx) + pin(2 * yz)); mm[8] = pin(pin(pin(w2 - x2) - y2) + z2);
        } else
        rot::rodrigues_smpl(pose + n * 3 * J + 3 * jj, Rs + j * 9);
        if (VAR == 6) asm volatile("s_nop 7\n s_nop 7\n s_setprio 0" ::: "memory");
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float s = m.jt[jj * 3 + c];
            const float *jsr = m.js + (size_t)(jj * 3 + c) * nb;
            for (int k = 0; k < nb; ++k) s += jsr[k] * beta[k];
            Js[j * 3 + c] = s;
        }
    }
    float *f = feat + n * KB;
    if (j >= 1 && j < J) {
#pragma unroll
        for (int e = 0; e < 9; ++e) f[(j - 1) * 9 + e] = Rs[j * 9 + e] - ((e == 0 || e == 4 || e == 8) ? 1.0f : 0.0f);
    }
    const int base = 9 * (J - 1);
    for (int k = j; k < KB - base; k += 64) f[base + k] = k < nb ? beta[k] : (k == nb ? 1.0f : 0.0f);
    __syncthreads();
    if (VAR == 1) asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier" ::: "memory");
    if (j < 12) {
        const int r = j >> 2, c = j & 3;
        Gs[j] = c < 3 ? Rs[r * 3 + c] : Js[r];
    }
    __syncthreads();
    if (VAR == 1) asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier" ::: "memory");
    for (int i = 1; i < J; ++i) {
        const int p = VAR == 3 ? par_s[i] : m.parents[i];
        if (VAR == 4 && j == 0 && i >= 49) dbg[n * 64 + (i - 49)] = (float)p;
        if (j < 12) {
            const int r = j >> 2, c = j & 3;
            const float *gp = Gs + p * 12 + r * 4;
            float v;
            if (c < 3)
                v = gp[0] * Rs[i * 9 + c] + gp[1] * Rs[i * 9 + 3 + c] + gp[2] * Rs[i * 9 + 6 + c];
            else
                v = gp[0] * (Js[i * 3] - Js[p * 3]) + gp[1] * (Js[i * 3 + 1] - Js[p * 3 + 1]) +
                    gp[2] * (Js[i * 3 + 2] - Js[p * 3 + 2]) + gp[3];
            Gs[i * 12 + j] = v;
            if (VAR == 4 && i >= 49) dbg[n * 64 + 4 + (i - 49) * 12 + j] = v;
        }
        __syncthreads();
        if (VAR == 1) asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier" ::: "memory");
    }
    if (VAR == 4 && j == 0) {          // where did this workgroup run?  HW_ID (hwreg 4): simd [5:4], cu [11:8], sh [12], se [15:13]; XCC_ID (hwreg 20): [3:0]
        dbg[n * 64 + 60] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4));
        dbg[n * 64 + 61] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 20));
    }
    if (VAR == 4 && j >= 49 && j < J) {
        if (j == 49) for (int c = 0; c < 9; ++c) dbg[n * 64 + 40 + c] = Rs[49 * 9 + c];        // joint 49's Rodrigues matrix as it sits in LDS after the chain
        dbg[n * 64 + 49 + (j - 49)] = Js[j * 3];
    }
    if (j < J) {
        const float *g = Gs + j * 12;
        const float jx = Js[j * 3], jy = Js[j * 3 + 1], jz = Js[j * 3 + 2];
        float *a = A + ((size_t)n * J + j) * 12;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            a[r * 4 + 0] = g[r * 4 + 0];
            a[r * 4 + 1] = g[r * 4 + 1];
            a[r * 4 + 2] = g[r * 4 + 2];
            a[r * 4 + 3] = g[r * 4 + 3] - (g[r * 4] * jx + g[r * 4 + 1] * jy + g[r * 4 + 2] * jz);
            jtr[((size_t)n * J + j) * 3 + r] = g[r * 4 + 3] + trans[n * 3 + r];
        }
    }
}

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
// KIND bit 0: f16 MFMAs, bit 1: fp32 MFMAs, bit 2: streaming global loads, bit 3: LDS b128 reads, bit 4: take 100 KiB of LDS (occupancy like the row block)
template <int KIND>
__global__ __launch_bounds__(256) void aggressor(const float4 *__restrict__ src, size_t n4, float *__restrict__ sink, int iters) {
    __shared__ float4 lds[(KIND & 16) ? 6400 : 512];
    const int tid = threadIdx.x;
    for (int i = tid; i < 512; i += 256) lds[i] = make_float4(0.001f * i, 1.f, -2.f, 0.5f);
    __syncthreads();
    h8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.01f * ((tid + e) % 37) - 0.15f); b[e] = (_Float16)(0.02f * ((tid * 3 + e) % 29) - 0.2f); }
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    size_t p = ((size_t)blockIdx.x * 256 + tid) % n4;
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND & 1) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, a, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, b, c3, 0, 0, 0);
        }
        if constexpr (KIND & 2) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32((float)a[0], (float)b[0], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32((float)b[1], (float)a[1], c1, 0, 0, 0);
        }
        if constexpr (KIND & 4) {
            const float4 v = src[p];
            p += 256 * 977;
            if (p >= n4) p -= n4;
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        if constexpr (KIND & 8) {
            const float4 v = lds[(tid * 7 + it) & 511];
            acc.x += v.x; acc.y += v.w;
        }
    }
    float s = acc.x + acc.y + acc.z + acc.w;
    for (int r = 0; r < 4; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    if (s == 123.456f) sink[0] = s;            // never true: keeps everything alive
}

template <int KIND>
void launch_aggr(hipStream_t s, const float4 *src, size_t n4, float *sink, int iters, int grid) {
    hipLaunchKernelGGL(aggressor<KIND>, dim3(grid), dim3(256), 0, s, src, n4, sink, iters);
}

int main(int argc, char **argv) {
    const int64_t N = 1600;
    const int J = 52, nb = 10, KB = 480, REP = 24;
    srand(3);
    auto rnd = [](float s) { return (rand() / (float)RAND_MAX - 0.5f) * 2.f * s; };
    std::vector<float> hjt(J * 3), hjs(J * 3 * nb), hpose(N * 156), hbeta(N * 10), htrans(N * 3);
    for (auto &v : hjt) v = rnd(0.5f);
    for (auto &v : hjs) v = rnd(0.02f);
    for (auto &v : hpose) v = rnd(0.6f);
    for (auto &v : hbeta) v = rnd(1.5f);
    for (auto &v : htrans) v = rnd(1.0f);
    std::vector<int32_t> par(J);
    for (int j = 0; j < J; ++j) par[j] = j == 0 ? 0 : (j < 22 ? (j - 1) / 2 : ((j - 22) % 3 == 0 ? (j < 37 ? 20 : 21) : j - 1));     // body tree + 10 finger chains of 3
    float *jt, *js, *pose, *betas, *trans, *feat, *A, *jtr, *sink;
    int32_t *parents;
    CK(hipMalloc(&jt, hjt.size() * 4)); CK(hipMalloc(&js, hjs.size() * 4)); CK(hipMalloc(&parents, J * 4));
    CK(hipMalloc(&pose, hpose.size() * 4)); CK(hipMalloc(&betas, hbeta.size() * 4)); CK(hipMalloc(&trans, htrans.size() * 4));
    CK(hipMemcpy(jt, hjt.data(), hjt.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(js, hjs.data(), hjs.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(parents, par.data(), J * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(pose, hpose.data(), hpose.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(betas, hbeta.data(), hbeta.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(trans, htrans.data(), htrans.size() * 4, hipMemcpyHostToDevice));
    const size_t nA = (size_t)N * J * 12, nJ = (size_t)N * J * 3, nF = (size_t)N * KB;
    CK(hipMalloc(&feat, nF * 4 * REP)); CK(hipMalloc(&A, nA * 4 * REP)); CK(hipMalloc(&jtr, nJ * 4 * REP)); CK(hipMalloc(&sink, 64));
    const size_t n4 = (size_t)64 << 20;                       // 1 GiB of float4 to stream through
    float4 *src;
    CK(hipMalloc(&src, n4 * 16)); CK(hipMemset(src, 0, n4 * 16));
    idf_smpl_model m{6890, J, nb, KB, 4, nullptr, jt, js, parents, nullptr, nullptr};
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    auto victims = [&]() {
        for (int r = 0; r < REP; ++r)
            hipLaunchKernelGGL(smpl_pose_kernel, dim3((unsigned)N), dim3(64), 0, sa, m, pose, betas, trans, feat + r * nF, A + r * nA, jtr + r * nJ);
    };
    victims();
    CK(hipDeviceSynchronize());
    std::vector<float> refA(nA), refJ(nJ), gotA(nA * REP), gotJ(nJ * REP);
    CK(hipMemcpy(refA.data(), A, nA * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(refJ.data(), jtr, nJ * 4, hipMemcpyDeviceToHost));
    const char *names[] = {"none", "f16 MFMA", "fp32 MFMA", "f16 MFMA + global loads", "global loads", "f16 MFMA + LDS reads", "f16 MFMA + loads + LDS", "f16 MFMA, 100 KiB LDS",
                           "f16 MFMA + loads + LDS reads, 100 KiB LDS", "fp32 MFMA + loads + LDS reads, 100 KiB LDS"};
    const int iters = argc > 1 ? atoi(argv[1]) : 20000, grid = argc > 2 ? atoi(argv[2]) : 1024, trials = argc > 3 ? atoi(argv[3]) : 6;
    for (int kind = 0; kind < 10; ++kind) {
        long bad_runs = 0, bad_vals = 0;
        int jhist[64] = {0};
        double worst = 0;
        for (int t = 0; t < trials; ++t) {
            CK(hipMemsetAsync(A, 0, nA * 4 * REP, sa)); CK(hipMemsetAsync(jtr, 0, nJ * 4 * REP, sa));
            CK(hipDeviceSynchronize());
            switch (kind) {
            case 1: launch_aggr<1>(sb, src, n4, sink, iters, grid); break;
            case 2: launch_aggr<2>(sb, src, n4, sink, iters, grid); break;
            case 3: launch_aggr<5>(sb, src, n4, sink, iters / 4, grid); break;
            case 4: launch_aggr<4>(sb, src, n4, sink, iters / 4, grid); break;
            case 5: launch_aggr<9>(sb, src, n4, sink, iters, grid); break;
            case 6: launch_aggr<13>(sb, src, n4, sink, iters / 4, grid); break;
            case 7: launch_aggr<17>(sb, src, n4, sink, iters, grid); break;
            case 8: launch_aggr<29>(sb, src, n4, sink, iters / 4, grid); break;
            case 9: launch_aggr<30>(sb, src, n4, sink, iters / 4, grid); break;
            default: break;
            }
            victims();
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(gotA.data(), A, nA * 4 * REP, hipMemcpyDeviceToHost)); CK(hipMemcpy(gotJ.data(), jtr, nJ * 4 * REP, hipMemcpyDeviceToHost));
            for (int r = 0; r < REP; ++r) {
                bool bad = false;
                for (size_t i = 0; i < nJ; ++i)
                    if (memcmp(&gotJ[r * nJ + i], &refJ[i], 4)) {
                        bad = true; ++bad_vals; ++jhist[(i / 3) % J];
                        const double d = fabs((double)gotJ[r * nJ + i] - refJ[i]);
                        if (d > worst) worst = d;
                    }
                if (memcmp(&gotA[r * nA], refA.data(), nA * 4)) bad = true;
                bad_runs += bad;
            }
        }
        printf("aggressor %-44s: %ld of %d victim launches differ, %ld joint coordinates, worst %.3g; joints:", names[kind], bad_runs, trials * REP, bad_vals, worst);
        for (int j = 0; j < J; ++j) if (jhist[j]) printf(" %d(%d)", j, jhist[j]);
        printf("\n");
        fflush(stdout);
    }
    // ---- which part of the victim is it?  the aggressor that reproduces ("f16 MFMA + global loads") against the victim's variants
    float *dbg;
    CK(hipMalloc(&dbg, (size_t)N * 64 * 4 * 2));
    const char *vn[] = {"as shipped", "real s_barrier at every __syncthreads", "4 KiB of unused LDS behind the arrays", "parent table in LDS (no scalar loads in the chain)", "debug stores", "all 64 lanes active in the Rodrigues part (no partial exec)",
                        "Rodrigues block between s_setprio 3 / s_nop fences", "quaternion -> matrix block with every product pinned to a scalar fp32 op (no packed fp32)"};
    std::vector<float> dref((size_t)N * 64), dgot((size_t)N * 64);
    for (int var = 0; var < 8; ++var) {
        auto vict = [&](float *Aout, float *Jout, float *D) {
            switch (var) {
            case 0: hipLaunchKernelGGL(pose_victim<0>, dim3((unsigned)N), dim3(64), 0, sa, m, pose, betas, trans, feat, Aout, Jout, D); break;
            case 1: hipLaunchKernelGGL(pose_victim<1>, dim3((unsigned)N), dim3(64), 0, sa, m, pose, betas, trans, feat, Aout, Jout, D); break;
            case 2: hipLaunchKernelGGL(pose_victim<2>, dim3((unsigned)N), dim3(64), 0, sa, m, pose, betas, trans, feat, Aout, Jout, D); break;
            case 3: hipLaunchKernelGGL(pose_victim<3>, dim3((unsigned)N), dim3(64), 0, sa, m, pose, betas, trans, feat, Aout, Jout, D); break;
            case 5: hipLaunchKernelGGL(pose_victim<5>, dim3((unsigned)N), dim3(64), 0, sa, m, pose, betas, trans, feat, Aout, Jout, D); break;
            case 6: hipLaunchKernelGGL(pose_victim<6>, dim3((unsigned)N), dim3(64), 0, sa, m, pose, betas, trans, feat, Aout, Jout, D); break;
            case 7: hipLaunchKernelGGL(pose_victim<7>, dim3((unsigned)N), dim3(64), 0, sa, m, pose, betas, trans, feat, Aout, Jout, D); break;
            default: hipLaunchKernelGGL(pose_victim<4>, dim3((unsigned)N), dim3(64), 0, sa, m, pose, betas, trans, feat, Aout, Jout, D); break;
            }
        };
        vict(A, jtr, dbg);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(refJ.data(), jtr, nJ * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(dref.data(), dbg, dref.size() * 4, hipMemcpyDeviceToHost));
        long bad_runs = 0, shown = 0;
        int jhist[64] = {0}, ahist[64] = {0}, ehist[12] = {0};
        CK(hipMemcpy(refA.data(), A, nA * 4, hipMemcpyDeviceToHost));
        for (int t = 0; t < trials * 4; ++t) {
            CK(hipDeviceSynchronize());
            launch_aggr<5>(sb, src, n4, sink, iters / 4, grid);
            for (int r = 0; r < REP; ++r) vict(A + r * nA, jtr + r * nJ, dbg + (r == REP - 1 ? (size_t)N * 64 : 0));
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(gotJ.data(), jtr, nJ * 4 * REP, hipMemcpyDeviceToHost));
            CK(hipMemcpy(gotA.data(), A, nA * 4 * REP, hipMemcpyDeviceToHost));
            CK(hipMemcpy(dgot.data(), dbg + (size_t)N * 64, dgot.size() * 4, hipMemcpyDeviceToHost));
            for (int r = 0; r < REP; ++r)
                for (size_t i = 0; i < nA; ++i)
                    if (memcmp(&gotA[r * nA + i], &refA[i], 4)) { ++ahist[(i / 12) % J]; ++ehist[i % 12]; }
            for (int r = 0; r < REP; ++r) {
                bool bad = false;
                for (size_t i = 0; i < nJ; ++i)
                    if (memcmp(&gotJ[r * nJ + i], &refJ[i], 4)) { bad = true; ++jhist[(i / 3) % J]; }
                bad_runs += bad;
            }
            if (var == 4) {
                static int where[8][8][2][16][4], total[8][8][2][16][4];
                for (int64_t fr = 0; fr < N; ++fr) {
                    unsigned hw, xc;
                    memcpy(&hw, &dgot[fr * 64 + 60], 4); memcpy(&xc, &dgot[fr * 64 + 61], 4);
                    const int simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7, x = xc & 7;
                    ++total[x][se][sh][cu][simd];
                    if (memcmp(&dgot[fr * 64], &dref[fr * 64], 52 * 4)) ++where[x][se][sh][cu][simd];
                }
                if (t == trials * 4 - 1) {
                    printf("    workgroups of the last victim launch of every trial with wrong values, by where they ran (xcc se sh cu simd: wrong / all):\n     ");
                    int shown2 = 0;
                    for (int x = 0; x < 8; ++x) for (int se = 0; se < 8; ++se) for (int sh = 0; sh < 2; ++sh) for (int cu = 0; cu < 16; ++cu) for (int sd = 0; sd < 4; ++sd)
                        if (where[x][se][sh][cu][sd]) { printf(" x%d se%d sh%d cu%d simd%d: %d/%d;", x, se, sh, cu, sd, where[x][se][sh][cu][sd], total[x][se][sh][cu][sd]); if (++shown2 % 6 == 0) printf("\n     "); }
                    printf("\n");
                }
            }
            if (var == 4 && shown < 3)
                for (int64_t fr = 0; fr < N && shown < 3; ++fr) {
                    if (!memcmp(&dgot[fr * 64], &dref[fr * 64], 52 * 4)) continue;
                    ++shown;
                    printf("    frame %lld debug words that differ (index: alone -> beside the aggressor):", (long long)fr);
                    for (int k = 0; k < 52; ++k)
                        if (memcmp(&dgot[fr * 64 + k], &dref[fr * 64 + k], 4)) printf(" [%d] %g -> %g", k, dref[fr * 64 + k], dgot[fr * 64 + k]);
                    printf("\n      (0..2 parent of joints 49..51; 4..39 chain rows of joints 49..51 as written; 40..48 Rs[49][0..8] after the chain; 49..51 Js[j][0])\n");
                }
        }
        printf("victim %-52s: %ld of %d launches differ; joints:", vn[var], bad_runs, trials * 4 * REP);
        for (int j = 0; j < J; ++j) if (jhist[j]) printf(" %d(%d)", j, jhist[j]);
        printf("\n    3x4 transforms A that differ, by joint:");
        for (int j = 0; j < J; ++j) if (ahist[j]) printf(" %d(%d)", j, ahist[j]);
        printf("; by entry (row-major 3x4):");
        for (int e = 0; e < 12; ++e) if (ehist[e]) printf(" %d(%d)", e, ehist[e]);
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
