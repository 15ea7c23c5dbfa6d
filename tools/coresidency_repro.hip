// Stand-alone reproducer of the co-residency effect described in DESIGN.md ("exclusive CU") -- no dependency on this repository:
//
//     hipcc --offload-arch=gfx950 -O3 tools/coresidency_repro.hip -o coresidency_repro && ./coresidency_repro
//     hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/coresidency_repro.hip -o coresidency_repro_noslp && ./coresidency_repro_noslp
//
// Observation (MI355X, gfx950, ROCm 7.2): a one-wave workgroup whose fp32 VALU code contains PACKED fp32 instructions (v_pk_mul_f32 / v_pk_fma_f32 /
// v_pk_add_f32, formed here by the SLP vectoriser) and runs with a PARTIAL exec mask (52 of 64 lanes) occasionally computes WRONG values in lanes 48..51
// while, on another stream, a kernel that issues v_mfma_f32_16x16x32_f16 is resident on the same CU.  The same victim next to an fp32-MFMA aggressor or next
// to loads alone is bit-stable; the same victim compiled WITHOUT packed fp32 instructions (-fno-slp-vectorize) is bit-stable next to every aggressor.
// Nothing is shared between the two kernels (no common buffer, no LDS overlap, different streams).
//
// VICTIM:    `victim`, 64 threads: lane j < 52 turns an axis-angle vector into a rotation matrix (quaternion route, ~60 fp32 VALU ops that the
//            SLP vectoriser turns into v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32), stores the nine entries.  1600 workgroups per launch.
// AGGRESSOR: `aggressor`, 256 threads: a loop of four v_mfma_f32_16x16x32_f16 on register operands + one 16-byte streaming load per lane.
// The program runs the victim alone (reference bits), then 48 victim launches beside the aggressor, and counts launches / lanes that differ;
// then the other aggressor forms (f16 MFMAs without loads; fp32 MFMAs; loads alone).  Seen on MI355X / ROCm 7.2 with the default build: one of the two f16-MFMA
// forms differs in 30 - 100 % of the victim launches, always lanes 48..51 (which of the two depends on how the two kernels' phases line up: this small victim reacts to
// "f16 MFMA only", the original SMPL kernel in tools/coresidency_probe.hip to "f16 MFMA + loads"); the fp32-MFMA and loads-only forms: 0.  With -fno-slp-vectorize
// (no v_pk_*_f32 instruction in the victim): 0 everywhere.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int J = 52;      // active lanes of the victim wave (the SMPL-H joint count of the kernel this was reduced from)

__global__ __launch_bounds__(64) void victim(const float *__restrict__ pose, float *__restrict__ out) {
    __shared__ float Rs[64 * 9];
    const int n = blockIdx.x, j = threadIdx.x;
    if (j < J) {
        const float *a = pose + ((size_t)n * J + j) * 3;
        const float ex = a[0] + 1e-8f, ey = a[1] + 1e-8f, ez = a[2] + 1e-8f;
        const float ang = sqrtf(ex * ex + ey * ey + ez * ez);
        const float half = ang * 0.5f, sn = sinf(half);
        float w = cosf(half), x = sn * (a[0] / ang), y = sn * (a[1] / ang), z = sn * (a[2] / ang);
        const float nq = sqrtf(w * w + x * x + y * y + z * z);
        w /= nq; x /= nq; y /= nq; z /= nq;
        const float w2 = w * w, x2 = x * x, y2 = y * y, z2 = z * z;
        const float wx = w * x, wy = w * y, wz = w * z, xy = x * y, xz = x * z, yz = y * z;
        float *m = Rs + j * 9;
        m[0] = w2 + x2 - y2 - z2; m[1] = 2 * xy - 2 * wz;    m[2] = 2 * wy + 2 * xz;
        m[3] = 2 * wz + 2 * xy;    m[4] = w2 - x2 + y2 - z2; m[5] = 2 * yz - 2 * wx;
        m[6] = 2 * xz - 2 * wy;    m[7] = 2 * wx + 2 * yz;    m[8] = w2 - x2 - y2 + z2;
    }
    __syncthreads();
    if (j < J)
        for (int e = 0; e < 9; ++e) out[((size_t)n * J + j) * 9 + e] = Rs[j * 9 + e];
}

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
// kind: 1 = f16 MFMA + loads, 2 = f16 MFMA only, 3 = fp32 MFMA + loads, 4 = loads only
__global__ __launch_bounds__(256) void aggressor(const float4 *__restrict__ src, size_t n4, float *__restrict__ sink, int iters, int kind) {
    const int tid = threadIdx.x;
    h8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.01f * ((tid + e) % 37) - 0.15f); b[e] = (_Float16)(0.02f * ((tid * 3 + e) % 29) - 0.2f); }
    f4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    size_t p = ((size_t)blockIdx.x * 256 + tid) % n4;
    for (int it = 0; it < iters; ++it) {
        if (kind == 1 || kind == 2) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, a, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, b, c3, 0, 0, 0);
        }
        if (kind == 3) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32((float)a[0], (float)b[0], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32((float)b[1], (float)a[1], c1, 0, 0, 0);
        }
        if (kind != 2) {
            const float4 v = src[p];
            p += 256 * 977;
            if (p >= n4) p -= n4;
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    float s = acc.x + acc.y + acc.z + acc.w;
    for (int r = 0; r < 4; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    if (s == 123.456f) sink[0] = s;            // never true: keeps everything alive
}

int main(int argc, char **argv) {
    const int N = 1600, REP = 24, trials = argc > 1 ? atoi(argv[1]) : 2, iters = argc > 2 ? atoi(argv[2]) : 5000, grid = argc > 3 ? atoi(argv[3]) : 1024;
    srand(3);
    std::vector<float> hpose((size_t)N * J * 3);
    for (auto &v : hpose) v = (rand() / (float)RAND_MAX - 0.5f) * 1.2f;
    float *pose, *out, *sink;
    const size_t no = (size_t)N * J * 9;
    CK(hipMalloc(&pose, hpose.size() * 4)); CK(hipMalloc(&out, no * 4 * REP)); CK(hipMalloc(&sink, 64));
    CK(hipMemcpy(pose, hpose.data(), hpose.size() * 4, hipMemcpyHostToDevice));
    const size_t n4 = (size_t)64 << 20;                       // 1 GiB to stream through
    float4 *src;
    CK(hipMalloc(&src, n4 * 16)); CK(hipMemset(src, 0, n4 * 16));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    hipLaunchKernelGGL(victim, dim3(N), dim3(64), 0, sa, pose, out);
    CK(hipDeviceSynchronize());
    std::vector<float> ref(no), got(no * REP);
    CK(hipMemcpy(ref.data(), out, no * 4, hipMemcpyDeviceToHost));
    const char *names[] = {"no aggressor", "f16 MFMA + global loads", "f16 MFMA only", "fp32 MFMA + global loads", "global loads only"};
    int affected = 0;
    for (int kind = 0; kind < 5; ++kind) {
        long bad_launches = 0, bad_words = 0;
        int lane_hist[64] = {0};
        for (int t = 0; t < trials; ++t) {
            CK(hipMemsetAsync(out, 0, no * 4 * REP, sa));
            CK(hipDeviceSynchronize());
            if (kind) hipLaunchKernelGGL(aggressor, dim3(grid), dim3(256), 0, sb, src, n4, sink, iters, kind);
            for (int r = 0; r < REP; ++r) hipLaunchKernelGGL(victim, dim3(N), dim3(64), 0, sa, pose, out + r * no);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(got.data(), out, no * 4 * REP, hipMemcpyDeviceToHost));
            for (int r = 0; r < REP; ++r) {
                bool bad = false;
                for (size_t i = 0; i < no; ++i)
                    if (memcmp(&got[r * no + i], &ref[i], 4)) { bad = true; ++bad_words; ++lane_hist[(i / 9) % J]; }
                bad_launches += bad;
            }
        }
        printf("%-28s: %ld of %d victim launches differ from the victim run alone, %ld words; lanes:", names[kind], bad_launches, trials * REP, bad_words);
        for (int l = 0; l < 64; ++l) if (lane_hist[l]) printf(" %d(%d)", l, lane_hist[l]);
        printf("\n");
        if ((kind == 1 || kind == 2) && bad_launches) affected = 1;
    }
    printf(affected ? "AFFECTED: a co-resident kernel issuing v_mfma_f32_16x16x32_f16 changed the victim's results\n" : "not reproduced on this system / build\n");
    return 0;
}
