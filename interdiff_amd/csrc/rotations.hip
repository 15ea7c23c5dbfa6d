// Elementwise rotation conversions behind the pytorch3d.transforms names the reference calls
// (row C2 of SURVEY.md §8).  One thread per rotation; inputs/outputs are tiny ([T,B,22,6]).
#include "common.h"
#include "rot_math.h"

namespace {

enum { OP_6D_MAT, OP_MAT_6D, OP_MAT_AA, OP_AA_MAT, OP_AA_QUAT, OP_6D_AA };

template <int OP, int NIN, int NOUT>
__global__ __launch_bounds__(256) void rot_kernel(const float *__restrict__ in, float *__restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a[NIN], r[NOUT];
#pragma unroll
    for (int k = 0; k < NIN; ++k) a[k] = in[i * NIN + k];
    if constexpr (OP == OP_6D_MAT) rot::rot6d_to_matrix(a, r);
    if constexpr (OP == OP_MAT_6D) {
#pragma unroll
        for (int k = 0; k < 6; ++k) r[k] = a[k];
    }
    if constexpr (OP == OP_MAT_AA) rot::matrix_to_axis_angle(a, r);
    if constexpr (OP == OP_AA_MAT) {
        float q[4];
        rot::axis_angle_to_quaternion(a, q);
        rot::quaternion_to_matrix(q, r);
    }
    if constexpr (OP == OP_AA_QUAT) rot::axis_angle_to_quaternion(a, r);
    if constexpr (OP == OP_6D_AA) {
        float m[9];
        rot::rot6d_to_matrix(a, m);
        rot::matrix_to_axis_angle(m, r);
    }
#pragma unroll
    for (int k = 0; k < NOUT; ++k) out[i * NOUT + k] = r[k];
}

template <int OP, int NIN, int NOUT>
int run(const float *in, float *out, int64_t n, void *stream) {
    if (!in || !out || n < 0) return IDF_E_INVAL;
    if (n == 0) return IDF_OK;
    hipLaunchKernelGGL((rot_kernel<OP, NIN, NOUT>), dim3((unsigned)idf_cdiv(n, 256)), dim3(256), 0, idf_stream(stream), in, out, n);
    IDF_CHECK_LAUNCH();
    return IDF_OK;
}

}  // namespace

extern "C" int interdiff_rotation_6d_to_matrix(const float *d6, float *m, int64_t n, void *s) { return run<OP_6D_MAT, 6, 9>(d6, m, n, s); }
extern "C" int interdiff_matrix_to_rotation_6d(const float *m, float *d6, int64_t n, void *s) { return run<OP_MAT_6D, 9, 6>(m, d6, n, s); }
extern "C" int interdiff_matrix_to_axis_angle(const float *m, float *aa, int64_t n, void *s) { return run<OP_MAT_AA, 9, 3>(m, aa, n, s); }
extern "C" int interdiff_axis_angle_to_matrix(const float *aa, float *m, int64_t n, void *s) { return run<OP_AA_MAT, 3, 9>(aa, m, n, s); }
extern "C" int interdiff_axis_angle_to_quaternion(const float *aa, float *q, int64_t n, void *s) { return run<OP_AA_QUAT, 3, 4>(aa, q, n, s); }
extern "C" int interdiff_rotation_6d_to_axis_angle(const float *d6, float *aa, int64_t n, void *s) { return run<OP_6D_AA, 6, 3>(d6, aa, n, s); }
