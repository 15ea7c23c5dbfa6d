// Contact-frame correction predictor ObjProjector.sample (rows D1, D2 of SURVEY.md §8): launchers; the device code is csrc/objproj.h.
#include "objproj.h"

namespace {
using namespace idf_objproj_dev;

template <int PART>
__global__ __launch_bounds__(NTHR) void objproj_kernel(const idf_objproj op, const float *__restrict__ obj_angles,
                                                      const float *__restrict__ obj_trans, const float *__restrict__ markers,
                                                      const int32_t *__restrict__ contact, int B, float *__restrict__ keep_g, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    objproj_body<PART>(sm, op, obj_angles, obj_trans, markers, contact, B, blockIdx.x, keep_g, out);
}

// node selection + IDCT of the selected node from the stacks' output `keep_g` (objproj_body<1>): the tail of objproj_body<0>, the same expressions on the same values
__global__ __launch_bounds__(256) void objproj_pick_kernel(const idf_objproj op, const float *__restrict__ keep_g, const int32_t *__restrict__ contact, int B, float *__restrict__ out) {
    __shared__ int pick_s;
    __shared__ float col[CH * NP];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int T = op.T, P = op.P, P1 = P + 1;
    const float *ar = op.arena, *Di = ar + op.idct;
    if (tid == 0) {                                     // correction_smpl.py:125-136: no contact -> node 0, else 1 + argmax(contact + hand bonus)
        const float *bonus = ar + op.hand_bonus;
        long csum = 0;
        float best = -1.f;
        int bi = 0;
        for (int p = 0; p < P; ++p) {
            const int cv = contact[(size_t)b * P + p];
            csum += cv;
            const float sc = (float)cv + bonus[p];
            if (sc > best) { best = sc; bi = p; }
        }
        pick_s = csum > 0 ? 1 + bi : 0;
    }
    __syncthreads();
    const int pick = pick_s;
    if (tid < CH * NP) col[tid] = keep_g[(size_t)b * (CH * NP * MAXN) + (tid / NP) * NP * P1 + (tid % NP) * P1 + pick];
    __syncthreads();
    for (int i = tid; i < T * CH; i += 256) {
        const int t = i / CH, c = i - t * CH;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NP; ++k) s += Di[t * NP + k] * col[c * NP + k];
        out[((size_t)t * B + b) * CH + c] = s;
    }
}

}  // namespace

int idf_objproj_check(const idf_objproj *op) {
    if (!op || op->n_pre != NP || op->P + 1 != MAXN || op->past_len < 1 || op->past_len > op->T) return IDF_E_INVAL;
    for (int l = 0; l < 12; ++l)
        if (op->cin[l] > 32 || op->cout[l] > 32 || op->cin[l] + op->cout[l] > POOL_CH) return IDF_E_INVAL;
    return IDF_OK;
}

size_t idf_objproj_keep_floats() { return (size_t)CH * NP * MAXN; }

int idf_objproj_pick(const idf_objproj *op, const float *keep, const int32_t *contact, int B, float *out, hipStream_t s) {
    if (!keep || !contact || !out || B <= 0 || idf_objproj_check(op) != IDF_OK) return IDF_E_INVAL;
    hipLaunchKernelGGL(objproj_pick_kernel, dim3(B), dim3(256), 0, s, *op, keep, contact, B, out);
    return IDF_OK;
}

extern "C" int interdiff_objprojector_sample(const idf_objproj *op, const float *obj_angles, const float *obj_trans,
                                             const float *markers, const int32_t *contact, int32_t B, float *out,
                                             void *stream) {
    if (!op || !obj_angles || !obj_trans || !markers || !contact || !out || B <= 0) return IDF_E_INVAL;
    if (idf_objproj_check(op) != IDF_OK) return IDF_E_INVAL;
    static std::atomic<uint64_t> lds_ok{0};
    if (idf_opt_in_lds(reinterpret_cast<const void *>(objproj_kernel<0>), (int)OBJPROJ_LDS, lds_ok) != IDF_OK) return IDF_E_LAUNCH;
    idf_prof_mark(IDF_K_OBJPROJ, idf_stream(stream));
    hipLaunchKernelGGL(objproj_kernel<0>, dim3(B), dim3(NTHR), OBJPROJ_LDS, idf_stream(stream), *op, obj_angles, obj_trans, markers, contact, B, nullptr, out);
    idf_prof_mark(-1, idf_stream(stream));
    IDF_CHECK_LAUNCH();
    return IDF_OK;
}
