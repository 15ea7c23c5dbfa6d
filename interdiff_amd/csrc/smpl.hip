// SMPL-H forward kinematics + linear blend skinning for gfx950 (rows B1-B3 of SURVEY.md §8).
//
// Behaviour restated from libsmpl/smplpytorch/pytorch/smpl_layer.py:72-175.  The reference runs
// ~1000 tiny torch kernels per call (Python loops over 52 joints) and materialises the per-vertex
// transforms th_T [N,4,4,6890] (705 MB at N=1600).  Here:
//   kernel 1 (one wave per frame): Rodrigues for all joints, joints = J_t + J_s.beta (the joint
//       regressor is folded into the shape basis on the host), the kinematic chain staged in LDS,
//       rest-pose removal; emits the blend-shape feature row [R_1..R_{J-1} - I | beta | 1] and the
//       per-joint 3x4 transforms A.
//   kernel 2 (32 frames x 64 vertices per workgroup): ONE fp32-MFMA GEMM feat[32,KB] x blend[KB,192]
//       gives v_posed (template, shape and pose blend shapes in one contraction: 95% of the FLOPs),
//       staged through LDS, then the <=S-bone skinning sum, the 3x4 apply and the translation in
//       the epilogue.  Nothing per-vertex except the final vertices ever reaches HBM.
#include "common.h"
#include "rot_math.h"
#include <algorithm>

namespace {

constexpr int MAXJ = 64;

__global__ __launch_bounds__(64) void smpl_pose_kernel(const idf_smpl_model m, const float *__restrict__ pose,
                                                       const float *__restrict__ betas, const float *__restrict__ trans,
                                                       float *__restrict__ feat, float *__restrict__ A,
                                                       float *__restrict__ jtr) {
    __shared__ float Rs[MAXJ * 9], Js[MAXJ * 3], Gs[MAXJ * 12];
    const int64_t n = blockIdx.x;
    const int j = threadIdx.x, J = m.J, nb = m.n_betas, KB = m.KB;
    const float *beta = betas + n * nb;
    if (j < J) {
        rot::rodrigues_smpl(pose + n * 3 * J + 3 * j, Rs + j * 9);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float s = m.jt[j * 3 + c];
            const float *jsr = m.js + (size_t)(j * 3 + c) * nb;
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
    if (j < 12) {
        const int r = j >> 2, c = j & 3;
        Gs[j] = c < 3 ? Rs[r * 3 + c] : Js[r];
    }
    __syncthreads();
    for (int i = 1; i < J; ++i) {
        const int p = m.parents[i];
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
        }
        __syncthreads();
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

constexpr int FT = 32, VT = 64, NTC = 3 * VT;          // frames / vertices / coordinates per workgroup
constexpr int FM = FT / 16;                             // 16-frame MFMA tiles per workgroup (each basis fragment is reused FM times)
constexpr int AQS = FT * 4 + 4;                         // padded quad stride of the feature image
constexpr int STS = NTC + 1;                            // stage row stride

__global__ __launch_bounds__(256) void smpl_blend_skin_kernel(const idf_smpl_model m, const float *__restrict__ feat,
                                                              const float *__restrict__ A, const float *__restrict__ trans,
                                                              int64_t N, float *__restrict__ verts,
                                                              float *__restrict__ v_posed) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int KB = m.KB, V = m.V, J = m.J, S = m.S, nq = KB / 4;
    float *As = sm, *stage = sm;                       // the stage image reuses the feature image once the contraction is done
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
    // frame tiles on blockIdx.x (fast) so that concurrently resident workgroups share one slice of the
    // blend basis in L2; vertex tiles on blockIdx.y
    const int64_t f0 = (int64_t)blockIdx.x * FT;
    const int v0 = blockIdx.y * VT;

    for (int idx = tid; idx < FT * nq; idx += 256) {
        const int row = idx / nq, q = idx - row * nq;
        const int64_t n = f0 + row;
        const float4 v = n < N ? *reinterpret_cast<const float4 *>(feat + n * KB + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4 *>(As + q * AQS + row * 4) = v;
    }
    __syncthreads();

    const float *brow[3];
    bool bval[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int nrow = 3 * v0 + (wave * 3 + t) * 16 + li;
        bval[t] = nrow < 3 * V;
        brow[t] = m.blend + (size_t)(bval[t] ? nrow : 0) * KB + kq * 4;
    }
    f32x4 acc[FM][3];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int t = 0; t < 3; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ng = KB / 16;
    // the basis fragments come straight from global memory (each is used by this wave only): three register sets keep the
    // loads of groups g+1 and g+2 in flight behind the 8*FM*3 MFMAs of group g
    float4 b0[3], b1[3], b2[3];
    auto ldg = [&](float4 (&dst)[3], int gi) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
            dst[t] = (bval[t] && gi < ng) ? *reinterpret_cast<const float4 *>(brow[t] + gi * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto mac = [&](const float4 (&bc)[3], int gi) {
        if (gi >= ng) return;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const float4 a = *reinterpret_cast<const float4 *>(As + (gi * 4 + kq) * AQS + (i * 16 + li) * 4);
#pragma unroll
            for (int t = 0; t < 3; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bc[t].x, acc[i][t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 3; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bc[t].y, acc[i][t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 3; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bc[t].z, acc[i][t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 3; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bc[t].w, acc[i][t], 0, 0, 0);
        }
    };
    ldg(b0, 0);
    ldg(b1, 1);
    for (int g = 0; g < ng; g += 3) {
        ldg(b2, g + 2);
        mac(b0, g);
        ldg(b0, g + 3);
        mac(b1, g + 1);
        ldg(b1, g + 4);
        mac(b2, g + 2);
    }
    __syncthreads();                                   // every wave is done reading As
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) stage[(i * 16 + kq * 4 + r) * STS + (wave * 3 + t) * 16 + li] = acc[i][t][r];
    __syncthreads();

#pragma unroll
    for (int k = 0; k < FT * VT / 256; ++k) {
        const int p = tid + 256 * k, f = p / VT, vl = p - f * VT;
        const int64_t n = f0 + f;
        const int v = v0 + vl;
        if (n >= N || v >= V) continue;
        const float px = stage[f * STS + 3 * vl], py = stage[f * STS + 3 * vl + 1], pz = stage[f * STS + 3 * vl + 2];
        float4 t0 = make_float4(0.f, 0.f, 0.f, 0.f), t1 = t0, t2 = t0;
        const float *An = A + (size_t)n * J * 12;
        for (int s = 0; s < S; ++s) {
            const float w = m.skin_w[(size_t)v * S + s];
            const float4 *a = reinterpret_cast<const float4 *>(An + m.skin_idx[(size_t)v * S + s] * 12);
            const float4 a0 = a[0], a1 = a[1], a2 = a[2];
            t0.x += w * a0.x; t0.y += w * a0.y; t0.z += w * a0.z; t0.w += w * a0.w;
            t1.x += w * a1.x; t1.y += w * a1.y; t1.z += w * a1.z; t1.w += w * a1.w;
            t2.x += w * a2.x; t2.y += w * a2.y; t2.z += w * a2.z; t2.w += w * a2.w;
        }
        float *o = verts + ((size_t)n * V + v) * 3;
        o[0] = (t0.x * px + t0.y * py + t0.z * pz + t0.w) + trans[n * 3 + 0];
        o[1] = (t1.x * px + t1.y * py + t1.z * pz + t1.w) + trans[n * 3 + 1];
        o[2] = (t2.x * px + t2.y * py + t2.z * pz + t2.w) + trans[n * 3 + 2];
        if (v_posed) {
            float *q = v_posed + ((size_t)n * V + v) * 3;
            q[0] = px; q[1] = py; q[2] = pz;
        }
    }
}

}  // namespace

extern "C" size_t interdiff_smpl_workspace_bytes(const idf_smpl_model *m, int64_t N) {
    if (!m || N < 0) return 0;
    return idf_align((size_t)N * m->KB * sizeof(float)) + idf_align((size_t)N * m->J * 12 * sizeof(float));
}

extern "C" int interdiff_smpl_forward(const idf_smpl_model *m, const float *pose, const float *betas, const float *trans,
                                      int64_t N, float *verts, float *jtr, float *v_posed, void *ws, size_t ws_bytes,
                                      void *stream) {
    if (!m || !pose || !betas || !trans || !verts || !jtr || !ws || N < 0) return IDF_E_INVAL;
    if (m->J < 1 || m->J > MAXJ || m->KB % 16 != 0 || m->KB < 9 * (m->J - 1) + m->n_betas + 1 || m->S < 1) return IDF_E_INVAL;
    if (ws_bytes < interdiff_smpl_workspace_bytes(m, N)) return IDF_E_NOMEM;
    if (N == 0) return IDF_OK;
    hipStream_t s = idf_stream(stream);
    float *feat = reinterpret_cast<float *>(ws);
    float *A = reinterpret_cast<float *>(reinterpret_cast<char *>(ws) + idf_align((size_t)N * m->KB * sizeof(float)));
    idf_prof_mark(IDF_K_SMPL_POSE, s);
    hipLaunchKernelGGL(smpl_pose_kernel, dim3((unsigned)N), dim3(64), 0, s, *m, pose, betas, trans, feat, A, jtr);
    const size_t lds = std::max((size_t)(m->KB / 4) * AQS, (size_t)FT * STS) * sizeof(float);
    if (lds > 150 * 1024) return IDF_E_INVAL;
    static std::atomic<uint64_t> lds_ok{0};
    if (lds > 64 * 1024 && idf_opt_in_lds(reinterpret_cast<const void *>(smpl_blend_skin_kernel), 150 * 1024, lds_ok) != IDF_OK) return IDF_E_LAUNCH;
    idf_prof_mark(IDF_K_SMPL_BLEND_SKIN, s);
    hipLaunchKernelGGL(smpl_blend_skin_kernel, dim3((unsigned)idf_cdiv(N, FT), (unsigned)idf_cdiv(m->V, VT)), dim3(256), lds, s, *m,
                       feat, A, trans, N, verts, v_posed);
    idf_prof_mark(-1, s);
    IDF_CHECK_LAUNCH();
    return IDF_OK;
}
