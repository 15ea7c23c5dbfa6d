// In-kernel Gaussian noise and the DDPM posterior update, shared by the stand-alone step kernels (sampler.hip) and by the
// denoiser's last GEMM when it applies the update in its epilogue (gemm.h E_HEADS_POST): both must produce the same bits.
#pragma once
#include "common.h"

__device__ __forceinline__ void philox_round(uint32_t &c0, uint32_t &c1, uint32_t &c2, uint32_t &c3, uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}

// four N(0,1) draws for element group g (= flat element index / 4) of step `step`: Philox4x32-10 + Box-Muller
__device__ __forceinline__ float4 randn4(uint64_t seed, uint64_t step, uint64_t g) {
    uint32_t c0 = (uint32_t)g, c1 = (uint32_t)(g >> 32), c2 = (uint32_t)step, c3 = (uint32_t)(step >> 32);
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    // (0,1] uniforms -> Box-Muller on the hardware transcendentals: v_log_f32 (log2, 1 ulp), v_sqrt_f32, and v_sin_f32 / v_cos_f32,
    // which take their argument in REVOLUTIONS -- sin(2 pi u) is one instruction, where libm's sincosf spends ~150 on range
    // reduction that a [0,1) argument never needs.  Noise only has to be N(0,1) and reproducible (tests: moments, independence).
    const float u0 = ((float)(c0 >> 8) + 1.0f) * (1.0f / 16777216.0f), u1 = (float)(c1 >> 8) * (1.0f / 16777216.0f);
    const float u2 = ((float)(c2 >> 8) + 1.0f) * (1.0f / 16777216.0f), u3 = (float)(c3 >> 8) * (1.0f / 16777216.0f);
    const float r0 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u0));      // -2 ln u = -2 ln2 log2 u
    const float r1 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u2));
    const float s0 = __builtin_amdgcn_sinf(u1), co0 = __builtin_amdgcn_cosf(u1);
    const float s1 = __builtin_amdgcn_sinf(u3), co1 = __builtin_amdgcn_cosf(u3);
    return make_float4(r0 * co0, r0 * s0, r1 * co1, r1 * s1);
}

// x_{t-1} = (c1 * x0 + c2 * x_t) + sigma * eps with every product and sum rounded on its own, as the reference's tensor
// expression does (gaussian_diffusion.py:253-275 posterior mean, :532-547 sample): no FMA contraction, so every caller agrees
__device__ __forceinline__ float posterior1(float c1, float c2, float sigma, float x0, float xt, float eps) {
#pragma clang fp contract(off)
    const float a = c1 * x0, b = c2 * xt, c = sigma * eps;
    return (a + b) + c;
}
__device__ __forceinline__ float4 posterior4(float c1, float c2, float sigma, const float4 x0, const float4 xt, const float4 e) {
    return make_float4(posterior1(c1, c2, sigma, x0.x, xt.x, e.x), posterior1(c1, c2, sigma, x0.y, xt.y, e.y),
                       posterior1(c1, c2, sigma, x0.z, xt.z, e.z), posterior1(c1, c2, sigma, x0.w, xt.w, e.w));
}

// sampler state {t, loop index, seed, arrival counter}: the LAST workgroup of a launch to arrive advances it (t -= 1, loop index
// += 1, ts[b] = max(t, 0)).  Every workgroup reads the state before it adds itself to the counter, so the update cannot race
// with a reader of the same launch.  Call with all threads of the workgroup.
__device__ __forceinline__ void sampler_advance_last(int64_t *__restrict__ state, int64_t *__restrict__ ts, int B, unsigned n_wg, int64_t t,
                                                     uint64_t it) {
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned *arrived = reinterpret_cast<unsigned *>(state + 3);          // (no fence: __syncthreads drained every load of the state,
                                                                                 // and an agent-scope release would write back this XCD's whole L2)
        if (atomicAdd(arrived, 1u) == n_wg - 1) {
            *arrived = 0u;
            const int64_t tn = t - 1;
            state[0] = tn;
            state[1] = (int64_t)it + 1;
            for (int b = 0; b < B; ++b) ts[b] = tn < 0 ? 0 : tn;
        }
    }
}

// Fused plain step (interdiff_mdm_forward_step): the last GEMM applies the update in its epilogue, and 250 workgroups arriving on
// one counter would serialise ~20 ns apiece at the memory side.  Instead ONE thread of a kernel in the MIDDLE of the forward --
// after the embedding has read ts, before the last GEMM reads the state -- does the bookkeeping for the whole step: it parks the
// current {t, loop index} in state[4..5] for the last GEMM, sets ts to the NEXT step's timestep and advances state[0..1].  Nothing
// else reads or writes these words while that kernel runs, so there is no counter and no fence.
__device__ __forceinline__ void sampler_prepare_step(int64_t *__restrict__ state, int64_t *__restrict__ ts, int B) {
    const int64_t t = state[0], it = state[1], tn = t - 1;
    state[4] = t;
    state[5] = it;
    state[0] = tn;
    state[1] = it + 1;
    for (int b = 0; b < B; ++b) ts[b] = tn < 0 ? 0 : tn;
}
