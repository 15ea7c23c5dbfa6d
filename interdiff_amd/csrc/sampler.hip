// DDPM reverse-step update (rows S4-S5 of SURVEY.md §8): inpainting of the x0 prediction,
// posterior mean and the noise add, one HBM pass each (elementwise, 16 B per lane).
// Reference behaviour: diffusion/gaussian_diffusion.py:307-311 (inpaint), :374 + :253-275
// (posterior mean = c1*x0 + c2*x_t), :532-547 (sample = mean + (t!=0) exp(.5 logvar) eps).
// The reference draws eps with torch's global generator (one randn_like per step); here eps
// is either handed in (deterministic parity runs) or produced in-kernel by Philox4x32-10 +
// Box-Muller keyed by (seed, step, element), so no noise tensor ever round-trips HBM.
#include "common.h"
#include "philox.h"

namespace {

__global__ __launch_bounds__(256) void inpaint_kernel(float *__restrict__ x0, const float *__restrict__ gt,
                                                      const uint8_t *__restrict__ mask, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        if (mask[i]) x0[i] = gt[i];
}

template <bool GEN>
__global__ __launch_bounds__(256) void posterior_kernel(float *__restrict__ x, const float *__restrict__ x0,
                                                        const float *__restrict__ noise, int64_t n, float c1, float c2,
                                                        float sigma, uint64_t seed, uint64_t step, uint64_t g0) {
    const int64_t n4 = (n + 3) >> 2, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n4; g += stride) {
        const int64_t i = g * 4;
        float4 e;
        if constexpr (GEN) e = randn4(seed, step, g0 + (uint64_t)g);
        if (i + 3 < n) {
            float4 xv = *reinterpret_cast<float4 *>(x + i);
            const float4 pv = *reinterpret_cast<const float4 *>(x0 + i);
            if constexpr (!GEN) e = *reinterpret_cast<const float4 *>(noise + i);
            *reinterpret_cast<float4 *>(x + i) = posterior4(c1, c2, sigma, pv, xv, e);
        } else {
            const float ev[4] = {e.x, e.y, e.z, e.w};
            for (int k = 0; k < 4 && i + k < n; ++k)
                x[i + k] = posterior1(c1, c2, sigma, x0[i + k], x[i + k], GEN ? ev[k] : noise[i + k]);
        }
    }
}

__global__ __launch_bounds__(256) void randn_kernel(float *__restrict__ out, int64_t n, uint64_t seed, uint64_t step, uint64_t g0) {
    const int64_t n4 = (n + 3) >> 2, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n4; g += stride) {
        const float4 e = randn4(seed, step, g0 + (uint64_t)g);
        const float ev[4] = {e.x, e.y, e.z, e.w};
        for (int k = 0; k < 4 && g * 4 + k < n; ++k) out[g * 4 + k] = ev[k];
    }
}

// Device-parameterised step update: every per-step scalar comes from HBM (coefficient table row of the current
// timestep, loop index and seed from the sampler state), so ONE captured hipGraph of [denoiser forward -> this ->
// advance] replays for every plain step of the loop.  state = {t, loop_index, seed, -, -, -, elem0}; table[t] = {c1, c2, sigma, t/1000};
// elem0 (a multiple of 4) = position of x[0] inside the whole sample: a chain of a split batch draws the whole batch's noise.
__global__ __launch_bounds__(256) void posterior_dev_kernel(float *__restrict__ x, const float *__restrict__ x0,
                                                            const float *__restrict__ gt, const uint8_t *__restrict__ mask,
                                                            int64_t n, const float *__restrict__ table,
                                                            int64_t *__restrict__ state, int64_t *__restrict__ ts, int B) {
    const int64_t t = state[0];
    const uint64_t it = (uint64_t)state[1], seed = (uint64_t)state[2], g0 = (uint64_t)state[6] >> 2;
    const float c1 = table[t * 4], c2 = table[t * 4 + 1], sigma = table[t * 4 + 2];
    const int64_t n4 = (n + 3) >> 2, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n4; g += stride) {
        const int64_t i = g * 4;
        const float4 e = randn4(seed, it, g0 + (uint64_t)g);
        const float ev[4] = {e.x, e.y, e.z, e.w};
        if (i + 3 < n) {
            float4 xv = *reinterpret_cast<float4 *>(x + i);
            float4 pv = *reinterpret_cast<const float4 *>(x0 + i);
            if (mask) {
                const uchar4 m = *reinterpret_cast<const uchar4 *>(mask + i);
                const float4 gv = *reinterpret_cast<const float4 *>(gt + i);
                pv.x = m.x ? gv.x : pv.x; pv.y = m.y ? gv.y : pv.y; pv.z = m.z ? gv.z : pv.z; pv.w = m.w ? gv.w : pv.w;
            }
            *reinterpret_cast<float4 *>(x + i) = posterior4(c1, c2, sigma, pv, xv, e);
        } else {
            for (int k = 0; k < 4 && i + k < n; ++k) {
                const float p = (mask && mask[i + k]) ? gt[i + k] : x0[i + k];
                x[i + k] = posterior1(c1, c2, sigma, p, x[i + k], ev[k]);
            }
        }
    }
    // advance (t -= 1, loop index += 1, ts[b] = max(t, 0)) by the LAST workgroup to arrive (philox.h)
    if (ts) sampler_advance_last(state, ts, B, gridDim.x, t, it);
}

__global__ __launch_bounds__(256) void advance_kernel(int64_t *__restrict__ state, int64_t *__restrict__ ts, int B) {
    const int64_t t = state[0] - 1;
    __syncthreads();                                        // every thread has read the old value (single workgroup)
    for (int b = threadIdx.x; b < B; b += 256) ts[b] = t < 0 ? 0 : t;
    if (threadIdx.x == 0) { state[0] = t; state[1] += 1; }
}

inline unsigned grid_for(int64_t work) {
    int64_t b = idf_cdiv(work, 256);
    return (unsigned)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace

extern "C" int interdiff_inpaint(float *x0, const float *gt, const uint8_t *mask, int64_t n, void *stream) {
    if (!x0 || !gt || !mask || n < 0) return IDF_E_INVAL;
    if (n == 0) return IDF_OK;
    idf_prof_mark(IDF_K_INPAINT, idf_stream(stream));
    hipLaunchKernelGGL(inpaint_kernel, dim3(grid_for(n)), dim3(256), 0, idf_stream(stream), x0, gt, mask, n);
    idf_prof_mark(-1, idf_stream(stream));
    IDF_CHECK_LAUNCH();
    return IDF_OK;
}

// elem0 (a multiple of 4) = position of x[0] inside the tensor whose noise stream is drawn: a shard of a batch (one rank's clips, one
// chain of a split batch) draws the WHOLE batch's noise at its own elements, so a sharded run equals the unsharded one bit for bit
extern "C" int interdiff_posterior_step_at(float *x, const float *x0, const float *noise, int64_t n, float c1, float c2,
                                           float sigma, uint64_t seed, uint64_t step_index, uint64_t elem0, void *stream) {
    if (!x || !x0 || n < 0 || (elem0 & 3)) return IDF_E_INVAL;
    if (n == 0) return IDF_OK;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(x0) | reinterpret_cast<uintptr_t>(noise)) & 15)
        return IDF_E_INVAL;
    const unsigned g = grid_for((n + 3) / 4);
    idf_prof_mark(IDF_K_POSTERIOR, idf_stream(stream));
    if (noise)
        hipLaunchKernelGGL((posterior_kernel<false>), dim3(g), dim3(256), 0, idf_stream(stream), x, x0, noise, n, c1, c2,
                           sigma, seed, step_index, elem0 >> 2);
    else
        hipLaunchKernelGGL((posterior_kernel<true>), dim3(g), dim3(256), 0, idf_stream(stream), x, x0, noise, n, c1, c2, sigma,
                           seed, step_index, elem0 >> 2);
    idf_prof_mark(-1, idf_stream(stream));
    IDF_CHECK_LAUNCH();
    return IDF_OK;
}

extern "C" int interdiff_posterior_step(float *x, const float *x0, const float *noise, int64_t n, float c1, float c2,
                                        float sigma, uint64_t seed, uint64_t step_index, void *stream) {
    return interdiff_posterior_step_at(x, x0, noise, n, c1, c2, sigma, seed, step_index, 0, stream);
}

extern "C" int interdiff_posterior_step_dev(float *x, const float *x0, const float *gt, const uint8_t *mask, int64_t n,
                                            const float *table, int64_t *state, int64_t *ts, int32_t B, void *stream) {
    if (!x || !x0 || !table || !state || n < 0 || (mask && !gt) || (ts && B <= 0)) return IDF_E_INVAL;   // (state[6] & 3 cannot be checked here without a sync: the host mirror asserts it)
    if (n == 0) return IDF_OK;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(x0) | reinterpret_cast<uintptr_t>(gt)) & 15) return IDF_E_INVAL;
    if (reinterpret_cast<uintptr_t>(mask) & 3) return IDF_E_INVAL;
    idf_prof_mark(IDF_K_POSTERIOR, idf_stream(stream));
    hipLaunchKernelGGL(posterior_dev_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, idf_stream(stream), x, x0, gt, mask, n, table,
                       state, ts, B);
    idf_prof_mark(-1, idf_stream(stream));
    IDF_CHECK_LAUNCH();
    return IDF_OK;
}

extern "C" int interdiff_sampler_advance(int64_t *state, int64_t *ts, int32_t B, void *stream) {
    if (!state || !ts || B <= 0) return IDF_E_INVAL;
    hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(256), 0, idf_stream(stream), state, ts, B);
    IDF_CHECK_LAUNCH();
    return IDF_OK;
}

extern "C" int interdiff_randn_at(float *out, int64_t n, uint64_t seed, uint64_t step_index, uint64_t elem0, void *stream) {
    if (!out || n < 0 || (elem0 & 3)) return IDF_E_INVAL;
    if (n == 0) return IDF_OK;
    hipLaunchKernelGGL(randn_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, idf_stream(stream), out, n, seed, step_index, elem0 >> 2);
    IDF_CHECK_LAUNCH();
    return IDF_OK;
}

extern "C" int interdiff_randn(float *out, int64_t n, uint64_t seed, uint64_t step_index, void *stream) {
    return interdiff_randn_at(out, n, seed, step_index, 0, stream);
}

#define IDF_ABI_VERSION 16
#define IDF_STR_(x) #x
#define IDF_STR(x) IDF_STR_(x)
extern "C" int interdiff_abi_version(void) { return IDF_ABI_VERSION; }
extern "C" const char *interdiff_build_info(void) { return "interdiff_hip gfx950 (hipcc, fp32 MFMA) abi " IDF_STR(IDF_ABI_VERSION); }
