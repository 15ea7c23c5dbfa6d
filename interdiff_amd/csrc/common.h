// Shared device/host helpers for the gfx950 kernels (wave = 64 lanes, always).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/interdiff_hip.h"

#define IDF_WAVE 64

#define IDF_CHECK_LAUNCH()                                                                        \
    do {                                                                                          \
        const hipError_t idf_e_ = hipGetLastError();                                              \
        if (idf_e_ != hipSuccess) {                                                               \
            fprintf(stderr, "interdiff_hip: %s (%s:%d)\n", hipGetErrorString(idf_e_), __FILE__, __LINE__); \
            return IDF_E_LAUNCH;                                                                  \
        }                                                                                         \
    } while (0)

static inline hipStream_t idf_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }
static inline int64_t idf_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t idf_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// profiling hook (prof.hip): no-op unless interdiff_profile_begin() armed it
extern bool g_idf_prof_on;
void idf_prof_mark_slow(int kind, hipStream_t s);
static inline void idf_prof_mark(int kind, hipStream_t s) { if (g_idf_prof_on) idf_prof_mark_slow(kind, s); }

extern int g_idf_tune[];          // denoiser.hip: tile-configuration overrides (interdiff_tune)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- wave-wide reductions (all 64 lanes participate, result in every lane) -------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// Branch-free erf-GELU for GEMM epilogues: erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7; the GELU value is
// within 5e-7 absolute of the exact one over the whole fp32 range, torch's own fp32 gelu is within 1.2e-6).
__device__ __forceinline__ float gelu_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __frcp_rn(1.0f + 0.3275911f * z);
    const float p = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
    const float er = 1.0f - p * __expf(-(z * z));
    return 0.5f * x * (1.0f + copysignf(er, x));
}

// LayerNorm statistics of a 256-wide row held 4 values per lane by one wave.
__device__ __forceinline__ void ln_row_stats(const float4 v, float &mean, float &rstd) {
    mean = wave_sum(v.x + v.y + v.z + v.w) * (1.0f / 256.0f);
    const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
    const float var = wave_sum(a * a + b * b + c * c + d * d) * (1.0f / 256.0f);
    rstd = 1.0f / sqrtf(var + 1e-5f);
}
