// Per-kernel-kind timing with HIP events on the launch stream (bench.py's roofline block only).
#include "common.h"
#include <vector>

bool g_idf_prof_on = false;

namespace {
std::vector<hipEvent_t> g_ev;
std::vector<int> g_kind;
size_t g_cap = 0;
}  // namespace

void idf_prof_mark_slow(int kind, hipStream_t s) {
    if (g_ev.size() >= g_cap) return;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, s);
    g_ev.push_back(e);
    g_kind.push_back(kind);
}

extern "C" int interdiff_profile_begin(int32_t capacity) {
    if (capacity <= 1) return IDF_E_INVAL;
    for (hipEvent_t e : g_ev) (void)hipEventDestroy(e);
    g_ev.clear();
    g_kind.clear();
    g_cap = (size_t)capacity;
    g_ev.reserve(g_cap);
    g_kind.reserve(g_cap);
    g_idf_prof_on = true;
    return IDF_OK;
}

extern "C" int interdiff_profile_end(double *ms_per_kind, int64_t *count_per_kind) {
    if (!ms_per_kind || !count_per_kind) return IDF_E_INVAL;
    g_idf_prof_on = false;
    for (int k = 0; k < IDF_K_COUNT; ++k) { ms_per_kind[k] = 0.0; count_per_kind[k] = 0; }
    if (g_ev.empty()) return IDF_OK;
    // closing event on the null-stream-synchronising path: wait for the device, then read pairs
    if (hipDeviceSynchronize() != hipSuccess) return IDF_E_LAUNCH;
    for (size_t i = 0; i + 1 < g_ev.size(); ++i) {
        float ms = 0.f;
        if (g_kind[i] < 0) continue;                       // -1 = "end of a call" marker: gap not attributed
        if (hipEventElapsedTime(&ms, g_ev[i], g_ev[i + 1]) == hipSuccess) {
            ms_per_kind[g_kind[i]] += ms;
            count_per_kind[g_kind[i]] += 1;
        }
    }
    for (hipEvent_t e : g_ev) (void)hipEventDestroy(e);
    g_ev.clear();
    g_kind.clear();
    return IDF_OK;
}

// ---- LDS sentinel (tools/lds_sentinel_probe.py): a one-wave workgroup fills its 6 KiB of LDS with a pattern and keeps re-reading it;
// any word that changes was written by somebody else.  Diagnostic only.
namespace {
__global__ __launch_bounds__(64) void lds_sentinel_kernel(uint32_t *__restrict__ out, int spin) {
    __shared__ uint32_t buf[1536];
    const int j = threadIdx.x;
    for (int i = j; i < 1536; i += 64) buf[i] = 0xA5000000u | (uint32_t)i;
    __syncthreads();
    for (int it = 0; it < spin; ++it) {
        for (int i = j; i < 1536; i += 64) {
            const uint32_t v = buf[i];
            if (v != (0xA5000000u | (uint32_t)i)) {
                const uint32_t slot = atomicAdd(out, 1u);
                if (slot < 1000) {
                    out[4 + 4 * slot] = blockIdx.x;
                    out[5 + 4 * slot] = (uint32_t)i;
                    out[6 + 4 * slot] = v;
                    out[7 + 4 * slot] = (uint32_t)it;
                }
                buf[i] = 0xA5000000u | (uint32_t)i;
            }
        }
        __syncthreads();
    }
}
}  // namespace

extern "C" int interdiff_debug_lds_sentinel(uint32_t *out, int32_t n_wg, int32_t spin, void *stream) {
    if (!out || n_wg <= 0) return IDF_E_INVAL;
    hipLaunchKernelGGL(lds_sentinel_kernel, dim3((unsigned)n_wg), dim3(64), 0, idf_stream(stream), out, spin);
    IDF_CHECK_LAUNCH();
    return IDF_OK;
}

// ---- f16-MFMA "aggressor" of the co-residency probes (tools/coresidency_probe.hip, tools/hook_stage_probe.py; DESIGN.md "exclusive CU"): a kernel that only
// loops over v_mfma_f32_16x16x32_f16 on register operands and streaming global loads, with small LDS / register needs so that it shares CUs with whatever
// runs on another stream -- the one pattern that has been seen to corrupt a co-resident wave's VALU results.  Diagnostic only: nothing in the product launches it.
namespace {
typedef _Float16 idf_h8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void f16_aggressor_kernel(const float4 *__restrict__ src, size_t n4, float *__restrict__ sink, int iters, int with_loads) {
    const int tid = threadIdx.x;
    idf_h8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.01f * ((tid + e) % 37) - 0.15f); b[e] = (_Float16)(0.02f * ((tid * 3 + e) % 29) - 0.2f); }
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    size_t p = ((size_t)blockIdx.x * 256 + tid) % n4;
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, a, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, b, c3, 0, 0, 0);
        if (with_loads) {
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
}  // namespace

extern "C" int interdiff_debug_f16_aggressor(const float *src, size_t n_floats, float *sink, int32_t iters, int32_t grid, int32_t with_loads, void *stream) {
    if (!src || !sink || n_floats < 4096 || iters <= 0 || grid <= 0) return IDF_E_INVAL;
    hipLaunchKernelGGL(f16_aggressor_kernel, dim3((unsigned)grid), dim3(256), 0, idf_stream(stream), reinterpret_cast<const float4 *>(src), n_floats / 4, sink, iters, with_loads);
    IDF_CHECK_LAUNCH();
    return IDF_OK;
}
