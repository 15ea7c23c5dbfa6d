// Shared device/host helpers for the gfx950 kernels (wave = 64 lanes, always).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>
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

// Opt a kernel into more than 64 KiB of dynamic LDS.  The attribute belongs to (kernel, device), so `done` is a per-kernel bit
// mask of the devices it has been set on: an idempotent, lock-free cache (a racing thread merely sets the attribute twice),
// not state -- a process that drives several GPUs gets the opt-in on each of them.
static inline int idf_opt_in_lds(const void *fn, int bytes, std::atomic<uint64_t> &done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return IDF_E_LAUNCH;
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return IDF_OK;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return IDF_E_LAUNCH;
    done.fetch_or(bit, std::memory_order_release);
    return IDF_OK;
}

// ---- EXCLUSIVE CU, verified (round 5) ------------------------------------------------------------------------------------------
// Every kernel of this library that issues the f16 MFMA claims its CU for itself (ffn_h2.h "EXCLUSIVE CU": next to such a kernel, waves of OTHER kernels on
// the same CU have been seen to compute wrong values): all 160 KiB of LDS (static + the dynamic bytes its launcher adds) and the whole register file
// (asm clobbers v255 / a255).  That used to be a property nobody checked.  idf_exclusive_cu() asks the runtime, once per (kernel, device):
//   * hipOccupancyMaxActiveBlocksPerMultiprocessor(kernel, threads, dyn) == 1            -- no second workgroup of the same kernel,
//   * static + dynamic LDS == 160 KiB                                                     -- no LDS-using workgroup of any other kernel,
//   * the compiler really allocated >= 256 registers per lane (hipFuncAttributes.numRegs) -- with threads / 256 waves per SIMD no LDS-free wave fits either
// and returns the dynamic LDS to launch with, or -1: the caller then takes its fp32-MFMA kernel instead (csrc/denoiser.hip) -- a configuration where
// the claim does not hold (LDS partitioning, a compiler that stops honouring the clobber) degrades to the exact arithmetic, never to an unprotected launch
// and never to IDF_E_LAUNCH.  Every verdict is kept in a table that interdiff_exclusive_cu_report() prints (tests, bench.py).
struct idf_excl_entry {
    const char *name;
    int threads, num_regs, static_lds, dyn_lds, blocks_per_cu, ok, dev, denied;
};
// one table per process (C++17 inline variables: every translation unit, and every probe build that includes a kernel file whole, sees the same one); not on
// any hot path: touched once per (kernel, device)
inline std::mutex g_idf_excl_mu;
inline std::vector<idf_excl_entry> g_idf_excl;
inline void idf_excl_record(const idf_excl_entry &e) {
    std::lock_guard<std::mutex> lk(g_idf_excl_mu);
    for (idf_excl_entry &o : g_idf_excl)
        if (o.dev == e.dev && std::string(o.name) == e.name) { o = e; return; }
    g_idf_excl.push_back(e);
}
inline int idf_excl_report(char *buf, int cap) {        // the table as text; returns the number of non-exclusive rows
    std::lock_guard<std::mutex> lk(g_idf_excl_mu);
    int bad = 0, n = 0;
    buf[0] = 0;
    for (const idf_excl_entry &e : g_idf_excl) {
        bad += e.ok ? 0 : 1;
        const int w = snprintf(buf + n, (size_t)(cap - n), "%-52s dev %d  threads %3d  regs %3d  lds %6d + %6d  workgroups_per_cu %d  %s\n", e.name, e.dev, e.threads, e.num_regs,
                               e.static_lds, e.dyn_lds, e.blocks_per_cu, e.ok ? "exclusive" : (e.denied ? "DENIED (debug deny list) -> fp32 kernel" : "NOT exclusive -> fp32 kernel"));
        if (w < 0 || w >= cap - n) break;
        n += w;
    }
    return bad;
}
// DEBUG deny list (round 6; interdiff_debug_deny_exclusive, tests only -- empty in the product): kernels whose name contains one of the comma-separated patterns are treated as if
// they did not get their CU, so that the fp32 fallbacks behind every split-f16 launcher can be EXECUTED on a device where every claim holds (MI355X: all of them).  Setting the list
// bumps a generation counter; a launcher's cached verdict of an older generation is dropped and asked again.
inline std::string g_idf_excl_deny;                     // guarded by g_idf_excl_mu
inline std::atomic<uint64_t> g_idf_excl_gen{1};
inline void idf_excl_set_deny(const char *patterns) {
    std::lock_guard<std::mutex> lk(g_idf_excl_mu);
    g_idf_excl_deny = patterns ? patterns : "";
    g_idf_excl.clear();
    g_idf_excl_gen.fetch_add(1, std::memory_order_acq_rel);
}
inline bool idf_excl_denied(const char *name) {
    std::lock_guard<std::mutex> lk(g_idf_excl_mu);
    if (g_idf_excl_deny.empty()) return false;
    const std::string nm(name);
    size_t a = 0;
    while (a <= g_idf_excl_deny.size()) {
        size_t b = g_idf_excl_deny.find(',', a);
        if (b == std::string::npos) b = g_idf_excl_deny.size();
        if (b > a && nm.find(g_idf_excl_deny.substr(a, b - a)) != std::string::npos) return true;
        a = b + 1;
    }
    return false;
}
struct idf_excl_cache {
    std::atomic<uint64_t> yes{0}, no{0}, gen{0};
    std::atomic<int> dyn{-1};
};
constexpr int IDF_CU_LDS_BYTES = 160 * 1024;
constexpr int IDF_NOT_EXCLUSIVE = 1;                    // internal (positive) return of the split-f16 launchers: "take the fp32 kernel"; never leaves the library (idf_public_rc)
static inline int idf_public_rc(int rc) { return rc == IDF_NOT_EXCLUSIVE ? IDF_E_LAUNCH : rc; }      // at a public entry point: a launcher that had no fp32 fallback left
static inline int idf_exclusive_cu(const void *fn, const char *name, int threads, idf_excl_cache &c) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    const uint64_t bit = 1ull << (dev & 63), gen = g_idf_excl_gen.load(std::memory_order_acquire);
    if (c.gen.load(std::memory_order_acquire) != gen) {  // (a new deny list: forget what was cached under the old one; racing threads both reset, both ask again)
        c.yes.store(0, std::memory_order_release);
        c.no.store(0, std::memory_order_release);
        c.gen.store(gen, std::memory_order_release);
    }
    if (c.yes.load(std::memory_order_acquire) & bit) return c.dyn.load(std::memory_order_acquire);
    if (c.no.load(std::memory_order_acquire) & bit) return -1;
    idf_excl_entry e{name, threads, -1, -1, -1, -1, 0, dev, 0};
    hipFuncAttributes at;
    if (hipFuncGetAttributes(&at, fn) == hipSuccess) {
        e.num_regs = at.numRegs;
        e.static_lds = (int)at.sharedSizeBytes;
        e.dyn_lds = IDF_CU_LDS_BYTES - e.static_lds;
        if (e.dyn_lds >= 0 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, e.dyn_lds) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&e.blocks_per_cu, fn, threads, (size_t)e.dyn_lds) == hipSuccess)
            e.ok = e.blocks_per_cu == 1 && e.num_regs >= (threads > 512 ? 128 : 256) ? 1 : 0;      // (a 1024-thread workgroup is four waves per SIMD: 4 x 128 = the file)
    }
    (void)hipGetLastError();                             // a refused attribute must not surface as the next launch's error
    if (e.ok && idf_excl_denied(name)) { e.ok = 0; e.denied = 1; }
    idf_excl_record(e);
    if (e.ok) {
        c.dyn.store(e.dyn_lds, std::memory_order_release);
        c.yes.fetch_or(bit, std::memory_order_release);
        return e.dyn_lds;
    }
    c.no.fetch_or(bit, std::memory_order_release);
    if (e.denied) fprintf(stderr, "interdiff_hip: %s is on the DEBUG deny list: the fp32-MFMA kernel runs instead\n", name);
    else fprintf(stderr, "interdiff_hip: %s does not get its CU to itself on device %d (regs %d, LDS %d + %d, %d workgroups per CU): the fp32-MFMA kernel runs instead\n",
                 name, dev, e.num_regs, e.static_lds, e.dyn_lds, e.blocks_per_cu);
    return -1;
}

// CUs of the current device (256 on MI355X): how many one-per-CU workgroups one round holds (launch geometry choices only; cached per device)
static inline int idf_cu_count() {
    static std::atomic<int> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    int n = cached[dev & 63].load(std::memory_order_acquire);
    if (n <= 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev & 63].store(n, std::memory_order_release);
    }
    return n;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- LDS-DMA (global -> LDS without a VGPR round trip) as INLINE ASM ----------------------------------------------------------
// hipcc's waitcnt pass cannot tell which LDS bytes a `global_load_lds` writes, so with the builtin it drains vmcnt to 0 before the
// first ds_read that follows ANY pending DMA -- a software pipeline that keeps chunks in flight across iterations silently degrades
// to "issue, wait for everything, compute" (round 1's GEMMs did: their ISA shows `s_waitcnt vmcnt(0)` right after the issue).
// Inline asm is invisible to that pass: the kernel owns the bookkeeping and must cover every DMA with its own counted
// `s_waitcnt vmcnt(N)` (+ an s_barrier before other waves read the bytes).  The compiler's waits for ITS OWN loads stay correct:
// memory returns in issue order, so a DMA it does not know about can only make its waits stricter, never looser.
// Every wave instruction moves 64 lanes x 16 B to LDS bytes [lds_base + 16*lane); lds_base (and gbase) must be wave-uniform.
typedef __attribute__((address_space(3))) float idf_lds_float;
__device__ __forceinline__ uint32_t idf_lds_addr(const float *p) {       // LDS byte address of a pointer into __shared__ memory
    return __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(idf_lds_float *)p);
}
__device__ __forceinline__ const float *idf_uniform_ptr(const float *p) {
    const uint64_t v = (uint64_t)(uintptr_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<const float *>(((uint64_t)hi << 32) | lo);
}
// global address = gbase (SGPR pair) + voff_bytes (per lane)
__device__ __forceinline__ void idf_dma16_s(const float *gbase, uint32_t voff_bytes, uint32_t lds_base) {
    gbase = idf_uniform_ptr(gbase);                                      // both are wave-uniform by contract: pin them to SGPRs
    lds_base = __builtin_amdgcn_readfirstlane(lds_base);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff_bytes), "s"(gbase), "s"(lds_base) : "memory", "m0");
}
// global address = gptr (per lane, 64-bit)
__device__ __forceinline__ void idf_dma16_v(const float *gptr, uint32_t lds_base) {
    lds_base = __builtin_amdgcn_readfirstlane(lds_base);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(lds_base) : "memory", "m0");
}

// 16-byte WRITE-THROUGH store (sc0 sc1): the line goes to the memory side right away instead of sitting dirty in this XCD's L2
// until the end-of-kernel write-back.  For tensors that the NEXT kernel reads from other XCDs (every layer output here) this takes
// the flush off the kernel boundary (MI355X_MICROARCH.md: a boundary costs + bytes-left-dirty / 6 TB/s) and overlaps it with the
// rest of the launch.  Invisible to the compiler's vmcnt bookkeeping like every asm memory op: only for data this kernel never
// reads back.
#ifndef IDF_WT_MODE
#define IDF_WT_MODE 0             // 0: system-scope write-through (shipped); 1: agent scope (sc1); 2: plain stores -- A/B builds only (tools/wt_mode_ab.sh)
#endif
#if IDF_WT_MODE == 0
#define IDF_WT_BITS " sc0 sc1"
#elif IDF_WT_MODE == 1
#define IDF_WT_BITS " sc1"
#else
#define IDF_WT_BITS ""
#endif
__device__ __forceinline__ void idf_store4_wt(float *p, const float v) {
    asm volatile("global_store_dword %0, %1, off" IDF_WT_BITS ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void idf_store16_wt(float *p, const float4 v) {
    const f32x4 t = {v.x, v.y, v.z, v.w};
    // s_nop: a store of more than 64 bits reads its data registers AFTER issue -- a VALU write to them needs wait states in
    // between.  The compiler's hazard recognizer pads its own stores; it cannot see into this asm (tools/lnlin_probe.hip caught
    // the next loop iteration's index landing in .x of ~0.1 % of the stores).
    asm volatile("global_store_dwordx4 %0, %1, off" IDF_WT_BITS "\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
}

// Kernel arguments live in the kernarg segment and reach SGPRs by s_load; the compiler sinks each s_load (+ s_waitcnt) next to its
// first use, often behind a branch, so a kernel with a dozen pointer arguments starts with a CHAIN of scalar-cache misses (the row
// block: ~5 k cycles before its first vector load was issued, tools/rowblock_probe.hip).  Naming the arguments in one empty asm
// at kernel entry forces all of them into SGPRs there: the s_loads go out back to back and the misses overlap.
template <class T>
__device__ __forceinline__ void idf_arg_now(T a) { asm volatile("" ::"s"(a)); }
template <class... A>
__device__ __forceinline__ void idf_args_now(A... a) { (idf_arg_now(a), ...); }

// exact nearest-vertex scan with block culling (correction.hip) as the post-optimisation uses it (optimize.hip), and its contact-radius mask
int idf_point_order(hipStream_t s, const float *obj_points, int B, int P, int32_t *porder);
int idf_nn_scan_opt(hipStream_t s, int64_t N, int frames_per_clip, const float *verts, int V, const float *pts_frame, const float *obj_points, int P,
                    const int32_t *porder, const idf_correction_ctx *c, int32_t *yidx);
int idf_near_mask_opt(hipStream_t s, int64_t N, int frames_per_clip, const float *verts, int V, const float *pts_frame, int P, const int32_t *porder,
                      const int32_t *vorder, float *psort, float *pbox, int32_t *near);

// the contact-frame predictor as the correction hook runs it (objproj.hip / objproj.h): its three stacks ride in the contact scan's launch (csrc/objproj.h objproj_body<1> ->
// keep [B][idf_objproj_keep_floats()]), then the selected node's IDCT once the contact labels exist.  Together bit-identical to interdiff_objprojector_sample.
size_t idf_objproj_keep_floats();
int idf_objproj_check(const idf_objproj *op);
int idf_objproj_pick(const idf_objproj *op, const float *keep, const int32_t *contact, int B, float *out, hipStream_t s);

// profiling hook (prof.hip): no-op unless interdiff_profile_begin() armed it
extern bool g_idf_prof_on;
void idf_prof_mark_slow(int kind, hipStream_t s);
static inline void idf_prof_mark(int kind, hipStream_t s) { if (g_idf_prof_on) idf_prof_mark_slow(kind, s); }


typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- cross-lane reductions on the DPP path (no LDS-crossbar permutes) --------------------------------------
// row16_*: all-reduce inside each 16-lane DPP row by rotations (row_ror:8,4,2,1); every lane of the row gets the result.
#define IDF_DPP_ROR(v, n) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + (n), 0xf, 0xf, false))
__device__ __forceinline__ float row16_sum(float v) {
    v += IDF_DPP_ROR(v, 8);
    v += IDF_DPP_ROR(v, 4);
    v += IDF_DPP_ROR(v, 2);
    v += IDF_DPP_ROR(v, 1);
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, IDF_DPP_ROR(v, 8));
    v = fmaxf(v, IDF_DPP_ROR(v, 4));
    v = fmaxf(v, IDF_DPP_ROR(v, 2));
    v = fmaxf(v, IDF_DPP_ROR(v, 1));
    return v;
}
#define IDF_LANE(v, l) __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l))
// wave-wide (all 64 lanes participate, result in every lane): four row results combined through scalar registers
__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    return (IDF_LANE(v, 0) + IDF_LANE(v, 16)) + (IDF_LANE(v, 32) + IDF_LANE(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = row16_max(v);
    return fmaxf(fmaxf(IDF_LANE(v, 0), IDF_LANE(v, 16)), fmaxf(IDF_LANE(v, 32), IDF_LANE(v, 48)));
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// Branch-free erf-GELU for GEMM epilogues: erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7; the GELU value is
// within 5e-7 absolute of the exact one over the whole fp32 range, torch's own fp32 gelu is within 1.2e-6).
__device__ __forceinline__ float gelu_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);      // v_rcp_f32 (1 ulp): __frcp_rn expands to the 10-instruction IEEE division
    const float p = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
    const float er = 1.0f - p * __expf(-(z * z));
    return 0.5f * x * (1.0f + copysignf(er, x));
}

// LayerNorm statistics of a 256-wide row held 4 values per lane by one wave.
__device__ __forceinline__ void ln_row_stats(const float4 v, float &mean, float &rstd) {
    mean = wave_sum(v.x + v.y + v.z + v.w) * (1.0f / 256.0f);
    const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
    const float var = wave_sum(a * a + b * b + c * c + d * d) * (1.0f / 256.0f);
    rstd = __builtin_amdgcn_rsqf(var + 1e-5f);          // v_rsq_f32 (1 ulp) instead of IEEE sqrt + IEEE division (~25 dependent instructions)
}

// LayerNorm of 256-wide rows held by 16-lane groups: lane l16 of the group owns the four float4 chunks
// {l16, 16+l16, 32+l16, 48+l16} of its row (so that each load/store instruction of a group covers 256 contiguous bytes).
struct Row16 {
    float4 c[4];
};
__device__ __forceinline__ void ln_row16(Row16 &r, const float *__restrict__ w, const float *__restrict__ b, int l16) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += (r.c[i].x + r.c[i].y) + (r.c[i].z + r.c[i].w);
    const float mean = row16_sum(s) * (1.0f / 256.0f);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a0 = r.c[i].x - mean, a1 = r.c[i].y - mean, a2 = r.c[i].z - mean, a3 = r.c[i].w - mean;
        q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
    }
    const float rstd = __builtin_amdgcn_rsqf(row16_sum(q) * (1.0f / 256.0f) + 1e-5f);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 g = *reinterpret_cast<const float4 *>(w + (i * 16 + l16) * 4), be = *reinterpret_cast<const float4 *>(b + (i * 16 + l16) * 4);
        r.c[i].x = (r.c[i].x - mean) * rstd * g.x + be.x;
        r.c[i].y = (r.c[i].y - mean) * rstd * g.y + be.y;
        r.c[i].z = (r.c[i].z - mean) * rstd * g.z + be.z;
        r.c[i].w = (r.c[i].w - mean) * rstd * g.w + be.w;
    }
}
__device__ __forceinline__ void row16_load(Row16 &r, const float *__restrict__ row, int l16) {
#pragma unroll
    for (int i = 0; i < 4; ++i) r.c[i] = *reinterpret_cast<const float4 *>(row + (i * 16 + l16) * 4);
}
// A layer output is either one [N,256] matrix (NP = 1) or the IDF_FFN_SLICES partial slabs the fused FFN leaves behind (ffn.h):
// element = ((((s0 + s1) + s2) + s3) + s4), slabs `stride` floats apart.  Summed in this fixed order by every reader: deterministic.
// NP is a COMPILE-TIME constant on purpose: with a run-time slab count the compiler parks a `s_waitcnt vmcnt(0)` in front of the
// branch and every slab of every row becomes its own round trip to memory (a row-block prologue of 16 dependent fetches).
template <int NP>
__device__ __forceinline__ float4 ld4_sum(const float *__restrict__ p, size_t stride) {
    float4 t[NP];
#pragma unroll
    for (int s = 0; s < NP; ++s) t[s] = *reinterpret_cast<const float4 *>(p + s * stride);      // all loads in flight before the first add
    float4 v = t[0];
#pragma unroll
    for (int s = 1; s < NP; ++s) { v.x += t[s].x; v.y += t[s].y; v.z += t[s].z; v.w += t[s].w; }
    return v;
}
// split form: request (all 4*NP loads, nothing consumed) ... reduce later, so that several rows' fetches share one trip to memory
template <int NP>
struct Row16Raw {
    float4 t[4][NP];
    __device__ __forceinline__ void request(const float *__restrict__ row, int l16, size_t stride) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int s = 0; s < NP; ++s) t[i][s] = *reinterpret_cast<const float4 *>(row + s * stride + (i * 16 + l16) * 4);
    }
    __device__ __forceinline__ void reduce(Row16 &r) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float4 v = t[i][0];
#pragma unroll
            for (int s = 1; s < NP; ++s) { v.x += t[i][s].x; v.y += t[i][s].y; v.z += t[i][s].z; v.w += t[i][s].w; }
            r.c[i] = v;
        }
    }
};
template <int NP>
__device__ __forceinline__ void row16_load_sum(Row16 &r, const float *__restrict__ row, int l16, size_t stride) {
    Row16Raw<NP> raw;
    raw.request(row, l16, stride);
    raw.reduce(r);
}
__device__ __forceinline__ void row16_store(const Row16 &r, float *__restrict__ row, int l16) {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<float4 *>(row + (i * 16 + l16) * 4) = r.c[i];
}
// (global rows that other XCDs read next: write-through, see idf_store16_wt)
__device__ __forceinline__ void row16_store_wt(const Row16 &r, float *__restrict__ row, int l16) {
#pragma unroll
    for (int i = 0; i < 4; ++i) idf_store16_wt(row + (i * 16 + l16) * 4, r.c[i]);
}
__device__ __forceinline__ void row16_zero(Row16 &r) {
#pragma unroll
    for (int i = 0; i < 4; ++i) r.c[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}
