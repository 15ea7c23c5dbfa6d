// Go / no-go measurement for "a QaN layer as ONE persistent launch"  on this workload's own geometry, before any product kernel is touched:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-inline-asm -I interdiff_amd/csrc tools/experiments/handoff_probe.hip -o build_tools/handoff_probe
// The seam: rowblock8_kernel (208 workgroups of 8 tokens, ~7 us) hands 1600 x2 rows (1 KiB each) to ffn_h2_kernel (250 workgroups = 50 M tiles of 32 rows x 5 hidden slices, ~11 us),
// which first fills its weight ring (3 x 32 KiB by LDS-DMA) and fetches its 32 rows (32 KiB by LDS-DMA).  What a persistent layer would change: no kernel boundary, no cold start of
// the second kernel, the ring fill issued BEFORE the rows exist (prefetch credit); what it adds: write-through rows -> drained flag per 8-token tile, a poll of the 4-6 tile flags an M
// tile's rows come from, and row loads that must be sc1 (a CU's L1 and another XCD's L2 are never refreshed by a store).  This probe runs exactly that traffic with the two kernels'
// compute replaced by timed waits (the seam does not care what the waves computed), in two forms:
//   A  two launches per layer  (produce; consume)  x 8 layers, stream order                      -- what the product does today
//   B  one launch per layer    (workgroup i: produce tile i, publish, prefetch the ring, poll, fetch rows, "compute") x 8 launches
// and reports us per layer pair for both, plus B's stamps: last needed flag seen, rows landed (from the consumer's entry).  Go: A - B >= 2 us.
#include "common.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

bool g_idf_prof_on = false;
void idf_prof_mark_slow(int, hipStream_t) {}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int T = 100, B = 16, M = B * T, TV = 8, NTILE = (T + TV - 1) / TV, NRB = NTILE * B, NSL = 5, NMT = M / 32, NFFN = NMT * NSL, D = 256;
constexpr int LDS_ALL = 160 * 1024;

__device__ __forceinline__ void busy(long long cycles) {
    const long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < cycles) __builtin_amdgcn_s_sleep(2);
}
__device__ __forceinline__ int xcd_logical(int id, int nwg) {
    const int xq = nwg >> 3, xr = nwg & 7, xcd = id & 7;
    return (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (id >> 3);
}
__device__ __forceinline__ void dma16_sc1(const float *gbase, uint32_t voff, uint32_t lds) {      // coherent (sc0 sc1) LDS-DMA: bypasses this CU's L1 and a stale line of this XCD's L2
    gbase = idf_uniform_ptr(gbase);
    lds = __builtin_amdgcn_readfirstlane(lds);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 sc0 sc1" ::"v"(voff), "s"(gbase), "s"(lds) : "memory", "m0");
}

// the row block's part: `work` cycles of "compute", then 8 rows (one 16-byte write-through store per thread), drained, (fused form) one flag per tile
__device__ __forceinline__ void produce_part(int tile, float *rows, unsigned *flags, unsigned epoch, long long work, bool publish) {
    const int b = tile / NTILE, t0 = (tile - b * NTILE) * TV, tid = threadIdx.x;
    busy(work);
    const int r = tid >> 6, t = t0 + r;
    if (t < T) idf_store16_wt(rows + ((size_t)b * T + t) * D + ((tid & 63) << 2), make_float4((float)epoch, (float)tile, (float)r, 1.f));
    if (publish) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(flags + tile, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// the feed-forward block's part up to "rows are in LDS": ring fill (3 x 32 KiB) -> (fused) poll the tiles the M tile's rows come from -> 32 rows by DMA -> wait; then `work` cycles and a 32 KiB slab store
__device__ __forceinline__ void consume_part(int wg, const float *weights, const float *rows, unsigned *flags, unsigned epoch, float *slabs, long long work, bool poll, long long *stamps,
                                             unsigned *bad) {
    extern __shared__ __attribute__((aligned(1024))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sl = wg % NSL, mt = wg / NSL, m0 = mt * 32;
    const long long t_in = __builtin_readcyclecounter();
    const float *stream = weights + (size_t)sl * 110592;
    const uint32_t ring = idf_lds_addr(smem + 32 * 256);
#pragma unroll
    for (int P = 0; P < 3; ++P)
#pragma unroll
        for (int j = 0; j < 4; ++j) idf_dma16_s(stream, (uint32_t)((P * 32 + wave + 8 * j) * 1024 + (lane << 4)), ring + (uint32_t)((P * 32 + wave + 8 * j) * 1024));
    long long t_flag = t_in;
    if (poll) {
        if (tid == 0) {
            const int r0 = m0, r1 = min(m0 + 31, M - 1);
            const int lo = (r0 / T) * NTILE + (r0 % T) / TV, hi = (r1 / T) * NTILE + (r1 % T) / TV;
            for (int k = lo; k <= hi; ++k) {
                unsigned spins = 0;
                while (__hip_atomic_load(flags + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1u << 22)) { atomicAdd(bad, 1u); break; }        // bounded: a lost flag must not hang the box
                }
            }
            t_flag = __builtin_readcyclecounter();
        }
        __syncthreads();
    }
    const uint32_t xs = idf_lds_addr(smem);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = wave + 8 * j;
        if (poll) dma16_sc1(rows + (size_t)min(m0 + i, M - 1) * D, (uint32_t)(lane << 4), xs + (uint32_t)(i * 1024));
        else idf_dma16_s(rows + (size_t)min(m0 + i, M - 1) * D, (uint32_t)(lane << 4), xs + (uint32_t)(i * 1024));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const long long t_rows = __builtin_readcyclecounter();
    // every row must carry this epoch (a stale line would show the previous one)
    const float v = smem[(tid >> 4) * 256 + (tid & 15) * 4];
    if (m0 + (tid >> 4) < M && v != (float)epoch) atomicAdd(bad + 1, 1u);
    busy(work);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = (tid >> 6) + it * 8;
        if (m0 + row < M) idf_store16_wt(slabs + ((size_t)sl * M + m0 + row) * D + ((tid & 63) << 2), make_float4(v, 0.f, 0.f, 0.f));
    }
    if (stamps && tid == 0) { stamps[wg * 4 + 0] = t_flag - t_in; stamps[wg * 4 + 1] = t_rows - t_in; stamps[wg * 4 + 2] = t_rows - t_flag; }
}

__global__ __launch_bounds__(512) void produce_kernel(float *rows, unsigned epoch, long long work) {
    asm volatile("" ::: "v255");
    produce_part(xcd_logical(blockIdx.x, NRB), rows, nullptr, epoch, work, false);
}
__global__ __launch_bounds__(512) void consume_kernel(const float *weights, const float *rows, unsigned epoch, float *slabs, long long work, long long *stamps, unsigned *bad) {
    asm volatile("" ::: "v255");
    consume_part(xcd_logical(blockIdx.x, NFFN), weights, rows, nullptr, epoch, slabs, work, false, stamps, bad);
}
// one launch: workgroup id < 208 (per XCD: local index < 26) produces a tile first; every workgroup then consumes
__global__ __launch_bounds__(512) void fused_kernel(const float *weights, float *rows, unsigned *flags, unsigned epoch, float *slabs, long long work_p, long long work_c, long long *stamps,
                                                    unsigned *bad) {
    asm volatile("" ::: "v255");
    const int id = blockIdx.x, xcd = id & 7, loc = id >> 3;
    if (loc < NRB / 8) produce_part(xcd * (NRB / 8) + loc, rows, flags, epoch, work_p, true);
    consume_part(xcd_logical(id, NFFN), weights, rows, flags, epoch, slabs, work_c, true, stamps, bad);
}

int main(int argc, char **argv) {
    const long long work_p = argc > 1 ? atoll(argv[1]) : 11000, work_c = argc > 2 ? atoll(argv[2]) : 16000;      // ~ the row block's / the feed-forward block's cycles behind their fetch phases
    float *weights, *rows, *slabs;
    unsigned *flags, *bad;
    long long *stamps;
    CK(hipMalloc(&weights, (size_t)8 * NSL * 110592 * 4));
    CK(hipMemset(weights, 0, (size_t)8 * NSL * 110592 * 4));
    CK(hipMalloc(&rows, (size_t)M * D * 4));
    CK(hipMalloc(&slabs, (size_t)NSL * M * D * 4));
    CK(hipMalloc(&flags, NRB * 4)); CK(hipMemset(flags, 0, NRB * 4));
    CK(hipMalloc(&bad, 8)); CK(hipMemset(bad, 0, 8));
    CK(hipMalloc(&stamps, (size_t)NFFN * 4 * 8));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&produce_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_ALL));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&consume_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_ALL));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_ALL));
    unsigned epoch = 0;
    auto form_a = [&](int layers) {
        for (int l = 0; l < layers; ++l) {
            ++epoch;
            hipLaunchKernelGGL(produce_kernel, dim3(NRB), dim3(512), LDS_ALL, 0, rows, epoch, work_p);
            hipLaunchKernelGGL(consume_kernel, dim3(NFFN), dim3(512), LDS_ALL, 0, weights + (size_t)(l % 8) * NSL * 110592, rows, epoch, slabs, work_c, stamps, bad);
        }
    };
    auto form_b = [&](int layers) {
        for (int l = 0; l < layers; ++l) {
            ++epoch;
            hipLaunchKernelGGL(fused_kernel, dim3(NFFN), dim3(512), LDS_ALL, 0, weights + (size_t)(l % 8) * NSL * 110592, rows, flags, epoch, slabs, work_p, work_c, stamps, bad);
        }
    };
    auto timeit = [&](auto f, int layers) {
        f(16);
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        f(layers);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        return 1e3f * ms / layers;
    };
    printf("row-block stand-in %lld cycles, feed-forward stand-in %lld cycles behind their fetches; %d + %d workgroups\n", work_p, work_c, NRB, NFFN);
    for (int rep = 0; rep < 3; ++rep) {
        const float a = timeit(form_a, 400), b = timeit(form_b, 400);
        printf("  A two launches per layer %7.2f us | B one launch per layer (flags) %7.2f us | A - B = %+.2f us per layer\n", a, b, a - b);
    }
    unsigned hb[2];
    CK(hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost));
    printf("  lost flags %u, stale row reads %u\n", hb[0], hb[1]);
    std::vector<long long> st((size_t)NFFN * 4);
    for (int form = 0; form < 2; ++form) {
        if (form == 0) form_a(4); else form_b(4);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
        double f = 0, r = 0, fr = 0;
        for (int w = 0; w < NFFN; ++w) { f += st[w * 4]; r += st[w * 4 + 1]; fr += st[w * 4 + 2]; }
        printf("  form %c, consumer part, mean cycles from its entry: last flag seen %.0f, rows landed %.0f (flag -> rows %.0f)\n", form ? 'B' : 'A', f / NFFN, r / NFFN, fr / NFFN);
    }
    return 0;
}
