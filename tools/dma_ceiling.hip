// Ceiling of the global -> LDS DMA stream per CU for L2-resident data with DEEP queues (tools only; asm DMA so that the compiler
// inserts no waits):   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-inline-asm -I interdiff_amd/csrc tools/dma_ceiling.hip -o build_tools/dma_ceiling
// Every workgroup streams `iters` rounds of (waves x DEPTH) 1-KiB pieces into an LDS ring; each wave keeps DEPTH pieces in flight
// (s_waitcnt vmcnt(DEPTH - 1) before reusing a slot).  SHARED = 1: all workgroups read the same 416 KiB window (the fused FFN's
// pattern: 50 workgroups share one weight slice); SHARED = 0: every workgroup has its own window (2 MiB apart, L2/MALL-resident).
#include "common.h"
#include <cstdio>
bool g_idf_prof_on = false;
void idf_prof_mark_slow(int, hipStream_t) {}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int NW, int DEPTH, int REG>
__global__ __launch_bounds__(NW * 64) void stream_kernel(const float *src, int shared, int iters, float *out) {
    __shared__ __attribute__((aligned(1024))) float lds[NW * DEPTH * 256];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float *base = idf_uniform_ptr(src + (shared ? (size_t)(blockIdx.x % 5) * 106496 : (size_t)blockIdx.x * 106496));   // 416 KiB windows
    const uint32_t l0 = idf_lds_addr(lds) + (uint32_t)(wave * DEPTH * 1024);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = 0; i < iters; ++i) {
        // window of 416 KiB = 416 pieces; wave w takes pieces (i*NW*DEPTH + d*NW + w) % 416
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int piece = (i * NW * DEPTH + d * NW + wave) % 416;
            if (REG) {
                const float4 v = *reinterpret_cast<const float4 *>(base + piece * 256 + lane * 4);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            } else {
                if (i > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");      // slot d's previous piece has landed
                idf_dma16_s(base + piece * 256, (uint32_t)(lane << 4), l0 + (uint32_t)(d * 1024));
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!REG) acc.x = lds[tid];
    out[blockIdx.x * NW * 64 + tid] = acc.x + acc.y + acc.z + acc.w;
}

template <int NW, int DEPTH, int REG>
int run(const float *src, float *out, int wgs, int shared) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 400;
    hipLaunchKernelGGL((stream_kernel<NW, DEPTH, REG>), dim3(wgs), dim3(NW * 64), 0, 0, src, shared, iters, out);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((stream_kernel<NW, DEPTH, REG>), dim3(wgs), dim3(NW * 64), 0, 0, src, shared, iters, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)wgs * iters * NW * DEPTH * 1024;
    printf("%-22s waves %2d depth %2d wgs %4d %-8s: %8.1f GB/s  %6.1f B/clk/CU (2.4 GHz x 256)  %.1f KiB in flight per WG\n", REG ? "global_load -> VGPR" : "LDS-DMA (asm)", NW, DEPTH, wgs,
           shared ? "shared" : "private", bytes / ms / 1e6, bytes / ms / 1e6 / 256 / 2.4, NW * DEPTH * 1.0);
    return 0;
}

int main() {
    float *src, *out;
    const size_t floats = (size_t)1024 * 106496 + 4096;
    CK(hipMalloc(&src, floats * 4)); CK(hipMalloc(&out, 2048 * 1024 * 4));
    CK(hipMemset(src, 0, floats * 4));
    for (int shared : {1, 0}) {
        for (int wgs : {250, 500}) {
            run<4, 4, 0>(src, out, wgs, shared);
            run<4, 8, 0>(src, out, wgs, shared);
            run<8, 4, 0>(src, out, wgs, shared);
            run<8, 8, 0>(src, out, wgs, shared);
            run<16, 4, 0>(src, out, wgs, shared);
            run<16, 8, 0>(src, out, wgs, shared);
            run<8, 8, 1>(src, out, wgs, shared);
            run<16, 8, 1>(src, out, wgs, shared);
        }
    }
    return 0;
}
