// Ceiling of global->LDS streaming per CU for L2-resident data (tools only).
//   hipcc --offload-arch=gfx950 -O3 tools/lds_dma_probe.hip -o build_tools/lds_dma_probe
// Every workgroup streams `bytes_per_wg` of an L2/MALL-resident buffer into LDS with (a) global_load_lds_dwordx4,
// (b) global_load_dwordx4 + ds_write_b128, varying waves per workgroup and workgroups per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef __attribute__((address_space(3))) void *lds_ptr_t;

template <int MODE, int NT>
__global__ __launch_bounds__(NT) void stream_kernel(const float *src, size_t span_floats, int iters, float *out) {
    __shared__ __attribute__((aligned(1024))) float lds[4 * NT * 4];            // 4 slots of NT*16 B
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const float *base = src + ((size_t)blockIdx.x * 8191 * 64) % (span_floats / 2);
    float acc = 0.f;
    for (int i = 0; i < iters; ++i) {
        const size_t off = ((size_t)i * NT * 4) % (span_floats / 2 - NT * 4);
        const float *p = base + off + tid * 4;
        float *slot = lds + (i & 3) * NT * 4;
        if (MODE == 0) {
            __builtin_amdgcn_global_load_lds((const void *)p, (lds_ptr_t)(slot + wave * 256), 16, 0, 0);
            if ((i & 3) == 3) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); acc += slot[lane]; }
        } else {
            const float4 v = *reinterpret_cast<const float4 *>(p);
            *reinterpret_cast<float4 *>(slot + tid * 4) = v;
            if ((i & 3) == 3) acc += slot[lane];
        }
    }
    out[blockIdx.x * NT + tid] = acc;
}

template <int MODE, int NT>
int run(const float *src, size_t span, float *out, int wgs, const char *name) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    hipLaunchKernelGGL((stream_kernel<MODE, NT>), dim3(wgs), dim3(NT), 0, 0, src, span, iters, out);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((stream_kernel<MODE, NT>), dim3(wgs), dim3(NT), 0, 0, src, span, iters, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)wgs * iters * NT * 16;
    printf("%-34s wgs %4d x %4d thr: %7.1f GB/s total  %6.1f B/clk/CU (2.4 GHz, 256 CUs)\n", name, wgs, NT, bytes / ms / 1e6, bytes / ms / 1e6 / 256 / 2.4);
    return 0;
}

int main() {
    float *src, *out;
    const size_t span = (size_t)2 << 20;                    // 8 MB of floats: fits the aggregate L2 / MALL
    CK(hipMalloc(&src, span * 4)); CK(hipMalloc(&out, 4096 * 1024 * 4));
    CK(hipMemset(src, 0, span * 4));
    for (int wgs : {256, 512, 1024}) {
        run<0, 256>(src, span, out, wgs, "LDS-DMA dwordx4, 4 waves/WG");
        run<0, 512>(src, span, out, wgs, "LDS-DMA dwordx4, 8 waves/WG");
        run<1, 256>(src, span, out, wgs, "load dwordx4 + ds_write, 4 waves/WG");
        run<1, 512>(src, span, out, wgs, "load dwordx4 + ds_write, 8 waves/WG");
    }
    return 0;
}
