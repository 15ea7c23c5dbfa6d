// Machine probe for the GPU box (not product code):  hipcc --offload-arch=gfx950 -O3 tools/clockprobe.hip -o /tmp/clockprobe
//  1. shader clock under a sustained fp32-MFMA load (s_memtime ticks / wall time) and the achieved TFLOP/s,
//  2. the same for a SHORT burst (what a 10-20 us kernel sees),
//  3. dependent global-load latency (L2-resident 1 MB and HBM-sized 1 GB pointer chase),
//  4. back-to-back empty-kernel launch cost.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void mfma_loop(float *out, int iters, long long *clk) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
    }
    long long t1 = clock64();
    out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

__global__ void chase(const int *next, int n, int *out, long long *clk) {
    int p = 0;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) p = next[p];
    long long t1 = clock64();
    out[0] = p;
    clk[0] = t1 - t0;
}

__global__ void empty_kernel() {}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s  CUs %d  clockRate %d kHz  memClock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate, prop.memoryClockRate);
    float *out; long long *clk; int *iout;
    CK(hipMalloc(&out, 4096 * 256 * 4)); CK(hipMalloc(&clk, 64)); CK(hipMalloc(&iout, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep)
        for (int iters : {200, 2000, 20000, 200000}) {
            const int grid = 1024;                     // 4 workgroups of 4 waves per CU: one wave... 4 per SIMD
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(mfma_loop, dim3(grid), dim3(256), 0, 0, out, iters, clk);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            long long c; CK(hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost));
            const double flops = (double)grid * 4 * iters * 4 * 2048.0;
            printf("mfma iters %7d: %9.3f ms  %7.1f TFLOP/s  wave0 ticks %lld  (ticks/iter %.1f)\n", iters, ms, flops / ms / 1e9, c, (double)c / iters);
        }
    // clock64 tick rate: run a long loop and compare ticks with event time
    {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(mfma_loop, dim3(1), dim3(64), 0, 0, out, 100000, clk);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        long long c; CK(hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost));
        printf("single wave 100000 iters x4 mfma: %.3f ms, ticks %lld -> tick rate %.1f MHz; cycles per mfma (if 32/mfma) => clock %.0f MHz\n", ms, c, c / ms / 1e3,
               100000.0 * 4 * 32 / ms / 1e3);
    }
    for (size_t bytes : {(size_t)1 << 20, (size_t)1 << 30}) {
        const int n = (int)(bytes / 4);
        std::vector<int> h(n);
        const int stride = 4099 * 16;                  // a permutation walk with a large odd-ish stride (in ints)
        for (int i = 0; i < n; ++i) h[i] = (int)(((long long)i + stride) % n);
        int *d; CK(hipMalloc(&d, bytes)); CK(hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice));
        for (int rep = 0; rep < 2; ++rep) {
            const int hops = 20000;
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(chase, dim3(1), dim3(1), 0, 0, d, hops, iout, clk);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            long long c; CK(hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost));
            printf("chase %4zu MB: %.1f ns/hop  (%.1f ticks/hop)\n", bytes >> 20, ms * 1e6 / hops, (double)c / hops);
        }
        CK(hipFree(d));
    }
    {
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, 0);
        CK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        const int n = 2000;
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, 0);
        auto t1 = std::chrono::steady_clock::now();
        CK(hipDeviceSynchronize());
        auto t2 = std::chrono::steady_clock::now();
        printf("empty kernel: host enqueue %.2f us each, end-to-end %.2f us each\n",
               std::chrono::duration<double, std::micro>(t1 - t0).count() / n, std::chrono::duration<double, std::micro>(t2 - t0).count() / n);
    }
    return 0;
}
