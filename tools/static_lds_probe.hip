// Does a kernel with more than 64 KiB of STATIC LDS launch on gfx950 without an opt-in?  (rowblock_kernel<.., H2> has 81 KiB.)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float *o) {
    __shared__ float a[20000];
    __shared__ float b[2000];
    a[threadIdx.x + 19000] = o[threadIdx.x];
    b[threadIdx.x] = 1.f;
    __syncthreads();
    o[threadIdx.x] = a[19255 - threadIdx.x] + b[3];
}
int main() {
    float *d, h[256];
    for (int i = 0; i < 256; ++i) h[i] = (float)i;
    if (hipMalloc(&d, 1024) != hipSuccess) return 2;
    (void)hipMemcpy(d, h, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d);
    const hipError_t e1 = hipGetLastError(), e2 = hipDeviceSynchronize();
    (void)hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    printf("static 88000-byte LDS kernel: launch %s, sync %s, out[0] = %g (expect 256)\n", hipGetErrorString(e1), hipGetErrorString(e2), h[0]);
    return e1 != hipSuccess || e2 != hipSuccess || h[0] != 256.f;
}
