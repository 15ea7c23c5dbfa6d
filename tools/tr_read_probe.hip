// What does gfx950's ds_read_b64_tr_b16 return?  (not product code)  LDS holds halves whose bit pattern is their own index; every lane hands in its own byte address
// and the four halves it gets back are printed as LDS indices.
//     hipcc --offload-arch=gfx950 -O2 tools/tr_read_probe.hip -o build_tools/tr_read_probe && build_tools/tr_read_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef __attribute__((address_space(3))) uint16_t lds_u16;
__global__ void k(uint16_t *out, int *idx_out, int mode, int stride) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int l = threadIdx.x;
    int idx;                                   // element index (halves) this lane addresses
    if (mode == 0) idx = l * 4;                // consecutive 8-byte chunks
    else if (mode == 1) idx = (l & 15) * stride + (l >> 4) * 4;              // lane (l & 15) -> row, group l >> 4 -> 4-half chunk of the row
    else idx = ((l >> 4) * 4 + (l & 3)) * stride + ((l & 15) >> 2) * 4;      // group g: rows 4g .. 4g+3 (lane & 3), chunk (lane & 15) >> 2 of the row
    const uint32_t addr = (uint32_t)(uintptr_t)(lds_u16 *)lds + (uint32_t)idx * 2u;
    uint64_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)(v >> (16 * j));
    idx_out[l] = idx;
}
int main() {
    uint16_t *out; int *io;
    hipMalloc(&out, 64 * 4 * 2); hipMalloc(&io, 64 * 4);
    uint16_t h[256]; int hi[64];
    for (int mode = 0; mode < 3; ++mode) {
        const int stride = 64;
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, io, mode, stride);
        hipDeviceSynchronize();
        hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(hi, io, sizeof(hi), hipMemcpyDeviceToHost);
        printf("mode %d (row stride %d halves): lane: addressed element -> four elements returned\n", mode, stride);
        for (int l = 0; l < 64; ++l) printf("  lane %2d: %4d -> %4d %4d %4d %4d%s", l, hi[l], h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3], (l & 1) ? "\n" : "   |");
    }
    return 0;
}
