// Does v_mfma_f32_16x16x32_f16 honour SUBNORMAL f16 inputs on gfx950, and does v_cvt_pk_f16_f32 produce them?  (not product code)
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_f16_subnormal_probe.hip -o build_tools/mfma_f16_subnormal_probe
// The split-f16 kernels (csrc/ffn_h2.h split1) zero the hi plane below 2^-14 so that the answer does not matter for them; this probe says whether
// that rule is needed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float a_val, float b_val, float *out, unsigned short *bits) {
    const _Float16 a = (_Float16)a_val, b = (_Float16)b_val;      // a_val = 2^-20: subnormal in f16
    h8 av = {a, 0, 0, 0, 0, 0, 0, 0}, bv = {b, 0, 0, 0, 0, 0, 0, 0};
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = acc[0]; bits[0] = __builtin_bit_cast(unsigned short, a); }
}
int main() {
    float *o; unsigned short *b;
    (void)hipMalloc(&o, 16); (void)hipMalloc(&b, 16);
    const float cases[3][2] = {{ldexpf(1.f, -20), 1024.f}, {ldexpf(1.f, -24), 1024.f}, {ldexpf(1.f, -10), 1024.f}};
    for (auto &c : cases) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, c[0], c[1], o, b);
        float r; unsigned short hb;
        (void)hipMemcpy(&r, o, 4, hipMemcpyDeviceToHost); (void)hipMemcpy(&hb, b, 2, hipMemcpyDeviceToHost);
        printf("a = %g (f16 bits 0x%04x) x b = %g  ->  mfma %g (exact %g)\n", c[0], hb, c[1], r, c[0] * c[1]);
    }
    return 0;
}
