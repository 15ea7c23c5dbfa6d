// PointNet++ object encoder of the conditioning path ("next" row N1 of SURVEY.md §8(f)).
//
// Behaviour restated from model/layers.py:111-175 (PointNet2Encoder with ONE key point) around the third-party
// pointnet2_ops 3.0.0 set-abstraction ops (furthest point sampling, ball query, grouping, shared MLP, max pool; SURVEY.md
// appendix B.5, oracle/pointnet2.py -- parity unpinned) and model/diffusion_smpl.py:210-211 (input feature = ||p||).
// The reference runs SA1 for all 1024 sampled centres and then throws almost all of it away: SA2 has a single key point
// (the first sampled centre) whose two ball queries keep at most 16 + 32 SA1 centres.  Here ONE 16-wave workgroup per
// clip keeps the cloud in LDS and
//   1. runs the furthest-point sampling exactly (its ORDER decides which centres a ball query sees first),
//   2. finds the <= 48 SA1 centres the key point groups,
//   3. evaluates the SA1 multi-scale grouping + shared MLPs for those centres only,
//   4. evaluates SA2 and the final Linear.
// Eval-mode BatchNorm is folded into the 1x1 convolutions on the host (interdiff_amd/mdm.py: pack_pointnet2).
#include "common.h"
#include <float.h>

namespace {

constexpr int PT = 1024, NWV = PT / 64;
constexpr int MAXP = 2048;                 // points per cloud (two per thread)
constexpr int NP1 = 1024;                  // SA1 npoint
constexpr int NS0 = 16, NS1 = 32, NSLOT = NS0 + NS1;
constexpr int F1 = 96;                     // SA1 output channels (32 + 64)
constexpr int CIN2 = F1 + 3;               // SA2 input channels

__device__ __forceinline__ float d2_exact(float ax, float ay, float az, float bx, float by, float bz) {
#pragma clang fp contract(off)
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    return (dx * dx + dy * dy) + dz * dz;
}

// rank of this thread's hit among all hits of the block in thread order (exclusive), and the block total
__device__ __forceinline__ int block_rank(bool hit, unsigned *wcount, int &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long m = __ballot(hit);
    const int below = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wcount[wave] = (unsigned)__popcll(m);
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NWV; ++w) {
        const int c = (int)wcount[w];
        if (w < wave) base += c;
        tot += c;
    }
    __syncthreads();                           // wcount is reused by the next call
    total = tot;
    return base + below;
}

// out[n][o] = relu(b[o] + sum_c W[o][c] in[n][c]) for n < ns, o < cout
__device__ __forceinline__ void mlp_layer(const float *in, int in_stride, const float *__restrict__ W, const float *__restrict__ bias,
                                          int cin, int cout, int ns, float *out, int out_stride) {
    for (int i = threadIdx.x; i < ns * cout; i += PT) {
        const int n = i / cout, o = i - n * cout;
        const float *w = W + (size_t)o * cin, *x = in + n * in_stride;
        float s = bias[o];
        for (int c = 0; c < cin; ++c) s += w[c] * x[c];
        out[n * out_stride + o] = fmaxf(s, 0.f);
    }
}

__device__ __forceinline__ void maxpool(const float *a, int stride, int ns, int cout, float *dst) {
    for (int o = threadIdx.x; o < cout; o += PT) {
        float m = a[o];
        for (int n = 1; n < ns; ++n) m = fmaxf(m, a[n * stride + o]);
        dst[o] = m;
    }
}

__global__ __launch_bounds__(PT) void pointnet2_kernel(const idf_pointnet2 pn, const float *__restrict__ obj_points, int P,
                                                       float *__restrict__ out) {
    __shared__ float4 pts[MAXP];                       // x, y, z, ||p||
    __shared__ unsigned short fps[NP1];
    __shared__ float redv[2][NWV];
    __shared__ int redi[2][NWV];
    __shared__ unsigned wcount[NWV];
    __shared__ int list[NSLOT];                        // SA1 centres (FPS positions) grouped by the key point: 16 | 32
    __shared__ int nb0[NS0], nb1[NS1];                 // SA1 neighbours (point indices) of the current centre
    __shared__ float feat1[NSLOT][F1];
    __shared__ float xin[NS1 * 100], a1[NS1 * 64], a2[NS1 * 96], a3[NS1 * 128];
    __shared__ float f2[256];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *ar = pn.arena;

    // ---- 0. the cloud
    for (int i = tid; i < MAXP; i += PT) {
        float x = 0.f, y = 0.f, z = 0.f;
        if (i < P) { const float *p = obj_points + ((size_t)b * P + i) * 3; x = p[0]; y = p[1]; z = p[2]; }
        pts[i] = make_float4(x, y, z, sqrtf(x * x + y * y + z * z));
    }
    __syncthreads();

    // ---- 1. furthest point sampling 2048 -> 1024 (start at 0, skip |p|^2 <= 1e-3, arg-max = lowest index on ties)
    {
        float px[2], py[2], pz[2], tmp[2];
        bool ok[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = tid + PT * k;
            const float4 p = pts[i];
            px[k] = p.x; py[k] = p.y; pz[k] = p.z;
            tmp[k] = 1e10f;
            ok[k] = i < P && d2_exact(p.x, p.y, p.z, 0.f, 0.f, 0.f) > 1e-3f;
        }
        if (tid == 0) fps[0] = 0;
        int old = 0;
        for (int j = 1; j < NP1; ++j) {
            const float4 o = pts[old];
            float bv = -1.0f;
            int bi = 0x7fffffff;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (ok[k]) {
                    const float d2 = fminf(d2_exact(px[k], py[k], pz[k], o.x, o.y, o.z), tmp[k]);
                    tmp[k] = d2;
                    if (d2 > bv) { bv = d2; bi = tid + PT * k; }      // k = 0 first: lower index wins ties inside the thread
                }
            }
            const float wv = wave_max(bv);
            const float wi = -wave_max(bv == wv ? -(float)bi : -3e9f);                   // lowest index among the wave's maxima
            const int buf = j & 1;
            if (lane == 0) { redv[buf][wave] = wv; redi[buf][wave] = wv > -1.0f ? (int)wi : 0x7fffffff; }
            __syncthreads();
            float fv = -1.0f;
            int fi = 0x7fffffff;
#pragma unroll
            for (int w = 0; w < NWV; ++w) {
                const float v = redv[buf][w];
                const int ii = redi[buf][w];
                if (v > fv || (v == fv && ii < fi)) { fv = v; fi = ii; }
            }
            old = fv > -1.0f ? fi : 0;
            if (tid == 0) fps[j] = (unsigned short)old;
        }
    }
    __syncthreads();

    // ---- 2. the key point (first sampled centre) and the SA1 centres it groups: r = 0.1 / 16 and r = 0.2 / 32, FPS order
    const float4 c0 = pts[fps[0]];
    {
        const float4 q = pts[fps[tid]];
        const float d2 = d2_exact(q.x, q.y, q.z, c0.x, c0.y, c0.z);
        const float r0 = 0.1f, r1 = 0.2f;
        int tot;
        int rk = block_rank(d2 < r0 * r0, wcount, tot);
        if (d2 < r0 * r0 && rk < NS0) list[rk] = tid;
        __syncthreads();
        if (tid < NS0 && tid >= tot) list[tid] = tot > 0 ? list[0] : 0;
        __syncthreads();
        rk = block_rank(d2 < r1 * r1, wcount, tot);
        if (d2 < r1 * r1 && rk < NS1) list[NS0 + rk] = tid;
        __syncthreads();
        if (tid < NS1 && tid >= tot) list[NS0 + tid] = tot > 0 ? list[NS0] : 0;
        __syncthreads();
    }

    // ---- 3. SA1 (radii 0.05 / 0.1, 16 / 32 samples, MLPs 4-16-16-32 and 4-32-32-64) for the grouped centres only
    for (int s = 0; s < NSLOT; ++s) {
        const int cj = list[s];
        int dup = -1;
        for (int e = 0; e < s; ++e)
            if (list[e] == cj) { dup = e; break; }
        if (dup >= 0) {                                                  // block-uniform: padded slots repeat a centre
            for (int o = tid; o < F1; o += PT) feat1[s][o] = feat1[dup][o];
            __syncthreads();
            continue;
        }
        const float4 pc = pts[fps[cj]];
        const float ra = 0.05f, rb = 0.1f;
        bool h0[2], h1[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = tid + PT * k;
            const float4 p = pts[i];
            const float d2 = d2_exact(pc.x, pc.y, pc.z, p.x, p.y, p.z);          // new_xyz - xyz, as the CUDA op computes it
            h0[k] = i < P && d2 < ra * ra;
            h1[k] = i < P && d2 < rb * rb;
        }
        int tlo, thi;
        int rk = block_rank(h0[0], wcount, tlo);
        if (h0[0] && rk < NS0) nb0[rk] = tid;
        rk = block_rank(h0[1], wcount, thi);
        if (h0[1] && tlo + rk < NS0) nb0[tlo + rk] = tid + PT;
        __syncthreads();
        const int t0 = tlo + thi;
        if (tid < NS0 && tid >= t0) nb0[tid] = t0 > 0 ? nb0[0] : 0;
        rk = block_rank(h1[0], wcount, tlo);
        if (h1[0] && rk < NS1) nb1[rk] = tid;
        rk = block_rank(h1[1], wcount, thi);
        if (h1[1] && tlo + rk < NS1) nb1[tlo + rk] = tid + PT;
        __syncthreads();
        const int t1 = tlo + thi;
        if (tid < NS1 && tid >= t1) nb1[tid] = t1 > 0 ? nb1[0] : 0;
        __syncthreads();
        // grouped inputs [xyz_j - centre | ||p_j||]: scale 0 in xin[0 .. 16*4), scale 1 in xin[64 .. 64 + 32*4)
        if (tid < NS0 + NS1) {
            const bool sc1 = tid >= NS0;
            const int n = sc1 ? tid - NS0 : tid;
            const float4 p = pts[sc1 ? nb1[n] : nb0[n]];
            float *x = xin + (sc1 ? 64 + n * 4 : n * 4);
            x[0] = p.x - pc.x; x[1] = p.y - pc.y; x[2] = p.z - pc.z; x[3] = p.w;
        }
        __syncthreads();
        const idf_pn_mlp &m0 = pn.sa1[0], &m1 = pn.sa1[1];
        mlp_layer(xin, 4, ar + m0.w[0], ar + m0.b[0], 4, 16, NS0, a1, 16);
        mlp_layer(xin + 64, 4, ar + m1.w[0], ar + m1.b[0], 4, 32, NS1, a1 + 256, 32);
        __syncthreads();
        mlp_layer(a1, 16, ar + m0.w[1], ar + m0.b[1], 16, 16, NS0, a2, 16);
        mlp_layer(a1 + 256, 32, ar + m1.w[1], ar + m1.b[1], 32, 32, NS1, a2 + 256, 32);
        __syncthreads();
        mlp_layer(a2, 16, ar + m0.w[2], ar + m0.b[2], 16, 32, NS0, a3, 32);
        mlp_layer(a2 + 256, 32, ar + m1.w[2], ar + m1.b[2], 32, 64, NS1, a3 + 512, 64);
        __syncthreads();
        maxpool(a3, 32, NS0, 32, &feat1[s][0]);
        maxpool(a3 + 512, 64, NS1, 64, &feat1[s][32]);
        __syncthreads();
    }

    // ---- 4. SA2 around the key point (MLPs 99-64-64-128 and 99-64-96-128), max pool, Linear 256 -> 253
    for (int sc = 0; sc < 2; ++sc) {
        const int ns = sc ? NS1 : NS0, s0 = sc ? NS0 : 0;
        const idf_pn_mlp &m = pn.sa2[sc];
        for (int i = tid; i < ns * CIN2; i += PT) {
            const int n = i / CIN2, c = i - n * CIN2;
            float v;
            if (c < 3) {
                const float4 q = pts[fps[list[s0 + n]]];
                v = (c == 0 ? q.x - c0.x : (c == 1 ? q.y - c0.y : q.z - c0.z));
            } else {
                v = feat1[s0 + n][c - 3];
            }
            xin[n * 100 + c] = v;
        }
        __syncthreads();
        mlp_layer(xin, 100, ar + m.w[0], ar + m.b[0], CIN2, m.c[1], ns, a1, m.c[1]);
        __syncthreads();
        mlp_layer(a1, m.c[1], ar + m.w[1], ar + m.b[1], m.c[1], m.c[2], ns, a2, m.c[2]);
        __syncthreads();
        mlp_layer(a2, m.c[2], ar + m.w[2], ar + m.b[2], m.c[2], m.c[3], ns, a3, m.c[3]);
        __syncthreads();
        maxpool(a3, m.c[3], ns, 128, f2 + sc * 128);
        __syncthreads();
    }
    float *o = out + (size_t)b * 256;
    if (tid < 3) o[tid] = tid == 0 ? c0.x : (tid == 1 ? c0.y : c0.z);
    if (tid < 253) {
        const float *w = ar + pn.lin_w + (size_t)tid * 256;
        float s = ar[pn.lin_b + tid];
        for (int c = 0; c < 256; ++c) s += w[c] * f2[c];
        o[3 + tid] = s;
    }
}

}  // namespace

extern "C" int interdiff_pointnet2_encode(const idf_pointnet2 *pn, const float *obj_points, int32_t B, int32_t P, float *out,
                                          void *stream) {
    if (!pn || !pn->arena || !obj_points || !out || B <= 0 || P < 1 || P > MAXP) return IDF_E_INVAL;
    if (pn->sa1[0].c[0] != 4 || pn->sa1[0].c[1] != 16 || pn->sa1[0].c[2] != 16 || pn->sa1[0].c[3] != 32 || pn->sa1[1].c[0] != 4 ||
        pn->sa1[1].c[1] != 32 || pn->sa1[1].c[2] != 32 || pn->sa1[1].c[3] != 64)
        return IDF_E_INVAL;
    for (int sc = 0; sc < 2; ++sc)
        if (pn->sa2[sc].c[0] != CIN2 || pn->sa2[sc].c[1] > 64 || pn->sa2[sc].c[2] > 96 || pn->sa2[sc].c[3] != 128) return IDF_E_INVAL;
    hipStream_t s = idf_stream(stream);
    idf_prof_mark(IDF_K_OTHER, s);
    hipLaunchKernelGGL(pointnet2_kernel, dim3(B), dim3(PT), 0, s, *pn, obj_points, P, out);
    idf_prof_mark(-1, s);
    IDF_CHECK_LAUNCH();
    return IDF_OK;
}
