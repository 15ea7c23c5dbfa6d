// THE DEFAULT self-attention of the split arithmetic since round 5 (tune[IDF_TUNE_MISC] = 6 selects the fp32 kernel of denoiser.hip instead).  Round 4 built it and
// left it off the route: owning the CU costs the overlap of two co-resident workgroups, and with V staged as fp32 and transposed through LDS it was 1.5 % slower over whole
// samples (profiles/r04_attn_split_f16_ab.txt).  With V kept row-major and read by ds_read_b64_tr_b16 -- no staging, no transposition, two barriers fewer: 17.4 k -> 15.2 k
// cycles per workgroup -- it is 1.2 % FASTER than the fp32 kernel (profiles/r05_attn_split_f16_ab.txt); a denoiser forward is 3.2e-7 from the fp64 answer with it, 4.6e-7 without.
//
// Temporal self-attention of the two standard layers with its head's slice of the out-projection, on the f16 matrix pipe (round 4):
//
//     ctx = softmax(Q K^T / 8) V   per (clip, head);   slab[head] = ctx . W_o[:, head]^T        (torch.nn.MultiheadAttention inside TransformerDecoderLayer)
//
// Same contract as denoiser.hip self_attn_kernel<OUTPROJ>: qkv [N][768] in, H partial slabs [N][256] out (the row block sums them).  What differs:
//  * 32 queries per workgroup and ONE workgroup per CU (ceil(T/32) x 4 heads x B = 256 workgroups at T = 100, B = 16: exactly the chip), 512 threads;
//    the fp32 kernel runs 448 workgroups of 16 queries, two per CU -- a kernel that issues the f16 MFMA must own its CU (ffn_h2.h "exclusive CU"), so the
//    overlap of two co-resident workgroups is replaced by twice the waves on twice the rows and K / V fetched once per 32 queries instead of per 16;
//  * the three contractions are split-f16 products (v = hi + lo' 2^-11: three v_mfma_f32_16x16x32_f16 per product, fp32 accumulate).  Q, K and V come out
//    of the QKV projection with no a-priori range: each of the three tiles is divided by the power of two that puts its largest magnitude in [2^13, 2^14)
//    (exact; an f16 pair then keeps 22 bits of every element down to 2^-27 of the largest) and the fp32 results are multiplied back -- S by 2^(eq + ek) / 8,
//    the out-projection by 2^ev (the context is a convex combination of V rows: it inherits V's scale and range).  Probabilities are split as they are.
//  * W_o arrives as pre-split plane fragments [head][16 column tiles][2 K steps][2 planes][64 lanes][8 halves] (mdm.py sa_out_fragments_h2).
// LDS (dynamic, sized by T; T <= 192): K planes [TP][72] x 2 | V planes [TPP][72] x 2, row-major | S fp32 [32][TP + 4] | Q planes [32][72] x 2; the probability
// planes [32][TPP + 8] x 2 overwrite K once S is complete, the context planes overwrite Q.  TP = T up to 16, TPP = T up to 32 (K steps over the keys).
#pragma once
#include <float.h>
#include "common.h"
#include "ffn_h2.h"

// phase stamps exist only in tools/rowblock_probe.hip (which defines the macro before including this file)
#ifndef IDF_AH2_STAMP
#define IDF_AH2_STAMP(i) do { } while (0)
#endif

namespace idf_attn_h2 {

using idf_ffn_h2::h8;
constexpr int D = IDF_MDM_D, H = IDF_MDM_HEADS, HD = D / H, QT = 32, NTH = 512, NWV = NTH / 64;
constexpr int KHS = HD + 8;                              // row stride (halves) of the Q / K / context planes
constexpr int VRS = HD + 8;                              // row stride (halves) of the V planes (row-major [key][dim]: the P V operand is read with the transposing LDS read)
constexpr int MAX_T = 192, NIT = (MAX_T * 16 + NTH - 1) / NTH;      // float4 sweeps of a K / V tile per thread: 6 (the planes of a longer clip do not fit the CU's LDS: the fp32 kernel takes it)
constexpr int NIT0 = 4;                                  // sweeps that cover T <= 128: requested unconditionally; the rest sit behind ONE workgroup-uniform branch
constexpr int WO_H2_FLOATS = H * 16 * 2 * 2 * 64 * 4;      // 32768
constexpr int OCS = D + 4;                               // row stride (floats) of the out-projection's staging tile [32][OCS] (33 280 B)

// halves of the region that holds the K planes and later the probability planes
__host__ __device__ inline int k_region_halves(int TP, int TPP) { return 2 * TP * KHS > 2 * QT * (TPP + 8) ? 2 * TP * KHS : 2 * QT * (TPP + 8); }
// the out-projection's staging tile [32][OCS] sits over the K / V planes when they are large enough to hold it (T > 32), else behind the Q planes
__host__ __device__ inline bool stage_over_kv(int TP, int TPP) { return (size_t)k_region_halves(TP, TPP) * 2 + (size_t)2 * TPP * VRS * 2 >= (size_t)QT * OCS * 4; }
inline size_t lds_bytes(int T) {
    const int TP = (T + 15) & ~15, TPP = (T + 31) & ~31;
    return (size_t)k_region_halves(TP, TPP) * 2 + (size_t)2 * TPP * VRS * 2 + (size_t)QT * (TP + 4) * 4 + (size_t)2 * QT * KHS * 2 + 64 + (stage_over_kv(TP, TPP) ? 0 : (size_t)QT * OCS * 4);
}

// The P V operand of one K step from the row-major V planes: four transposing LDS reads (ds_read_b64_tr_b16, four halves each: keys k0 .. k0 + 3 and k0 + 4 .. k0 + 7 of the
// lane's column, hi plane and lo plane) and ONE wait.  `hi` / `lo`: the lane's 8-byte-aligned chunk address in either plane (see the call site); the second read of a plane
// is the immediate offset of four rows.  (The wait is inside: the compiler does not count an asm's LDS operations, and nothing of this asm is in flight when it ends.)
__device__ __forceinline__ void tr_read_v(const _Float16 *hi, const _Float16 *lo, h8 &bh, h8 &bl) {
    typedef __attribute__((address_space(3))) const _Float16 lds_h;
    uint64_t h0, h1, l0, l1;
    asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %4 offset:%6\n\tds_read_b64_tr_b16 %2, %5\n\tds_read_b64_tr_b16 %3, %5 offset:%6\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(h0), "=&v"(h1), "=&v"(l0), "=&v"(l1)
                 : "v"((uint32_t)(uintptr_t)(lds_h *)hi), "v"((uint32_t)(uintptr_t)(lds_h *)lo), "i"(4 * VRS * 2)
                 : "memory");
    struct { uint64_t a, b; } ph = {h0, h1}, pl = {l0, l1};
    bh = __builtin_bit_cast(h8, ph);
    bl = __builtin_bit_cast(h8, pl);
}

// power of two that brings amax into [2^13, 2^14): returns the multiplier 2^-e and, through `up`, 2^e (1 for an all-zero or non-finite tile)
__device__ __forceinline__ float pow2_scale(float amax, float &up) {
    int e = (amax > 0.f && amax < INFINITY) ? (int)((__builtin_bit_cast(uint32_t, amax) >> 23) & 0xff) - 127 - 13 : 0;
    e = max(-100, min(100, e));                          // (a tile below 2^-87 is as good as zero; keeps both powers of two finite)
    up = __builtin_bit_cast(float, (uint32_t)((127 + e) << 23));
    return __builtin_bit_cast(float, (uint32_t)((127 - e) << 23));
}

// PLANES_IN: `qkv` holds the plane pairs the QKV kernel wrote (ffn_h2.h ln_linear_h2_kernel<.., PLANES>: per token row and (q / k / v, head) group [hi 64 halves | lo' 64 halves])
// and `scales` [N][4] the power of two each row's q / k / v were divided by -- nothing is split here, the planes go from memory to LDS as they are; S is multiplied back per
// (query row, key), a probability takes its key's V scale relative to the largest of the clip (<= 1) before it is split, the largest scales the output.  Otherwise `qkv` is the fp32
// [N][768] matrix and the kernel splits Q, K, V itself under one power of two per tile (the route of a QKV projection that ran as fp32).
template <int PLACED = 0, bool PLANES_IN = false>      // (PLACED: a template so that denoiser.hip can instantiate the kernel explicitly, next to the other kernels of a step: code placement)
__global__ __launch_bounds__(NTH) void self_attn_h2_kernel(const float *__restrict__ qkv, int T, int nwg, const float *__restrict__ wo_h2,
                                                           float *__restrict__ slabs, size_t pstride, const float *__restrict__ scales) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    asm volatile("" ::: "v255");                         // exclusive CU: 2 waves per SIMD x 256 registers (+ the launcher's 160 KiB of LDS)
    __shared__ float red[3][NWV];
    __shared__ float sqs[PLANES_IN ? QT : 1], sks[PLANES_IN ? MAX_T : 1], svs[PLANES_IN ? MAX_T + 32 : 1];      // 2^e of the query rows / keys (K, V) of this (clip, head)
    // (all ten argument dwords, the grid size among them, arrive preloaded in SGPRs: build.py)
    const int TP = (T + 15) & ~15, TPP = (T + 31) & ~31, VTS = TPP + 8, SS = TP + 4;
    _Float16 *kh = reinterpret_cast<_Float16 *>(smraw), *kl = kh + TP * KHS;                 // K planes [TP][KHS]
    _Float16 *vh = kh + k_region_halves(TP, TPP), *vl = vh + TPP * VRS;                      // V planes [TPP][VRS], row-major (rows T .. TPP - 1 zero)
    float *Ss = reinterpret_cast<float *>(vl + TPP * VRS);                                   // S [32][SS]
    _Float16 *qh = reinterpret_cast<_Float16 *>(Ss + QT * SS), *ql = qh + QT * KHS;          // Q planes [32][KHS], later the context planes
    _Float16 *ph = kh, *pl = kh + QT * VTS;                                                  // probability planes [32][VTS] over K (the region is sized for the larger of the two)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
    const int id = blockIdx.x, xq = nwg >> 3, xr = nwg & 7, xcd = id & 7;
    const int lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (id >> 3);      // XCD-affine: the query tiles of a (clip, head) share its K / V in one L2
    const int nqt = (T + QT - 1) / QT, b = lid / (nqt * H), h = (lid / nqt) % H, q0 = (lid % nqt) * QT;
    const size_t rowbase = (size_t)b * T;

    IDF_AH2_STAMP(0);
    float uq = 1.f, uk = 1.f, uv = 1.f;                  // (tile scales of the fp32-input route)
    float4 wo[2][2][2];                                  // out-projection fragments of this wave's two column tiles: [tile][K step][plane]
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int p = 0; p < 2; ++p)
                wo[c][s][p] = *reinterpret_cast<const float4 *>(wo_h2 + (size_t)(((((h * 16 + 2 * wave + c) * 2 + s) * 2 + p) * 64) + lane) * 4);

    if constexpr (PLANES_IN) {
        // ---- planes from memory to LDS: thread (row qr of a 32-row sweep, 16-byte chunk c of the row's 256 bytes: c < 8 the hi plane, else the lo' plane)
        const int qr = tid >> 4, c = tid & 15;
        const size_t coff = (size_t)c * 4;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 qv = *reinterpret_cast<const float4 *>(qkv + ((rowbase + min(q0 + qr, T - 1)) * 12 + h) * 64 + coff);
        float4 kreg[NIT], vreg[NIT];
#pragma unroll
        for (int u = 0; u < NIT0; ++u) {
            const size_t rw = (rowbase + min(qr + 32 * u, T - 1)) * 12;
            kreg[u] = *reinterpret_cast<const float4 *>(qkv + (rw + 4 + h) * 64 + coff);
            vreg[u] = *reinterpret_cast<const float4 *>(qkv + (rw + 8 + h) * 64 + coff);
        }
        if (TP > 32 * NIT0) {
#pragma unroll
            for (int u = NIT0; u < NIT; ++u) {
                const size_t rw = (rowbase + min(qr + 32 * u, T - 1)) * 12;
                kreg[u] = *reinterpret_cast<const float4 *>(qkv + (rw + 4 + h) * 64 + coff);
                vreg[u] = *reinterpret_cast<const float4 *>(qkv + (rw + 8 + h) * 64 + coff);
            }
        } else {
#pragma unroll
            for (int u = NIT0; u < NIT; ++u) kreg[u] = vreg[u] = z;
        }
        float4 sc4 = z;
        if (tid < TPP) sc4 = *reinterpret_cast<const float4 *>(scales + (rowbase + min(tid, T - 1)) * 4);
        const float sq1 = tid < QT ? scales[(rowbase + min(q0 + tid, T - 1)) * 4] : 0.f;
        IDF_AH2_STAMP(1);
        // (component-wise selects and integer plane offsets: a `cond ? float4 : float4` or a `cond ? ptr : ptr` here made the compiler park both candidates in scratch
        // memory and index them -- 32 scratch instructions and 7 us per launch)
        auto keep = [](bool on, const float4 v) { return make_float4(on ? v.x : 0.f, on ? v.y : 0.f, on ? v.z : 0.f, on ? v.w : 0.f); };
        const int po = (c & 7) * 8;                      // halves into the row of either plane
        const bool lo = c >= 8;
        *reinterpret_cast<float4 *>(qh + (lo ? QT * KHS : 0) + qr * KHS + po) = keep(q0 + qr < T, qv);
        const int kplane = lo ? TP * KHS : 0, vplane = lo ? TPP * VRS : 0;
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int j = qr + 32 * u;
            if (32 * u < TPP && j < TPP) {               // (first condition workgroup-uniform)
                if (j < TP) *reinterpret_cast<float4 *>(kh + kplane + j * KHS + po) = keep(j < T, kreg[u]);
                *reinterpret_cast<float4 *>(vh + vplane + j * VRS + po) = keep(j < T, vreg[u]);      // keys T .. TPP - 1: zero rows (they meet zero probabilities)
            }
        }
        if (tid < TPP) { sks[tid] = tid < T ? sc4.y : 1.f; svs[tid] = tid < T ? sc4.z : 0.f; }
        if (tid < QT) sqs[tid] = sq1;
    } else {
        // ---- operand fetch: everything requested at once with clamped addresses (no guard around a load)
        const int qr = tid >> 4, c4 = (tid & 15) * 4;        // thread (row, 4-float chunk) of a [rows][64] tile
        const float4 qv = *reinterpret_cast<const float4 *>(qkv + (rowbase + min(q0 + qr, T - 1)) * (3 * D) + h * HD + c4);
        float4 kreg[NIT], vreg[NIT];
    #pragma unroll
        for (int u = 0; u < NIT0; ++u) {
            const int j = min(qr + 32 * u, T - 1);
            const float *src = qkv + (rowbase + j) * (3 * D) + h * HD + c4;
            kreg[u] = *reinterpret_cast<const float4 *>(src + D);
            vreg[u] = *reinterpret_cast<const float4 *>(src + 2 * D);
        }
        if (TP > 32 * NIT0) {                                // (clips longer than 128 frames; at T = 100 these sweeps were three more rounds of clamped -- repeated -- requests)
    #pragma unroll
            for (int u = NIT0; u < NIT; ++u) {
                const int j = min(qr + 32 * u, T - 1);
                const float *src = qkv + (rowbase + j) * (3 * D) + h * HD + c4;
                kreg[u] = *reinterpret_cast<const float4 *>(src + D);
                vreg[u] = *reinterpret_cast<const float4 *>(src + 2 * D);
            }
        } else {
    #pragma unroll
            for (int u = NIT0; u < NIT; ++u) kreg[u] = vreg[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        // ---- the three tile scales
        auto amax4 = [](const float4 v) { return fmaxf(fmaxf(__builtin_fabsf(v.x), __builtin_fabsf(v.y)), fmaxf(__builtin_fabsf(v.z), __builtin_fabsf(v.w))); };
        float aq = q0 + qr < T ? amax4(qv) : 0.f, ak = 0.f, av = 0.f;
    #pragma unroll
        for (int u = 0; u < NIT; ++u)
            if (qr + 32 * u < T) {
                ak = fmaxf(ak, amax4(kreg[u]));
                av = fmaxf(av, amax4(vreg[u]));
            }
        IDF_AH2_STAMP(1);                                    // operands landed (the amax code above consumed them)
        aq = wave_max(aq); ak = wave_max(ak); av = wave_max(av);
        if (lane == 0) { red[0][wave] = aq; red[1][wave] = ak; red[2][wave] = av; }
        __syncthreads();
        aq = ak = av = 0.f;
    #pragma unroll
        for (int w = 0; w < NWV; ++w) { aq = fmaxf(aq, red[0][w]); ak = fmaxf(ak, red[1][w]); av = fmaxf(av, red[2][w]); }
        const float dq = pow2_scale(aq, uq), dk = pow2_scale(ak, uk), dv = pow2_scale(av, uv);

        // ---- planes: Q, K and V row-major [row][dim], each thread splits the chunks it fetched -- no staging, no transposition: the P V contraction reads its V operand
        // with gfx950's transposing LDS read (ds_read_b64_tr_b16, below).  (A first version staged V as fp32, transposed it with eight strided scalar reads per octet of
        // keys and needed two more barriers: 6.7 k of the launch's 17.4 k cycles went into this phase, tools/rowblock_probe.hip.)
        {
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            uint2 hi, lo;
            idf_ffn_h2::split4_pk(q0 + qr < T ? make_float4(qv.x * dq, qv.y * dq, qv.z * dq, qv.w * dq) : z, hi, lo);
            *reinterpret_cast<uint2 *>(qh + qr * KHS + c4) = hi;
            *reinterpret_cast<uint2 *>(ql + qr * KHS + c4) = lo;
    #pragma unroll
            for (int u = 0; u < NIT; ++u) {
                const int j = qr + 32 * u;
                if (32 * u < TPP && j < TPP) {               // (first condition workgroup-uniform)
                    if (j < TP) {
                        idf_ffn_h2::split4_pk(j < T ? make_float4(kreg[u].x * dk, kreg[u].y * dk, kreg[u].z * dk, kreg[u].w * dk) : z, hi, lo);
                        *reinterpret_cast<uint2 *>(kh + j * KHS + c4) = hi;
                        *reinterpret_cast<uint2 *>(kl + j * KHS + c4) = lo;
                    }
                    idf_ffn_h2::split4_pk(j < T ? make_float4(vreg[u].x * dv, vreg[u].y * dv, vreg[u].z * dv, vreg[u].w * dv) : z, hi, lo);      // keys T .. TPP - 1: zero rows (they meet zero probabilities)
                    *reinterpret_cast<uint2 *>(vh + j * VRS + c4) = hi;
                    *reinterpret_cast<uint2 *>(vl + j * VRS + c4) = lo;
                }
            }
        }
    }
    __syncthreads();
    IDF_AH2_STAMP(2);                                    // scales + planes in LDS

    // ---- S = Q K^T / 8: wave w owns key tiles w, w + 8, ... for both query tiles
    const float sscale = uq * uk * 0.125f;
    for (int ct = wave; ct < TP / 16; ct += NWV) {
        f32x4 am[2], ac[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) am[rt] = ac[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int ko = 32 * s + 8 * kq;
            const h8 bh = *reinterpret_cast<const h8 *>(kh + (ct * 16 + li) * KHS + ko), bl = *reinterpret_cast<const h8 *>(kl + (ct * 16 + li) * KHS + ko);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const h8 ah = *reinterpret_cast<const h8 *>(qh + (rt * 16 + li) * KHS + ko), al = *reinterpret_cast<const h8 *>(ql + (rt * 16 + li) * KHS + ko);
                IDF_H2_MFMA(am[rt], ah, bh);
                IDF_H2_MFMA(ac[rt], ah, bl);
                IDF_H2_MFMA(ac[rt], al, bh);
            }
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float sc = PLANES_IN ? sqs[rt * 16 + kq * 4 + r] * (sks[ct * 16 + li] * 0.125f) : sscale;
                Ss[(rt * 16 + kq * 4 + r) * SS + ct * 16 + li] = (am[rt][r] + ac[rt][r] * idf_ffn_h2::LO_UNSCALE) * sc;
            }
    }
    __syncthreads();
    IDF_AH2_STAMP(3);                                    // S

    // ---- row softmax: one 16-lane group per query row (32 groups = 32 rows); lane l16 owns columns l16, 16 + l16, ...; probabilities leave as planes over K.
    // The first NC0 = 8 columns per lane cover T <= 128 and run unconditionally; the columns of longer clips sit behind ONE workgroup-uniform branch per pass
    // (same operations in the same order: max, then the sum over ascending columns).
    {
        constexpr int NC = ((MAX_T + 31) & ~31) / 16, NC0 = 8;      // 12 columns per lane cover TPP of the longest clip
        const int row = wave * 4 + kq;
        const float *srow = Ss + row * SS;
        const bool tail = TP > 16 * NC0;
        float v[NC], mx = -FLT_MAX;
        float svl[PLANES_IN ? NC : 1], svm = 0.f;        // PLANES_IN: 2^e of the V rows of this lane's columns (0 past the clip), and their maximum
#pragma unroll
        for (int c = 0; c < NC0; ++c) {
            const int j = 16 * c + li;
            v[c] = srow[min(j, TP - 1)];
            v[c] = j < T ? v[c] : -FLT_MAX;
            mx = fmaxf(mx, v[c]);
            if constexpr (PLANES_IN) { svl[c] = svs[min(j, TPP - 1)]; svm = fmaxf(svm, svl[c]); }
        }
        if (tail) {
#pragma unroll
            for (int c = NC0; c < NC; ++c) {
                const int j = 16 * c + li;
                v[c] = srow[min(j, TP - 1)];
                v[c] = j < T ? v[c] : -FLT_MAX;
                mx = fmaxf(mx, v[c]);
                if constexpr (PLANES_IN) { svl[c] = svs[min(j, TPP - 1)]; svm = fmaxf(svm, svl[c]); }
            }
        }
        mx = row16_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < NC0; ++c) {
            v[c] = 16 * c + li < T ? __expf(v[c] - mx) : 0.f;
            sum += v[c];
        }
        if (tail) {
#pragma unroll
            for (int c = NC0; c < NC; ++c) {
                v[c] = 16 * c + li < T ? __expf(v[c] - mx) : 0.f;
                sum += v[c];
            }
        }
        const float inv = __builtin_amdgcn_rcpf(row16_sum(sum));
        float rsvm = 1.f;
        if constexpr (PLANES_IN) {                       // a probability carries its key's V scale relative to the clip's largest (powers of two: exact, <= 1)
            svm = row16_max(svm);
            rsvm = __builtin_bit_cast(float, 0x7F000000u - __builtin_bit_cast(uint32_t, svm));
            if (tid == 0) red[0][0] = svm;               // (every row of every workgroup of the (clip, head) finds the same maximum)
        }
        auto put = [&](int c) {                          // columns T .. TPP - 1 are written as zeros: they are part of the last K step
            _Float16 a, cc;
            idf_ffn_h2::split1_nf(PLANES_IN ? v[c] * inv * (svl[c] * rsvm) : v[c] * inv, a, cc);
            ph[row * VTS + 16 * c + li] = a;
            pl[row * VTS + 16 * c + li] = cc;
        };
#pragma unroll
        for (int c = 0; c < NC0; ++c)
            if (16 * c < TPP) put(c);                    // (workgroup-uniform: TPP is a multiple of 32)
        if (tail) {
#pragma unroll
            for (int c = NC0; c < NC; ++c)
                if (16 * c < TPP) put(c);
        }
    }
    __syncthreads();
    IDF_AH2_STAMP(4);                                    // softmax + probability planes

    // ---- ctx' = P (V 2^-ev): eight 16 x 16 tiles, one per wave (query tile w >> 2, head-dim tile w & 3); left scaled, split, parked over Q
    {
        const int rt = wave >> 2, dt = wave & 3;
        f32x4 am = {0.f, 0.f, 0.f, 0.f}, ac = am;
        for (int s = 0; s < TPP / 32; ++s) {
            const int ko = 32 * s + 8 * kq;
            const h8 ah = *reinterpret_cast<const h8 *>(ph + (rt * 16 + li) * VTS + ko), al = *reinterpret_cast<const h8 *>(pl + (rt * 16 + li) * VTS + ko);
            // B operand: lane (li = head-dim column, kq) needs V[32 s + 8 kq .. + 7][16 dt + li] -- eight KEYS of one column of the row-major image.  ds_read_b64_tr_b16: the 16 lanes of
            // a group each name four contiguous halves, lane c the chunk (key k0 + (c >> 2), dims 16 dt + 4 (c & 3) ..), and lane i receives element i & 3 of the chunks (i >> 2) + 4 j,
            // j = 0..3 = keys k0 .. k0 + 3 of column 16 dt + i (tools/tr_read_probe.hip); two reads (k0 = 32 s + 8 kq, + 4) make the fragment
            const int vo = (ko + (li >> 2)) * VRS + dt * 16 + 4 * (li & 3);
            h8 bh, bl;
            tr_read_v(vh + vo, vl + vo, bh, bl);
            IDF_H2_MFMA(am, ah, bh);
            IDF_H2_MFMA(ac, ah, bl);
            IDF_H2_MFMA(ac, al, bh);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            _Float16 a, c;
            idf_ffn_h2::split1_nf(am[r] + ac[r] * idf_ffn_h2::LO_UNSCALE, a, c);
            qh[(rt * 16 + kq * 4 + r) * KHS + dt * 16 + li] = a;      // (every wave finished reading the Q planes two barriers ago)
            ql[(rt * 16 + kq * 4 + r) * KHS + dt * 16 + li] = c;
        }
    }
    __syncthreads();
    IDF_AH2_STAMP(5);                                    // P V + context planes

    // ---- out-projection partial of this head: [32 x 64] . [64 x 256]; wave w owns output column tiles 2w, 2w + 1 for both query tiles
    {
        f32x4 om[2][2], oc[2][2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int c = 0; c < 2; ++c) om[rt][c] = oc[rt][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int ko = 32 * s + 8 * kq;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const h8 ah = *reinterpret_cast<const h8 *>(qh + (rt * 16 + li) * KHS + ko), al = *reinterpret_cast<const h8 *>(ql + (rt * 16 + li) * KHS + ko);
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    IDF_H2_MFMA(om[rt][c], ah, __builtin_bit_cast(h8, wo[c][s][0]));
                    IDF_H2_MFMA(oc[rt][c], ah, __builtin_bit_cast(h8, wo[c][s][1]));
                    IDF_H2_MFMA(oc[rt][c], al, __builtin_bit_cast(h8, wo[c][s][0]));
                }
            }
        }
        // The [32 x 256] partial leaves through LDS as 16-byte row stores (round 6; staged over the K / V planes, which nobody reads past the barrier above).  Rounds 4-5 stored it
        // straight from the accumulators, one dword per lane and instruction: 8 192 four-byte write-through stores per workgroup, each its own fabric write (MI355X_MICROARCH.md:
        // a dword sc1 store costs ~6x a dwordx4's time per byte).
        float *Cs = stage_over_kv(TP, TPP) ? reinterpret_cast<float *>(smraw) : reinterpret_cast<float *>(ql + QT * KHS);       // [32][OCS]
        const float osc = PLANES_IN ? red[0][0] : uv;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    Cs[(rt * 16 + kq * 4 + r) * OCS + (2 * wave + c) * 16 + li] = (om[rt][c][r] + oc[rt][c][r] * idf_ffn_h2::LO_UNSCALE) * osc;
    }
    __syncthreads();
    {
        float *slab = slabs + (size_t)h * pstride;
        const float *Cs = stage_over_kv(TP, TPP) ? reinterpret_cast<const float *>(smraw) : reinterpret_cast<const float *>(ql + QT * KHS);
#pragma unroll
        for (int it = 0; it < QT * (D / 4) / NTH; ++it) {
            const int row = (tid >> 6) + it * NWV, c4 = (tid & 63) << 2, t = q0 + row;
            if (t < T) idf_store16_wt(slab + (rowbase + t) * D + c4, *reinterpret_cast<const float4 *>(Cs + row * OCS + c4));
        }
    }
    IDF_AH2_STAMP(6);                                    // out-projection + stores issued
}

// qkv: fp32 [N][768], scales null -- or the QKV kernel's plane pairs with their per-row scales [N][4] (see the kernel).  qkv null: availability query (the launch-time check of the
// kernel that WOULD run, nothing launched): the caller decides the QKV projection's output form before it launches that.
inline int launch_self_attn_h2(hipStream_t s, const float *qkv, int B, int T, const float *wo_h2, float *slabs, size_t pstride, const float *scales = nullptr, bool planes = false) {
    static idf_excl_cache excl, excl_p;
    const bool pl = planes || scales != nullptr;
    const int dyn = pl ? idf_exclusive_cu(reinterpret_cast<const void *>(&self_attn_h2_kernel<0, true>), "self_attn_h2_kernel<planes in>", NTH, excl_p)
                       : idf_exclusive_cu(reinterpret_cast<const void *>(&self_attn_h2_kernel<0, false>), "self_attn_h2_kernel", NTH, excl);      // the whole CU's LDS minus the kernel's static words (exclusive CU)
    if (dyn < 0) return IDF_NOT_EXCLUSIVE;
    if (T > MAX_T || (int)lds_bytes(T) > dyn) return IDF_E_INVAL;
    if (!qkv) return IDF_OK;
    const int nwg = (int)(idf_cdiv(T, QT) * H * B);
    if (pl) hipLaunchKernelGGL((self_attn_h2_kernel<0, true>), dim3((unsigned)nwg), dim3(NTH), (size_t)dyn, s, qkv, T, nwg, wo_h2, slabs, pstride, scales);
    else hipLaunchKernelGGL((self_attn_h2_kernel<0, false>), dim3((unsigned)nwg), dim3(NTH), (size_t)dyn, s, qkv, T, nwg, wo_h2, slabs, pstride, (const float *)nullptr);
    return IDF_OK;
}

}  // namespace idf_attn_h2
