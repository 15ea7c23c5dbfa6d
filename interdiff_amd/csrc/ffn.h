// Fused feed-forward block of a decoder / encoder layer for gfx950:
//
//     u3 = x2 + gelu(x2 . W1^T + b1) . W2^T + b2          (model/diffusion_smpl.py:73-120 linear1 / gelu / linear2 + residual;
//                                                           torch.nn.TransformerDecoderLayer._ff_block, sublayers.py:331-341)
//
// ONE launch per layer instead of two GEMMs with a [N,1024] round trip through HBM in between.  The token matrix of the benchmark
// (N = 1600 rows) is far too small to fill 256 CUs with M tiles alone, so the grid is (32-row M tile) x (slice of the 1024 hidden
// units): 50 x 5 = 250 workgroups at N = 1600, one per CU, each
//     phase 1   hid[32, HS]   = gelu(x2[32,256] . W1[slice]^T + b1[slice])      HS = 208 (13 MFMA column tiles; the last slice 192)
//     phase 2   part[32, 256] = hid[32, HS] . W2[:, slice]^T                     (K = HS)
// and writes its partial [32,256] tile to slab `slice` of `parts[5][N][256]` (slab 0 also carries x2 + b2).  The five slabs are summed
// by whoever reads the layer output next (LayerNorm-on-load of the next row block / QKV / heads GEMM: common.h ld4_sum), in a fixed
// order, so the result is deterministic -- there is no atomic and no cross-workgroup hand-off (MI355X_MICROARCH.md prices a GEMM->GEMM
// seam inside one launch at 5-13 us; a consumer-side sum costs four extra L2 reads per element).
//
// Work balance: 6400 (16-row, 16-hidden) units of 128 MFMAs each = 25.6 per workgroup; the 13/13/13/13/12 split is within 1.5 % of
// even, 250 of 256 CUs are busy.  Per workgroup 3328 MFMAs = 832 per SIMD = 26.6 k cycles of matrix pipe.
//
// Operand movement:
//  * x2 rows (32 x 1 KiB) go global -> LDS once by LDS-DMA; the GELU output overwrites them in place (same XOR-swizzled image), so the
//    hidden activations never leave the CU;
//  * the weights of a slice are PRE-PACKED on the host (interdiff_amd/mdm.py: pack_ffn) in exactly the order and LDS image the kernel
//    consumes: a stream of k-group chunks ([HS rows][16 k] of W1, then [256 rows][16 k] of W2), each already swizzled, so a DMA
//    instruction is a linear 1-KiB copy (perfectly coalesced, no address arithmetic in the loop);
//  * ring of 6 chunk slots (3 pairs); chunks are issued two pairs ahead, the only synchronisation in the loop is ONE counted
//    s_waitcnt vmcnt + ONE s_barrier per PAIR of k-groups, placed in the middle of the pair's MFMA stream: the fragments of the second
//    k-group are already in registers when the barrier is taken and cover the first LDS reads of the next pair.
// Arithmetic: v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulate), gelu = erf form (common.h gelu_fast).
#pragma once
#include "common.h"

namespace idf_ffn {

constexpr int D = IDF_MDM_D, FF = IDF_MDM_FF;
constexpr int BM = 32;                 // token rows per workgroup
constexpr int NSL = IDF_FFN_SLICES;    // hidden slices = partial slabs (5)
constexpr int NW = 8, NT = NW * 64;
constexpr int SLOT = 256 * 16;         // floats per ring slot: one k-group chunk (<= 256 rows x 16 k)
constexpr int NSLOT = 6;
constexpr int XS = BM * D;
constexpr int CSS = D + 4;             // row stride of the output staging tile
static_assert(BM * CSS <= NSLOT * SLOT, "output tile is staged in the ring");

__host__ __device__ inline int slice_h0(int s) { return 208 * s; }
__host__ __device__ inline int slice_hs(int s) { return s < NSL - 1 ? 208 : FF - 208 * (NSL - 1); }

__device__ __forceinline__ float4 ldsv4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

// s_waitcnt vmcnt(n) for a wave-uniform runtime n (the literal has to be an immediate)
__device__ __forceinline__ void wait_vmcnt_dyn(int n) {
    switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    }
}

#define IDF_FFN_MFMA(acc, a, b) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0)

// acc[j] += a(16 x 16k) . b[j](16k x 16) for the first `n` (3 or 4) tiles; four rounds so that consecutive MFMAs never share an accumulator
__device__ __forceinline__ void mma_group(f32x4 (&acc)[4], const float4 &a, const float4 (&b)[4], bool four) {
    IDF_FFN_MFMA(acc[0], a.x, b[0].x); IDF_FFN_MFMA(acc[1], a.x, b[1].x); IDF_FFN_MFMA(acc[2], a.x, b[2].x);
    if (four) IDF_FFN_MFMA(acc[3], a.x, b[3].x);
    IDF_FFN_MFMA(acc[0], a.y, b[0].y); IDF_FFN_MFMA(acc[1], a.y, b[1].y); IDF_FFN_MFMA(acc[2], a.y, b[2].y);
    if (four) IDF_FFN_MFMA(acc[3], a.y, b[3].y);
    IDF_FFN_MFMA(acc[0], a.z, b[0].z); IDF_FFN_MFMA(acc[1], a.z, b[1].z); IDF_FFN_MFMA(acc[2], a.z, b[2].z);
    if (four) IDF_FFN_MFMA(acc[3], a.z, b[3].z);
    IDF_FFN_MFMA(acc[0], a.w, b[0].w); IDF_FFN_MFMA(acc[1], a.w, b[1].w); IDF_FFN_MFMA(acc[2], a.w, b[2].w);
    if (four) IDF_FFN_MFMA(acc[3], a.w, b[3].w);
}

// x2 [M][256] token rows (LayerNorm-2 output: FFN input and residual), pack = the layer's packed weight stream (2 MiB),
// b1 [1024], b2 [256], parts [NSL][M][256].  grid = ceil(M/32) * NSL workgroups of 512 threads.
__global__ __launch_bounds__(NT) void ffn_fused_kernel(const float *__restrict__ x2, int M, const float *__restrict__ pack,
                                                        const float *__restrict__ b1, const float *__restrict__ b2,
                                                        float *__restrict__ parts) {
    __shared__ __attribute__((aligned(1024))) float smem[XS + NSLOT * SLOT + 256];
    float *Xs = smem, *ring = smem + XS, *Bs = ring + NSLOT * SLOT;          // Bs: the slice's linear1 bias (<= 208 floats)

    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = blockIdx.x, mt = wg / NSL, sl = wg - mt * NSL, m0 = mt * BM;
    const int HS = slice_hs(sl), h0 = slice_h0(sl), nt = HS >> 4;          // hidden column tiles of this slice = W2 k-groups
    const float *stream = idf_uniform_ptr(pack + (size_t)2 * D * h0);      // every slice occupies 2*256*HS floats of the stream
    const uint32_t lane16 = lane << 4, ring_lds = idf_lds_addr(ring);
    const int w1sz = HS * 16;                                              // floats of one W1 k-group chunk ([HS][16])
    const int nch = 16 + nt, npairs = (nch + 1) >> 1;                      // 16 k-groups of phase 1, nt of phase 2
    const int key = (4 - (li >> 2)) & 3;                                   // g4 = {0,3,2,1}[(row >> 2) & 3]: 16-B position swizzle of a [rows][16] chunk (mdm.py pack_ffn)

    auto ch_off = [&](int c) { return c < 16 ? c * w1sz : 16 * w1sz + (c - 16) * SLOT; };
    auto ch_ins = [&](int c) { return c >= nch ? 0 : (c < 16 ? nt : 16); };       // DMA instructions (1 KiB each) of chunk c
    auto pair_cnt = [&](int P) { return P < npairs ? (ch_ins(2 * P) + ch_ins(2 * P + 1) - wave + NW - 1) / NW : 0; };   // this wave's share
    auto issue_pair = [&](int P) {
        const int ca = 2 * P, na = ch_ins(ca), ntot = na + ch_ins(ca + 1);
        for (int i = wave; i < ntot; i += NW) {
            const int c = i < na ? ca : ca + 1, loc = i < na ? i : i - na;
            idf_dma16_s(stream + ch_off(c) + loc * 256, lane16, ring_lds + (uint32_t)(((c % NSLOT) * SLOT + loc * 256) * 4));
        }
    };

    // ---- phase-1 tile map: SIMD (w & 3) holds waves w and w+4; row tile (w>>1)&1, the column tiles of its half split 4|3 or 3|3
    const int r1 = (wave >> 1) & 1, half = wave & 1, hi = wave >> 2;
    const int left = (nt + 1) >> 1, hcnt = half ? nt - left : left, n_a = (hcnt + 1) >> 1;
    const int c0 = (half ? left : 0) + (hi ? n_a : 0), nct = hi ? hcnt - n_a : n_a;
    const bool four = nct == 4;

    // ---- prologue: the bias slice (one DMA, oldest in the queue: HS*4 <= 1 KiB; the pad lanes read the following bias entries /
    // the next arena block, in bounds), the x2 rows (row i = DMA instruction i, source chunk = position ^ (i & 15)), the first two
    // chunk pairs.  Every global->LDS move of this kernel is an asm DMA: no compiler-visible vector load is in flight in the loops.
    if (wave == 0) idf_dma16_s(idf_uniform_ptr(b1 + h0), lane16, idf_lds_addr(Bs));
    const uint32_t xs_lds = idf_lds_addr(Xs);
#pragma unroll
    for (int j = 0; j < BM / NW; ++j) {
        const int i = wave + NW * j;
        idf_dma16_s(idf_uniform_ptr(x2 + (size_t)min(m0 + i, M - 1) * D), (uint32_t)((lane ^ (i & 15)) << 4), xs_lds + (uint32_t)(i * D * 4));
    }
    issue_pair(0);
    issue_pair(1);
    wait_vmcnt_dyn(pair_cnt(1));                       // bias, x2 rows and pair 0 have landed (they are older); pair 1 may still fly
    __builtin_amdgcn_s_barrier();

    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 a0, a1, b0[4], b1f[4];
    const float *xrow1 = Xs + (r1 * 16 + li) * D;
    auto read1 = [&](int c, float4 &a, float4 (&b)[4]) {        // fragments of W1 k-group c
        a = ldsv4(xrow1 + (((4 * c + kq) ^ li) << 2));
        const float *sb = ring + (c % NSLOT) * SLOT + ((kq ^ key) << 2) + li * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < 3 || four) b[j] = ldsv4(sb + (c0 + j) * 256);
    };
    read1(0, a0, b0);
    for (int P = 0; P < 8; ++P) {
        issue_pair(P + 2);                               // into the slots of pair P-1 (every wave is past the barrier that followed its reads)
        read1(2 * P + 1, a1, b1f);
        mma_group(acc, a0, b0, four);
        wait_vmcnt_dyn(pair_cnt(P + 2));                 // pair P+1 has landed; pair P+2 may keep flying
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (P + 1 < 8) read1(2 * P + 2, a0, b0);
        mma_group(acc, a1, b1f, four);
    }

    // ---- hid = gelu(acc + b1) overwrites the x2 rows (every read of them is behind the last barrier), same swizzled image
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (j < 3 || four) {
            const int col = (c0 + j) * 16 + li;
            const float bv = Bs[col];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int row = r1 * 16 + kq * 4 + rr;
                Xs[row * D + ((((col >> 2) ^ (row & 15))) << 2) + (col & 3)] = gelu_fast(acc[j][rr] + bv);
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ---- phase 2: part[32,256] = hid[32,HS] . W2[:, slice]^T ; wave w: row tile w & 1, output column tiles 4 (w>>1) .. +3
    const int r2 = wave & 1, nb = (wave >> 1) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float *xrow2 = Xs + (r2 * 16 + li) * D;
    auto read2 = [&](int q, float4 &a, float4 (&b)[4]) {        // fragments of W2 k-group q (hidden units 16q .. 16q+15 of the slice)
        a = ldsv4(xrow2 + (((4 * q + kq) ^ li) << 2));
        const float *sb = ring + ((16 + q) % NSLOT) * SLOT + ((kq ^ key) << 2) + (nb * 16 + li) * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = ldsv4(sb + j * 256);
    };
    read2(0, a0, b0);
    for (int P = 8; P < npairs; ++P) {
        const int q = 2 * (P - 8);
        const bool two = q + 1 < nt;
        issue_pair(P + 2);
        if (two) read2(q + 1, a1, b1f);
        mma_group(acc, a0, b0, true);
        wait_vmcnt_dyn(pair_cnt(P + 2));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (q + 2 < nt) read2(q + 2, a0, b0);
        if (two) mma_group(acc, a1, b1f, true);
    }

    // ---- partial tile leaves through LDS as 16-byte row stores (the ring is idle: every chunk has landed and been consumed)
    float *Cs = ring;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) Cs[(r2 * 16 + kq * 4 + rr) * CSS + (nb + j) * 16 + li] = acc[j][rr];
    __syncthreads();
    float *out = parts + (size_t)sl * M * D;
    for (int idx = tid; idx < BM * (D / 4); idx += NT) {
        const int row = idx >> 6, c4 = (idx & 63) << 2, gr = m0 + row;
        if (gr >= M) continue;
        float4 v = ldsv4(Cs + row * CSS + c4);
        if (sl == 0) {                                    // slab 0 carries the residual and the output bias
            const float4 x = *reinterpret_cast<const float4 *>(x2 + (size_t)gr * D + c4), bb = *reinterpret_cast<const float4 *>(b2 + c4);
            v.x += x.x + bb.x; v.y += x.y + bb.y; v.z += x.z + bb.z; v.w += x.w + bb.w;
        }
        *reinterpret_cast<float4 *>(out + (size_t)gr * D + c4) = v;
    }
}

inline void launch_ffn(hipStream_t s, const float *x2, int M, const float *pack, const float *b1, const float *b2, float *parts) {
    hipLaunchKernelGGL(ffn_fused_kernel, dim3((unsigned)(idf_cdiv(M, BM) * NSL)), dim3(NT), 0, s, x2, M, pack, b1, b2, parts);
}

}  // namespace idf_ffn
