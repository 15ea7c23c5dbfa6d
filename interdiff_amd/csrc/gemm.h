// Generic tiled fp32-MFMA GEMM for the token-matrix contractions of the denoiser (gfx950).
//
//   C[M,N] = epi( pro(A)[M,K] . W[N,K]^T + bias )
//
// * arithmetic: v_mfma_f32_16x16x4_f32 -- exact fp32 products, fp32 accumulate (no reduced precision);
// * workgroup tile BM x BN, WM x WN waves, each wave owns a (BM/WM) x (BN/WN) tile = TM x TN accumulators
//   of 16x16 (>= 2 independent accumulators per wave keep the 40-cycle dependent latency hidden);
// * LDS images are quad-major ([k/4][row][4], padded by one quad per plane): the four k-steps of a 16-wide
//   k-group come from ONE ds_read_b128 per operand tile, and the register->LDS write of a 128-B row segment
//   is conflict-free;
// * global->register prefetch of chunk k+1 is issued before the MFMAs of chunk k (double-buffered LDS,
//   one barrier per chunk).
// A producers:  A_PLAIN row-major; A_LN LayerNorm over the 256-wide row on load (whole normalised rows are
// parked in LDS once); A_TOKT gathers token rows from the sampler layout x[b][c][t] (row = b*T+t, k = c).
// Epilogues: bias | gelu | +residual | heads (transposed store back to [b][c][t]) | embed (+temb[ts[b]]+pe[t]).
#pragma once
#include "common.h"
#include "philox.h"

namespace idf_gemm {

enum { A_PLAIN = 0, A_LN = 1, A_TOKT = 2, A_TOKT_R = 3 };      // A_TOKT_R: token gather for a token width that is no multiple of 4 (Args.Ka)
constexpr bool is_tokt(int a) { return a == A_TOKT || a == A_TOKT_R; }
enum { E_BIAS = 0, E_GELU = 1, E_RESID = 2, E_HEADS = 3, E_EMBED = 4, E_HEADS_POST = 5, E_HEADS_POST_RAGGED = 6 };      // 6: E_HEADS_POST for T % 4 != 0 (per-row update)
constexpr bool is_post(int epi) { return epi == E_HEADS_POST || epi == E_HEADS_POST_RAGGED; }

struct Args {
    const float *A;
    int lda, K;
    int Ka;                         // A_TOKT_R: channels of the gathered tensor x [b][Ka][t] (K = Ka rounded up to 4: W's zero-padded row length)
    const float *lnw, *lnb;         // A_LN (null lnw: rows copied unnormalised)
    size_t a_pstride;               // A_LN with NP > 1 (template parameter): A is NP partial slabs a_pstride floats apart, summed on load (common.h ld4_sum)
    const float *W;                 // [N][K]
    const float *bias;              // [N] or null
    float *C;
    int ldc, M, N;
    float *xn_out;                  // A_LN: normalised rows [M][256] written by the blockIdx.y == 0 column (nullable)
    const float *resid;             // E_RESID, leading dimension ldc
    int T;                          // tokens per clip (A_TOKT, E_HEADS, E_EMBED)
    const int64_t *ts;              // E_EMBED: timestep per clip
    const float *temb, *pe;         // E_EMBED: [n_steps][N], [max_T][N]
    int n_steps;
    // E_HEADS_POST: the x0 tile never reaches HBM -- inpainting, posterior mean and the noise add run on it in the epilogue and
    // overwrite the sampler state x in place ([b][c][t], same indexing as C).  LDS-DMA kernel only.
    float *post_x;
    const float *post_gt;           // nullable together with post_mask
    const uint8_t *post_mask;
    const float *post_table;        // [steps][4] = {c1, c2, sigma, .}
    const int64_t *post_state;      // int64[8]: [2] seed, [4..5] this step's {t, loop index}
#ifdef IDF_GEMM_PROBE
    long long *probe;               // tools/gemm_probe.hip only (built with -DIDF_GEMM_PROBE): per-workgroup s_memtime stamps
#endif
};

// XCD-aware tile order (workgroup b is dispatched to XCD b % 8, each XCD has its own L2): remap the linear
// workgroup id so that one XCD works through CONSECUTIVE logical tiles, with the N tiles of an M tile adjacent --
// the A rows of an M tile are then fetched into ONE L2 instead of up to eight (speed only; any order is correct).
__device__ __forceinline__ void xcd_tile(int ntn, int &mt, int &nt, int &wg) {
    const int nwg = gridDim.x, id = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = id & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
    mt = wg / ntn;
    nt = wg - mt * ntn;
}

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// phase stamps exist only in the probe build of tools/gemm_probe.hip; the product kernels carry no pointer and no branch for them
#ifdef IDF_GEMM_PROBE
#define IDF_PROBE_STAMP(g, wg, slot) do { if ((g).probe && threadIdx.x == 0) (g).probe[(wg) * 4 + (slot)] = clock64(); } while (0)
#else
#define IDF_PROBE_STAMP(g, wg, slot) do { } while (0)
#endif

#define IDF_MFMA4(acc, a, b)                                            \
    do {                                                                \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0); \
    } while (0)


// ---- epilogue shared by both kernels.  C/D layout of 16x16x4: col = lane & 15, row = (lane >> 4) * 4 + reg.
// rbase0 / col0: first row / column of this lane in accumulator tile (0,0); tiles step by 16.
template <int TN>
__device__ __forceinline__ void load_bias(const Args &g, float (&bv)[TN], int col0) {
#pragma unroll
    for (int j = 0; j < TN; ++j) bv[j] = g.bias ? g.bias[min(col0 + j * 16, g.N - 1)] : 0.f;
}

template <int TM, int TN>
__device__ __forceinline__ void load_resid(const Args &g, float (&rres)[TM][TN][4], int rbase0, int col0) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = min(col0 + j * 16, g.N - 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) rres[i][j][r] = g.resid[(size_t)min(rbase0 + i * 16 + r, g.M - 1) * g.ldc + col];
        }
}

// E_EMBED: temb[ts[b]][col] + pe[t][col] of every accumulator element, requested before the k-loop like the residual (the
// ts -> temb row chain is two dependent round trips; in the epilogue they would follow the matrix work)
template <int TM, int TN>
__device__ __forceinline__ void load_embed_add(const Args &g, float (&rres)[TM][TN][4], int rbase0, int col0) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rc = min(rbase0 + i * 16 + r, g.M - 1), b = rc / g.T, t = rc - b * g.T;
            int64_t step = g.ts[b];
            step = step < 0 ? 0 : (step >= g.n_steps ? g.n_steps - 1 : step);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int cc = min(col0 + j * 16, g.N - 1);
                rres[i][j][r] = g.temb[(size_t)step * g.N + cc] + g.pe[(size_t)t * g.N + cc];
            }
        }
}

// E_HEADS_POST operands of one lane, requested BEFORE the k-loop (the sampler state -> coefficient row -> x / gt / mask chain is
// three dependent memory round trips, and the Philox + Box-Muller noise is ~500 VALU instructions: in the epilogue they would
// run after the matrix work of a workgroup that has the CU to itself; up front they hide behind the first operand fetches)
template <int TM, int TN>
struct PostOperands {
    float4 xv[TM][TN], gv[TM][TN], e[TM][TN];
    uchar4 mk[TM][TN];
    float c1, c2, sigma;
};
// component k (0..7) of the eight normals of two consecutive Philox groups
__device__ __forceinline__ float sel8(const float4 a, const float4 b, int k) {
    const float lo = k == 0 ? a.x : (k == 1 ? a.y : (k == 2 ? a.z : a.w));
    const float hi = k == 4 ? b.x : (k == 5 ? b.y : (k == 6 ? b.z : b.w));
    return k < 4 ? lo : hi;
}
// A lane's four accumulator rows are four consecutive frames of one clip = one aligned float4 of x[b][col][t..t+3] when T % 4 == 0
// (the timed shapes).  Any other clip length (the reference's default T = 35) takes the per-row form: frame r of the lane sits at
// its own flat index -- unaligned, possibly in the next clip -- and draws component (index & 3) of Philox group (index >> 2), exactly
// what interdiff_posterior_step_dev gives that element: two groups cover four consecutive elements, a third call only for the rows
// that cross into the next clip.
template <int TM, int TN, bool ragged>
__device__ __forceinline__ void post_prefetch(const Args &g, PostOperands<TM, TN> &po, int rbase0, int col0) {
    const int64_t st = g.post_state[4];                 // {t, loop index} of THIS step, parked by sampler_prepare_step (philox.h)
    const uint64_t it = (uint64_t)g.post_state[5], seed = (uint64_t)g.post_state[2];
    const size_t elem0 = (size_t)g.post_state[6];       // position of x[0] inside the whole sample (a chain of a split batch draws the whole batch's noise)
    po.c1 = g.post_table[st * 4]; po.c2 = g.post_table[st * 4 + 1]; po.sigma = g.post_table[st * 4 + 2];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = min(col0 + j * 16, g.N - 1);
            if constexpr (!ragged) {
                const int rbase = min(rbase0 + i * 16, g.M - 4);      // clamped: out-of-tile lanes load valid addresses and store nothing
                const int b = rbase / g.T, t = rbase - b * g.T;
                const size_t flat = ((size_t)b * g.N + col) * g.T + t;
                po.xv[i][j] = ld4(g.post_x + flat);
                po.gv[i][j] = g.post_mask ? ld4(g.post_gt + flat) : zero4();
                po.mk[i][j] = g.post_mask ? *reinterpret_cast<const uchar4 *>(g.post_mask + flat) : make_uchar4(0, 0, 0, 0);
                po.e[i][j] = randn4(seed, it, (uint64_t)((flat + elem0) >> 2));
            } else {
                const int r0 = min(rbase0 + i * 16, g.M - 1), b0 = r0 / g.T;
                const size_t idx0 = ((size_t)b0 * g.N + col) * g.T + (r0 - b0 * g.T) + elem0;
                const float4 ea = randn4(seed, it, (uint64_t)(idx0 >> 2)), eb = randn4(seed, it, (uint64_t)(idx0 >> 2) + 1);
                float xr[4], gr[4], er[4];
                unsigned char mr[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = min(rbase0 + i * 16 + r, g.M - 1), b = row / g.T;
                    const size_t flat = ((size_t)b * g.N + col) * g.T + (row - b * g.T);
                    xr[r] = g.post_x[flat];
                    gr[r] = g.post_mask ? g.post_gt[flat] : 0.f;
                    mr[r] = g.post_mask ? g.post_mask[flat] : (unsigned char)0;
                    if (b == b0) er[r] = sel8(ea, eb, (int)(idx0 & 3) + (row - r0));
                    else {
                        const size_t idx = flat + elem0;
                        const float4 ec = randn4(seed, it, (uint64_t)(idx >> 2));
                        er[r] = sel8(ec, ec, (int)(idx & 3));
                    }
                }
                po.xv[i][j] = make_float4(xr[0], xr[1], xr[2], xr[3]);
                po.gv[i][j] = make_float4(gr[0], gr[1], gr[2], gr[3]);
                po.mk[i][j] = make_uchar4(mr[0], mr[1], mr[2], mr[3]);
                po.e[i][j] = make_float4(er[0], er[1], er[2], er[3]);
            }
            asm volatile("" : "+v"(po.e[i][j].x), "+v"(po.e[i][j].y), "+v"(po.e[i][j].z), "+v"(po.e[i][j].w));     // computed here, not sunk into the epilogue
        }
}
template <int TM, int TN, bool ragged>
__device__ __forceinline__ void epilogue_post(const Args &g, const f32x4 (&acc)[TM][TN], const float (&bvs)[TN], const PostOperands<TM, TN> &po,
                                              int rbase0, int col0, float *lds_tile = nullptr, int lds_stride = 0) {
    // lds_tile (tail_h2.h, TM = TN = 1): the updated values also go to an LDS tile -- lds_tile[r * lds_stride] = row r of the lane's four -- for the next step's embedding
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int rbase = rbase0 + i * 16, col = col0 + j * 16;
            if (col >= g.N || rbase >= g.M) continue;
            const float bv = bvs[j];
            float4 pv = make_float4(acc[i][j][0] + bv, acc[i][j][1] + bv, acc[i][j][2] + bv, acc[i][j][3] + bv);
            const uchar4 m = po.mk[i][j];
            const float4 gv = po.gv[i][j];
            pv.x = m.x ? gv.x : pv.x; pv.y = m.y ? gv.y : pv.y; pv.z = m.z ? gv.z : pv.z; pv.w = m.w ? gv.w : pv.w;
            const float4 out = posterior4(po.c1, po.c2, po.sigma, pv, po.xv[i][j], po.e[i][j]);
            if (lds_tile) {
                lds_tile[0] = out.x; lds_tile[lds_stride] = out.y; lds_tile[2 * lds_stride] = out.z; lds_tile[3 * lds_stride] = out.w;
            }
            if constexpr (!ragged) {
                const int b = rbase / g.T, t = rbase - b * g.T;
                idf_store16_wt(g.post_x + ((size_t)b * g.N + col) * g.T + t, out);      // the next step's embedding reads x from other XCDs
            } else {
                const float ov[4] = {out.x, out.y, out.z, out.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = rbase + r, b = row / g.T;
                    if (row < g.M) idf_store4_wt(g.post_x + ((size_t)b * g.N + col) * g.T + (row - b * g.T), ov[r]);
                }
            }
        }
}

template <int TM, int TN, int EPI>
__device__ __forceinline__ void epilogue(const Args &g, const f32x4 (&acc)[TM][TN], const float (&rres)[TM][TN][4],
                                         const float (&bvs)[TN], int rbase0, int col0) {
    const int M = g.M, N = g.N;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int rbase = rbase0 + i * 16;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = col0 + j * 16;
            const bool colok = col < N;
            const float bv = bvs[j];
            if constexpr (EPI == E_HEADS) {
                // rows rbase..rbase+3 are 4 consecutive frames of one clip when T % 4 == 0: one 16-B store
                if (!colok) continue;
                const int b = rbase / g.T, t = rbase - b * g.T;
                if ((g.T & 3) == 0 && rbase + 3 < M) {
                    *reinterpret_cast<float4 *>(g.C + ((size_t)b * N + col) * g.T + t) =
                        make_float4(acc[i][j][0] + bv, acc[i][j][1] + bv, acc[i][j][2] + bv, acc[i][j][3] + bv);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = rbase + r;
                        if (row >= M) continue;
                        const int bb = row / g.T, tt = row - bb * g.T;
                        g.C[((size_t)bb * N + col) * g.T + tt] = acc[i][j][r] + bv;
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = rbase + r;
                    float v = acc[i][j][r] + bv;
                    if constexpr (EPI == E_GELU) v = gelu_fast(v);
                    if constexpr (EPI == E_RESID) v += rres[i][j][r];
                    if constexpr (EPI == E_EMBED) {
                        const int rc = min(row, M - 1), b = rc / g.T, t = rc - b * g.T, cc = min(col, N - 1);
                        int64_t step = g.ts[b];
                        step = step < 0 ? 0 : (step >= g.n_steps ? g.n_steps - 1 : step);
                        v += g.temb[(size_t)step * N + cc] + g.pe[(size_t)t * N + cc];
                    }
                    if (row < M && colok) g.C[(size_t)row * g.ldc + col] = v;
                }
            }
        }
    }
}

// Row-major epilogues (bias | gelu | +residual) leave through LDS: the accumulator layout (16 columns x 4 rows per lane group)
// would store 64-B row segments four bytes per lane; staged in `cs` (BM x (BN+4) floats, free once the k-loop's last barrier
// has passed), every lane stores 16 B and a wave covers whole 128-B lines.  Same arithmetic as epilogue<>: bit-identical output.
// Needs N % 4 == 0, ldc % 4 == 0 and a 16-byte aligned C (true at every call site; checked in interdiff_gemm_f32).
template <int BM, int BN, int TM, int TN, int EPI, int NT>
__device__ __forceinline__ void epilogue_rows(const Args &g, float *cs, const f32x4 (&acc)[TM][TN], const float (&rres)[TM][TN][4],
                                              const float (&bvs)[TN], bool writer, int wrow0, int wcol0, int kq, int li, int m0, int n0,
                                              int tid) {
    constexpr int CS = BN + 4, Q = BN / 4;
    if (writer) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[i][j][r] + bvs[j];
                    if constexpr (EPI == E_GELU) v = gelu_fast(v);
                    if constexpr (EPI == E_RESID || EPI == E_EMBED) v += rres[i][j][r];
                    cs[(wrow0 + i * 16 + kq * 4 + r) * CS + wcol0 + j * 16 + li] = v;
                }
    }
    __syncthreads();
    for (int idx = tid; idx < BM * Q; idx += NT) {
        const int row = idx / Q, c4 = idx - row * Q, gr = m0 + row, gc = n0 + c4 * 4;
        if (gr < g.M && gc < g.N) idf_store16_wt(g.C + (size_t)gr * g.ldc + gc, ld4(cs + row * CS + c4 * 4));      // read next by other XCDs: write through
    }
}

template <int BM, int BN, int WM, int WN, int KC, int APRO, int EPI, int NP = 1>
__global__ __launch_bounds__(WM *WN * 64) void gemm_kernel(const Args g) {
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int QC = KC / 4;                            // quads per k-chunk
    constexpr int AQ = BM * 4 + 4, BQ = BN * 4 + 4;       // padded plane strides (floats)
    constexpr int KLN = 256;                              // A_LN row width
    constexpr int A_FLOATS = APRO == A_LN ? (KLN / 4) * AQ : 2 * QC * AQ;
    constexpr int LA = (BM * QC + NT - 1) / NT, LB = (BN * QC + NT - 1) / NT;
    static_assert(TM >= 1 && TN >= 1 && QC % 4 == 0, "tile shape");
    __shared__ __attribute__((aligned(16))) float smem[A_FLOATS + 2 * QC * BQ];
    float *As = smem, *Bs = smem + A_FLOATS;

    idf_args_now(g.A, g.lda, g.K, g.lnw, g.lnb, g.a_pstride, g.W, g.bias, g.C, g.ldc, g.M, g.N, g.xn_out, g.resid, g.T, g.ts, g.temb, g.pe, g.n_steps,
                 g.post_x, g.post_gt, g.post_mask, g.post_table, g.post_state, gridDim.x);       // the whole argument block into SGPRs now (common.h)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int mt_, nt_, wg;
    xcd_tile((g.N + BN - 1) / BN, mt_, nt_, wg);
    const int m0 = mt_ * BM, n0 = nt_ * BN;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 15, kq = lane >> 4;
    const int K = g.K, M = g.M, N = g.N;
    const int nk = (K + KC - 1) / KC;

    float4 areg[LA], breg[LB];
    auto load_b = [&](int kc) {
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const int f = tid + NT * i;
            if ((BN * QC) % NT == 0 || f < BN * QC) {
                const int col = f / QC, q = f % QC, n = n0 + col, k = kc * KC + q * 4;
                breg[i] = (n < N && k < K) ? ld4(g.W + (size_t)n * K + k) : zero4();
            }
        }
    };
    auto store_b = [&](int buf) {
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const int f = tid + NT * i;
            if ((BN * QC) % NT == 0 || f < BN * QC) {
                const int col = f / QC, q = f % QC;
                *reinterpret_cast<float4 *>(Bs + (buf * QC + q) * BQ + col * 4) = breg[i];
            }
        }
    };
    auto load_a = [&](int kc) {
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const int f = tid + NT * i;
            if ((BM * QC) % NT == 0 || f < BM * QC) {
                if constexpr (APRO == A_TOKT) {
                    const int row = f % BM, q = f / BM, m = m0 + row, k = kc * KC + q * 4;
                    float4 v = zero4();
                    if (m < M && k < K) {
                        const int b = m / g.T, t = m - b * g.T;
                        const float *p = g.A + ((size_t)b * K + k) * g.T + t;
                        v.x = p[0];
                        v.y = p[(size_t)g.T];
                        v.z = p[(size_t)2 * g.T];
                        v.w = p[(size_t)3 * g.T];
                    }
                    areg[i] = v;
                } else if constexpr (APRO == A_TOKT_R) {      // x has Ka channels (BASELINE config #1: 106); channels Ka .. K-1 are zeros against W's zero columns
                    const int row = f % BM, q = f / BM, m = m0 + row, k = kc * KC + q * 4, Ka = g.Ka;
                    float4 v = zero4();
                    if (m < M && k < Ka) {
                        const int b = m / g.T, t = m - b * g.T;
                        const float *p = g.A + ((size_t)b * Ka + k) * g.T + t;
                        const float y = p[(size_t)min(1, Ka - 1 - k) * g.T], z = p[(size_t)min(2, Ka - 1 - k) * g.T], w_ = p[(size_t)min(3, Ka - 1 - k) * g.T];
                        v.x = p[0];
                        v.y = k + 1 < Ka ? y : 0.f;
                        v.z = k + 2 < Ka ? z : 0.f;
                        v.w = k + 3 < Ka ? w_ : 0.f;
                    }
                    areg[i] = v;
                } else {
                    const int row = f / QC, q = f % QC, m = m0 + row, k = kc * KC + q * 4;
                    areg[i] = (m < M && k < K) ? ld4(g.A + (size_t)m * g.lda + k) : zero4();
                }
            }
        }
    };
    auto store_a = [&](int buf) {
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const int f = tid + NT * i;
            if ((BM * QC) % NT == 0 || f < BM * QC) {
                const int row = is_tokt(APRO) ? f % BM : f / QC, q = is_tokt(APRO) ? f / BM : f % QC;
                *reinterpret_cast<float4 *>(As + (buf * QC + q) * AQ + row * 4) = areg[i];
            }
        }
    };

    IDF_PROBE_STAMP(g, wg, 0);
    load_b(0);
    // epilogue operands requested up front: they are in registers long before the k-loop ends
    float rres[TM][TN][4], bvs[TN];
    load_bias<TN>(g, bvs, n0 + wn * TN * 16 + li);
    if constexpr (EPI == E_RESID) load_resid<TM, TN>(g, rres, m0 + wm * TM * 16 + kq * 4, n0 + wn * TN * 16 + li);
    if constexpr (EPI == E_EMBED) load_embed_add<TM, TN>(g, rres, m0 + wm * TM * 16 + kq * 4, n0 + wn * TN * 16 + li);
    if constexpr (APRO == A_LN) {
        // whole rows: wave w owns rows w, w+NW, ...; a lane holds 4 consecutive features of the 256-wide row
        constexpr int NW = WM * WN;
        const float4 gw = g.lnw ? ld4(g.lnw + lane * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 gb = g.lnw ? ld4(g.lnb + lane * 4) : zero4();
#pragma unroll
        for (int i = 0; i < BM / NW; ++i) {
            const int row = wave + NW * i, m = m0 + row;
            float4 v = ld4_sum<NP>(g.A + (size_t)min(m, M - 1) * g.lda + lane * 4, g.a_pstride);     // clamped, unguarded: all rows in flight together (rows >= M are never stored)
            if (g.lnw) {
                float mean, rstd;
                ln_row_stats(v, mean, rstd);
                v.x = (v.x - mean) * rstd * gw.x + gb.x;
                v.y = (v.y - mean) * rstd * gw.y + gb.y;
                v.z = (v.z - mean) * rstd * gw.z + gb.z;
                v.w = (v.w - mean) * rstd * gw.w + gb.w;
            }
            *reinterpret_cast<float4 *>(As + lane * AQ + row * 4) = v;
            if (g.xn_out && nt_ == 0 && m < M) *reinterpret_cast<float4 *>(g.xn_out + (size_t)m * KLN + lane * 4) = v;
        }
    } else {
        load_a(0);
        store_a(0);
    }
    store_b(0);
    __syncthreads();
    IDF_PROBE_STAMP(g, wg, 1);

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int kc = 0; kc < nk; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < nk) {
            load_b(kc + 1);
            if constexpr (APRO != A_LN) load_a(kc + 1);
        }
        const float *Ab = APRO == A_LN ? As + kc * QC * AQ : As + buf * QC * AQ;
        const float *Bb = Bs + buf * QC * BQ;
#pragma unroll
        for (int s = 0; s < QC / 4; ++s) {
            float4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = ld4(Ab + (s * 4 + kq) * AQ + ((wm * TM + i) * 16 + li) * 4);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = ld4(Bb + (s * 4 + kq) * BQ + ((wn * TN + j) * 16 + li) * 4);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) IDF_MFMA4(acc[i][j], a[i].x, b[j].x);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) IDF_MFMA4(acc[i][j], a[i].y, b[j].y);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) IDF_MFMA4(acc[i][j], a[i].z, b[j].z);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) IDF_MFMA4(acc[i][j], a[i].w, b[j].w);
        }
        if (kc + 1 < nk) {
            store_b(buf ^ 1);
            if constexpr (APRO != A_LN) store_a(buf ^ 1);
        }
        __syncthreads();
    }

    IDF_PROBE_STAMP(g, wg, 2);
    if constexpr (EPI == E_BIAS || EPI == E_GELU || EPI == E_RESID || EPI == E_EMBED) {
        static_assert(BM * (BN + 4) <= A_FLOATS + 2 * QC * BQ, "C tile must fit the operand buffers");
        epilogue_rows<BM, BN, TM, TN, EPI, NT>(g, smem, acc, rres, bvs, true, wm * TM * 16, wn * TN * 16, kq, li, m0, n0, tid);
    } else {
        epilogue<TM, TN, EPI>(g, acc, rres, bvs, m0 + wm * TM * 16 + kq * 4, n0 + wn * TN * 16 + li);
    }
    IDF_PROBE_STAMP(g, wg, 3);
}

template <int BM, int BN, int WM, int WN, int KC, int APRO, int EPI, int NP = 1>
inline void launch(hipStream_t s, const Args &g) {
    dim3 grid((unsigned)(idf_cdiv(g.M, BM) * idf_cdiv(g.N, BN)));
    hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, KC, APRO, EPI, NP>), grid, dim3(WM * WN * 64), 0, s, g);
}

// ------------------------------------------------------------------------------------------------------------
// LDS-DMA pipelined variant (A_PLAIN / A_LN, K % KC == 0).
//  * both operands go global -> LDS with global_load_lds_dwordx4 (no VGPR round trip, no ds_write pass).  The LDS
//    image of a wave instruction is lane-linear (1 KiB), so tiles are stored ROW-MAJOR and unpadded: KC/4 consecutive
//    lanes fetch one contiguous KC*4-byte row segment (coalesced), and bank conflicts are removed by an XOR swizzle
//    of the 16-B chunk index with the row number, applied to the SOURCE address on the way in and to the ds_read
//    address on the way out (the same involution on both sides);
//  * an MFMA operand fragment for a 16-wide k-group is still ONE ds_read_b128 (lane (i, kq) reads chunk 4s+kq of
//    row i: k = 16s + 4kq + {0..3});
//  * NS (3..6) LDS stages: chunks k+1 .. k+NS-1 are in flight while chunk k is on the MFMAs; the only wait in the loop
//    is a COUNTED s_waitcnt vmcnt((NS-2)*LPC) (chunk k+1 landed, the later ones still flying) followed by ONE raw
//    s_barrier per chunk -- the lookahead has to cover ~1 us of loaded global->LDS latency.  The DMA instruction is issued from
//    inline asm (common.h): only then does the counted wait really keep chunks in flight;
//  * KS = 2 / 4 splits the 16-wide k-groups of every chunk over that many wave sets (2 / 4 waves per SIMD at one
//    workgroup per CU), reduced through LDS at the end -- for the GEMMs whose grid cannot fill the chip twice.
// ------------------------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N <= 16, "vmcnt literal");
#define IDF_VMCNT_CASE(n) if constexpr (N == n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory");
    IDF_VMCNT_CASE(0) IDF_VMCNT_CASE(1) IDF_VMCNT_CASE(2) IDF_VMCNT_CASE(3) IDF_VMCNT_CASE(4) IDF_VMCNT_CASE(5) IDF_VMCNT_CASE(6)
    IDF_VMCNT_CASE(7) IDF_VMCNT_CASE(8) IDF_VMCNT_CASE(9) IDF_VMCNT_CASE(10) IDF_VMCNT_CASE(11) IDF_VMCNT_CASE(12)
    IDF_VMCNT_CASE(13) IDF_VMCNT_CASE(14) IDF_VMCNT_CASE(15) IDF_VMCNT_CASE(16)
#undef IDF_VMCNT_CASE
}

template <int BM, int BN, int WM, int WN, int KS, int KC, int APRO, int EPI, int NS = 3, int NP = 1>
__global__ __launch_bounds__(WM *WN *KS * 64) void gemm_glds_kernel(const Args g) {
    constexpr int NWT = WM * WN, NW = NWT * KS;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int CH = KC / 4;                                  // 16-B chunks per tile row (NS = LDS stages)
    constexpr int RSL = 256 + 4;                                // A_LN: padded row stride of the normalised rows
    constexpr int A_STAGE = APRO == A_LN ? 0 : BM * KC, STAGE = A_STAGE + BN * KC;
    constexpr int A_LN_FLOATS = APRO == A_LN ? BM * RSL : 0;
    constexpr int IA = A_STAGE / 256, IB = BN * KC / 256, IPC = IA + IB, LPC = IPC / NW;
    constexpr int RED = KS > 1 ? (KS - 1) * NWT * 64 * TM * TN * 4 : 0;
    static_assert(IPC % NW == 0 && LPC >= 1 && (NS - 2) * LPC <= 16 && NS >= 3 && NS <= 6, "chunk loads must split evenly over the waves");
    static_assert((CH / 4) % KS == 0 && TM >= 1 && TN >= 1 && (CH == 8 || CH == 16), "tile shape");
    static_assert(APRO != A_TOKT, "token gather uses the register-staged kernel");
    constexpr int SMEM = A_LN_FLOATS + NS * STAGE;
    __shared__ __attribute__((aligned(1024))) float smem[SMEM > RED ? SMEM : RED];
    float *Aln = smem, *stages = smem + A_LN_FLOATS;

    idf_args_now(g.A, g.lda, g.K, g.lnw, g.lnb, g.a_pstride, g.W, g.bias, g.C, g.ldc, g.M, g.N, g.xn_out, g.resid, g.T, g.ts, g.temb, g.pe, g.n_steps,
                 g.post_x, g.post_gt, g.post_mask, g.post_table, g.post_state, gridDim.x);       // the whole argument block into SGPRs now (common.h)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int mt_, nt_, wg;
    xcd_tile((g.N + BN - 1) / BN, mt_, nt_, wg);
    const int m0 = mt_ * BM, n0 = nt_ * BN;
    const int ks = wave / NWT, wt = wave % NWT, wm = wt / WN, wn = wt % WN;
    const int li = lane & 15, kq = lane >> 4;
    const int K = g.K, M = g.M, N = g.N;
    const int nk = K / KC;
    IDF_PROBE_STAMP(g, wg, 0);

    // per-wave DMA descriptors: instruction i = wave + NW*j of the IPC that make up one chunk; lane l of it fills
    // the 16-B LDS cell p = 64 i + l, i.e. (row p / CH, position p % CH), from source chunk (p % CH) ^ (row % CH)
    const float *src[LPC];
    int dst[LPC];
#pragma unroll
    for (int j = 0; j < LPC; ++j) {
        const int i = wave + NW * j;
        if (i < IA) {
            const int p = i * 64 + lane, row = p / CH, c = (p % CH) ^ (row % CH);
            src[j] = g.A + (size_t)min(m0 + row, M - 1) * g.lda + c * 4;
            dst[j] = i * 256;
        } else {
            const int ib = i - IA, p = ib * 64 + lane, row = p / CH, c = (p % CH) ^ (row % CH);
            src[j] = g.W + (size_t)min(n0 + row, N - 1) * K + c * 4;
            dst[j] = A_STAGE + ib * 256;
        }
    }
    // wait until chunk `c` has landed: the chunks issued after it (at most NS-2, fewer near the end) may keep flying
    auto wait_landed = [&](int c) {
        const int ahead = min(NS - 2, nk - 1 - c);
        if (ahead <= 0) wait_vmcnt<0>();
        else if (ahead == 1) wait_vmcnt<LPC>();
        else if (ahead == 2) { if constexpr (NS >= 4) wait_vmcnt<2 * LPC>(); }
        else if (ahead == 3) { if constexpr (NS >= 5) wait_vmcnt<3 * LPC>(); }
        else { if constexpr (NS >= 6) wait_vmcnt<4 * LPC>(); }
    };
    // asm DMA (common.h idf_dma16_v): with the builtin, hipcc drains vmcnt to 0 before the first ds_read after every issue and
    // the counted waits below never get to keep a chunk in flight
    const uint32_t stages_lds = idf_lds_addr(stages);
    auto issue = [&](int kc, int st) {
#pragma unroll
        for (int j = 0; j < LPC; ++j) idf_dma16_v(src[j] + kc * KC, stages_lds + (uint32_t)((st * STAGE + dst[j]) * 4));
    };

#pragma unroll
    for (int c = 0; c < NS - 1; ++c)
        if (c < nk) issue(c, c);
    float rres[TM][TN][4], bvs[TN];
    load_bias<TN>(g, bvs, n0 + wn * TN * 16 + li);
    if constexpr (EPI == E_RESID) {
        if (ks == 0) load_resid<TM, TN>(g, rres, m0 + wm * TM * 16 + kq * 4, n0 + wn * TN * 16 + li);
    }
    PostOperands<is_post(EPI) ? TM : 1, is_post(EPI) ? TN : 1> po;
    if constexpr (is_post(EPI)) {
        if (ks == 0) post_prefetch<TM, TN, EPI == E_HEADS_POST_RAGGED>(g, po, m0 + wm * TM * 16 + kq * 4, n0 + wn * TN * 16 + li);
    }
    if constexpr (APRO == A_LN) {
        const float4 gw = g.lnw ? ld4(g.lnw + lane * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 gb = g.lnw ? ld4(g.lnb + lane * 4) : zero4();
#pragma unroll
        for (int i = 0; i < BM / NW; ++i) {
            const int row = wave + NW * i, m = m0 + row;
            float4 v = ld4_sum<NP>(g.A + (size_t)min(m, M - 1) * g.lda + lane * 4, g.a_pstride);     // clamped, unguarded: all rows in flight together (rows >= M are never stored)
            if (g.lnw) {
                float mean, rstd;
                ln_row_stats(v, mean, rstd);
                v.x = (v.x - mean) * rstd * gw.x + gb.x;
                v.y = (v.y - mean) * rstd * gw.y + gb.y;
                v.z = (v.z - mean) * rstd * gw.z + gb.z;
                v.w = (v.w - mean) * rstd * gw.w + gb.w;
            }
            *reinterpret_cast<float4 *>(Aln + row * RSL + lane * 4) = v;
            if (g.xn_out && nt_ == 0 && m < M) *reinterpret_cast<float4 *>(g.xn_out + (size_t)m * 256 + lane * 4) = v;
        }
    }
    // chunk 0 landed (later chunks may still fly); the compiler's own waits for the ordinary loads above can only be stricter
    wait_landed(0);
    __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0) as a builtin: visible to the compiler's own wait counting (ffn.h)
    __builtin_amdgcn_s_barrier();
    IDF_PROBE_STAMP(g, wg, 1);

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // per-lane fragment bases: row offsets and the swizzle key of each operand tile row
    int aoff[TM], akey[TM], boff[TN], bkey[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = (wm * TM + i) * 16 + li;
        aoff[i] = APRO == A_LN ? row * RSL : row * KC;
        akey[i] = row % CH;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = (wn * TN + j) * 16 + li;
        boff[j] = row * KC;
        bkey[j] = row % CH;
    }

    int st = 0;
    for (int kc = 0; kc < nk; ++kc) {
        if (kc + NS - 1 < nk) issue(kc + NS - 1, st >= 1 ? st - 1 : NS - 1);   // stage (kc-1) % NS, last read in iteration kc-1
        const float *Ab = APRO == A_LN ? Aln + kc * KC : stages + st * STAGE;
        const float *Bb = stages + st * STAGE + A_STAGE;
#pragma unroll
        for (int s2 = 0; s2 < CH / 4 / KS; ++s2) {
            const int c = (s2 * KS + ks) * 4 + kq;                     // chunk index within the tile row
            float4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = ld4(Ab + aoff[i] + (APRO == A_LN ? c : (c ^ akey[i])) * 4);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = ld4(Bb + boff[j] + (c ^ bkey[j]) * 4);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) IDF_MFMA4(acc[i][j], a[i].x, b[j].x);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) IDF_MFMA4(acc[i][j], a[i].y, b[j].y);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) IDF_MFMA4(acc[i][j], a[i].z, b[j].z);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) IDF_MFMA4(acc[i][j], a[i].w, b[j].w);
        }
        // chunk kc+1 must have landed before anyone reads it; the chunks issued after it may keep flying
        wait_landed(kc + 1);
        __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0) as a builtin: visible to the compiler's own wait counting (ffn.h)
        __builtin_amdgcn_s_barrier();
        st = st == NS - 1 ? 0 : st + 1;
    }
    IDF_PROBE_STAMP(g, wg, 2);

    if constexpr (KS > 1) {
        // all LDS reads of the k-loop are behind the last barrier: reuse the buffer for the partial tiles
        constexpr int SLAB = NWT * 64 * TM * TN * 4;
        float *red = smem + (size_t)(wt * 64 + lane) * (TM * TN * 4);
        if (ks > 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) *reinterpret_cast<f32x4 *>(red + (ks - 1) * SLAB + (i * TN + j) * 4) = acc[i][j];
        }
        __syncthreads();
        if (ks == 0) {
#pragma unroll
            for (int q = 0; q < KS - 1; ++q)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] += *reinterpret_cast<const f32x4 *>(red + q * SLAB + (i * TN + j) * 4);
        }
    }
    if constexpr (EPI == E_BIAS || EPI == E_GELU || EPI == E_RESID) {
        static_assert(BM * (BN + 4) <= (SMEM > RED ? SMEM : RED), "C tile must fit the operand buffers");
        if constexpr (KS > 1) __syncthreads();                    // the split-K partials have been read
        epilogue_rows<BM, BN, TM, TN, EPI, NW * 64>(g, smem, acc, rres, bvs, ks == 0, wm * TM * 16, wn * TN * 16, kq, li, m0, n0, tid);
    } else {
        if constexpr (is_post(EPI)) {
            if (ks == 0) epilogue_post<TM, TN, EPI == E_HEADS_POST_RAGGED>(g, acc, bvs, po, m0 + wm * TM * 16 + kq * 4, n0 + wn * TN * 16 + li);
        } else {
            if (ks == 0) epilogue<TM, TN, EPI>(g, acc, rres, bvs, m0 + wm * TM * 16 + kq * 4, n0 + wn * TN * 16 + li);
        }
    }
    IDF_PROBE_STAMP(g, wg, 3);
}

template <int BM, int BN, int WM, int WN, int KS, int KC, int APRO, int EPI, int NS = 3, int NP = 1>
inline void launch_glds(hipStream_t s, const Args &g) {
    dim3 grid((unsigned)(idf_cdiv(g.M, BM) * idf_cdiv(g.N, BN)));
    hipLaunchKernelGGL((gemm_glds_kernel<BM, BN, WM, WN, KS, KC, APRO, EPI, NS, NP>), grid, dim3(WM * WN * KS * 64), 0, s, g);
}

}  // namespace idf_gemm
