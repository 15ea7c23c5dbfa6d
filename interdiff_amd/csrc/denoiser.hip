// MDM denoiser, decoder path, for gfx950 (rows A1-A4 of SURVEY.md §8).
//
// Reference behaviour restated: model/diffusion_smpl.py:226-246 (_decode/forward),
// model/sublayers.py:311-375 (QaN layer), torch.nn.TransformerDecoderLayer (post-norm, gelu).
// NOT a translation: the reference runs ~40 eager torch kernels per layer over [T,B,D]
// tensors and re-projects the constant memory every step.  Here
//   * tokens are clip-major rows (row = b*T + t) of [N,256] fp32 matrices that stay L2-resident,
//   * EVERY contraction runs on the matrix pipe.  Exact form: the fp32 MFMA (v_mfma_f32_16x16x4_f32) -- the six token GEMMs (gemm.h), the temporal
//     self-attention (QK^T and PV), and the whole "row block" between two FFNs: learned-query local attention, LayerNorms and the cross attention to the
//     10-token memory.  Shipped form (tune[IDF_TUNE_FFN_MATH] = 1, round 4): the feed-forward block, the QKV projection, the row block's contractions and the
//     two token GEMMs at the ends of a step as split-f16 products on v_mfma_f32_16x16x32_f16 (ffn_h2.h, tail_h2.h: every fp32 operand as two f16 planes, three
//     MFMAs per product, fp32 accumulate: fp32-grade results); the self-attention stays on the fp32 MFMA (attn_h2.h: its split-f16 form measured slower),
//   * the learned-query local attention collapses to a [16 x 256] x [256 x 30] contraction with
//     constant pre-rotated queries Qc (interdiff_amd/mdm.py: qan_constants) + a 3-tap stencil,
//   * cross-attention to the constant memory is folded per sample into
//     scores = x.G^T + g0 and out = P.VW (interdiff_mdm_prepare_memory),
//   * LayerNorm never gets its own launch: the row-block kernel owns whole rows (LN_prev on load,
//     LN1, LN2 in place) and the QKV / heads kernels normalise on load.
//   * the feed-forward block is ONE launch (ffn.h / ffn_h2.h): linear1 -> gelu -> linear2 per (32-row tile, hidden slice), the hidden
//     activations never leave the CU; it leaves IDF_FFN_SLICES partial slabs that the next reader sums on load.
// Per step: embedding + 2 x 4 (standard layers: QKV, self-attention + out-projection, row block, FFN) + 6 x 2 (QaN layers: row block, FFN) + heads = 22 launches;
// 21 inside a captured run of plain steps, where a step's last launch also computes the next step's embedding (tail_h2.h).
#include "common.h"
#include "gemm.h"
#include "ffn.h"
#include "ffn_h2.h"
#include "tail_h2.h"
#include "attn_h2.h"

// the feed-forward block of one layer: arithmetic by tune[IDF_TUNE_FFN_MATH] (and whether the packer set the layer's split-f16 stream),
// row tile by tune[IDF_TUNE_FFN] (0: by this launch's rows)
static inline int idf_launch_layer_ffn(hipStream_t s, const idf_mdm_layer &ly, const float *ar, const int32_t *tune, const float *x2, int M, float *parts) {
    int rows = idf_ffn::ffn_rows_of_tune(tune[IDF_TUNE_FFN]);
    if (tune[IDF_TUNE_FFN_MATH] != 0 && ly.ffn_pack_h2 != 0) {
        const int rc = idf_ffn_h2::launch_ffn_h2(s, x2, M, ar + ly.ffn_pack_h2, ar + ly.ffn_b1p, ar + ly.ff2_b, parts, rows ? rows : idf_ffn::ffn_tile_for_rows(M), tune[IDF_TUNE_MISC] == 2 ? 1 : (tune[IDF_TUNE_MISC] == 3 ? 2 : (tune[IDF_TUNE_MISC] == 6 ? 3 : (tune[IDF_TUNE_MISC] == 10 ? 4 : 0))));      // (MISC = 2 / 3 / 6 / 10: slice-major affine ids / plain ids / three ring slots / eight waves without loaders, A/B only: ffn_h2.h)
        if (rc != IDF_NOT_EXCLUSIVE) return rc;        // (the kernel does not get its CU on this device -- common.h idf_exclusive_cu: the exact kernel below)
    }
    idf_ffn::launch_ffn(s, x2, M, ar + ly.ffn_pack, ar + ly.ffn_b1p, ar + ly.ff2_b, parts, rows);
    return IDF_OK;
}
#include <float.h>
#ifndef IDF_LDS_STRIDE_SET
#define IDF_LDS_STRIDE_SET 5          // LDS row strides of the row block's planes and the attention's Q / K / score images -- 5: round 5's (conflict-free fragment reads, see HS / ASK);
#endif                                // 4: round 4's (one 16-byte slot mod 16 everywhere): A/B builds only (tools/r05_ab.py, IDF_EXTRA_HIPCC_FLAGS=-DIDF_LDS_STRIDE_SET=4)

// phase stamps of the row-block kernel exist only in tools/rowblock_probe.hip (which defines the macro before including this file)
#ifndef IDF_AT_STAMP
#define IDF_AT_STAMP(i) do { } while (0)        // tools/rowblock_probe.hip: phase stamps of the self-attention kernel
#endif
#ifndef IDF_RB_STAMP
#define IDF_RB_STAMP(i) do { } while (0)
#endif

namespace {

using namespace idf_gemm;

constexpr int D = IDF_MDM_D;          // 256
constexpr int H = IDF_MDM_HEADS;      // 4
constexpr int HD = D / H;             // 64
constexpr int NQ = IDF_MDM_NQ;        // 10
constexpr int MEM = IDF_MDM_MEM;      // 10: the memory length of every BASELINE config -- the COMPACT layout of the folded memory below
constexpr int MEMX = IDF_MDM_MEM_MAX; // 16: longest memory the generic layout takes (eval_smpl_short.py:376 --past_len is a CLI argument)
constexpr int HM = H * MEM;           // 40
constexpr int HMP = 48;               // HM padded to a multiple of 16 (k-groups of the P.VW contraction)
// Layout of the folded memory's (head, slot) score columns, by the template parameter MS of the kernels that touch it:
//   MS == MEM (compact, the shipped fast path):  column = head * 10 + slot, 40 columns in 3 tiles of 16 (8 zero columns);
//   MS == MEMX (generic, any memory length 1..16 given at run time):  column = head * 16 + slot, one 16-column tile per head, slots >= mem_len masked.
// Everything that depends on it -- fragment sizes, column tiles, slots per softmax lane -- comes from here; with MS == MEM every expression below folds to
// the constants the kernels had before the generic path existed (same instructions, same bits).
template <int MS>
struct MemLay {
    static_assert(MS == MEM || MS == MEMX, "compact or generic");
    static constexpr bool GEN = MS != MEM;
    static constexpr int NCT = GEN ? 4 : 3;                 // 16-column tiles of the score matrix
    static constexpr int HMPX = 16 * NCT;                   // padded score columns = K of the P.VW contraction (fp32 form)
    static constexpr int NSLOT = GEN ? 4 : 3;               // memory slots per lane of the head softmax (quad q takes slots q, q + 4, ...)
    static constexpr int G0N = GEN ? 64 : HM;               // g0 floats per (layer, clip), indexed like the columns
    static constexpr int G_FRAG = 4 * 4 * NCT * 64 * 4;     // fp32 G fragments per (layer, clip)
    static constexpr int G_H2 = 4 * 2 * NCT * 2 * 64 * 4;   // split-f16 G plane fragments (as floats)
    static constexpr int VWT_F = D * HMPX;                  // fp32 VW fragments
    __host__ __device__ static constexpr int col(int h, int m) { return GEN ? h * 16 + m : h * MEM + m; }
};
constexpr int L = IDF_MDM_LAYERS;
constexpr int NSL = IDF_FFN_SLICES;   // partial output slabs of the fused FFN (ffn.h)

// XCD-affine workgroup order for the kernels below (speed only; any order is correct): workgroup `id` of a launch runs on XCD
// id % 8 and every XCD has its own L2, so the LOGICAL index is permuted to make one XCD work through consecutive logical
// workgroups -- the 7 token tiles of a clip (which all read the clip's folded memory G[b] / VW[b]) or the 4 query tiles of a
// (clip, head) (which all read its K and V) then fetch those operands into ONE L2 instead of up to eight.  Each XCD reaches
// HBM / Infinity Cache at only ~1/8 of the chip's rate, and at one workgroup per CU that fetch is most of what these kernels wait for.
__device__ __forceinline__ int xcd_logical_id() {
    const int nwg = gridDim.x * gridDim.y * gridDim.z, id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const int q = nwg >> 3, r = nwg & 7, xcd = id & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
}

// four dependent-free rounds of MFMAs over NA accumulators: acc[i] += a[i](k-group) . b[i](k-group)
template <int NA>
__device__ __forceinline__ void mma_rounds(f32x4 (&acc)[NA], const float4 (&a)[NA], const float4 (&b)[NA]) {
#pragma unroll
    for (int i = 0; i < NA; ++i) IDF_MFMA4(acc[i], a[i].x, b[i].x);
#pragma unroll
    for (int i = 0; i < NA; ++i) IDF_MFMA4(acc[i], a[i].y, b[i].y);
#pragma unroll
    for (int i = 0; i < NA; ++i) IDF_MFMA4(acc[i], a[i].z, b[i].z);
#pragma unroll
    for (int i = 0; i < NA; ++i) IDF_MFMA4(acc[i], a[i].w, b[i].w);
}

__device__ __forceinline__ float4 ln_apply(const float4 v, const float4 g, const float4 b) {
    float mean, rstd;
    ln_row_stats(v, mean, rstd);
    return make_float4((v.x - mean) * rstd * g.x + b.x, (v.y - mean) * rstd * g.y + b.y, (v.z - mean) * rstd * g.z + b.z,
                       (v.w - mean) * rstd * g.w + b.w);
}

// LayerNorm with gamma / beta read from LDS (staged there by DMA at kernel entry; same arithmetic as common.h ln_row16): lane l16
// of a 16-lane row group handles the chunks {l16, 16+l16, 32+l16, 48+l16} of its row
__device__ __forceinline__ void ln_row16_lds(Row16 &r, const float *w, const float *b, int l16) {
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) sum += (r.c[i].x + r.c[i].y) + (r.c[i].z + r.c[i].w);
    const float mean = row16_sum(sum) * (1.0f / 256.0f);
    float qv = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a0 = r.c[i].x - mean, a1 = r.c[i].y - mean, a2 = r.c[i].z - mean, a3 = r.c[i].w - mean;
        qv += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
    }
    const float rstd = __builtin_amdgcn_rsqf(row16_sum(qv) * (1.0f / 256.0f) + 1e-5f);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 g = *reinterpret_cast<const float4 *>(w + (i * 16 + l16) * 4), be = *reinterpret_cast<const float4 *>(b + (i * 16 + l16) * 4);
        r.c[i].x = (r.c[i].x - mean) * rstd * g.x + be.x;
        r.c[i].y = (r.c[i].y - mean) * rstd * g.y + be.y;
        r.c[i].z = (r.c[i].z - mean) * rstd * g.z + be.z;
        r.c[i].w = (r.c[i].w - mean) * rstd * g.w + be.w;
    }
}

using idf_ffn_h2::row16_store_planes;
// one K = 32 step of a split-f16 product on NA tiles: main += ah.bh, corr += ah.bl' + al'.bh   (result = main + corr 2^-11)
template <int NA>
__device__ __forceinline__ void mma_h2(f32x4 (&am)[NA], f32x4 (&ac)[NA], const idf_ffn_h2::h8 (&ah)[NA], const idf_ffn_h2::h8 (&al)[NA],
                                       const float4 (&bh)[NA], const float4 (&bl)[NA]) {
    using idf_ffn_h2::h8;
#pragma unroll
    for (int i = 0; i < NA; ++i) IDF_H2_MFMA(am[i], ah[i], __builtin_bit_cast(h8, bh[i]));
#pragma unroll
    for (int i = 0; i < NA; ++i) IDF_H2_MFMA(ac[i], ah[i], __builtin_bit_cast(h8, bl[i]));
#pragma unroll
    for (int i = 0; i < NA; ++i) IDF_H2_MFMA(ac[i], al[i], __builtin_bit_cast(h8, bh[i]));
}

// ------------------------------------------------------------------------------------
// Row block shared by QaN layers and standard layers, 16 tokens of one clip per workgroup:
//   [QAN]  x = LN_prev(u_in rows t-1 .. t+16);  logits[t][n][j] = <Qc[n][j], x[t+j-1]>  (MFMA, K split over waves)
//          u1 = x_t + sum_j c_j(t) x_{t+j-1},  c_j = sum_n wk[n] softmax_j(logits[t][n][:])
//   [!QAN] u1 = u_in row
//   x1 = LN1(u1);  scores = x1.G^T + g0 (MFMA);  P = softmax per head;  u2 = x1 + P.VW + b_out (MFMA)
//   x2 = LN2(u2)  -> HBM (the FFN input AND its residual)
// grid (ceil(T/16), B), 256 threads.
// ------------------------------------------------------------------------------------
constexpr int TR = 16;
// MFMA B-operand fragments of the row block are stored IN FRAGMENT ORDER -- [wave = K quarter][k-group of 16][tile][lane][4 floats] --
// so that one wave load instruction reads 1 KiB contiguous (8 cache lines).  Fetched from the row-major matrices the same
// instruction gathered sixteen 64-byte pieces, and the kernel spent 5.9 k of its 20 k cycles just ISSUING its first batch of
// requests (tools/rowblock_probe.hip): the vector memory pipe handles about one line request every two cycles.
//   Qc  (weights, mdm.py qan_fragments):      [4][4][3 taps][4 kq][NQ][4]: only lanes li < NQ fetch (logit columns >= NQ are never read)
//   G   (per sample, mem_fold_kernel):        [4][4][3 column tiles][64][4], column = 16 ct + li of the 40 (head, slot) pairs (+8 zero)
//   VWT (per sample, mem_fold_kernel):        [4 waves = output column quarter][3 k-groups][4 tiles][64][4]
//   (sizes per (layer, clip): MemLay<MS>::G_FRAG / G_H2 / VWT_F -- 12288 / 12288 / 12288 floats in the compact layout)
// split-f16 forms (rowblock_kernel<.., H2>): 16-byte plane fragments = the v_mfma_f32_16x16x32_f16 operand of a lane (8 halves, k = 8 kq .. 8 kq + 7 of a K = 32 step)
//   Qc  (mdm.py qan_fragments_h2):            [4 waves = K quarter][2 K steps][3 taps][2 planes][4 kq][NQ][8 halves]
//   G   (mem_fold_h2_kernel):                 [4 waves = K quarter][2 K steps][3 column tiles][2 planes][64 lanes][8 halves], values divided by 2^e
//   VWT (mem_fold_h2_kernel):                 [4 waves = output column quarter][2 K steps (probability columns 0..63, zero from 40)][4 tiles][2 planes][64][8 halves]
constexpr int VW_H2 = 4 * 2 * 4 * 2 * 64 * 4;  // 16384 floats
constexpr int RS = D + 4;             // LDS row stride of token rows (floats)

// H2 (decoder layers, tune[IDF_TUNE_FFN_MATH] == 1 and the packer's range proof): the three contractions on the f16 matrix pipe with every fp32
// operand as two f16 planes (ffn_h2.h: v = hi + lo' 2^-11, three v_mfma_f32_16x16x32_f16 per product, fp32 accumulate).  48 fp32 MFMAs of 32 cycles per
// wave and contraction -- 1 536 of the ~2 000 cycles each of those three phases took -- become 18 / 18 / 24 of 16.  Qc / G / VWT then point at the
// plane fragments (mdm.py qan_fragments_h2; mem_fold_h2_kernel) and h2_scale[b][2] holds the powers of two that G[b] / VW[b] were divided by.
// Token rows need no scaling: they are LayerNorm outputs, bounded by 16 max|gamma| + max|beta| (the packer checks that against the f16 range).
template <bool QAN, bool CROSS = true, int NP = 1, bool H2 = false, int MS = MEM>
__global__ __launch_bounds__(256) void rowblock_kernel(const float *__restrict__ u_in, const float *__restrict__ lnp_w,
                                                       const float *__restrict__ lnp_b, const float *__restrict__ Qc,
                                                       const float *__restrict__ wk, const float *__restrict__ ln1_w,
                                                       const float *__restrict__ ln1_b, const float *__restrict__ G,
                                                       const float *__restrict__ g0, const float *__restrict__ VWT,
                                                       const float *__restrict__ bout, const float *__restrict__ ln2_w,
                                                       const float *__restrict__ ln2_b, float *__restrict__ x2_out, int T,
                                                       int out_frame_major /* encoder output: row = t*B + b */,
                                                       size_t u_pstride /* NP > 1: u_in is NP partial slabs this many floats apart */,
                                                       const float *__restrict__ sa_resid /* !QAN, nullable: u1 = sum(u_in slabs) + sa_resid row + sa_bias */,
                                                       const float *__restrict__ sa_bias, const float *__restrict__ h2_scale = nullptr,
                                                       int mem_len = MEM /* MS == MEMX: slots per head actually present (1..16) */) {
    static_assert(!H2 || CROSS, "the split-f16 form exists for the decoder's row blocks");
    using idf_ffn_h2::h8;
    using ML = MemLay<MS>;
    constexpr int NCT = ML::NCT, HMPX = ML::HMPX, NSLOT = ML::NSLOT, PSX = HMPX + 4;
    const int mlen = ML::GEN ? mem_len : MEM;
    constexpr int XS = QAN ? (TR + 2) * RS : 0;
    // Row strides (halves) of the token-row planes / probability planes.  A fragment read is one ds_read_b128 per lane, lane (li, kq) -> row li, 16-byte slot s0 + kq;
    // the instruction is served in lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS table), i.e. rows {0-3, 12-15} at slot s and
    // rows {4-11} at slot s + 1 together.  With a row stride of 2 slots mod 16 (32 bytes mod 256) the first set lands on the even slots and the second on the odd ones:
    // conflict-free.  (Round 4 had a stride of 1 slot -- D + 8 halves, 64 + 8 -- where row 12 at slot s and row 11 at slot s + 1 collide in every group: 2 x the LDS cycles.)
    constexpr int HS = D + (IDF_LDS_STRIDE_SET == 4 ? 8 : 16), PHS = 64 + (IDF_LDS_STRIDE_SET == 4 ? 8 : 16);
    __shared__ __attribute__((aligned(16))) _Float16 xpl[H2 ? 2 * (TR + 2) * HS : 8];       // [hi | lo'][TR+2][HS]: LN_prev rows for the logits, then x1 for the scores
    __shared__ __attribute__((aligned(16))) _Float16 ppl[H2 ? 2 * TR * PHS : 8];            // [hi | lo'][TR][PHS]: probabilities, columns >= HM stay zero
    _Float16 *const xh = xpl, *const xl = xpl + (H2 ? (TR + 2) * HS : 0), *const ph = ppl, *const pl = ppl + (H2 ? TR * PHS : 0);
    __shared__ __attribute__((aligned(1024))) float prm[8 * 256];        // LN_prev / LN1 / LN2 gamma, beta + cross-attention output bias + self-attention output bias (DMA targets)
    __shared__ __attribute__((aligned(16))) float sm[XS + TR * RS + 4 * (NCT > 3 ? NCT : 3) * 256 + TR * PSX];
    float *xs = sm;                               // [TR+2][RS]  LN_prev rows t0-1 .. t0+16 (QAN)
    float *x1s = sm + XS;                         // [TR][RS]    x1, then u2 in place
    float *part = x1s + TR * RS;                  // [4 waves][3 taps | NCT score tiles][16x16] K-split partial tiles
    float *Ps = part + 4 * (NCT > 3 ? NCT : 3) * 256;               // [TR][PSX]

    idf_args_now(u_in, lnp_w, lnp_b, Qc, wk, ln1_w, ln1_b, G, g0, VWT, bout, ln2_w, ln2_b, x2_out, T, out_frame_major, u_pstride, sa_resid, sa_bias, h2_scale, mem_len, gridDim.x, gridDim.y);
    const int lid = xcd_logical_id(), b = lid / (int)gridDim.x, t0 = (lid - b * (int)gridDim.x) * TR;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const size_t rowbase = (size_t)b * T;
    // row passes (LayerNorms): every 16-lane group owns one token row, four rows per wave, all 16 rows in one sweep
    const int rown = wave * 4 + kq;
    const float *Gb = CROSS ? G + (size_t)b * (H2 ? ML::G_H2 : ML::G_FRAG) : nullptr, *g0b = CROSS ? g0 + b * ML::G0N : nullptr;
    const float *VWTb = CROSS ? VWT + (size_t)b * (H2 ? VW_H2 : ML::VWT_F) : nullptr;
    IDF_RB_STAMP(0);
    if constexpr (H2) {
        // EXCLUSIVE CU, like the other kernels that issue the f16 MFMA (ffn_h2.h "exclusive CU": next to such a kernel, workgroups of OTHER kernels on the same CU
        // have been seen to compute wrong values -- again with this kernel before it took its CU: the staggered-chains bit-identity test at B = 32).  The whole
        // register file (one wave per SIMD x 512 registers) and, by the dynamic LDS the launcher adds, all 160 KiB of LDS: nothing else can be resident here.
        asm volatile("" ::: "v255", "a255");
        // probability planes: the columns past HM are never written again
        for (int i = threadIdx.x; i < 2 * TR * PHS / 8; i += 256) reinterpret_cast<float4 *>(ppl)[i] = zero4();
    }

    // ---- Operand fetch, organised for memory-level parallelism: everything is requested with clamped (always valid) addresses and
    // no lane-divergent guard around a load, in three batches that each have whole phases of work to hide behind --
    //   entry:            token rows (all slabs), the learned-query fragments, and by DMA the small shared vectors (LayerNorm
    //                     gamma / beta x3, output bias) into LDS;
    //   after barrier 1:  the folded-score fragments G (used two phases later);
    //   after barrier 2:  the P.VW fragments (used four phases later).
    // An earlier version fetched operands "one phase ahead" behind per-lane guards; the compiler parked a full s_waitcnt vmcnt(0)
    // at every guard and the kernel made ~15 dependent round trips to L2 / Infinity Cache (tools/rowblock_probe.hip: 18 k of its
    // 26 k cycles).  Rows / columns duplicated by the clamps are never consumed (logit columns >= NQ, score columns >= HM) or are
    // zeroed below.
    {
        const float *srcs[8] = {lnp_w ? lnp_w : ln1_w, lnp_w ? lnp_b : ln1_b, ln1_w, ln1_b, CROSS ? ln2_w : ln1_w, CROSS ? ln2_b : ln1_b,
                                CROSS ? bout : ln1_b, sa_bias ? sa_bias : ln1_b};
        const uint32_t prm_lds = idf_lds_addr(prm);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if ((i & 3) == wave) idf_dma16_s(idf_uniform_ptr(srcs[i]), (uint32_t)(lane << 4), prm_lds + (uint32_t)(i * 1024));
    }
    Row16 ra, rb;
    float4 q[4][3], gv[4][NCT], vw[NCT][4];      // fp32 fragments; H2: the same registers hold 16-byte plane fragments -- q[2 s + pl][j], gv[2 s + pl][ct], vw2[s][c][pl]
    float4 vw2[2][4][2];
    float wk_n = 0.f, g0v[NSLOT], gsc = 1.f, vsc = 1.f;                     // g0 of (head = wave, memory slots q, q+4, q+8 with q = lane & 3)
    const int ta = QAN ? t0 - 1 + rown : t0 + rown, tb = t0 + 15 + kq;
    const bool va = ta >= 0 && ta < T, vb = QAN && wave == 0 && kq < 2 && tb < T;
    Row16Raw<NP> raw_a, raw_b;
    // issue order is pinned (sched_barrier) because the wait below COUNTS: small vectors, then the token rows, then -- youngest -- the
    // twelve learned-query fragments, which are not needed before the logits and may keep flying across the first barrier
    if constexpr (QAN) wk_n = wk[min(li, NQ - 1)];
    if constexpr (CROSS) {
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) g0v[i] = g0b[ML::col(wave, min((lane & 3) + 4 * i, mlen - 1))];
    }
    if constexpr (H2) {
        gsc = h2_scale[2 * b];
        vsc = h2_scale[2 * b + 1];
    }
    if constexpr (QAN) {
        if (wave == 0 && kq < 2) raw_b.request(u_in + (rowbase + min(tb, T - 1)) * D, li, u_pstride);     // halo rows t0+15, t0+16: two lane groups of wave 0 (exec-masked loads)
    }
    raw_a.request(u_in + (rowbase + min(max(ta, 0), T - 1)) * D, li, u_pstride);
    Row16Raw<1> raw_r;                            // standard layers with the out-projection folded into the attention kernel: the residual row
    if constexpr (!QAN) {
        if (sa_resid) raw_r.request(sa_resid + (rowbase + min(max(ta, 0), T - 1)) * D, li, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (QAN) {
#pragma unroll
        for (int ss = 0; ss < 4; ++ss)
#pragma unroll
            for (int j = 0; j < 3; ++j) q[ss][j] = zero4();
        if (li < NQ) {                                   // compact fragment order: only the NQ valid query columns are stored and fetched
#pragma unroll
            for (int ss = 0; ss < 4; ++ss)               // (H2: ss = 2 s + plane of [wave][K step s][tap][plane][kq][NQ][8 halves] -- the same twelve 16-byte loads)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    q[ss][j] = H2 ? ld4(Qc + (((((wave * 2 + (ss >> 1)) * 3 + j) * 2 + (ss & 1)) * 4 + kq) * NQ + li) * 4)
                                  : ld4(Qc + (((wave * 4 + ss) * 3 + j) * (4 * NQ) + kq * NQ + li) * 4);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    auto fetch_g = [&]() {
        if constexpr (CROSS) {
#pragma unroll
            for (int ss = 0; ss < 4; ++ss)               // (H2: ss = 2 s + plane of [wave][K step s][column tile][plane][lane][8 halves])
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
                    gv[ss][ct] = H2 ? ld4(Gb + (((((wave * 2 + (ss >> 1)) * NCT + ct) * 2 + (ss & 1)) * 64) + lane) * 4)
                                    : ld4(Gb + (((wave * 4 + ss) * NCT + ct) * 64 + lane) * 4);
        }
    };
    auto fetch_vw = [&]() {
        if constexpr (H2) {                              // [wave = output column quarter][K step s][column tile c][plane][lane][8 halves]
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int pl2 = 0; pl2 < 2; ++pl2) vw2[s2][c][pl2] = ld4(VWTb + (((((wave * 2 + s2) * 4 + c) * 2 + pl2) * 64) + lane) * 4);
        } else if constexpr (CROSS) {
#pragma unroll
            for (int sidx = 0; sidx < NCT; ++sidx)
#pragma unroll
                for (int c = 0; c < 4; ++c) vw[sidx][c] = ld4(VWTb + (((wave * NCT + sidx) * 4 + c) * 64 + lane) * 4);
        }
    };
    // the DMA'd vectors must have landed for every wave before anyone reads them: each wave drains its own queue (its rows come
    // with it -- they are needed now anyway), then one barrier
    // (H2 owns the whole register file, so the G / VW fragments could be requested here as well: measured, slower -- the CU's address path takes ~16 cycles per
    //  sixteen-byte wave load whoever waits for it, a wave cannot start waiting for its rows before it has issued everything, and the first phase grew from
    //  7.5 k to 9.0 k cycles, the kernel from 15.6 k to 16.6 k: the three batches stay.  This kernel moves ~330 KB per workgroup through that path: 5.2 k of its cycles.)
    IDF_RB_STAMP(9);                                     // every request of the first batch issued
    if constexpr (QAN) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");      // everything older than the 12 Qc fragment loads has landed
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    IDF_RB_STAMP(10);                                    // this wave's share has landed
    __syncthreads();
    raw_a.reduce(ra);
    if constexpr (QAN) {
        if (wave == 0) raw_b.reduce(rb);
    }
    const float *P_lnp_w = prm, *P_lnp_b = prm + 256, *P_ln1_w = prm + 512, *P_ln1_b = prm + 768, *P_ln2_w = prm + 1024, *P_ln2_b = prm + 1280,
                *P_bout = prm + 1536, *P_sab = prm + 1792;

    if constexpr (QAN) {
        // rows t0-1 .. t0+14 by all groups, halo rows t0+15, t0+16 by the first two groups of wave 0
        if (lnp_w) {
            ln_row16_lds(ra, P_lnp_w, P_lnp_b, li);
            if (wave == 0) ln_row16_lds(rb, P_lnp_w, P_lnp_b, li);
        }
        if (!va) row16_zero(ra);
        if (!vb) row16_zero(rb);
        row16_store(ra, xs + rown * RS, li);
        if (wave == 0 && kq < 2) row16_store(rb, xs + (16 + kq) * RS, li);
        if constexpr (H2) {
            row16_store_planes(ra, xh + rown * HS, xl + rown * HS, li);
            if (wave == 0 && kq < 2) row16_store_planes(rb, xh + (16 + kq) * HS, xl + (16 + kq) * HS, li);
        }
        fetch_g();
        __syncthreads();
        IDF_RB_STAMP(1);                                 // rows loaded (+ slab sum), LN_prev
        // logits: three 16x16 tiles (j = 0,1,2), each wave contracts a 64-wide slice of K
        f32x4 acc[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        if constexpr (H2) {
            f32x4 acc_c[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int koff = 64 * wave + 32 * s2 + 8 * kq;
                h8 ah[3], al[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    ah[j] = *reinterpret_cast<const h8 *>(xh + (li + j) * HS + koff);
                    al[j] = *reinterpret_cast<const h8 *>(xl + (li + j) * HS + koff);
                }
                mma_h2<3>(acc, acc_c, ah, al, q[2 * s2], q[2 * s2 + 1]);
            }
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[j][r] += acc_c[j][r] * idf_ffn_h2::LO_UNSCALE;
        } else {
#pragma unroll
            for (int ss = 0; ss < 4; ++ss) {
                const int koff = 16 * (wave * 4 + ss) + 4 * kq;
                float4 a[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) a[j] = ld4(xs + (li + j) * RS + koff);
                mma_rounds<3>(acc, a, q[ss]);
            }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[(wave * 3 + j) * 256 + (kq * 4 + r) * 16 + li] = acc[j][r];
        fetch_vw();
        __syncthreads();
        IDF_RB_STAMP(2);                                 // logits MFMA
        // tap softmax + coefficients: the 16-lane group of token `rown` holds query n = li; c_j = sum_n wk[n] softmax_j(logits) is a
        // row reduction on the DPP path, so the coefficients stay in the registers of the group that applies them below (an earlier
        // version went (t, n) -> LDS -> barrier -> 64 threads summing over n -> LDS -> barrier)
        float c0, c1, c2;
        {
            const int t = rown, n = min(li, NQ - 1);
            float l[3];
#pragma unroll
            for (int j = 0; j < 3; ++j)
                l[j] = (part[(0 * 3 + j) * 256 + t * 16 + n] + part[(1 * 3 + j) * 256 + t * 16 + n]) +
                       (part[(2 * 3 + j) * 256 + t * 16 + n] + part[(3 * 3 + j) * 256 + t * 16 + n]);
            const int tg = t0 + t;
            if (tg <= 0) l[0] = -FLT_MAX;
            if (tg + 1 >= T) l[2] = -FLT_MAX;
            const float mx = fmaxf(l[0], fmaxf(l[1], l[2]));
            const float e0 = __expf(l[0] - mx), e1 = __expf(l[1] - mx), e2 = __expf(l[2] - mx);
            const float w = li < NQ ? wk_n / (e0 + e1 + e2) : 0.f;
            c0 = row16_sum(w * e0);
            c1 = row16_sum(w * e1);
            c2 = row16_sum(w * e2);
        }
        IDF_RB_STAMP(3);                                 // tap softmax + coefficient sums
        {   // u1 = x_t + sum_j c_j x_{t+j-1} ;  x1 = LN1(u1)
            Row16 xm, xc, xp;
            row16_load(xm, xs + rown * RS, li);
            row16_load(xc, xs + (rown + 1) * RS, li);
            row16_load(xp, xs + (rown + 2) * RS, li);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                xc.c[i].x = xc.c[i].x + (c0 * xm.c[i].x + c1 * xc.c[i].x + c2 * xp.c[i].x);
                xc.c[i].y = xc.c[i].y + (c0 * xm.c[i].y + c1 * xc.c[i].y + c2 * xp.c[i].y);
                xc.c[i].z = xc.c[i].z + (c0 * xm.c[i].z + c1 * xc.c[i].z + c2 * xp.c[i].z);
                xc.c[i].w = xc.c[i].w + (c0 * xm.c[i].w + c1 * xc.c[i].w + c2 * xp.c[i].w);
            }
            ln_row16_lds(xc, P_ln1_w, P_ln1_b, li);
            if constexpr (H2) row16_store_planes(xc, xh + rown * HS, xl + rown * HS, li);      // (the logits' reads of these planes ended at the barrier above)
            if constexpr (CROSS) row16_store(xc, x1s + rown * RS, li);
            else if (t0 + rown < T) row16_store(xc, x2_out + (out_frame_major ? (size_t)(t0 + rown) * gridDim.y + b : rowbase + t0 + rown) * D, li);
        }
    } else {
        const int t = t0 + rown;
        if (sa_resid) {                               // u1 = (head partials) + xn + b_o   (workgroup-uniform branch)
            Row16 rr;
            raw_r.reduce(rr);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 bo = *reinterpret_cast<const float4 *>(P_sab + (i * 16 + li) * 4);
                ra.c[i].x += rr.c[i].x + bo.x;
                ra.c[i].y += rr.c[i].y + bo.y;
                ra.c[i].z += rr.c[i].z + bo.z;
                ra.c[i].w += rr.c[i].w + bo.w;
            }
        }
        if (!va) row16_zero(ra);
        fetch_g();
        fetch_vw();
        ln_row16_lds(ra, P_ln1_w, P_ln1_b, li);
        if constexpr (H2) row16_store_planes(ra, xh + rown * HS, xl + rown * HS, li);
        if constexpr (CROSS) row16_store(ra, x1s + rown * RS, li);
        else if (t < T) row16_store(ra, x2_out + (out_frame_major ? (size_t)t * gridDim.y + b : rowbase + t) * D, li);
    }
    if constexpr (!CROSS) return;                  // encoder layers: no memory to attend to, x1 is the FFN input
    __syncthreads();
    IDF_RB_STAMP(4);                                     // stencil + LN1
    {   // folded cross-attention scores: NCT 16x16 tiles over the (head, memory slot) columns (compact layout: three tiles over 40 columns)
        f32x4 acc[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (H2) {
            f32x4 acc_c[NCT];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) acc_c[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int koff = 64 * wave + 32 * s2 + 8 * kq;
                const h8 a_h = *reinterpret_cast<const h8 *>(xh + li * HS + koff), a_l = *reinterpret_cast<const h8 *>(xl + li * HS + koff);
                h8 ah[NCT], al[NCT];
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) { ah[ct] = a_h; al[ct] = a_l; }
                mma_h2<NCT>(acc, acc_c, ah, al, gv[2 * s2], gv[2 * s2 + 1]);
            }
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[ct][r] = (acc[ct][r] + acc_c[ct][r] * idf_ffn_h2::LO_UNSCALE) * gsc;      // G[b] was divided by gsc (a power of two)
        } else {
#pragma unroll
            for (int ss = 0; ss < 4; ++ss) {
                const float4 av = ld4(x1s + li * RS + 16 * (wave * 4 + ss) + 4 * kq);
                float4 a[NCT];
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) a[ct] = av;
                mma_rounds<NCT>(acc, a, gv[ss]);
            }
        }
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[(wave * NCT + ct) * 256 + (kq * 4 + r) * 16 + li] = acc[ct][r];
    }
    __syncthreads();
    IDF_RB_STAMP(5);                                     // folded scores MFMA
    {   // softmax over the MEM memory slots of each head: wave = head, token = lane >> 2, the four lanes of a quad take slots
        // q, q+4, q+8 and combine by two quad permutes (all 256 lanes busy; one wave doing all 40 columns of its 16 tokens took 2 k cycles)
        static_assert(H == 4 && MEM <= 12 && MEMX <= 16, "one wave per head, <= 3 (compact) / 4 (generic) slots per lane");
#define IDF_QUAD_XOR1(v) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false))
#define IDF_QUAD_XOR2(v) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false))
        const int tq = lane >> 2, qd = lane & 3;
        float sc[NSLOT], mx = -FLT_MAX;
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
            const int m = qd + 4 * i, idx = ML::col(wave, min(m, mlen - 1)), ct = idx >> 4, cl = idx & 15, o = tq * 16 + cl;
            const float v = ((part[(0 * NCT + ct) * 256 + o] + part[(1 * NCT + ct) * 256 + o]) +
                             (part[(2 * NCT + ct) * 256 + o] + part[(3 * NCT + ct) * 256 + o])) + g0v[i];
            sc[i] = m < mlen ? v : -FLT_MAX;
            mx = fmaxf(mx, sc[i]);
        }
        mx = fmaxf(mx, IDF_QUAD_XOR1(mx));
        mx = fmaxf(mx, IDF_QUAD_XOR2(mx));
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
            sc[i] = qd + 4 * i < mlen ? __expf(sc[i] - mx) : 0.f;
            sum += sc[i];
        }
        sum += IDF_QUAD_XOR1(sum);
        sum += IDF_QUAD_XOR2(sum);
        const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
        for (int i = 0; i < NSLOT; ++i)
            if (qd + 4 * i < (ML::GEN ? MEMX : MEM)) {        // (generic layout: a head's 16 columns are all written -- zero past mem_len; the planes of the split form were zeroed at entry)
                const float pv = qd + 4 * i < mlen ? sc[i] * inv : 0.f;
                if constexpr (H2) {
                    _Float16 hv, lv;
                    idf_ffn_h2::split1_nf(pv, hv, lv);
                    ph[tq * PHS + ML::col(wave, qd + 4 * i)] = hv;
                    pl[tq * PHS + ML::col(wave, qd + 4 * i)] = lv;
                } else {
                    Ps[tq * PSX + ML::col(wave, qd + 4 * i)] = pv;
                }
            }
        if (!ML::GEN && !H2 && wave == 0 && qd < (HMP - HM + 3) / 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (HM + qd * 4 + i < HMP) Ps[tq * PSX + HM + qd * 4 + i] = 0.f;
        }
    }
    __syncthreads();
    IDF_RB_STAMP(6);                                     // head softmax
    {   // u2 = x1 + P.VW + b_out : wave w owns output columns [64w, 64w+64)
        f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        if constexpr (H2) {
            f32x4 acc_c[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const h8 p_h = *reinterpret_cast<const h8 *>(ph + li * PHS + 32 * s2 + 8 * kq), p_l = *reinterpret_cast<const h8 *>(pl + li * PHS + 32 * s2 + 8 * kq);
                const h8 ah[4] = {p_h, p_h, p_h, p_h}, al[4] = {p_l, p_l, p_l, p_l};
                const float4 bh[4] = {vw2[s2][0][0], vw2[s2][1][0], vw2[s2][2][0], vw2[s2][3][0]}, bl[4] = {vw2[s2][0][1], vw2[s2][1][1], vw2[s2][2][1], vw2[s2][3][1]};
                mma_h2<4>(acc, acc_c, ah, al, bh, bl);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[c][r] = (acc[c][r] + acc_c[c][r] * idf_ffn_h2::LO_UNSCALE) * vsc;       // VW[b] was divided by vsc
        } else {
#pragma unroll
            for (int s = 0; s < NCT; ++s) {
                const float4 pv = ld4(Ps + li * PSX + 16 * s + 4 * kq);
                float4 a[4] = {pv, pv, pv, pv};
                mma_rounds<4>(acc, a, vw[s]);
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int col = (wave * 4 + c) * 16 + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) x1s[(kq * 4 + r) * RS + col] += acc[c][r] + P_bout[col];
        }
    }
    __syncthreads();
    IDF_RB_STAMP(7);                                     // P.VW MFMA
    {
        const int t = t0 + rown;
        Row16 r;
        row16_load(r, x1s + rown * RS, li);
        ln_row16_lds(r, P_ln2_w, P_ln2_b, li);
        if (t < T) row16_store_wt(r, x2_out + (rowbase + t) * D, li);
    }
    IDF_RB_STAMP(8);                                     // LN2 + store
}

// ------------------------------------------------------------------------------------
// The split-f16 row block on EIGHT waves (round 5).  Same arithmetic per product, operand layouts, grid and outputs as rowblock_kernel<QAN, true, NP, true, MS>;
// what differs is how a workgroup's work is cut.  Round 4's kernel (four waves, one per SIMD) was a chain of short phases -- 15.6 k cycles per workgroup of which the
// matrix pipe ran 0.4 k -- on 112 of 256 CUs; its phases are latency chains over whatever one wave holds:
//   * row passes (slab sums, the three LayerNorms, the stencil, the f16 splits): one token row per 32 lanes (two 16-byte chunks per lane) instead of per 16 lanes
//     (four) -- every pass is half as deep; a row's reductions are a 16-lane DPP reduction + one ds_swizzle exchange with the other half;
//   * the logits and scores contractions split K over eight waves (one K = 32 step each: three / NCT dependent-free MFMA groups instead of two rounds), P.VW gives a wave
//     two output tiles instead of four; every wave fetches half the fragments (the fragment orders in memory are unchanged: [K quarter][K step] IS [K eighth]);
//   * two waves per SIMD: one's loads and LDS traffic run beside the other's VALU work.
// It still owns its CU (512 threads x 256 registers = the register file, LDS topped up to 160 KiB by the launcher).  Results differ from the four-wave kernel by
// summation order only (eight K partials instead of four; 32-lane LayerNorm sums); every route of a process takes the same kernel, so bit-identity between routes holds.
// Reference work being replaced: model/sublayers.py:311-352 (QaN block + cross-attention + the two LayerNorms around them).
// ------------------------------------------------------------------------------------
// Token rows of the eight-wave row block: LPR lanes per row (32 with 16 tokens per workgroup, 64 with 8), lane lr owns the 4-float chunks lr, LPR + lr, ...
// The two forms compute the SAME bits: every per-element operation is the same, and a row's two LayerNorm sums are taken over ONE fixed tree whatever LPR is --
// leaves = the 64 chunks' partial sums; T_r = the 16-lane rotation tree (row16_sum) over chunks 16 r .. 16 r + 15, r = 0..3; sum = (T0 + T1) + (T2 + T3) -- so the number
// of tokens a workgroup takes is a pure performance choice (like the row tile of ffn_h2.h): shards / chains of a batch that pick differently still agree bit for bit.
// With 64 lanes per row that tree is common.h's wave_sum (DPP + four v_readlane: no LDS-crossbar instruction on the row passes' critical path -- a first version
// exchanged halves with ds_bpermute / ds_swizzle and spent ~250 cycles per reduction, six reductions per workgroup in a row); with 32 lanes a lane's two chunks belong
// to T_r and T_(r+2), so both are reduced over the 16-lane row and the halves exchanged with one ds_swizzle each.
template <int LPR>
struct RowL {
    static constexpr int CPL = 64 / LPR;          // chunks per lane: 2 or 1
    float4 c[CPL];
};
#define IDF_SWZ_XOR16(v) __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F))      // lane i <- lane i ^ 16 (bit-mask mode: and 0x1f, or 0, xor 0x10)
// p0 / p1: this lane's partial sums of its chunk(s) (p1 only with two chunks per lane)
template <int LPR>
__device__ __forceinline__ float rowl_tree(float p0, float p1) {
    if constexpr (LPR == 64) {
        return wave_sum(p0);                      // row16_sum, then (T0 + T1) + (T2 + T3) through scalar registers
    } else {
        const float t0 = row16_sum(p0), t1 = row16_sum(p1);      // lanes 0..15: T0, T2; lanes 16..31: T1, T3
        const float s01 = t0 + IDF_SWZ_XOR16(t0), s23 = t1 + IDF_SWZ_XOR16(t1);
        return s01 + s23;
    }
}
template <int LPR, int NP>
struct RowLRaw {
    static constexpr int CPL = 64 / LPR;
    float4 t[CPL][NP];
    __device__ __forceinline__ void request(const float *__restrict__ row, int lr, size_t stride) {
#pragma unroll
        for (int i = 0; i < CPL; ++i)
#pragma unroll
            for (int sl = 0; sl < NP; ++sl) t[i][sl] = ld4(row + sl * stride + (i * LPR + lr) * 4);
    }
    __device__ __forceinline__ void reduce(RowL<LPR> &r) const {      // slabs summed in the order of common.h ld4_sum
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            float4 v = t[i][0];
#pragma unroll
            for (int sl = 1; sl < NP; ++sl) { v.x += t[i][sl].x; v.y += t[i][sl].y; v.z += t[i][sl].z; v.w += t[i][sl].w; }
            r.c[i] = v;
        }
    }
};
template <int LPR>
__device__ __forceinline__ void ln_rowl_lds(RowL<LPR> &r, const float *w, const float *b, int lr) {
    // no FMA contraction in the row passes of this kernel: WHICH product of `a * a + b * b` (or of a * b + c written over two statements) the backend fuses is its own choice per
    // instantiation, and the 8- and the 16-token form must round alike (every product and every sum rounded on its own, like the sampler update of gemm.h)
#pragma clang fp contract(off)
    constexpr int CPL = 64 / LPR;
    float p[2] = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < CPL; ++i) p[i] = (r.c[i].x + r.c[i].y) + (r.c[i].z + r.c[i].w);
    const float mean = rowl_tree<LPR>(p[0], p[1]) * (1.0f / 256.0f);
    float q[2] = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const float a0 = r.c[i].x - mean, a1 = r.c[i].y - mean, a2 = r.c[i].z - mean, a3 = r.c[i].w - mean;
        q[i] = (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
    }
    const float rstd = __builtin_amdgcn_rsqf(rowl_tree<LPR>(q[0], q[1]) * (1.0f / 256.0f) + 1e-5f);
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const float4 g = *reinterpret_cast<const float4 *>(w + (i * LPR + lr) * 4), be = *reinterpret_cast<const float4 *>(b + (i * LPR + lr) * 4);
        r.c[i].x = (r.c[i].x - mean) * rstd * g.x + be.x;
        r.c[i].y = (r.c[i].y - mean) * rstd * g.y + be.y;
        r.c[i].z = (r.c[i].z - mean) * rstd * g.z + be.z;
        r.c[i].w = (r.c[i].w - mean) * rstd * g.w + be.w;
    }
}
// The same LayerNorm on TWO independent rows at once (a token row and a halo row of the same lanes): same operations per row in the same order -- same bits as two
// calls --, written side by side so that the two rows' reduction chains overlap instead of following each other behind a branch (the waves that own a halo row were
// the last to reach the barrier: tools/rowblock_probe.hip).
template <int LPR>
__device__ __forceinline__ void ln2_rowl_lds(RowL<LPR> &ra, RowL<LPR> &rb, const float *w, const float *b, int lr) {
#pragma clang fp contract(off)
    constexpr int CPL = 64 / LPR;
    float pa[2] = {0.f, 0.f}, pb[2] = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        pa[i] = (ra.c[i].x + ra.c[i].y) + (ra.c[i].z + ra.c[i].w);
        pb[i] = (rb.c[i].x + rb.c[i].y) + (rb.c[i].z + rb.c[i].w);
    }
    const float ma = rowl_tree<LPR>(pa[0], pa[1]) * (1.0f / 256.0f), mb = rowl_tree<LPR>(pb[0], pb[1]) * (1.0f / 256.0f);
    float qa[2] = {0.f, 0.f}, qb[2] = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const float a0 = ra.c[i].x - ma, a1 = ra.c[i].y - ma, a2 = ra.c[i].z - ma, a3 = ra.c[i].w - ma;
        const float b0 = rb.c[i].x - mb, b1 = rb.c[i].y - mb, b2 = rb.c[i].z - mb, b3 = rb.c[i].w - mb;
        qa[i] = (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        qb[i] = (b0 * b0 + b1 * b1) + (b2 * b2 + b3 * b3);
    }
    const float rsa = __builtin_amdgcn_rsqf(rowl_tree<LPR>(qa[0], qa[1]) * (1.0f / 256.0f) + 1e-5f), rsb = __builtin_amdgcn_rsqf(rowl_tree<LPR>(qb[0], qb[1]) * (1.0f / 256.0f) + 1e-5f);
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const float4 g = *reinterpret_cast<const float4 *>(w + (i * LPR + lr) * 4), be = *reinterpret_cast<const float4 *>(b + (i * LPR + lr) * 4);
        ra.c[i].x = (ra.c[i].x - ma) * rsa * g.x + be.x;
        ra.c[i].y = (ra.c[i].y - ma) * rsa * g.y + be.y;
        ra.c[i].z = (ra.c[i].z - ma) * rsa * g.z + be.z;
        ra.c[i].w = (ra.c[i].w - ma) * rsa * g.w + be.w;
        rb.c[i].x = (rb.c[i].x - mb) * rsb * g.x + be.x;
        rb.c[i].y = (rb.c[i].y - mb) * rsb * g.y + be.y;
        rb.c[i].z = (rb.c[i].z - mb) * rsb * g.z + be.z;
        rb.c[i].w = (rb.c[i].w - mb) * rsb * g.w + be.w;
    }
}
template <int LPR>
__device__ __forceinline__ void rowl_load(RowL<LPR> &r, const float *row, int lr) {
#pragma unroll
    for (int i = 0; i < 64 / LPR; ++i) r.c[i] = *reinterpret_cast<const float4 *>(row + (i * LPR + lr) * 4);
}
template <int LPR>
__device__ __forceinline__ void rowl_store(const RowL<LPR> &r, float *row, int lr) {
#pragma unroll
    for (int i = 0; i < 64 / LPR; ++i) *reinterpret_cast<float4 *>(row + (i * LPR + lr) * 4) = r.c[i];
}
template <int LPR>
__device__ __forceinline__ void rowl_store_wt(const RowL<LPR> &r, float *row, int lr) {
#pragma unroll
    for (int i = 0; i < 64 / LPR; ++i) idf_store16_wt(row + (i * LPR + lr) * 4, r.c[i]);
}
template <int LPR>
__device__ __forceinline__ void rowl_zero(RowL<LPR> &r) {
#pragma unroll
    for (int i = 0; i < 64 / LPR; ++i) r.c[i] = zero4();
}
template <int LPR>
__device__ __forceinline__ void rowl_store_planes(const RowL<LPR> &r, _Float16 *hi_row, _Float16 *lo_row, int lr) {
#pragma unroll
    for (int i = 0; i < 64 / LPR; ++i) {
        uint2 h, l;
        idf_ffn_h2::split4_pk(r.c[i], h, l);
        *reinterpret_cast<uint2 *>(hi_row + (i * LPR + lr) * 4) = h;
        *reinterpret_cast<uint2 *>(lo_row + (i * LPR + lr) * 4) = l;
    }
}

// TV = tokens per workgroup: 16 (grid ceil(T/16) x B) or 8 (grid ceil(T/8) x B).  The row passes are bound by the CU's VALU issue (tools/rowblock_probe.hip: eight waves
// instead of four changed them by 6 % -- the same instructions on the same four SIMDs), so what shortens them is FEWER ROWS PER CU: with eight tokens the launch is
// 208 workgroups at B = 16, T = 100 -- one round on 256 CUs instead of 112 -- each with half the row work; the MFMA tiles stay 16 rows tall (rows >= TV + 2 of the A
// operands are whatever LDS holds: they only reach output rows nobody stores).  The launcher takes 8 whenever the launch still fits the chip in one round.
template <bool QAN, int NP, int MS = MEM, int TV = 16>
__global__ __launch_bounds__(512) void rowblock8_kernel(const float *__restrict__ u_in, int T, int ntile, int nclip, int mem_len, size_t u_pstride,
                                                        const float *__restrict__ sa_resid, const float *__restrict__ Qc,
                                                        const float *__restrict__ lnp_w, const float *__restrict__ lnp_b,
                                                        const float *__restrict__ wk, const float *__restrict__ ln1_w,
                                                        const float *__restrict__ ln1_b, const float *__restrict__ G,
                                                        const float *__restrict__ g0, const float *__restrict__ VWT,
                                                        const float *__restrict__ bout, const float *__restrict__ ln2_w,
                                                        const float *__restrict__ ln2_b, float *__restrict__ x2_out,
                                                        const float *__restrict__ sa_bias, const float *__restrict__ h2_scale) {
    // Argument order: what the token-row addresses need comes first -- the library is built with kernel-argument preloading (build.py), the leading 14 dwords are in
    // SGPRs when the wave starts, and the rows are requested before anything else is even read from the argument segment.  1-D grid of ntile * nclip workgroups.
#pragma clang fp contract(off)                    // (see ln_rowl_lds: the two token counts must compute the same bits)
    using idf_ffn_h2::h8;
    using ML = MemLay<MS>;
    static_assert(TV == 16 || TV == 8, "tokens per workgroup");
    constexpr int NCT = ML::NCT, NSLOT = ML::NSLOT, NW8 = 8, NPT = NCT > 3 ? NCT : 3;
    constexpr int LPR = 512 / TV;                 // lanes per token row in the row passes
    constexpr int PTR = 20, PTS = 16 * PTR;       // K-split partial tiles: row stride 20 floats.  The MFMA result layout stores row 4 kq + r, column li per lane: 4 rows = 80 floats = 16 banks
                                                  // further, so the four kq groups of a store cover 64 different banks (17 put them 4 banks apart: up to 4 lanes per bank, ~0.6 k conflict
                                                  // cycles per workgroup = most of the kernel's SQ_LDS_BANK_CONFLICT); the head softmax's column-wise reads (token = lane >> 2: 20 tq mod 64 are
                                                  // the sixteen multiples of 4) and the tap softmax's row reads stay conflict-free
    constexpr int XS = QAN ? (TR + 2) * RS : 0;
    constexpr int HS = D + (IDF_LDS_STRIDE_SET == 4 ? 8 : 16), PHS = 64 + (IDF_LDS_STRIDE_SET == 4 ? 8 : 16);     // plane strides: conflict-free fragment reads (rowblock_kernel)
    __shared__ __attribute__((aligned(16))) _Float16 xpl[2 * (TR + 2) * HS];       // [hi | lo'][TR+2][HS]: LN_prev rows for the logits, then x1 for the scores (TV + 2 rows are written)
    __shared__ __attribute__((aligned(16))) _Float16 ppl[2 * TR * PHS];            // [hi | lo'][TR][PHS]: probabilities, unwritten columns stay zero
    _Float16 *const xh = xpl, *const xl = xpl + (TR + 2) * HS, *const ph = ppl, *const pl = ppl + TR * PHS;
    __shared__ __attribute__((aligned(1024))) float prm[8 * 256];        // LN_prev / LN1 / LN2 gamma, beta + cross-attention output bias + self-attention output bias (DMA targets)
    __shared__ __attribute__((aligned(16))) float sm[XS + TR * RS + NW8 * NPT * PTS];
    float *xs = sm;                               // [TV+2 of TR+2][RS]  LN_prev rows t0-1 .. t0+TV (QAN)
    float *x1s = sm + XS;                         // [TR][RS]    x1, then u2 in place
    float *part = x1s + TR * RS;                  // [8 waves][3 taps | NCT score tiles][16][17] K-split partial tiles

    IDF_RB_STAMP(0);
    asm volatile("" ::: "v255");                  // EXCLUSIVE CU (ffn_h2.h): 2 waves per SIMD x 256 registers = the register file; the launcher tops the LDS up to 160 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rg = tid / LPR, lr = tid % LPR;     // row passes: token row rg of the TV, lane lr of its LPR; threads of rows 0 and 1 also take the two halo rows (QAN)
    // ---- first batch of requests (pinned order, the wait below counts).  Oldest: the token rows (five slabs each; behind them everything else of the prologue runs
    // while they are in flight); then the shared vectors by DMA, the scalars, and -- youngest -- the learned-query fragments
    const int nwg = ntile * nclip, xq = nwg >> 3, xr = nwg & 7, xcd = (int)blockIdx.x & 7;      // XCD-affine logical id (xcd_logical_id) from the preloaded counts
    const int lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + ((int)blockIdx.x >> 3);
    const int b = lid / ntile, t0 = (lid - b * ntile) * TV;
    const size_t rowbase = (size_t)b * T;
    const int ta = QAN ? t0 - 1 + rg : t0 + rg, tb = t0 + TV - 1 + rg;
    const bool halo = QAN && rg < 2;
    const bool va = ta >= 0 && ta < T, vb = halo && tb < T;
    RowLRaw<LPR, NP> raw_a, raw_b;
    RowLRaw<LPR, 1> raw_r;
    if constexpr (QAN) {
        if (halo) raw_b.request(u_in + (rowbase + min(tb, T - 1)) * D, lr, u_pstride);     // halo rows t0 + TV - 1, t0 + TV: the threads of rows 0 and 1 (wave-uniform: LPR >= 32)
    }
    raw_a.request(u_in + (rowbase + min(max(ta, 0), T - 1)) * D, lr, u_pstride);
    if constexpr (!QAN) {
        if (sa_resid) raw_r.request(sa_resid + (rowbase + min(max(ta, 0), T - 1)) * D, lr, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    IDF_RB_STAMP(9);                                     // token rows requested (from the preloaded arguments alone)
    idf_args_now(lnp_b, wk, ln1_w, ln1_b, G, g0, VWT, bout, ln2_w, ln2_b, x2_out, sa_bias, h2_scale);
#ifdef IDF_RB_STAMP_ARGS
    IDF_RB_STAMP(14);                                    // (probe variant) the rest of the argument segment has arrived
#endif
    const int li = lane & 15, kq = lane >> 4;
    const int mlen = ML::GEN ? mem_len : MEM;
    const float *Gb = G + (size_t)b * ML::G_H2, *g0b = g0 + b * ML::G0N, *VWTb = VWT + (size_t)b * VW_H2;
    {   // wave w brings shared vector w (a scalar choice: one DMA instruction per wave)
        const int ws = __builtin_amdgcn_readfirstlane(wave);
        const float *src = ws == 0 ? (lnp_w ? lnp_w : ln1_w) : ws == 1 ? (lnp_w ? lnp_b : ln1_b) : ws == 2 ? ln1_w : ws == 3 ? ln1_b : ws == 4 ? ln2_w : ws == 5 ? ln2_b
                         : ws == 6 ? bout : (sa_bias ? sa_bias : ln1_b);
        idf_dma16_s(idf_uniform_ptr(src), (uint32_t)(lane << 4), idf_lds_addr(prm) + (uint32_t)(ws * 1024));
    }
    float4 q[2][3], gv[2][NCT], vw2[2][2][2];     // plane fragments: q[plane][tap], gv[plane][column tile] of this wave's K step; vw2[K step][tile][plane] of its two output tiles
    float wk_n = 0.f, g0v[NSLOT];
    const float gsc = h2_scale[2 * b], vsc = h2_scale[2 * b + 1];
    const int hh = wave & 3;                      // head softmax: waves 0..3, wave = head
    RowL<LPR> ra, rb;
    if constexpr (QAN) wk_n = wk[min(li, NQ - 1)];
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) g0v[i] = g0b[ML::col(hh, min((lane & 3) + 4 * i, mlen - 1))];
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (QAN) {
#pragma unroll
        for (int p2 = 0; p2 < 2; ++p2)
#pragma unroll
            for (int j = 0; j < 3; ++j) q[p2][j] = zero4();
        if (li < NQ) {                            // [K eighth = wave][tap][plane][kq][NQ][8 halves]
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2)
#pragma unroll
                for (int j = 0; j < 3; ++j) q[p2][j] = ld4(Qc + ((((wave * 3 + j) * 2 + p2) * 4 + kq) * NQ + li) * 4);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    for (int i = tid; i < 2 * TR * PHS / 8; i += 512) reinterpret_cast<float4 *>(ppl)[i] = zero4();
    IDF_RB_STAMP(15);                                    // the rest of the argument segment read, every request of the first batch issued
    // (A CU's vector-memory path accepts ~64 B per clock for all its waves and a wave waits in its issue slot until its request is taken: the ~100 KB of this batch keep
    // wave 0 here until ~4.4 k cycles.  Sending the learned-query fragments behind the barrier below instead moved that wait into the LayerNorm pass, same total:
    // tools/rowblock_probe.hip, profiles/r05_rowblock_probe.txt.  The kernel's ~200 KB of operands per workgroup are ~3.2 k cycles of such issue time.)
    if constexpr (QAN) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");       // everything older than the six learned-query fragment loads has landed
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    IDF_RB_STAMP(10);                                    // this wave's share has landed
    __syncthreads();
    IDF_RB_STAMP(11);                                    // every wave's share has landed
    auto fetch_g = [&]() {                        // [K eighth = wave][column tile][plane][lane][8 halves]
#pragma unroll
        for (int p2 = 0; p2 < 2; ++p2)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
#ifdef IDF_RB_EXPERIMENT_SKIP_LO      // (timing experiment only, wrong results: how much of the kernel is the operand stream?  tools/rowblock_probe.hip)
                if (p2 == 1) { gv[1][ct] = gv[0][ct]; continue; }
#endif
                gv[p2][ct] = ld4(Gb + ((((wave * NCT + ct) * 2 + p2) * 64) + lane) * 4);
            }
    };
    raw_a.reduce(ra);
    if constexpr (QAN) {
        if (halo) raw_b.reduce(rb);
        else rowl_zero<LPR>(rb);                  // (waves without a halo row run the two-row LayerNorm below on zeros: a few VALU slots, no second latency chain)
    }
    const float *P_lnp_w = prm, *P_lnp_b = prm + 256, *P_ln1_w = prm + 512, *P_ln1_b = prm + 768, *P_ln2_w = prm + 1024, *P_ln2_b = prm + 1280,
                *P_bout = prm + 1536, *P_sab = prm + 1792;
    auto fetch_vw = [&]() {                       // [output column quarter = wave >> 1][K step][column tile 2 (wave & 1) + c][plane][lane][8 halves]
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int p2 = 0; p2 < 2; ++p2) {
#ifdef IDF_RB_EXPERIMENT_SKIP_LO
                    if (p2 == 1) { vw2[s2][c][1] = vw2[s2][c][0]; continue; }
#endif
                    vw2[s2][c][p2] = ld4(VWTb + ((((((wave >> 1) * 2 + s2) * 4 + 2 * (wave & 1) + c) * 2 + p2) * 64) + lane) * 4);
                }
    };

    if constexpr (QAN) {
        if (lnp_w) {
            if (halo) ln2_rowl_lds<LPR>(ra, rb, P_lnp_w, P_lnp_b, lr);      // its own row and its halo row side by side (same bits as two single-row calls)
            else ln_rowl_lds<LPR>(ra, P_lnp_w, P_lnp_b, lr);
        }
        if (!va) rowl_zero<LPR>(ra);
        if (!vb) rowl_zero<LPR>(rb);
        rowl_store<LPR>(ra, xs + rg * RS, lr);
        rowl_store_planes<LPR>(ra, xh + rg * HS, xl + rg * HS, lr);
        if (halo) {
            rowl_store<LPR>(rb, xs + (TV + rg) * RS, lr);
            rowl_store_planes<LPR>(rb, xh + (TV + rg) * HS, xl + (TV + rg) * HS, lr);
        }
        fetch_g();
        IDF_RB_STAMP(12);                                // this wave is done with its LN_prev rows (before the barrier)
        __syncthreads();
        IDF_RB_STAMP(1);                                 // rows loaded (+ slab sum), LN_prev, planes
        {   // logits: three 16x16 tiles (taps), this wave contracts K = [32 wave, 32 wave + 32)
            f32x4 acc[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, acc_c[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            const int koff = 32 * wave + 8 * kq;
            h8 ah[3], al[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                ah[j] = *reinterpret_cast<const h8 *>(xh + (li + j) * HS + koff);
                al[j] = *reinterpret_cast<const h8 *>(xl + (li + j) * HS + koff);
            }
            mma_h2<3>(acc, acc_c, ah, al, q[0], q[1]);
            IDF_RB_STAMP(13);                                // (probe) logits: operands read, MFMAs issued
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) part[(wave * 3 + j) * PTS + (kq * 4 + r) * PTR + li] = acc[j][r] + acc_c[j][r] * idf_ffn_h2::LO_UNSCALE;
        }
        fetch_vw();
#ifndef IDF_RB_STAMP_ARGS
        IDF_RB_STAMP(14);                                // (probe) logits: partial tiles stored, VW requested
#endif
        __syncthreads();
        IDF_RB_STAMP(2);                                 // logits MFMA
        float c0, c1, c2;
        {   // tap softmax + coefficients: every 16-lane group of a row computes them (query n = lane & 15), so no exchange is needed before the stencil
            const int n = min(li, NQ - 1);
            float l[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float *pp = part + j * PTS + rg * PTR + n;
                constexpr int WS3 = 3 * PTS;
                l[j] = ((pp[0 * WS3] + pp[1 * WS3]) + (pp[2 * WS3] + pp[3 * WS3])) + ((pp[4 * WS3] + pp[5 * WS3]) + (pp[6 * WS3] + pp[7 * WS3]));
            }
            const int tg = t0 + rg;
            if (tg <= 0) l[0] = -FLT_MAX;
            if (tg + 1 >= T) l[2] = -FLT_MAX;
            const float mx = fmaxf(l[0], fmaxf(l[1], l[2]));
            const float e0 = __expf(l[0] - mx), e1 = __expf(l[1] - mx), e2 = __expf(l[2] - mx);
            const float w = li < NQ ? wk_n / (e0 + e1 + e2) : 0.f;
            c0 = row16_sum(w * e0);
            c1 = row16_sum(w * e1);
            c2 = row16_sum(w * e2);
        }
        IDF_RB_STAMP(3);                                 // tap softmax + coefficient sums
        {   // u1 = x_t + sum_j c_j x_{t+j-1} ;  x1 = LN1(u1)
            RowL<LPR> xm, xc, xp;
            rowl_load<LPR>(xm, xs + rg * RS, lr);
            rowl_load<LPR>(xc, xs + (rg + 1) * RS, lr);
            rowl_load<LPR>(xp, xs + (rg + 2) * RS, lr);
#pragma unroll
            for (int i = 0; i < 64 / LPR; ++i) {
                xc.c[i].x = xc.c[i].x + (c0 * xm.c[i].x + c1 * xc.c[i].x + c2 * xp.c[i].x);
                xc.c[i].y = xc.c[i].y + (c0 * xm.c[i].y + c1 * xc.c[i].y + c2 * xp.c[i].y);
                xc.c[i].z = xc.c[i].z + (c0 * xm.c[i].z + c1 * xc.c[i].z + c2 * xp.c[i].z);
                xc.c[i].w = xc.c[i].w + (c0 * xm.c[i].w + c1 * xc.c[i].w + c2 * xp.c[i].w);
            }
            ln_rowl_lds<LPR>(xc, P_ln1_w, P_ln1_b, lr);
            rowl_store_planes<LPR>(xc, xh + rg * HS, xl + rg * HS, lr);      // (the logits' reads of these planes ended at the barrier above)
            rowl_store<LPR>(xc, x1s + rg * RS, lr);
        }
    } else {
        if (sa_resid) {                               // u1 = (head partials) + xn + b_o   (workgroup-uniform branch)
            RowL<LPR> rr;
            raw_r.reduce(rr);
#pragma unroll
            for (int i = 0; i < 64 / LPR; ++i) {
                const float4 bo = *reinterpret_cast<const float4 *>(P_sab + (i * LPR + lr) * 4);
                ra.c[i].x += rr.c[i].x + bo.x;
                ra.c[i].y += rr.c[i].y + bo.y;
                ra.c[i].z += rr.c[i].z + bo.z;
                ra.c[i].w += rr.c[i].w + bo.w;
            }
        }
        if (!va) rowl_zero<LPR>(ra);
        fetch_g();
        fetch_vw();
        ln_rowl_lds<LPR>(ra, P_ln1_w, P_ln1_b, lr);
        rowl_store_planes<LPR>(ra, xh + rg * HS, xl + rg * HS, lr);
        rowl_store<LPR>(ra, x1s + rg * RS, lr);
    }
    __syncthreads();
    IDF_RB_STAMP(4);                                     // stencil + LN1
    {   // folded cross-attention scores: NCT 16x16 tiles, this wave's K step
        f32x4 acc[NCT], acc_c[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[ct] = acc_c[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int koff = 32 * wave + 8 * kq;
        const h8 a_h = *reinterpret_cast<const h8 *>(xh + li * HS + koff), a_l = *reinterpret_cast<const h8 *>(xl + li * HS + koff);
        h8 ah[NCT], al[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) { ah[ct] = a_h; al[ct] = a_l; }
        mma_h2<NCT>(acc, acc_c, ah, al, gv[0], gv[1]);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[(wave * NCT + ct) * PTS + (kq * 4 + r) * PTR + li] = (acc[ct][r] + acc_c[ct][r] * idf_ffn_h2::LO_UNSCALE) * gsc;      // G[b] was divided by gsc (a power of two)
    }
    __syncthreads();
    IDF_RB_STAMP(5);                                     // folded scores MFMA
    if (wave < 4) {   // softmax over the memory slots of each head: wave = head, token = lane >> 2, the four lanes of a quad take slots q, q+4, ... (rowblock_kernel)
        const int tq = lane >> 2, qd = lane & 3;
        float sc[NSLOT], mx = -FLT_MAX;
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
            const int m = qd + 4 * i, idx = ML::col(wave, min(m, mlen - 1)), ct = idx >> 4, cl = idx & 15;
            const float *pp = part + ct * PTS + tq * PTR + cl;
            constexpr int WS8 = NCT * PTS;
            const float v = (((pp[0 * WS8] + pp[1 * WS8]) + (pp[2 * WS8] + pp[3 * WS8])) + ((pp[4 * WS8] + pp[5 * WS8]) + (pp[6 * WS8] + pp[7 * WS8]))) + g0v[i];
            sc[i] = m < mlen ? v : -FLT_MAX;
            mx = fmaxf(mx, sc[i]);
        }
        mx = fmaxf(mx, IDF_QUAD_XOR1(mx));
        mx = fmaxf(mx, IDF_QUAD_XOR2(mx));
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
            sc[i] = qd + 4 * i < mlen ? __expf(sc[i] - mx) : 0.f;
            sum += sc[i];
        }
        sum += IDF_QUAD_XOR1(sum);
        sum += IDF_QUAD_XOR2(sum);
        const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
        for (int i = 0; i < NSLOT; ++i)
            if (qd + 4 * i < (ML::GEN ? MEMX : MEM)) {
                _Float16 hv, lv;
                idf_ffn_h2::split1_nf(qd + 4 * i < mlen ? sc[i] * inv : 0.f, hv, lv);
                ph[tq * PHS + ML::col(wave, qd + 4 * i)] = hv;
                pl[tq * PHS + ML::col(wave, qd + 4 * i)] = lv;
            }
    }
    __syncthreads();
    IDF_RB_STAMP(6);                                     // head softmax
    {   // u2 = x1 + P.VW + b_out : wave w owns output columns [32 w, 32 w + 32)
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, acc_c[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const h8 p_h = *reinterpret_cast<const h8 *>(ph + li * PHS + 32 * s2 + 8 * kq), p_l = *reinterpret_cast<const h8 *>(pl + li * PHS + 32 * s2 + 8 * kq);
            const h8 ah[2] = {p_h, p_h}, al[2] = {p_l, p_l};
            const float4 bh[2] = {vw2[s2][0][0], vw2[s2][1][0]}, bl[2] = {vw2[s2][0][1], vw2[s2][1][1]};
            mma_h2<2>(acc, acc_c, ah, al, bh, bl);
        }
        if (TV == 16 || kq * 4 < TV) {            // (8 tokens: rows 0..7 = the lanes of kq 0, 1.  ONE branch around all eight updates: a guard per element made eight serial LDS round trips)
            float xo[2][4], bo[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int col = (wave * 2 + c) * 16 + li;
                bo[c] = P_bout[col];
#pragma unroll
                for (int r = 0; r < 4; ++r) xo[c][r] = x1s[(kq * 4 + r) * RS + col];
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int col = (wave * 2 + c) * 16 + li;
#pragma unroll
                for (int r = 0; r < 4; ++r) x1s[(kq * 4 + r) * RS + col] = xo[c][r] + ((acc[c][r] + acc_c[c][r] * idf_ffn_h2::LO_UNSCALE) * vsc + bo[c]);       // VW[b] was divided by vsc
            }
        }
    }
    __syncthreads();
    IDF_RB_STAMP(7);                                     // P.VW MFMA
    {
        const int t = t0 + rg;
        RowL<LPR> r;
        rowl_load<LPR>(r, x1s + rg * RS, lr);
        ln_rowl_lds<LPR>(r, P_ln2_w, P_ln2_b, lr);
        if (t < T) rowl_store_wt<LPR>(r, x2_out + (rowbase + t) * D, lr);
    }
    IDF_RB_STAMP(8);                                     // LN2 + store
}

// ------------------------------------------------------------------------------------
// Temporal self-attention of the two standard layers: softmax(Q K^T / 8) V per (clip, head), on the MFMA.
// grid (ceil(T/QT), H, B), 256 threads.  LDS: K,V [TP][68], Q [QT][68], S [QT][TP+4]; TP = T rounded up to 16, QT = 16 or 32 query rows.
// ------------------------------------------------------------------------------------
constexpr int AS = HD + 4;            // row stride (floats) of the V image: its P.V operand reads are scalar (4 k rows x 16 columns per instruction: 68 keeps the four 16-lane groups on disjoint banks)
constexpr int SPAD = IDF_LDS_STRIDE_SET == 5 ? 8 : 4;         // padding of a score row (floats); set 6 = set 5 with round 4's score rows (their softmax sweeps prefer 4)
constexpr int ASK = HD + (IDF_LDS_STRIDE_SET == 4 ? 4 : 8);        // row stride of the Q and K images: their S = Q K^T operands are ds_read_b128 of lane (li, kq) -> row li, slot s0 + kq, and a stride of 2 slots mod 16
                                      // makes those conflict-free (see the row block's HS; round 4 used 68 for all three: every Q / K fragment read took twice its LDS cycles)
constexpr int ATTN_MAX_T = 208;
#ifndef IDF_ATTN_RT
#define IDF_ATTN_RT 1
#endif
constexpr int ATTN_RT = IDF_ATTN_RT;     // 16-query tiles per workgroup of the fused attention + out-projection kernel

// OUTPROJ: the workgroup also multiplies its [32 x 64] context tile with its head's 64 rows of W_o^T (K = 64, wave w owns output
// columns [64w, 64w+64)) and writes a [32 x 256] PARTIAL of the out-projection into slab `head` of `slabs`; the row block that follows
// sums the H slabs + residual + bias (one launch, one kernel boundary and the ctx round trip less than a separate GEMM).
// wo_frag: this layer's W_o in fragment order [head][wave][k-group of 16][column tile][lane][4] (mdm.py sa_out_fragments).
// RT: 16-query tiles per workgroup (grid.x = ceil(T / (16 RT))).  RT = 1 halves a workgroup's LDS image (73 KB at T = 100), so that two
// workgroups share a CU and one's loads / softmax run beside the other's MFMA phases.
template <bool OUTPROJ, int RT>
__global__ __launch_bounds__(256) void self_attn_kernel(const float *__restrict__ qkv, float *__restrict__ ctx, int T, int nwg,
                                                        const float *__restrict__ wo_frag, float *__restrict__ slabs, size_t pstride) {
    // 1-D grid of nwg = ceil(T / QT) * H * B workgroups; all twelve argument dwords arrive preloaded in SGPRs (build.py): no argument-segment read, no gridDim
    extern __shared__ __attribute__((aligned(16))) float smx[];
    IDF_AT_STAMP(0);
    const int TP = (T + 15) & ~15, SS = TP + SPAD;       // (TP + 8: the same 2-slots-mod-16 rule for the P.V A-operand reads of the score rows)
    constexpr int QT = 16 * RT;
    float *Ks = smx, *Vs = Ks + TP * ASK, *Qs = Vs + TP * AS, *Ss = Qs + QT * ASK;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = (int)blockIdx.x & 7;                   // XCD-affine logical id (xcd_logical_id)
    const int lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + ((int)blockIdx.x >> 3);
    const int nqt = (T + QT - 1) / QT, b = lid / (nqt * H), h = (lid / nqt) % H, q0 = (lid % nqt) * QT, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
    const size_t rowbase = (size_t)b * T;
    // Operand fetch, all requests of a batch in flight together: clamped (always valid) addresses, no guard around a load -- a guarded
    // load inside a run-time loop costs one full memory round trip per iteration (7 at T = 100).  TP * 16 is a multiple of 256, so a
    // sweep `it` covers key rows 16 it .. 16 it + 15 for the whole workgroup; 8 sweeps per batch (T <= 128: one batch).
    float4 qreg[RT];
#pragma unroll
    for (int u = 0; u < RT; ++u) {
        const int i = tid + 256 * u, r = i >> 4, d4 = (i & 15) * 4;
        qreg[u] = ld4(qkv + (rowbase + min(q0 + r, T - 1)) * (3 * D) + h * HD + d4);
    }
    const int nsweep = TP >> 4;
    for (int it0 = 0; it0 < nsweep; it0 += 8) {
        float4 kreg[8], vreg[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = tid + 256 * (it0 + u), j = i >> 4, d4 = (i & 15) * 4;
            const float *src = qkv + (rowbase + min(j, T - 1)) * (3 * D) + h * HD + d4;
            kreg[u] = ld4(src + D);
            vreg[u] = ld4(src + 2 * D);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = tid + 256 * (it0 + u), j = i >> 4, d4 = (i & 15) * 4;
            if (it0 + u < nsweep) {                                    // workgroup-uniform
                *reinterpret_cast<float4 *>(Ks + j * ASK + d4) = j < T ? kreg[u] : zero4();
                *reinterpret_cast<float4 *>(Vs + j * AS + d4) = j < T ? vreg[u] : zero4();
            }
        }
    }
    float4 wo[4][4];                                     // out-projection fragments: requested now (behind K / V, pinned), consumed after P V
    if constexpr (OUTPROJ) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx)
#pragma unroll
            for (int c = 0; c < 4; ++c) wo[sidx][c] = ld4(wo_frag + ((((size_t)(h * 4 + wave) * 4 + sidx) * 4 + c) * 64 + lane) * 4);
    }
#pragma unroll
    for (int u = 0; u < RT; ++u) {
        const int i = tid + 256 * u, r = i >> 4, d4 = (i & 15) * 4;
        *reinterpret_cast<float4 *>(Qs + r * ASK + d4) = q0 + r < T ? qreg[u] : zero4();
    }
    __syncthreads();
    IDF_AT_STAMP(1);                                     // K, V, Q in LDS
    // S = Q K^T / 8: wave w owns key tiles w, w+4, ... for both 16-query tiles
    for (int ct = wave; ct < TP / 16; ct += 4) {
        f32x4 acc[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < HD / 16; ++s) {
            const int koff = 16 * s + 4 * kq;
            const float4 kv = ld4(Ks + (ct * 16 + li) * ASK + koff);
            float4 a[RT], bb[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) { a[rt] = ld4(Qs + (rt * 16 + li) * ASK + koff); bb[rt] = kv; }
            mma_rounds<RT>(acc, a, bb);
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) Ss[(rt * 16 + kq * 4 + r) * SS + ct * 16 + li] = acc[rt][r] * 0.125f;
    }
    __syncthreads();
    IDF_AT_STAMP(2);                                     // S = Q K^T
    {   // row softmax: one 16-lane group per row, 16 rows per sweep; lane l16 owns columns l16, 16+l16, ... and keeps them in
        // registers between the max, the exp/sum and the normalisation (one LDS read and one write per element; three run-time loops
        // over LDS took 7.2 k of the kernel's 19 k cycles)
        constexpr int NC = ATTN_MAX_T / 16, NC0 = 8;     // columns per lane at the longest clip; the first NC0 cover T <= 128
        const int ncol = TP >> 4;
        const bool tail = ncol > NC0;                    // workgroup-uniform: ONE branch around the columns past 128
        for (int i = wave * 4 + kq; i < QT; i += 16) {
            float *row = Ss + i * SS;
            float v[NC], mx = -FLT_MAX;
            // reads are unconditional with clamped addresses (a guard per column makes the compiler wait for every read separately)
#pragma unroll
            for (int c = 0; c < NC0; ++c) {
                const int j = 16 * c + li;
                v[c] = row[min(j, TP - 1)];
                v[c] = j < T ? v[c] : -FLT_MAX;
            }
            if (tail) {
#pragma unroll
                for (int c = NC0; c < NC; ++c) {
                    const int j = 16 * c + li;
                    v[c] = row[min(j, TP - 1)];
                    v[c] = j < T ? v[c] : -FLT_MAX;
                }
            } else {
#pragma unroll
                for (int c = NC0; c < NC; ++c) v[c] = -FLT_MAX;
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) mx = fmaxf(mx, v[c]);
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
#pragma unroll
            for (int c = 0; c < NC0; ++c)
                if (16 * c + li < TP) row[16 * c + li] = v[c] * inv;
            if (tail) {
#pragma unroll
                for (int c = NC0; c < NC; ++c)
                    if (16 * c + li < TP) row[16 * c + li] = v[c] * inv;
            }
        }
    }
    __syncthreads();
    IDF_AT_STAMP(3);                                     // softmax
    {   // ctx = P V: wave w owns the 16 head-dim columns [16w, 16w+16) for both query tiles
        f32x4 acc[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int dcol = wave * 16 + li;
        for (int s = 0; s < TP / 16; ++s) {
            const int koff = 16 * s + 4 * kq;
            const float *vp = Vs + koff * AS + dcol;
            const float4 vv = make_float4(vp[0], vp[AS], vp[2 * AS], vp[3 * AS]);
            float4 a[RT], bb[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) { a[rt] = ld4(Ss + (rt * 16 + li) * SS + koff); bb[rt] = vv; }
            mma_rounds<RT>(acc, a, bb);
        }
        if constexpr (!OUTPROJ) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int t = q0 + rt * 16 + kq * 4 + r;
                    if (t < T) idf_store4_wt(ctx + (rowbase + t) * D + h * HD + dcol, acc[rt][r]);
                }
        } else {
            // context tile -> LDS in A-operand (row-major) form; the Q image is dead since the S phase
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) Qs[(rt * 16 + kq * 4 + r) * ASK + dcol] = acc[rt][r];
        }
    }
    IDF_AT_STAMP(4);                                     // P V (+ store)
    if constexpr (OUTPROJ) {
        __syncthreads();
        IDF_AT_STAMP(7);                                 // (probe) context tile in LDS, every wave there
        f32x4 o[4 * RT];
#pragma unroll
        for (int i = 0; i < 4 * RT; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) {
            const int koff = 16 * sidx + 4 * kq;
            float4 a[4 * RT], bb[4 * RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const float4 av = ld4(Qs + (rt * 16 + li) * ASK + koff);
#pragma unroll
                for (int c = 0; c < 4; ++c) { a[rt * 4 + c] = av; bb[rt * 4 + c] = wo[sidx][c]; }
            }
            mma_rounds<4 * RT>(o, a, bb);
        }
        IDF_AT_STAMP(6);                                 // (probe) out-projection MFMAs issued
        float *slab = slabs + (size_t)h * pstride;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = q0 + rt * 16 + kq * 4 + r;
                if (t < T) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) idf_store4_wt(slab + (rowbase + t) * D + (wave * 4 + c) * 16 + li, o[rt * 4 + c][r]);
                }
            }
        IDF_AT_STAMP(5);                                 // out-projection partial + store
    }
}

// ------------------------------------------------------------------------------------
// Clips longer than ATTN_MAX_T frames: the same attention with K / V streamed through LDS in tiles of 64 keys and a running row maximum / sum
// ("flash" form), fp32 MFMA, the out-projection partial in its tail like self_attn_kernel<true, 1>.  The one-shot kernel parks K and V of a whole
// (clip, head) in LDS (149 KiB at T = 208); the reference's own bound is its positional table, PositionalEncoding(max_len = 5000) (model/layers.py:10).
// A correct path for every length the table allows, not a tuned one: per key tile four barriers and one pass over S; the shipped shapes never take it.
// grid (ceil(T/16), H, B), 256 threads; static LDS ~45 KiB.
// ------------------------------------------------------------------------------------
constexpr int KT = 64;                       // keys per tile
__global__ __launch_bounds__(256) void self_attn_tiled_kernel(const float *__restrict__ qkv, int T, const float *__restrict__ wo_frag,
                                                              float *__restrict__ slabs, size_t pstride) {
    __shared__ __attribute__((aligned(16))) float Ks[KT * ASK], Vs[KT * AS], Qs[16 * ASK], Ss[16 * (KT + 8)];
    __shared__ float m_run[16], l_run[16], alpha[16];
    constexpr int SS = KT + 8;
    idf_args_now(qkv, T, wo_frag, slabs, pstride, gridDim.x);
    const int lid = xcd_logical_id(), nqt = gridDim.x, b = lid / (nqt * H), h = (lid / nqt) % H, q0 = (lid % nqt) * 16, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
    const size_t rowbase = (size_t)b * T;
    float4 wo[4][4];
#pragma unroll
    for (int sidx = 0; sidx < 4; ++sidx)
#pragma unroll
        for (int c = 0; c < 4; ++c) wo[sidx][c] = ld4(wo_frag + ((((size_t)(h * 4 + wave) * 4 + sidx) * 4 + c) * 64 + lane) * 4);
    {   // the 16 query rows (zero past the clip), running statistics
        const int r = tid >> 4, d4 = (tid & 15) * 4;
        const float4 qv = ld4(qkv + (rowbase + min(q0 + r, T - 1)) * (3 * D) + h * HD + d4);
        *reinterpret_cast<float4 *>(Qs + r * ASK + d4) = q0 + r < T ? qv : zero4();
        if (tid < 16) { m_run[tid] = -FLT_MAX; l_run[tid] = 0.f; }
    }
    f32x4 o_acc = {0.f, 0.f, 0.f, 0.f};                  // O[16 queries][16 head-dim columns of this wave]: lane (li = column, kq) holds rows 4 kq .. 4 kq + 3
    const int dcol = wave * 16 + li;
    for (int k0 = 0; k0 < T; k0 += KT) {
        __syncthreads();                                 // the previous tile's P.V reads of Ks / Vs / Ss are done (first pass: Qs, statistics written)
#pragma unroll
        for (int u = 0; u < KT / 16; ++u) {
            const int i = tid + 256 * u, j = i >> 4, d4 = (i & 15) * 4;
            const float *src = qkv + (rowbase + min(k0 + j, T - 1)) * (3 * D) + h * HD + d4;
            const float4 kv = ld4(src + D), vv = ld4(src + 2 * D);
            *reinterpret_cast<float4 *>(Ks + j * ASK + d4) = k0 + j < T ? kv : zero4();
            *reinterpret_cast<float4 *>(Vs + j * AS + d4) = k0 + j < T ? vv : zero4();
        }
        __syncthreads();
        {   // S tile = Q K^T / 8: wave w owns key tile w of the four
            f32x4 acc[1] = {{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int sg = 0; sg < HD / 16; ++sg) {
                const int koff = 16 * sg + 4 * kq;
                const float4 a[1] = {ld4(Qs + li * ASK + koff)}, bb[1] = {ld4(Ks + (wave * 16 + li) * ASK + koff)};
                mma_rounds<1>(acc, a, bb);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) Ss[(kq * 4 + r) * SS + wave * 16 + li] = k0 + wave * 16 + li < T ? acc[0][r] * 0.125f : -FLT_MAX;
        }
        __syncthreads();
        {   // running softmax: the 16-lane group of row i holds columns li, 16 + li, 32 + li, 48 + li of the tile
            const int i = wave * 4 + kq;
            float *row = Ss + i * SS;
            float v[4], mx = -FLT_MAX;
#pragma unroll
            for (int c = 0; c < 4; ++c) { v[c] = row[16 * c + li]; mx = fmaxf(mx, v[c]); }
            mx = row16_max(mx);
            const float m_old = m_run[i], m_new = fmaxf(m_old, mx);
            float sum = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                v[c] = k0 + 16 * c + li < T ? __expf(v[c] - m_new) : 0.f;
                sum += v[c];
                row[16 * c + li] = v[c];
            }
            sum = row16_sum(sum);
            if (li == 0) {
                const float al = __expf(m_old - m_new);     // first tile: exp(-FLT_MAX - m) = 0
                alpha[i] = al;
                l_run[i] = l_run[i] * al + sum;
                m_run[i] = m_new;
            }
        }
        __syncthreads();
        {   // O = O alpha + P V
#pragma unroll
            for (int r = 0; r < 4; ++r) o_acc[r] *= alpha[kq * 4 + r];
            f32x4 acc[1] = {o_acc};
#pragma unroll
            for (int sg = 0; sg < KT / 16; ++sg) {
                const int koff = 16 * sg + 4 * kq;
                const float *vp = Vs + koff * AS + dcol;
                const float4 a[1] = {ld4(Ss + li * SS + koff)}, bb[1] = {make_float4(vp[0], vp[AS], vp[2 * AS], vp[3 * AS])};
                mma_rounds<1>(acc, a, bb);
            }
            o_acc = acc[0];
        }
    }
    __syncthreads();                                     // every wave is past its reads of Qs (S phase of the last tile)
#pragma unroll
    for (int r = 0; r < 4; ++r) Qs[(kq * 4 + r) * ASK + dcol] = o_acc[r] * __builtin_amdgcn_rcpf(l_run[kq * 4 + r]);       // context tile, A-operand (row-major) form
    __syncthreads();
    f32x4 o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int sidx = 0; sidx < 4; ++sidx) {
        const float4 av = ld4(Qs + li * ASK + 16 * sidx + 4 * kq);
        const float4 a[4] = {av, av, av, av};
        mma_rounds<4>(o, a, wo[sidx]);
    }
    float *slab = slabs + (size_t)h * pstride;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int t = q0 + kq * 4 + r;
        if (t < T) {
#pragma unroll
            for (int c = 0; c < 4; ++c) idf_store4_wt(slab + (rowbase + t) * D + (wave * 4 + c) * 16 + li, o[c][r]);
        }
    }
}

// ------------------------------------------------------------------------------------
// Per-sample memory folding (interdiff_mdm_prepare_memory)
// ------------------------------------------------------------------------------------
// kv[l][r][c] = cond[r][:] . Wkv_l[c][:] + bkv_l[c];  r = m*B + b (reference layout [mem_len,B,D]), c < 512
__global__ __launch_bounds__(256) void mem_kv_kernel(const float *__restrict__ arena, const idf_mdm_weights w,
                                                     const float *__restrict__ cond, int R, float *__restrict__ kv) {
    const int l = blockIdx.z, r = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    __shared__ float xr[D];
    xr[threadIdx.x] = cond[(size_t)r * D + threadIdx.x];
    __syncthreads();
    const float *Wr = arena + w.layer[l].ca_kv_w + (size_t)c * D;
    float s = 0.f;
    for (int k = 0; k < D; k += 4) {
        const float4 wv = *reinterpret_cast<const float4 *>(Wr + k);
        s += xr[k] * wv.x + xr[k + 1] * wv.y + xr[k + 2] * wv.z + xr[k + 3] * wv.w;
    }
    kv[((size_t)l * R + r) * 512 + c] = s + arena[w.layer[l].ca_kv_b + c];
}

// fragment order of the fp32 forms (see G_FRAG): the reader's lane is kq * 16 + li; NCT = column tiles of the layout (MemLay)
template <int NCT>
__device__ __forceinline__ int g_slot(int col, int k) {            // score column col (0 .. 16 NCT - 1), feature k (0..255)
    const int ks = k >> 4, kq = (k >> 2) & 3, e = k & 3, ct = col >> 4, li = col & 15;
    return ((ks * NCT + ct) * 64 + kq * 16 + li) * 4 + e;
}
template <int NCT>
__device__ __forceinline__ int vw_slot(int o, int col) {           // output feature o (0..255), probability column col (0 .. 16 NCT - 1)
    const int wave = o >> 6, c = (o >> 4) & 3, li = o & 15, sidx = col >> 4, kq = (col >> 2) & 3, e = col & 3;
    return (((wave * NCT + sidx) * 4 + c) * 64 + kq * 16 + li) * 4 + e;
}

// G[l][b][h*MEM+m][i] = 1/8 sum_d Wq[h*64+d][i] K[m,b][h*64+d];  g0 = 1/8 sum_d bq[h*64+d] K[..]
// VWT[l][b][o][h*MEM+m] = sum_d V[m,b][h*64+d] Wo[o][h*64+d]   (columns 40..47 zero)
// grid (H * mem_len, B, L), 256 threads (thread = i / o).  Generic layout (MS == MEMX): the buffers were zeroed by the caller, only the columns of present slots are written.
template <int MS>
__global__ __launch_bounds__(256) void mem_fold_kernel(const float *__restrict__ arena, const idf_mdm_weights w,
                                                       const float *__restrict__ kv, int B, float *__restrict__ G,
                                                       float *__restrict__ g0, float *__restrict__ VWT, int mem_len) {
    using ML = MemLay<MS>;
    const int mlen = ML::GEN ? mem_len : MEM;
    const int hm = blockIdx.x, b = blockIdx.y, l = blockIdx.z, h = hm / mlen, m = hm - h * mlen, i = threadIdx.x, col = ML::col(h, m);
    __shared__ float kd[HD], vd[HD];
    const float *row = kv + ((size_t)l * (mlen * B) + (size_t)m * B + b) * 512;
    if (i < HD) kd[i] = row[h * HD + i];
    else if (i < 2 * HD) vd[i - HD] = row[D + h * HD + (i - HD)];
    __syncthreads();
    const float *Wq = arena + w.layer[l].ca_q_w, *bq = arena + w.layer[l].ca_q_b, *Wo = arena + w.layer[l].ca_out_w;
    float sg = 0.f, sv = 0.f;
    for (int d = 0; d < HD; ++d) {
        sg += Wq[(size_t)(h * HD + d) * D + i] * kd[d];
        sv += vd[d] * Wo[(size_t)i * D + h * HD + d];
    }
    float *Gf = G + ((size_t)l * B + b) * ML::G_FRAG, *Vf = VWT + ((size_t)l * B + b) * ML::VWT_F;
    Gf[g_slot<ML::NCT>(col, i)] = sg * 0.125f;
    Vf[vw_slot<ML::NCT>(i, col)] = sv;
    if (!ML::GEN && hm < HMP - HM) {
        Gf[g_slot<ML::NCT>(HM + hm, i)] = 0.f;
        Vf[vw_slot<ML::NCT>(i, HM + hm)] = 0.f;
    }
    if (i == 0) {
        float s = 0.f;
        for (int d = 0; d < HD; ++d) s += bq[h * HD + d] * kd[d];
        g0[((size_t)l * B + b) * ML::G0N + col] = s * 0.125f;
    }
}

// The folded memory as split-f16 plane fragments (rowblock_kernel<.., H2>; layouts at G_H2).  Its values are data (encoder output folded with
// weights), so each (layer, clip) matrix is divided by a power of two that puts its largest magnitude in [2^13, 2^14) -- exact, the f16 pair then keeps
// 22 bits of every element down to 2^-27 of the largest -- and the row block multiplies the fp32 result back (sc[l][b] = {2^eG, 2^eV}).
// grid (B, L), 256 threads; once per sample.
template <int MS>
__global__ __launch_bounds__(256) void mem_fold_h2_kernel(const float *__restrict__ G, const float *__restrict__ VWT, int B,
                                                          float *__restrict__ Gh2, float *__restrict__ VWh2, float *__restrict__ sc) {
    using ML = MemLay<MS>;
    constexpr int NCT = ML::NCT, HMPX = ML::HMPX;
    const int b = blockIdx.x, l = blockIdx.y, tid = threadIdx.x;
    const float *Gf = G + ((size_t)l * B + b) * ML::G_FRAG, *Vf = VWT + ((size_t)l * B + b) * ML::VWT_F;
    __shared__ float red[2][256];
    float ag = 0.f, av = 0.f;
    for (int i = tid; i < ML::G_FRAG; i += 256) ag = fmaxf(ag, fabsf(Gf[i]));
    for (int i = tid; i < ML::VWT_F; i += 256) av = fmaxf(av, fabsf(Vf[i]));
    red[0][tid] = ag;
    red[1][tid] = av;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) {
            red[0][tid] = fmaxf(red[0][tid], red[0][tid + st]);
            red[1][tid] = fmaxf(red[1][tid], red[1][tid + st]);
        }
        __syncthreads();
    }
    // amax 2^-e in [2^13, 2^14); a matrix of zeros (or non-finite values, which then stay what they are) is left alone
    const int eg = (red[0][0] > 0.f && red[0][0] < INFINITY) ? ilogbf(red[0][0]) - 13 : 0, ev = (red[1][0] > 0.f && red[1][0] < INFINITY) ? ilogbf(red[1][0]) - 13 : 0;
    const float dg = ldexpf(1.0f, -eg), dv = ldexpf(1.0f, -ev);
    float *Go = Gh2 + ((size_t)l * B + b) * ML::G_H2, *Vo = VWh2 + ((size_t)l * B + b) * VW_H2;
    for (int it = tid; it < 4 * 2 * NCT * 64; it += 256) {
        const int lane = it & 63, ct = (it >> 6) % NCT, s2 = (it / (64 * NCT)) & 1, wv = it / (128 * NCT), kq = lane >> 4, li = lane & 15;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = Gf[g_slot<NCT>(16 * ct + li, 64 * wv + 32 * s2 + 8 * kq + e)] * dg;
        uint2 h0, l0, h1, l1;
        idf_ffn_h2::split4(make_float4(v[0], v[1], v[2], v[3]), h0, l0);
        idf_ffn_h2::split4(make_float4(v[4], v[5], v[6], v[7]), h1, l1);
        uint4 *dst = reinterpret_cast<uint4 *>(Go) + (((wv * 2 + s2) * NCT + ct) * 2) * 64 + lane;
        dst[0] = make_uint4(h0.x, h0.y, h1.x, h1.y);
        dst[64] = make_uint4(l0.x, l0.y, l1.x, l1.y);
    }
    for (int it = tid; it < 4 * 2 * 4 * 64; it += 256) {
        const int lane = it & 63, c = (it >> 6) & 3, s2 = (it >> 8) & 1, wv = it >> 9, kq = lane >> 4, li = lane & 15;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int col = 32 * s2 + 8 * kq + e;
            v[e] = col < HMPX ? Vf[vw_slot<NCT>(64 * wv + 16 * c + li, min(col, HMPX - 1))] * dv : 0.f;
        }
        uint2 h0, l0, h1, l1;
        idf_ffn_h2::split4(make_float4(v[0], v[1], v[2], v[3]), h0, l0);
        idf_ffn_h2::split4(make_float4(v[4], v[5], v[6], v[7]), h1, l1);
        uint4 *dst = reinterpret_cast<uint4 *>(Vo) + (((wv * 2 + s2) * 4 + c) * 2) * 64 + lane;
        dst[0] = make_uint4(h0.x, h0.y, h1.x, h1.y);
        dst[64] = make_uint4(l0.x, l0.y, l1.x, l1.y);
    }
    if (tid == 0) {
        sc[((size_t)l * B + b) * 2] = ldexpf(1.0f, eg);
        sc[((size_t)l * B + b) * 2 + 1] = ldexpf(1.0f, ev);
    }
}

// self_attn_kernel needs more than 64 KiB of dynamic LDS for long clips: per-device opt-in (common.h)
int attn_opt_in() {
    static std::atomic<uint64_t> lds_ok{0};
    static std::atomic<uint64_t> lds_ok_op{0};
    const int bytes = (int)(((size_t)ATTN_MAX_T * (ASK + AS) + 32 * ASK + 32 * (ATTN_MAX_T + SPAD)) * sizeof(float));
    const int rc = idf_opt_in_lds(reinterpret_cast<const void *>(self_attn_kernel<false, 2>), bytes, lds_ok);
    return rc != IDF_OK ? rc : idf_opt_in_lds(reinterpret_cast<const void *>(self_attn_kernel<true, ATTN_RT>), bytes, lds_ok_op);
}
inline size_t attn_lds_bytes(int T, int rt) {
    const int TP = (T + 15) & ~15;
    return ((size_t)TP * (ASK + AS) + 16 * rt * ASK + 16 * rt * (TP + SPAD)) * sizeof(float);
}

// self-attention + out-projection partials of one standard layer: the one-shot kernel up to ATTN_MAX_T frames, the K/V-tiled one beyond
inline void launch_self_attn_outproj(hipStream_t s, const float *qkv, int B, int T, const float *wo_frag, float *parts, size_t pstride) {
    if (T <= ATTN_MAX_T)
        hipLaunchKernelGGL((self_attn_kernel<true, ATTN_RT>), dim3((unsigned)(idf_cdiv(T, 16 * ATTN_RT) * H * B)), dim3(256), attn_lds_bytes(T, ATTN_RT), s, qkv, nullptr, T,
                           (int)(idf_cdiv(T, 16 * ATTN_RT) * H * B), wo_frag, parts, pstride);
    else
        hipLaunchKernelGGL(self_attn_tiled_kernel, dim3((unsigned)idf_cdiv(T, 16), H, B), dim3(256), 0, s, qkv, T, wo_frag, parts, pstride);
}

struct Ws {
    float *uA, *uB, *xn, *x2, *ctx, *qkv, *parts;
};
constexpr size_t WS_ROW_FLOATS = (size_t)(5 + 3 + NSL) * D;      // uA uB xn x2 ctx | qkv | FFN partial slabs
Ws carve(void *ws, int64_t N) {
    float *p = reinterpret_cast<float *>(ws);
    Ws r;
    r.uA = p; p += N * D;
    r.uB = p; p += N * D;
    r.xn = p; p += N * D;
    r.x2 = p; p += N * D;
    r.ctx = p; p += N * D;
    r.qkv = p; p += N * 3 * D;
    r.parts = p;
    return r;
}

// tile configurations of the token GEMMs (A_PLAIN / A_LN call sites: the LDS-DMA pipeline of gemm.h).  The default
// per call site was picked on MI355X with tools/gemm_probe.hip + tools/kbench.py (profiles/); interdiff_tune()
// overrides it for A/B runs.  Config ids: BM x BN, waves, k-slices per workgroup (ks), chunk depth (kc).
template <int APRO, int EPI, int NP = 1>
void run_gemm(int cfg, hipStream_t s, const Args &g) {
    switch (cfg) {
    case 1: launch_glds<32, 64, 2, 2, 1, 32, APRO, EPI, 3, NP>(s, g); break;
    case 2: launch_glds<32, 64, 2, 2, 1, 64, APRO, EPI, 3, NP>(s, g); break;
    case 3: launch_glds<32, 64, 2, 2, 2, 64, APRO, EPI, 3, NP>(s, g); break;
    case 4: launch_glds<64, 64, 2, 2, 1, 32, APRO, EPI, 3, NP>(s, g); break;
    case 5: launch_glds<32, 32, 2, 2, 1, 64, APRO, EPI, 3, NP>(s, g); break;
    case 6: launch_glds<32, 32, 2, 2, 2, 64, APRO, EPI, 3, NP>(s, g); break;
    case 7: launch_glds<64, 32, 2, 2, 1, 32, APRO, EPI, 3, NP>(s, g); break;
    case 8: launch_glds<64, 32, 2, 2, 2, 64, APRO, EPI, 3, NP>(s, g); break;
    case 9: launch<32, 64, 2, 2, 32, APRO, EPI, NP>(s, g); break;               // register-staged double buffer
    default: launch_glds<32, 64, 2, 2, 1, 32, APRO, EPI, 3, NP>(s, g); break;
    }
}
// A_LN call sites: the layer input is one matrix (layer 0) or the previous layer's NSL partial slabs
template <int EPI>
void run_gemm_ln(int cfg, hipStream_t s, const Args &g, int np) {
    if (np == NSL) run_gemm<A_LN, EPI, NSL>(cfg, s, g);
    else run_gemm<A_LN, EPI, 1>(cfg, s, g);
}
constexpr int CFG_FFN1 = 7, CFG_FFN2 = 6, CFG_HEADS = 6;          // (the QKV projection has its own kernel: run_qkv; the out-projection rides in the attention kernel, as a separate GEMM -- tune != 0 -- configuration 5 was the fastest)
inline int pick(int tuned, int dflt) { return tuned ? tuned : dflt; }
// the last GEMM with the sampler update in its epilogue (gemm.h E_HEADS_POST): LDS-DMA kernel configurations only
template <int NP>
void run_heads_post_np(int cfg, hipStream_t s, const Args &g) {
    if (g.T & 3) {                                   // clip lengths that are not a multiple of 4: the per-row form of the update (one configuration)
        launch_glds<32, 32, 2, 2, 2, 64, A_LN, E_HEADS_POST_RAGGED, 3, NP>(s, g);
        return;
    }
    switch (cfg) {
    case 1: launch_glds<32, 64, 2, 2, 1, 32, A_LN, E_HEADS_POST, 3, NP>(s, g); break;
    case 3: launch_glds<32, 64, 2, 2, 2, 64, A_LN, E_HEADS_POST, 3, NP>(s, g); break;
    case 5: launch_glds<32, 32, 2, 2, 1, 64, A_LN, E_HEADS_POST, 3, NP>(s, g); break;
    default: launch_glds<32, 32, 2, 2, 2, 64, A_LN, E_HEADS_POST, 3, NP>(s, g); break;          // CFG_HEADS
    }
}
void run_heads_post(int cfg, hipStream_t s, const Args &g, int np) {
    if (np == NSL) run_heads_post_np<NSL>(cfg, s, g);
    else run_heads_post_np<1>(cfg, s, g);
}
// QKV projection: the LayerNorm+linear kernel of ffn.h (tune 0) or, for A/B runs, one of the generic GEMM configurations
// step_state != null (layer 0 of interdiff_mdm_forward_step): one thread of the launch does the step's sampler bookkeeping (philox.h)
// pack_h2 != null: the split-f16 form (ffn_h2.h ln_linear_h2_kernel: tune[IDF_TUNE_FFN_MATH] == 1 and the layer's sa_in_pack_h2 is set)
// planes / scales != null (with pack_h2): the output leaves as the self-attention's f16 plane pairs + per-row scales instead of fp32 rows (ffn_h2.h ln_linear_h2_kernel<.., PLANES>);
// the caller has checked (qkv_planes_ok) that this kernel and the attention kernel that reads the planes both run here -- there is no fp32 fallback behind a planes launch
int run_qkv(int tuned, hipStream_t s, const Args &g, const float *pack, int np, int64_t *step_state = nullptr, int64_t *step_ts = nullptr,
            int step_B = 0, const float *pack_h2 = nullptr, float *planes = nullptr, float *scales = nullptr) {
    if (planes) {
        if (!pack_h2) return IDF_E_INVAL;
        const int rc = np == NSL ? idf_ffn_h2::launch_ln_linear_h2<NSL>(s, g.A, g.a_pstride, g.lnw, g.lnb, g.M, g.N, pack_h2, g.bias, g.C, g.ldc, g.xn_out, step_state, step_ts, step_B, planes, scales)
                                 : idf_ffn_h2::launch_ln_linear_h2<1>(s, g.A, g.a_pstride, g.lnw, g.lnb, g.M, g.N, pack_h2, g.bias, g.C, g.ldc, g.xn_out, step_state, step_ts, step_B, planes, scales);
        return idf_public_rc(rc);
    }
    if (tuned && !step_state) { run_gemm_ln<E_BIAS>(tuned, s, g, np); return IDF_OK; }
    if (pack_h2) {
        const int rc = np == NSL ? idf_ffn_h2::launch_ln_linear_h2<NSL>(s, g.A, g.a_pstride, g.lnw, g.lnb, g.M, g.N, pack_h2, g.bias, g.C, g.ldc, g.xn_out, step_state, step_ts, step_B)
                                 : idf_ffn_h2::launch_ln_linear_h2<1>(s, g.A, g.a_pstride, g.lnw, g.lnb, g.M, g.N, pack_h2, g.bias, g.C, g.ldc, g.xn_out, step_state, step_ts, step_B);
        if (rc != IDF_NOT_EXCLUSIVE) return rc;        // (not exclusive on this device: the fp32 kernel below)
    }
    if (np == NSL)
        idf_ffn::launch_ln_linear<NSL>(s, g.A, g.a_pstride, g.lnw, g.lnb, g.M, g.N, pack, g.bias, g.C, g.ldc, g.xn_out, step_state, step_ts, step_B);
    else
        idf_ffn::launch_ln_linear<1>(s, g.A, g.a_pstride, g.lnw, g.lnb, g.M, g.N, pack, g.bias, g.C, g.ldc, g.xn_out, step_state, step_ts, step_B);
    return IDF_OK;
}


// May the QKV projection of a standard layer hand the self-attention PLANES (np = slabs of its input, T = clip length)?  Both kernels of the pair must get their CUs here.
bool qkv_planes_ok(int np, int B, int T) {
    float dummy = 0.f;
    const int rq = np == NSL ? idf_ffn_h2::launch_ln_linear_h2<NSL>(nullptr, nullptr, 0, nullptr, nullptr, 0, 3 * D, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0, &dummy, &dummy)
                             : idf_ffn_h2::launch_ln_linear_h2<1>(nullptr, nullptr, 0, nullptr, nullptr, 0, 3 * D, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0, &dummy, &dummy);
    return rq == IDF_OK && idf_attn_h2::launch_self_attn_h2(nullptr, nullptr, B, T, nullptr, nullptr, 0, nullptr, true) == IDF_OK;
}

// ---- The seven kernels of a plain denoising step at the shipped shapes, instantiated HERE, next to each other (two inside this namespace, five right behind it).  Explicit
// instantiations are emitted where they stand, so the seven end up CONTIGUOUS in the code object (57 KB); implicit ones land wherever the compiler gets to them -- scattered over
// this file's 1.1 MB of code, where 16-39 % of their cache lines shared an instruction-cache set with more lines than it has ways (64 KB per CU pair; a step cycles through all
// seven kernels, 22 launches at a time: with LRU such a set misses on every pass).  Contiguous code maps onto the cache without a collision.
// (Which instantiations: mdm_forward_impl_t at B = 16, T = 100, memory length 10.)
template __global__ void rowblock8_kernel<false, H, MEM, 8>(const float *, int, int, int, int, size_t, const float *, const float *, const float *, const float *, const float *, const float *,
                                                            const float *, const float *, const float *, const float *, const float *, const float *, const float *, float *, const float *,
                                                            const float *);
template __global__ void rowblock8_kernel<true, NSL, MEM, 8>(const float *, int, int, int, int, size_t, const float *, const float *, const float *, const float *, const float *, const float *,
                                                             const float *, const float *, const float *, const float *, const float *, const float *, const float *, float *, const float *,
                                                             const float *);
}  // namespace
template __global__ void idf_attn_h2::self_attn_h2_kernel<0, true>(const float *, int, int, const float *, float *, size_t, const float *);
template __global__ void idf_ffn_h2::ffn_h2_kernel<2, 4, 0, 8>(const float *, int, int, const float *, const float *, const float *, float *, int);
template __global__ void idf_ffn_h2::ln_linear_h2_kernel<1, true, IDF_QKV_LOADER_WAVES>(const float *, size_t, int, int, const float *, const float *, const float *, int, int, const float *, float *, int, int, float *,
                                                                  int64_t *, int64_t *, float *, float *);
template __global__ void idf_ffn_h2::ln_linear_h2_kernel<IDF_FFN_SLICES, true, IDF_QKV_LOADER_WAVES>(const float *, size_t, int, int, const float *, const float *, const float *, int, int, const float *, float *, int,
                                                                               int, float *, int64_t *, int64_t *, float *, float *);
template __global__ void idf_tail_h2::step_tail_h2_kernel<3, false>(const float *, size_t, const float *, int, int, int, int, const idf_tail_h2::TailArgs);

extern "C" int interdiff_mdm_ffn(const idf_mdm_weights *w, int32_t layer, int32_t encoder, const float *x2, int32_t M, float *parts,
                                 void *stream) {
    if (!w || !x2 || !parts || M <= 0 || layer < 0 || layer >= L || (encoder && !w->has_encoder)) return IDF_E_INVAL;
    if ((reinterpret_cast<uintptr_t>(x2) & 15) || (reinterpret_cast<uintptr_t>(parts) & 15)) return IDF_E_INVAL;
    const idf_mdm_layer &ly = encoder ? w->enc_layer[layer] : w->layer[layer];
    const int rc = idf_launch_layer_ffn(idf_stream(stream), ly, w->arena, w->tune, x2, M, parts);
    if (rc != IDF_OK) return rc;
    IDF_CHECK_LAUNCH();
    return IDF_OK;
}

// C[M,N] = epi(A[M,K] . W[N,K]^T + bias): the token GEMM of the denoiser as a standalone op (fp32 MFMA, LDS-DMA
// pipeline).  epi: 0 bias, 1 bias + erf-GELU, 2 bias + residual (resid has leading dimension ldc).  cfg 0 = the tile
// configuration the FFN call sites ship with.  K % 64 == 0, N % 16 == 0, lda / ldc % 4 == 0, A / C 16-byte aligned.
extern "C" int interdiff_gemm_f32(const float *A, int32_t lda, const float *W, const float *bias, const float *resid, float *C,
                                  int32_t ldc, int32_t M, int32_t N, int32_t K, int32_t epi, int32_t cfg, void *stream) {
    if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0 || (K & 63) || (N & 15) || (epi == 2 && !resid) || epi < 0 || epi > 2) return IDF_E_INVAL;
    if ((ldc & 3) || (lda & 3) || (reinterpret_cast<uintptr_t>(C) & 15) || (reinterpret_cast<uintptr_t>(A) & 15)) return IDF_E_INVAL;   // 16-B row stores / loads
    Args g{};
    g.A = A; g.lda = lda; g.K = K; g.W = W; g.bias = bias; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.resid = resid; g.T = 1;
    hipStream_t s = idf_stream(stream);
    if (epi == 0) run_gemm<A_PLAIN, E_BIAS>(pick(cfg, CFG_FFN2), s, g);
    else if (epi == 1) run_gemm<A_PLAIN, E_GELU>(pick(cfg, CFG_FFN1), s, g);
    else run_gemm<A_PLAIN, E_RESID>(pick(cfg, CFG_FFN2), s, g);
    IDF_CHECK_LAUNCH();
    return IDF_OK;
}

// memctx: G | VWT | g0 (fp32 forms) | G planes | VWT planes | scales (split-f16 forms, always folded: the arithmetic is chosen per forward)
struct MemCtx {
    const float *G, *VWT, *g0, *Gh2, *VWh2, *sc;
};
template <int MS>
static inline MemCtx memctx_carve(const float *m, int B) {
    using ML = MemLay<MS>;
    MemCtx c;
    c.G = m;
    c.VWT = c.G + (size_t)L * B * ML::G_FRAG;
    c.g0 = c.VWT + (size_t)L * B * ML::VWT_F;
    c.Gh2 = c.g0 + (size_t)L * B * ML::G0N;
    c.VWh2 = c.Gh2 + (size_t)L * B * ML::G_H2;
    c.sc = c.VWh2 + (size_t)L * B * VW_H2;
    return c;
}
template <int MS>
static inline size_t memctx_floats_t(int32_t B) {
    using ML = MemLay<MS>;
    return (size_t)L * B * (ML::G_FRAG + ML::VWT_F + ML::G0N + ML::G_H2 + VW_H2 + 2);
}
// memory length of a handle: w->mem_len (0 = the default IDF_MDM_MEM); the compact layout serves exactly IDF_MDM_MEM, the generic one every 1 .. IDF_MDM_MEM_MAX
static inline int idf_mem_len(const idf_mdm_weights *w) { return w->mem_len ? w->mem_len : MEM; }
extern "C" size_t interdiff_mdm_memctx_floats(int32_t B) { return memctx_floats_t<MEM>(B); }
extern "C" size_t interdiff_mdm_memctx_floats_for(int32_t B, int32_t mem_len) {
    if (mem_len < 1 || mem_len > MEMX) return 0;
    return mem_len == MEM ? memctx_floats_t<MEM>(B) : memctx_floats_t<MEMX>(B);
}

extern "C" size_t interdiff_mdm_workspace_bytes(int32_t B, int32_t T) {
    const size_t N = (size_t)B * T;
    const size_t fwd = N * WS_ROW_FLOATS * sizeof(float);
    const size_t prep = (size_t)L * MEMX * B * 512 * sizeof(float);         // interdiff_mdm_prepare_memory's K/V projections at the longest memory
    return idf_align(fwd > prep ? fwd : prep);
}

namespace {
template <int MS>
int prepare_memory_t(const idf_mdm_weights *w, const float *cond, int32_t B, float *memctx, void *ws, size_t ws_bytes, hipStream_t s) {
    const int mlen = idf_mem_len(w);
    if (ws_bytes < (size_t)L * mlen * B * 512 * sizeof(float)) return IDF_E_NOMEM;
    float *kv = reinterpret_cast<float *>(ws);
    const MemCtx mc = memctx_carve<MS>(memctx, B);
    float *G = const_cast<float *>(mc.G), *VWT = const_cast<float *>(mc.VWT), *g0 = const_cast<float *>(mc.g0);
    idf_prof_mark(IDF_K_MEM_PREP, s);
    if (MemLay<MS>::GEN) {       // generic layout: only the columns of present slots are written below
        if (hipMemsetAsync(memctx, 0, memctx_floats_t<MS>(B) * sizeof(float), s) != hipSuccess) return IDF_E_LAUNCH;
    }
    hipLaunchKernelGGL(mem_kv_kernel, dim3(2, mlen * B, L), dim3(256), 0, s, w->arena, *w, cond, mlen * B, kv);
    hipLaunchKernelGGL(mem_fold_kernel<MS>, dim3(H * mlen, B, L), dim3(256), 0, s, w->arena, *w, kv, B, G, g0, VWT, mlen);
    hipLaunchKernelGGL(mem_fold_h2_kernel<MS>, dim3(B, L), dim3(256), 0, s, G, VWT, B, const_cast<float *>(mc.Gh2), const_cast<float *>(mc.VWh2), const_cast<float *>(mc.sc));
    idf_prof_mark(-1, s);
    IDF_CHECK_LAUNCH();
    return IDF_OK;
}
}  // namespace

// cond [mem_len,B,256] with mem_len = w->mem_len (0 = IDF_MDM_MEM = 10): the reference takes the memory length from the command line (eval_smpl_short.py:376
// --past_len; model/diffusion_smpl.py:195-223 encodes that many past frames).  Length 10 -- every BASELINE config -- takes the compact layout of the folded memory
// (40 score columns); any other length 1 .. IDF_MDM_MEM_MAX a generic one (64 columns, one 16-column tile per head, absent slots masked): same kernels, one more
// column tile in the row block.  memctx must hold interdiff_mdm_memctx_floats_for(B, mem_len) floats.
extern "C" int interdiff_mdm_prepare_memory(const idf_mdm_weights *w, const float *cond, int32_t B, float *memctx,
                                            void *ws, size_t ws_bytes, void *stream) {
    if (!w || !cond || !memctx || !ws || B <= 0 || w->mem_len < 0 || w->mem_len > MEMX) return IDF_E_INVAL;
    return idf_mem_len(w) == MEM ? prepare_memory_t<MEM>(w, cond, B, memctx, ws, ws_bytes, idf_stream(stream))
                                 : prepare_memory_t<MEMX>(w, cond, B, memctx, ws, ws_bytes, idf_stream(stream));
}

namespace {
__global__ void iota_kernel(int64_t *p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i;
}
}  // namespace

extern "C" size_t interdiff_mdm_encode_workspace_bytes(int32_t B, int32_t Tp) {
    const size_t N = (size_t)B * Tp;
    return idf_align(N * WS_ROW_FLOATS * sizeof(float)) + idf_align((size_t)B * sizeof(int64_t));
}

// Encoder side: u0 = [body | obj].W_in^T + b_in + pc[b] + pe[t] over the Tp past frames, then the 8 encoder layers
// (model/diffusion_smpl.py:217-221).  Same kernels as the decoder; the row block runs without the cross-attention
// phases, the last LayerNorm writes the frame-major [Tp,B,256] layout the decoder's memory folding expects.
extern "C" int interdiff_mdm_encode(const idf_mdm_weights *w, const float *pc, const float *x_past, int32_t B, int32_t Tp,
                                    float *cond, void *ws, size_t ws_bytes, void *stream) {
    if (!w || !pc || !x_past || !cond || !ws || B <= 0 || Tp <= 0) return IDF_E_INVAL;
    if (!w->has_encoder || Tp > w->max_T || w->C > 256 || w->C < 1) return IDF_E_INVAL;
    if (ws_bytes < interdiff_mdm_encode_workspace_bytes(B, Tp)) return IDF_E_NOMEM;
    hipStream_t s = idf_stream(stream);
    const float *ar = w->arena;
    const int N = B * Tp, C = w->C, T = Tp;
    Ws k = carve(ws, N);
    int64_t *iota = reinterpret_cast<int64_t *>(reinterpret_cast<char *>(ws) + idf_align((size_t)N * WS_ROW_FLOATS * sizeof(float)));
    hipLaunchKernelGGL(iota_kernel, dim3((unsigned)idf_cdiv(B, 256)), dim3(256), 0, s, iota, B);
    {
        Args g{};
        g.A = x_past; g.K = (C + 3) & ~3; g.Ka = C; g.W = ar + w->in_w; g.bias = ar + w->in_b; g.C = k.uA; g.ldc = D; g.M = N; g.N = D; g.T = T;
        g.ts = iota; g.temb = pc; g.pe = ar + w->pe; g.n_steps = B;            // "+ temb[ts[b]]" adds pc[b]
        if (C & 3) launch<32, 64, 2, 2, 32, A_TOKT_R, E_EMBED>(s, g);           // (W_in is packed with its rows zero-padded to a multiple of 4: mdm.py)
        else launch<32, 64, 2, 2, 32, A_TOKT, E_EMBED>(s, g);
    }
    const float *u_in = k.uA;                  // layer input: plain [N,256] for layer 0, then the FFN's partial slabs
    int u_np = 1;
    const size_t pstride = (size_t)N * D;
    const float *lnp_w = nullptr, *lnp_b = nullptr;
    if (attn_opt_in() != IDF_OK) return IDF_E_LAUNCH;
    const dim3 rb_grid((unsigned)idf_cdiv(T, TR), B);
    for (int l = 0; l < L; ++l) {
        const idf_mdm_layer &ly = w->enc_layer[l];
        if (ly.is_qan) {
            if (u_np == NSL)
                hipLaunchKernelGGL((rowblock_kernel<true, false, NSL>), rb_grid, dim3(256), 0, s, u_in, lnp_w, lnp_b, ar + ly.qc, ar + ly.wk,
                                   ar + ly.ln_w[0], ar + ly.ln_b[0], nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, k.x2, T, 0, pstride, nullptr, nullptr, nullptr, MEM);
            else
                hipLaunchKernelGGL((rowblock_kernel<true, false, 1>), rb_grid, dim3(256), 0, s, u_in, lnp_w, lnp_b, ar + ly.qc, ar + ly.wk,
                                   ar + ly.ln_w[0], ar + ly.ln_b[0], nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, k.x2, T, 0, pstride, nullptr, nullptr, nullptr, MEM);
        } else {
            Args g{};
            g.A = u_in; g.lda = D; g.K = D; g.lnw = lnp_w; g.lnb = lnp_b; g.W = ar + ly.sa_in_w; g.bias = ar + ly.sa_in_b;
            g.C = k.qkv; g.ldc = 3 * D; g.M = N; g.N = 3 * D; g.xn_out = k.xn; g.T = T; g.a_pstride = pstride;
            if (const int rc = run_qkv(0, s, g, ar + ly.sa_in_pack, u_np, nullptr, nullptr, 0,
                                       (w->tune[IDF_TUNE_FFN_MATH] != 0 && ly.sa_in_pack_h2) ? ar + ly.sa_in_pack_h2 : nullptr); rc != IDF_OK) return rc;
            launch_self_attn_outproj(s, k.qkv, B, T, ar + ly.sa_out_frag, k.parts, pstride);
            hipLaunchKernelGGL((rowblock_kernel<false, false, H>), rb_grid, dim3(256), 0, s, k.parts, nullptr, nullptr, nullptr, nullptr,
                               ar + ly.ln_w[0], ar + ly.ln_b[0], nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, k.x2, T, 0, pstride,
                               k.xn, ar + ly.sa_out_b, nullptr, MEM);
        }
        if (const int rc = idf_launch_layer_ffn(s, ly, ar, w->tune, k.x2, N, k.parts); rc != IDF_OK) return rc;
        u_in = k.parts;
        u_np = NSL;
        lnp_w = ar + ly.ln_w[1];
        lnp_b = ar + ly.ln_b[1];
    }
    // cond[t][b][:] = LN2_last(u[b*T + t])
    hipLaunchKernelGGL((rowblock_kernel<false, false, NSL>), rb_grid, dim3(256), 0, s, u_in, nullptr, nullptr, nullptr, nullptr, lnp_w, lnp_b,
                       nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, cond, T, 1, pstride, nullptr, nullptr, nullptr, MEM);       // after 8 layers u_in is always the slabs
    IDF_CHECK_LAUNCH();
    return IDF_OK;
}

namespace {
// dynamic LDS that tops a split-f16 row block's static LDS up to the CU's whole 160 KiB (exclusive CU, see the kernel), verified per (kernel, device):
// -1 = the kernel does not get its CU there (common.h idf_exclusive_cu) and the caller launches the fp32 row block
template <int MS>
int rb_h2_qan_dyn() {
    static idf_excl_cache excl;
    return idf_exclusive_cu(reinterpret_cast<const void *>(rowblock_kernel<true, true, NSL, true, MS>), MS == MEM ? "rowblock_kernel<QaN, split-f16>" : "rowblock_kernel<QaN, split-f16, any memory length>", 256, excl);
}
// the eight-wave form (rowblock8_kernel: the shipped split-f16 row block since round 5; tune[IDF_TUNE_MISC] == 8 keeps round 4's four-wave kernel for A/B)
template <int MS, int TV>
int rb8_qan_dyn() {
    static idf_excl_cache excl;
    static const char *const nm[4] = {"rowblock8_kernel<QaN, split-f16, 16 tokens>", "rowblock8_kernel<QaN, split-f16, 8 tokens>", "rowblock8_kernel<QaN, split-f16, 16 tokens, any memory length>",
                                      "rowblock8_kernel<QaN, split-f16, 8 tokens, any memory length>"};
    return idf_exclusive_cu(reinterpret_cast<const void *>(rowblock8_kernel<true, NSL, MS, TV>), nm[(MS == MEM ? 0 : 2) + (TV == 8 ? 1 : 0)], 512, excl);
}
template <int MS, int TV>
int rb8_std_dyn() {
    static idf_excl_cache excl;
    static const char *const nm[4] = {"rowblock8_kernel<std, split-f16, 16 tokens>", "rowblock8_kernel<std, split-f16, 8 tokens>", "rowblock8_kernel<std, split-f16, 16 tokens, any memory length>",
                                      "rowblock8_kernel<std, split-f16, 8 tokens, any memory length>"};
    return idf_exclusive_cu(reinterpret_cast<const void *>(rowblock8_kernel<false, H, MS, TV>), nm[(MS == MEM ? 0 : 2) + (TV == 8 ? 1 : 0)], 512, excl);
}
// tokens per workgroup of the eight-wave row block: w->rb_tokens (8 / 16), or 0 = eight whenever the launch still fits the chip in one round of workgroups (the two forms
// compute the same bits: rowblock8_kernel)
inline int rb8_tokens(const idf_mdm_weights *w, int B, int T) {
    if (w->rb_tokens == 8 || w->rb_tokens == 16) return w->rb_tokens;
    return (int64_t)B * idf_cdiv(T, 8) <= idf_cu_count() ? 8 : 16;
}
template <int MS>
int rb_h2_std_dyn() {
    static idf_excl_cache excl;
    return idf_exclusive_cu(reinterpret_cast<const void *>(rowblock_kernel<false, true, H, true, MS>), MS == MEM ? "rowblock_kernel<std, split-f16>" : "rowblock_kernel<std, split-f16, any memory length>", 256, excl);
}

// the sampler-step operands of interdiff_mdm_forward_step (null x: plain forward, x0 written out)
struct StepPost {
    float *x;
    const float *gt;
    const uint8_t *mask;
    const float *table;
    int64_t *state, *ts;
};

template <int MS>
int mdm_forward_impl_t(const idf_mdm_weights *w, const float *memctx, const float *x, const int64_t *ts, int32_t B, int32_t T, float *x0,
                       void *ws, size_t ws_bytes, void *stream, const StepPost &post, int32_t flags) {
    using ML = MemLay<MS>;
    constexpr int G_FRAG = ML::G_FRAG, G_H2 = ML::G_H2;
    const int mlen = idf_mem_len(w);
    if (!w || !memctx || !x || !ts || (!x0 && !post.x) || !ws || B <= 0 || T <= 0) return IDF_E_INVAL;
    if (T > w->max_T || w->C > 256 || w->C < 1 || mlen < 1 || mlen > MEMX || (MS == MEM) != (mlen == MEM)) return IDF_E_INVAL;
    if (T > ATTN_MAX_T && w->tune[IDF_TUNE_GEMM_OUTPROJ] != 0) return IDF_E_INVAL;          // (the A/B route with the out-projection as its own GEMM exists for T <= ATTN_MAX_T only)
    if (ws_bytes < interdiff_mdm_workspace_bytes(B, T)) return IDF_E_NOMEM;
    hipStream_t s = idf_stream(stream);
    const float *ar = w->arena;
    const int N = B * T, C = w->C;
    Ws k = carve(ws, N);
    const MemCtx mc = memctx_carve<MS>(memctx, B);
    const float *G = mc.G, *VWT = mc.VWT, *g0 = mc.g0;
    const int32_t *tune = w->tune;

    // The two ends of the step as split-f16 "step tail" launches (tail_h2.h): with the split arithmetic, the SMPL token width and the packer's plane
    // fragments.  flags (interdiff_mdm_forward_step_ex) then chain consecutive plain steps: IDF_STEP_EMBED_READY = the previous call's tail has already
    // written this step's embedding into the workspace, IDF_STEP_EMBED_NEXT = this call's tail writes the next step's.  Ignored otherwise.
    const bool tail_h2 = tune[IDF_TUNE_FFN_MATH] != 0 && w->out_w_h2 != 0 && w->in_w_h2 != 0 && C == idf_tail_h2::CW && w->tail_h2_ok != 0 && idf_tail_h2::tail_exclusive_ok((T & 3) != 0);
    idf_tail_h2::TailArgs ta{};
    if (tail_h2) {
        ta.win = ar + w->in_w_h2; ta.in_b = ar + w->in_b; ta.temb = ar + w->temb_table; ta.pe = ar + w->pe; ta.ts = ts; ta.n_steps = w->n_steps;
        ta.u0 = k.uA; ta.M = N; ta.T = T; ta.x_tok = x; ta.plain_ids = tune[IDF_TUNE_MISC] == 7 ? 1 : 0;
        if (!(post.x && (flags & IDF_STEP_EMBED_READY))) {
            idf_prof_mark(IDF_K_EMBED, s);
            if (const int rc = idf_tail_h2::launch_tail(s, 0, ta); rc != IDF_OK) return idf_public_rc(rc);      // (tail_exclusive_ok was asked first: IDF_NOT_EXCLUSIVE cannot come back, and never leaves the library if it does)
        }
    } else
    {   // u0 = [x_body | x_obj].W_in^T + b_in + temb[ts] + pe   (tokens gathered from x[b][c][t])
        Args g{};
        // token width C: any (BASELINE config #1, the HO-GCN skeleton tokens of model/diffusion_skeleton.py:236-253, has C = 106); W_in is packed
        // with rows of (C + 3) & ~3 floats, zero-padded (mdm.py), and a width that is no multiple of 4 takes the guarded gather (gemm.h A_TOKT_R)
        g.A = x; g.K = (C + 3) & ~3; g.Ka = C; g.W = ar + w->in_w; g.bias = ar + w->in_b; g.C = k.uA; g.ldc = D; g.M = N; g.N = D; g.T = T;
        g.ts = ts; g.temb = ar + w->temb_table; g.pe = ar + w->pe; g.n_steps = w->n_steps;
        idf_prof_mark(IDF_K_EMBED, s);
        if (C & 3) launch<32, 64, 2, 2, 32, A_TOKT_R, E_EMBED>(s, g);
        else
        // K = 144: KC = 144 is the whole contraction in ONE chunk -- every operand load of the workgroup in flight at once (one memory
        // round trip instead of one per 32-deep chunk of the double buffer)
        switch (tune[IDF_TUNE_GEMM_EMBED]) {
        case 1: launch<64, 64, 2, 2, 32, A_TOKT, E_EMBED>(s, g); break;
        case 2: launch<32, 64, 2, 2, 144, A_TOKT, E_EMBED>(s, g); break;
        case 3: launch<32, 32, 2, 2, 144, A_TOKT, E_EMBED>(s, g); break;
        case 4: launch<32, 64, 2, 2, 48, A_TOKT, E_EMBED>(s, g); break;
        default: launch<32, 64, 2, 2, 32, A_TOKT, E_EMBED>(s, g); break;
        }
    }
    const float *u_in = k.uA;                  // layer input (pre-norm sum of the previous layer): plain for layer 0, then FFN partial slabs
    int u_np = 1;
    const size_t pstride = (size_t)N * D;
    float *u_tmp = k.uB;
    const float *lnp_w = nullptr, *lnp_b = nullptr;   // LayerNorm still to be applied to u_in (none for layer 0)
    if (attn_opt_in() != IDF_OK) return IDF_E_LAUNCH;
    const dim3 rb_grid((unsigned)idf_cdiv(T, TR), B);
    const int tv8 = rb8_tokens(w, B, T);
    const dim3 rb8_grid((unsigned)idf_cdiv(T, 8), B);
    for (int l = 0; l < L; ++l) {
        const idf_mdm_layer &ly = w->layer[l];
        const float *Gl = G + (size_t)l * B * G_FRAG, *VWTl = VWT + (size_t)l * B * ML::VWT_F, *g0l = g0 + (size_t)l * B * ML::G0N;
        // split-f16 row block: tune[IDF_TUNE_FFN_MATH] == 1 (2 = split-f16 feed-forward and QKV only), for layers whose LayerNorm outputs the packer proved to stay in the f16 range
        const bool rb_h2 = tune[IDF_TUNE_FFN_MATH] == 1 && ly.rb_h2_ok != 0 && (!ly.is_qan || (ly.qc_h2 != 0 && u_np == NSL));
        const float *Gh = mc.Gh2 + (size_t)l * B * G_H2, *VWh = mc.VWh2 + (size_t)l * B * VW_H2, *scl = mc.sc + (size_t)l * B * 2;
        if (ly.is_qan) {
            idf_prof_mark(IDF_K_ROWBLOCK_QAN, s);
            const bool rb8 = tune[IDF_TUNE_MISC] != 8;
            const int dyn8 = rb_h2 && rb8 ? (tv8 == 8 ? rb8_qan_dyn<MS, 8>() : rb8_qan_dyn<MS, 16>()) : -1;
            const int dyn = rb_h2 && dyn8 < 0 ? rb_h2_qan_dyn<MS>() : -1;
            if (dyn8 >= 0 && tv8 == 8) {
                rowblock8_kernel<true, NSL, MS, 8><<<dim3(rb8_grid.x * B), dim3(512), (size_t)dyn8, s>>>(u_in, T, (int)rb8_grid.x, B, mlen, pstride, nullptr, ar + ly.qc_h2,
                                   lnp_w, lnp_b, ar + ly.wk, ar + ly.ln_w[0], ar + ly.ln_b[0], Gh, g0l, VWh, ar + ly.ca_out_b, ar + ly.ln_w[1],
                                   ar + ly.ln_b[1], k.x2, nullptr, scl);
            } else if (dyn8 >= 0) {
                rowblock8_kernel<true, NSL, MS, 16><<<dim3(rb_grid.x * B), dim3(512), (size_t)dyn8, s>>>(u_in, T, (int)rb_grid.x, B, mlen, pstride, nullptr, ar + ly.qc_h2,
                                   lnp_w, lnp_b, ar + ly.wk, ar + ly.ln_w[0], ar + ly.ln_b[0], Gh, g0l, VWh, ar + ly.ca_out_b, ar + ly.ln_w[1],
                                   ar + ly.ln_b[1], k.x2, nullptr, scl);
            } else if (dyn >= 0) {
                rowblock_kernel<true, true, NSL, true, MS><<<rb_grid, dim3(256), (size_t)dyn, s>>>(u_in, lnp_w, lnp_b, ar + ly.qc_h2, ar + ly.wk,
                                   ar + ly.ln_w[0], ar + ly.ln_b[0], Gh, g0l, VWh, ar + ly.ca_out_b, ar + ly.ln_w[1],
                                   ar + ly.ln_b[1], k.x2, T, 0, pstride, nullptr, nullptr, scl, mlen);
            } else if (u_np == NSL)
                hipLaunchKernelGGL((rowblock_kernel<true, true, NSL, false, MS>), rb_grid, dim3(256), 0, s, u_in, lnp_w, lnp_b, ar + ly.qc, ar + ly.wk,
                                   ar + ly.ln_w[0], ar + ly.ln_b[0], Gl, g0l, VWTl, ar + ly.ca_out_b, ar + ly.ln_w[1],
                                   ar + ly.ln_b[1], k.x2, T, 0, pstride, nullptr, nullptr, nullptr, mlen);
            else
                hipLaunchKernelGGL((rowblock_kernel<true, true, 1, false, MS>), rb_grid, dim3(256), 0, s, u_in, lnp_w, lnp_b, ar + ly.qc, ar + ly.wk,
                                   ar + ly.ln_w[0], ar + ly.ln_b[0], Gl, g0l, VWTl, ar + ly.ca_out_b, ar + ly.ln_w[1],
                                   ar + ly.ln_b[1], k.x2, T, 0, pstride, nullptr, nullptr, nullptr, mlen);
        } else {
            // xn = LN_prev(u_in) ; qkv = xn.Win^T + b
            Args g{};
            g.A = u_in; g.lda = D; g.K = D; g.lnw = lnp_w; g.lnb = lnp_b; g.W = ar + ly.sa_in_w; g.bias = ar + ly.sa_in_b;
            g.C = k.qkv; g.ldc = 3 * D; g.M = N; g.N = 3 * D; g.xn_out = k.xn; g.T = T; g.a_pstride = pstride;
            idf_prof_mark(IDF_K_GEMM_QKV, s);
            const float *qkv_h2 = (tune[IDF_TUNE_FFN_MATH] != 0 && ly.sa_in_pack_h2) ? ar + ly.sa_in_pack_h2 : nullptr;
            // Round 5: when the split-f16 self-attention follows, the projection writes the attention's f16 plane pairs (+ per-row scales, in the unused context buffer) instead of
            // fp32 rows -- the four query tiles of a (clip, head) then fetch planes instead of each splitting K and V again.  tune[IDF_TUNE_MISC] == 9 keeps fp32 rows (A/B).
            const bool attn_h2 = tune[IDF_TUNE_FFN_MATH] != 0 && ly.sa_out_frag_h2 != 0 && tune[IDF_TUNE_MISC] != 6 && T <= idf_attn_h2::MAX_T && tune[IDF_TUNE_GEMM_OUTPROJ] == 0;
            const bool planes = attn_h2 && qkv_h2 && ly.qkv_bounds_ok != 0 && tune[IDF_TUNE_MISC] != 9 && (tune[IDF_TUNE_GEMM_QKV] == 0 || (post.x && l == 0)) && qkv_planes_ok(u_np, B, T);
            int rcq;
            if (post.x && l == 0) rcq = run_qkv(0, s, g, ar + ly.sa_in_pack, u_np, post.state, post.ts, B, qkv_h2, planes ? k.qkv : nullptr, planes ? k.ctx : nullptr);
            else rcq = run_qkv(tune[IDF_TUNE_GEMM_QKV], s, g, ar + ly.sa_in_pack, u_np, nullptr, nullptr, 0, qkv_h2, planes ? k.qkv : nullptr, planes ? k.ctx : nullptr);
            if (rcq != IDF_OK) return rcq;
            idf_prof_mark(IDF_K_SELF_ATTN, s);
            if (tune[IDF_TUNE_GEMM_OUTPROJ] == 0) {
                // u1 = xn + ctx.Wo^T + bo with the product taken per head inside the attention kernel: H partial slabs in the FFN's
                // slab buffer (its previous contents were consumed by the QKV kernel), summed with xn + bo by the row block
                int rc_ah2 = IDF_NOT_EXCLUSIVE;
                // the split-f16 form (attn_h2.h: 32 queries per workgroup, one workgroup per CU) is the default since round 5 (row-major V planes read with the transposing LDS read:
                // -1.2 % per step against the fp32 kernel, profiles/r05_attn_split_f16_ab.txt); tune[IDF_TUNE_MISC] == 6 keeps the fp32 kernel for A/B; clips longer than its LDS
                // budget (T > 192), the exact arithmetic and a device where it does not get its CU take the fp32 kernel
                if (attn_h2) {
                    rc_ah2 = idf_attn_h2::launch_self_attn_h2(s, k.qkv, B, T, ar + ly.sa_out_frag_h2, k.parts, pstride, planes ? k.ctx : nullptr);
                    if (rc_ah2 != IDF_OK && (rc_ah2 != IDF_NOT_EXCLUSIVE || planes)) return idf_public_rc(rc_ah2);
                }
                if (rc_ah2 == IDF_NOT_EXCLUSIVE) launch_self_attn_outproj(s, k.qkv, B, T, ar + ly.sa_out_frag, k.parts, pstride);
                idf_prof_mark(IDF_K_ROWBLOCK_STD, s);
                const bool rb8 = tune[IDF_TUNE_MISC] != 8;
                const int dyn8 = rb_h2 && rb8 ? (tv8 == 8 ? rb8_std_dyn<MS, 8>() : rb8_std_dyn<MS, 16>()) : -1;
                const int dyn = rb_h2 && dyn8 < 0 ? rb_h2_std_dyn<MS>() : -1;
                if (dyn8 >= 0 && tv8 == 8) {
                    rowblock8_kernel<false, H, MS, 8><<<dim3(rb8_grid.x * B), dim3(512), (size_t)dyn8, s>>>(k.parts, T, (int)rb8_grid.x, B, mlen, pstride, k.xn, nullptr,
                                   nullptr, nullptr, nullptr, ar + ly.ln_w[0], ar + ly.ln_b[0], Gh, g0l, VWh, ar + ly.ca_out_b, ar + ly.ln_w[1],
                                   ar + ly.ln_b[1], k.x2, ar + ly.sa_out_b, scl);
                } else if (dyn8 >= 0) {
                    rowblock8_kernel<false, H, MS, 16><<<dim3(rb_grid.x * B), dim3(512), (size_t)dyn8, s>>>(k.parts, T, (int)rb_grid.x, B, mlen, pstride, k.xn, nullptr,
                                   nullptr, nullptr, nullptr, ar + ly.ln_w[0], ar + ly.ln_b[0], Gh, g0l, VWh, ar + ly.ca_out_b, ar + ly.ln_w[1],
                                   ar + ly.ln_b[1], k.x2, ar + ly.sa_out_b, scl);
                } else if (dyn >= 0) {
                    rowblock_kernel<false, true, H, true, MS><<<rb_grid, dim3(256), (size_t)dyn, s>>>(k.parts, nullptr, nullptr, nullptr, nullptr,
                                   ar + ly.ln_w[0], ar + ly.ln_b[0], Gh, g0l, VWh, ar + ly.ca_out_b, ar + ly.ln_w[1],
                                   ar + ly.ln_b[1], k.x2, T, 0, pstride, k.xn, ar + ly.sa_out_b, scl, mlen);
                } else
                hipLaunchKernelGGL((rowblock_kernel<false, true, H, false, MS>), rb_grid, dim3(256), 0, s, k.parts, nullptr, nullptr, nullptr, nullptr,
                                   ar + ly.ln_w[0], ar + ly.ln_b[0], Gl, g0l, VWTl, ar + ly.ca_out_b, ar + ly.ln_w[1],
                                   ar + ly.ln_b[1], k.x2, T, 0, pstride, k.xn, ar + ly.sa_out_b, nullptr, mlen);
            } else {                                   // A/B runs: the out-projection as a separate GEMM (tools/kbench.py)
                hipLaunchKernelGGL((self_attn_kernel<false, 2>), dim3((unsigned)(idf_cdiv(T, 32) * H * B)), dim3(256), attn_lds_bytes(T, 2), s, k.qkv, k.ctx, T,
                                   (int)(idf_cdiv(T, 32) * H * B), nullptr, nullptr, (size_t)0);
                Args o{};
                o.A = k.ctx; o.lda = D; o.K = D; o.W = ar + ly.sa_out_w; o.bias = ar + ly.sa_out_b; o.C = u_tmp; o.ldc = D; o.M = N;
                o.N = D; o.resid = k.xn; o.T = T;
                idf_prof_mark(IDF_K_GEMM_OUTPROJ, s);
                run_gemm<A_PLAIN, E_RESID>(tune[IDF_TUNE_GEMM_OUTPROJ], s, o);
                idf_prof_mark(IDF_K_ROWBLOCK_STD, s);
                hipLaunchKernelGGL((rowblock_kernel<false, true, 1, false, MS>), rb_grid, dim3(256), 0, s, u_tmp, nullptr, nullptr, nullptr, nullptr,
                                   ar + ly.ln_w[0], ar + ly.ln_b[0], Gl, g0l, VWTl, ar + ly.ca_out_b, ar + ly.ln_w[1],
                                   ar + ly.ln_b[1], k.x2, T, 0, (size_t)0, nullptr, nullptr, nullptr, mlen);
            }
        }
        // u3 = x2 + linear2(gelu(linear1(x2))) as NSL partial slabs (ffn.h); their sum is taken by the next reader
        idf_prof_mark(IDF_K_FFN_FUSED, s);
        if (const int rc = idf_launch_layer_ffn(s, ly, ar, tune, k.x2, N, k.parts); rc != IDF_OK) return rc;
        u_in = k.parts;
        u_np = NSL;
        lnp_w = ar + ly.ln_w[2];
        lnp_b = ar + ly.ln_b[2];
    }
    if (tail_h2) {
        ta.u_in = u_in; ta.pstride = pstride; ta.ln_w = lnp_w; ta.ln_b = lnp_b; ta.wout = ar + w->out_w_h2; ta.out_b = ar + w->out_b; ta.x0 = x0;
        idf_prof_mark(IDF_K_GEMM_HEADS, s);
        int mode = 1;
        if (post.x) {
            ta.post.N = C; ta.post.M = N; ta.post.T = T;
            ta.post.post_x = post.x; ta.post.post_gt = post.gt; ta.post.post_mask = post.mask; ta.post.post_table = post.table; ta.post.post_state = post.state;
            mode = (flags & IDF_STEP_EMBED_NEXT) ? 3 : 2;
        }
        if (const int rc = idf_tail_h2::launch_tail(s, mode, ta); rc != IDF_OK) return idf_public_rc(rc);
    } else
    {   // heads: x0[b][c][t] = LN3_last(u).Wout^T + b
        Args g{};
        g.A = u_in; g.lda = D; g.K = D; g.lnw = lnp_w; g.lnb = lnp_b; g.W = ar + w->out_w; g.bias = ar + w->out_b; g.C = x0;
        g.ldc = C; g.M = N; g.N = C; g.T = T; g.a_pstride = pstride;
        idf_prof_mark(IDF_K_GEMM_HEADS, s);
        if (post.x) {
            g.post_x = post.x; g.post_gt = post.gt; g.post_mask = post.mask; g.post_table = post.table; g.post_state = post.state;
            run_heads_post(tune[IDF_TUNE_GEMM_HEADS], s, g, u_np);
        } else {
            run_gemm_ln<E_HEADS>(pick(tune[IDF_TUNE_GEMM_HEADS], CFG_HEADS), s, g, u_np);
        }
    }
    idf_prof_mark(-1, s);
    IDF_CHECK_LAUNCH();
    return IDF_OK;
}
int mdm_forward_impl(const idf_mdm_weights *w, const float *memctx, const float *x, const int64_t *ts, int32_t B, int32_t T, float *x0,
                     void *ws, size_t ws_bytes, void *stream, const StepPost &post, int32_t flags = 0) {
    if (!w || w->mem_len < 0 || w->mem_len > MEMX) return IDF_E_INVAL;
    return idf_mem_len(w) == MEM ? mdm_forward_impl_t<MEM>(w, memctx, x, ts, B, T, x0, ws, ws_bytes, stream, post, flags)
                                 : mdm_forward_impl_t<MEMX>(w, memctx, x, ts, B, T, x0, ws, ws_bytes, stream, post, flags);
}
}  // namespace

extern "C" int interdiff_mdm_forward(const idf_mdm_weights *w, const float *memctx, const float *x, const int64_t *ts,
                                     int32_t B, int32_t T, float *x0, void *ws, size_t ws_bytes, void *stream) {
    if (!x0) return IDF_E_INVAL;
    return mdm_forward_impl(w, memctx, x, ts, B, T, x0, ws, ws_bytes, stream, StepPost{});
}

// One plain DDPM reverse step in the denoiser's own launches: the forward above with the x0 tile of the last GEMM consumed in its
// epilogue (inpaint, posterior mean, in-kernel noise) -- x is updated in place and the sampler state advanced, exactly what
// interdiff_mdm_forward + interdiff_posterior_step_dev(ts != NULL) do in two more HBM passes and one more launch.
extern "C" int interdiff_mdm_forward_step(const idf_mdm_weights *w, const float *memctx, float *x, int64_t *ts, int32_t B, int32_t T,
                                          const float *gt, const uint8_t *mask, const float *table, int64_t *state, void *ws,
                                          size_t ws_bytes, void *stream) {
    if (!x || !ts || !table || !state || (mask && !gt) || T <= 0) return IDF_E_INVAL;
    // T % 4 == 0: the update is one aligned 16-byte access per lane; other clip lengths take the per-row form of the epilogue (gemm.h)
    if (!(T & 3) && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gt)) & 15 || (reinterpret_cast<uintptr_t>(mask) & 3))) return IDF_E_INVAL;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gt)) & 3) return IDF_E_INVAL;
    if (!w || w->layer[0].is_qan) return IDF_E_INVAL;         // the step bookkeeping rides on layer 0's QKV kernel
    return mdm_forward_impl(w, memctx, x, ts, B, T, nullptr, ws, ws_bytes, stream, StepPost{x, gt, mask, table, state, ts});
}

// The same with flags that chain CONSECUTIVE plain steps on one workspace (include/interdiff_hip.h IDF_STEP_*): a step's last launch then also computes
// the next step's embedding from the token rows it has just updated (tail_h2.h), and the next call starts at its QKV projection.
extern "C" int interdiff_mdm_forward_step_ex(const idf_mdm_weights *w, const float *memctx, float *x, int64_t *ts, int32_t B, int32_t T,
                                             const float *gt, const uint8_t *mask, const float *table, int64_t *state, void *ws,
                                             size_t ws_bytes, int32_t flags, void *stream) {
    if (!x || !ts || !table || !state || (mask && !gt) || T <= 0 || (flags & ~(IDF_STEP_EMBED_READY | IDF_STEP_EMBED_NEXT))) return IDF_E_INVAL;
    if (!(T & 3) && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gt)) & 15 || (reinterpret_cast<uintptr_t>(mask) & 3))) return IDF_E_INVAL;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gt)) & 3) return IDF_E_INVAL;
    if (!w || w->layer[0].is_qan) return IDF_E_INVAL;
    return mdm_forward_impl(w, memctx, x, ts, B, T, nullptr, ws, ws_bytes, stream, StepPost{x, gt, mask, table, state, ts}, flags);
}

// 1 when interdiff_mdm_forward_step_ex honours its flags for this handle (split arithmetic selected, token width 144, plane fragments packed); 0: they are ignored
extern "C" int interdiff_mdm_step_chaining(const idf_mdm_weights *w) {
    return w && w->tune[IDF_TUNE_FFN_MATH] != 0 && w->out_w_h2 != 0 && w->in_w_h2 != 0 && w->C == idf_tail_h2::CW && w->tail_h2_ok != 0 &&
                   idf_tail_h2::tail_exclusive_ok(false) && idf_tail_h2::tail_exclusive_ok(true) ? 1 : 0;
}

// DEBUG (tests only): kernels whose name -- as interdiff_exclusive_cu_report prints it -- contains one of the comma-separated patterns are treated as NOT getting their CU from now on,
// so that the fp32 kernel behind every split-f16 launcher can be executed on a device where all claims hold; null / "" clears the list.  Process-wide; not for product use.
extern "C" int interdiff_debug_deny_exclusive(const char *patterns) {
    idf_excl_set_deny(patterns);
    return IDF_OK;
}

// Verdicts of the exclusive-CU check (common.h idf_exclusive_cu) for EVERY kernel of the library that issues the f16 MFMA, on the current device: forces
// the check for each (no launch), writes one text line per kernel into buf (<= cap bytes, NUL-terminated) and returns the number of kernels that do NOT
// get their CU to themselves (those run as their fp32-MFMA counterparts), or a negative IDF_E_* code.
extern "C" int interdiff_exclusive_cu_report(char *buf, int32_t cap) {
    if (!buf || cap <= 0) return IDF_E_INVAL;
    {
        static idf_excl_cache c[7];
        using namespace idf_ffn_h2;
        idf_exclusive_cu(reinterpret_cast<const void *>(&ffn_h2_kernel<1, FFN_H2_SLOTS, 0>), "ffn_h2_kernel<16 rows>", NT, c[0]);
        idf_exclusive_cu(reinterpret_cast<const void *>(&ffn_h2_kernel<2, FFN_H2_SLOTS, 0>), "ffn_h2_kernel<32 rows>", NT, c[1]);
        idf_exclusive_cu(reinterpret_cast<const void *>(&ffn_h2_kernel<2, FFN_H2_SLOTS, 0, 8>), "ffn_h2_kernel<32 rows, loader waves>", NT + 512, c[6]);
        idf_exclusive_cu(reinterpret_cast<const void *>(&ffn_h2_kernel<4, 2, 0>), "ffn_h2_kernel<64 rows>", NT, c[2]);
        idf_exclusive_cu(reinterpret_cast<const void *>(&ln_linear_h2_kernel<1, false, IDF_QKV_LOADER_WAVES>), "ln_linear_h2_kernel<1 slab>", NT + 64 * IDF_QKV_LOADER_WAVES, c[3]);
        idf_exclusive_cu(reinterpret_cast<const void *>(&ln_linear_h2_kernel<IDF_FFN_SLICES, false, IDF_QKV_LOADER_WAVES>), "ln_linear_h2_kernel<5 slabs>", NT + 64 * IDF_QKV_LOADER_WAVES, c[4]);
        idf_exclusive_cu(reinterpret_cast<const void *>(&idf_attn_h2::self_attn_h2_kernel<0, false>), "self_attn_h2_kernel", idf_attn_h2::NTH, c[5]);
    }
    qkv_planes_ok(1, 1, 16);
    qkv_planes_ok(NSL, 1, 16);
    rb_h2_qan_dyn<MEM>();
    rb_h2_std_dyn<MEM>();
    rb_h2_qan_dyn<MEMX>();
    rb_h2_std_dyn<MEMX>();
    rb8_qan_dyn<MEM, 16>();
    rb8_std_dyn<MEM, 16>();
    rb8_qan_dyn<MEM, 8>();
    rb8_std_dyn<MEM, 8>();
    rb8_qan_dyn<MEMX, 16>();
    rb8_std_dyn<MEMX, 16>();
    rb8_qan_dyn<MEMX, 8>();
    rb8_std_dyn<MEMX, 8>();
    idf_tail_h2::tail_exclusive_ok(false);
    idf_tail_h2::tail_exclusive_ok(true);
    return idf_excl_report(buf, cap);
}
