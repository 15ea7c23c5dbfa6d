/*
 * interdiff_hip.h -- C-ABI of the MI355X-native InterDiff denoising-sampler hot path.
 *
 * The reference (Sirui-Xu/InterDiff) has no FFI: its seams are Python callables on torch
 * tensors.  This library is what a binding for those seams would load:
 * every entry point below names the reference callable it replaces (file:line relative to
 * /root/reference/interdiff).  INTEGRATION.md shows the ctypes stub for each.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer (gfx950 HBM) unless its name starts with h_;
 *  - tensors are dense, row-major, fp32 unless stated; shapes are given in comments;
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream); all work is
 *    enqueued asynchronously on it, nothing synchronises, nothing allocates: scratch comes
 *    from the caller (`ws`, sized by the matching *_workspace_bytes function);
 *  - return value: 0 on success, a negative code on error (IDF_E_*); no exceptions, no
 *    callbacks, no mutable process state (the only statics are idempotent per-device caches of a
 *    kernel attribute, see csrc/common.h idf_opt_in_lds) => safe to capture in a hipGraph;
 *  - inputs are borrowed; outputs are caller-allocated.
 */
#ifndef INTERDIFF_HIP_H
#define INTERDIFF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IDF_OK            0
#define IDF_E_INVAL      -22   /* bad shape / null pointer / unsupported size            */
#define IDF_E_NOMEM      -12   /* workspace too small                                     */
#define IDF_E_LAUNCH     -5    /* hipGetLastError() != hipSuccess after a launch          */

int         interdiff_abi_version(void);          /* bumped on any signature change       */
const char *interdiff_build_info(void);           /* "gfx950 hipcc ..."                   */

/* ------------------------------------------------------------------------------------
 * Rotation conversions  (pytorch3d.transforms 0.7.2 semantics; eval_smpl_short.py:18,
 * 33,65-66,90-91,157-162; model/diffusion_smpl.py:212-213).  n = number of rotations.
 * quaternions are (w,x,y,z); matrices row-major 3x3; rot6d = first two ROWS.
 * ---------------------------------------------------------------------------------- */
int interdiff_rotation_6d_to_matrix   (const float *d6, float *m,  int64_t n, void *stream);
int interdiff_matrix_to_rotation_6d   (const float *m,  float *d6, int64_t n, void *stream);
int interdiff_matrix_to_axis_angle    (const float *m,  float *aa, int64_t n, void *stream);
int interdiff_axis_angle_to_matrix    (const float *aa, float *m,  int64_t n, void *stream);
int interdiff_axis_angle_to_quaternion(const float *aa, float *q,  int64_t n, void *stream);
int interdiff_rotation_6d_to_axis_angle(const float *d6, float *aa, int64_t n, void *stream);

/* ------------------------------------------------------------------------------------
 * SMPL-H forward kinematics + linear blend skinning
 * replaces SMPL_Layer.forward  (libsmpl/smplpytorch/pytorch/smpl_layer.py:72-175).
 *
 * The model is passed as PACKED constants built once on the host
 * (interdiff_amd/smpl.py: pack_smpl_model):
 *   blend   [3V][KB]   KB = round_up(9(J-1)+n_betas+1, 8): per output coordinate the row
 *                      [posedirs(9(J-1)) | shapedirs(n_betas) | v_template | 0-pad]
 *   jt      [J][3]     J_regressor . v_template
 *   js      [J][3][n_betas]  J_regressor . shapedirs
 *   parents [J] int32  (parents[0] ignored)
 *   skin_idx[V][S] int32, skin_w [V][S]   ELL form of the skinning weights (zeros dropped,
 *                      padded with weight 0 / index 0); S = max non-zeros per vertex
 * ---------------------------------------------------------------------------------- */
typedef struct {
    int32_t V, J, n_betas, KB, S;
    const float   *blend;      /* blend basis [posedirs | shapedirs | template] (KB columns) in the kernel's MFMA fragment order: [ceil(V/64)][4][3][KB/16][64][4], rows past 3V zero (smpl.py pack_smpl_model) */
    const float   *jt;
    const float   *js;
    const int32_t *parents;
    const int32_t *skin_idx;
    const float   *skin_w;
} idf_smpl_model;

size_t interdiff_smpl_workspace_bytes(const idf_smpl_model *m, int64_t N);
/* pose [N,3J] axis-angle, betas [N,n_betas], trans [N,3] -> verts [N,V,3], jtr [N,J,3],
 * v_posed [N,V,3] (may be NULL). */
int interdiff_smpl_forward(const idf_smpl_model *m, const float *pose, const float *betas,
                           const float *trans, int64_t N, float *verts, float *jtr,
                           float *v_posed, void *ws, size_t ws_bytes, void *stream);

/* ------------------------------------------------------------------------------------
 * Geometry
 * vertex_normals        replaces data/tools.py:4-40   (faces NOT repeated per frame)
 *   adj_ptr [V+1], adj_face [nnz], adj_corner [nnz] (uint8 as int32): vertex->incident
 *   faces in the reference's accumulation order (corner 1, corner 2, corner 0; ascending
 *   face index inside each) -- built by interdiff_amd/geometry.py: build_vertex_adjacency.
 * nn_argmin / point2point_signed replace tools.py:11-76 + chamfer_distance (brute force,
 *   d2 = (dx*dx+dy*dy)+dz*dz without FMA, lowest index wins ties).
 * ---------------------------------------------------------------------------------- */
int interdiff_vertex_normals(const float *verts, int64_t N, int32_t V, const int32_t *faces,
                             const int32_t *adj_ptr, const int32_t *adj_face,
                             const int32_t *adj_corner, float *normals, void *stream);
/* q [N,Pq,3], r [N,Pr,3] -> idx int32 [N,Pq] : nearest r for every q */
int interdiff_nn_argmin(const float *q, int32_t Pq, const float *r, int32_t Pr, int64_t N,
                        int32_t *idx, void *stream);
/* x [N,P1,3], y [N,P2,3], optional normals (NULL = unsigned).  Outputs as tools.py:73-76:
 * y2x_signed [N,P2], x2y_signed [N,P1], yidx [N,P2], xidx [N,P1], y2x [N,P2,3], x2y [N,P1,3]
 * (vector outputs may be NULL). */
int interdiff_point2point_signed(const float *x, int32_t P1, const float *y, int32_t P2, int64_t N,
                                 const float *x_normals, const float *y_normals,
                                 float *y2x_signed, float *x2y_signed, int32_t *yidx, int32_t *xidx,
                                 float *y2x, float *x2y, void *stream);

/* ------------------------------------------------------------------------------------
 * MDM denoiser, decoder path   replaces MDM.forward / _decode
 * (model/diffusion_smpl.py:226-246; layers [std, QaN x6, std], sublayers.py:206-375).
 *
 * Weights are packed once from a state_dict with the reference's key names
 * (interdiff_amd/mdm.py: pack_mdm_weights) into ONE fp32 arena; idf_mdm_weights holds
 * offsets (in floats) into it.  Token rows are clip-major: row = b*T + t.
 * ---------------------------------------------------------------------------------- */
#define IDF_MDM_LAYERS 8
#define IDF_MDM_D      256
#define IDF_MDM_FF     1024
#define IDF_MDM_HEADS  4
#define IDF_MDM_NQ     10      /* learned queries per QaN layer                          */
#define IDF_MDM_MEM    10      /* memory (past) tokens: the default (every BASELINE config), compact layout of the folded memory */
#define IDF_MDM_MEM_MAX 16     /* longest memory (idf_mdm_weights.mem_len): the reference takes --past_len from the CLI (eval_smpl_short.py:376) */
#define IDF_FFN_SLICES 5       /* hidden-unit slices of the fused FFN = partial output slabs (csrc/ffn.h) */

typedef struct {
    int64_t is_qan;            /* 0: torch TransformerDecoderLayer, 1: QaN               */
    int64_t sa_in_w, sa_in_b, sa_out_w, sa_out_b;   /* std only: [768,256],[768],[256,256],[256] */
    int64_t qc, wk;            /* QaN only: Qc [NQ][3][256] (pre-rotated, pre-scaled) in MFMA fragment order [16][3][4][NQ][4] (mdm.py qan_fragments), wk [NQ] */
    int64_t ca_q_w, ca_q_b;    /* cross-attn query proj [256,256],[256]                  */
    int64_t ca_kv_w, ca_kv_b;  /* cross-attn key|value proj [512,256],[512]              */
    int64_t ca_out_w, ca_out_b;
    int64_t ff1_w, ff1_b, ff2_w, ff2_b;             /* [1024,256],[1024],[256,1024],[256] */
    int64_t ffn_pack;          /* linear1 + linear2 weights in the fused FFN kernel's stream order (5 x 106496 floats, mdm.py pack_ffn) */
    int64_t ffn_b1p;           /* linear1 bias zero-padded to 5 x 208 (+ 256 spare floats)                                      */
    int64_t sa_in_pack;        /* std only: sa_in_w in the LayerNorm+linear kernel's stream order (5 slices x 16 chunks [160][16], rows past 768 zero; mdm.py pack_linear160) */
    int64_t sa_out_frag;       /* std only: sa_out_w in MFMA fragment order [head][4 column quarters][4 k-groups][4 column tiles][64 lanes][4] (mdm.py sa_out_fragments): the attention kernel's out-projection tail */
    int64_t ln_w[3], ln_b[3];
    int64_t ffn_pack_h2;       /* 0, or linear1 + linear2 split into two f16 planes (hi, residual x 2^11) in the split-f16 FFN kernel's stream order
                                  (5 x 110592 floats, mdm.py pack_ffn_h2; csrc/ffn_h2.h).  Only set when the packer has PROVED that no operand of
                                  this layer's feed-forward block can leave the f16 range (mdm.py ffn_h2_range_ok) */
    int64_t sa_in_pack_h2;     /* std only: 0, or sa_in_w as two f16 planes in the split-f16 QKV kernel's stream order (5 slices x 8 K steps
                                  [10 tiles][2 planes][64 lanes][8 halves], rows past 768 zero; mdm.py pack_linear160_h2; csrc/ffn_h2.h
                                  ln_linear_h2_kernel).  The kernel scales every input row by a power of two, so only the weights' range matters */
    int64_t qc_h2;             /* QaN only: 0, or Qc as two f16 planes in the split-f16 row block's fragment order [4 K quarters][2 K steps][3 taps][2 planes][4 kq][NQ][8 halves]
                                  (mdm.py qan_fragments_h2; csrc/denoiser.hip rowblock_kernel<.., H2>) */
    int64_t rb_h2_ok;          /* decoder layers: 1 when the packer has proved that the LayerNorm outputs this layer's row block contracts (LN_prev, norm1) stay inside
                                  the f16 range (16 max|gamma| + max|beta| < 65504; mdm.py ln_h2_range_ok) -- with tune[IDF_TUNE_FFN_MATH] == 1 the row block then runs
                                  its three contractions as split-f16 products; 0 = always exact fp32 */
    int64_t sa_out_frag_h2;    /* std only: 0, or sa_out_w as two f16 planes in the split-f16 attention kernel's fragment order [head][16 output column tiles][2 K steps][2 planes]
                                  [64 lanes][8 halves] (mdm.py sa_out_fragments_h2; csrc/attn_h2.h) */
    int64_t qkv_bounds_ok;     /* std only: 1 when eight floats stand right behind the sa_in_pack_h2 stream (arena offset sa_in_pack_h2 + 5 x 40960): max over the q, k, v rows
                                  of sa_in_w of their L1 norm (x 1.001), max |sa_in_b| of the q, k, v rows, two spare (mdm.py qkv_bounds).  With them the QKV kernel may write
                                  its output as the self-attention's f16 plane pairs -- |q_c| <= 2^e_row ||W_c||_1 + |b_c| is the per-row power of two it divides by --
                                  instead of fp32 rows (csrc/ffn_h2.h ln_linear_h2_kernel<.., PLANES>); 0: fp32 rows, the attention splits them itself */
} idf_mdm_layer;

typedef struct {
    int32_t C;                 /* token width (144; 106 for the skeleton tokens of config #1) */
    int32_t n_steps;           /* rows of temb_table                                     */
    const float *arena;
    int64_t in_w, in_b;        /* [256][(C + 3) & ~3] = [bodyEmbedding | objEmbedding | 0], summed bias  */
    int64_t out_w, out_b;      /* [C][256] = [bodyFinalLinear ; objFinalLinear]          */
    int64_t temb_table;        /* [n_steps][256] = time_embed(pe[t]) (weights-only const) */
    int64_t pe;                /* [max_T][256] positional table rows 0..max_T-1          */
    int32_t max_T, has_encoder;
    idf_mdm_layer layer[IDF_MDM_LAYERS];
    /* encoder side ("next" row, MDM._get_embeddings): [std, QaN x6, std] without cross-attention; uses sa_*, qc, wk,
     * ff*, ln_w/ln_b[0..1] (= norm1, norm2); valid when has_encoder != 0 */
    idf_mdm_layer enc_layer[IDF_MDM_LAYERS];
    /* tile-configuration overrides for A/B measurements (tools/kbench.py), indexed by IDF_TUNE_*; all zero = the shipped
     * configuration.  A field of the handle, not process state: two models in one process never see each other's overrides.
     * One entry is NOT only for A/B runs -- tune[IDF_TUNE_FFN], the row tile of the fused feed-forward block: 0 = by the rows of
     * the launch (16-row tiles up to 800 rows, 64-row tiles from 2800 rows on, 32-row tiles
     * otherwise: csrc/ffn.h ffn_tile_for_rows), 1 = 32-row tiles, 2 = 16-row tiles, 3 = 64-row tiles.  The 32-row kernel agrees with
     * the other two to rounding (7e-7 of the output scale), not bit for bit: a caller that steps ONE batch as several calls on row subsets (the
     * sampler's half-batch chains) sets 1 or 2 from the rows of the whole batch so that every call takes the same kernel
     * (interdiff_amd/mdm.py: MDM._pick_ffn_tile does this before every forward / forward_step / encode / ffn call).
     * tune[IDF_TUNE_FFN_MATH] selects the arithmetic of the feed-forward block, of the QKV projection (layers whose sa_in_pack_h2 is set) AND of the row block's three contractions (layers whose rb_h2_ok -- and, QaN, qc_h2 -- is set): 0 = exact fp32 MFMA (v_mfma_f32_16x16x4_f32, csrc/ffn.h),
     * 1 = split-f16 (every fp32 operand as two f16 planes, three v_mfma_f32_16x16x32_f16 per product, fp32 accumulate: fp32-grade
     * results at 1/43 of the matrix-pipe time, csrc/ffn_h2.h) for every layer whose ffn_pack_h2 is set, exact fp32 for the others.  Its
     * 16-, 32- and 64-row tiles are bit-identical, so with 1 the row tile is a pure performance choice.  2 = like 1 for the feed-forward block and the QKV
     * projection, exact fp32 in the row block (A/B runs). */
    int32_t tune[8];
    /* split-f16 plane fragments of the two token GEMMs at the ends of a step (csrc/tail_h2.h; mdm.py pack_tail_h2), 0 = not packed (token width != 144
     * or a value outside the f16 range): out_w as [9 output tiles][8 K steps][2 planes][64 lanes][8 halves], in_w as [16][5][2][64][8] (K = 144 padded to 160) */
    int64_t out_w_h2, in_w_h2;
    /* 1 when the packer has proved that the heads GEMM's A operand -- LayerNorm norm3 of the LAST decoder layer -- stays inside the f16 range
     * (mdm.py ln_h2_range_ok on its gamma / beta); 0: the step tail keeps the fp32 token GEMMs (and interdiff_mdm_forward_step_ex ignores its flags) */
    int64_t tail_h2_ok;
    /* memory length of the NEXT interdiff_mdm_prepare_memory / forward calls on this handle: rows of cond [mem_len,B,256].  0 = IDF_MDM_MEM (10).  Any 1 ..
     * IDF_MDM_MEM_MAX is served; 10 takes the compact (fast) layout of the folded memory, other lengths a generic one (one more score-column tile in the row
     * block).  A memctx folded at one length must be consumed at the same length (its size differs: interdiff_mdm_memctx_floats_for). */
    int32_t mem_len;
    /* tokens per workgroup of the split-f16 row block (csrc/denoiser.hip rowblock8_kernel): 0 = by the launch (8 while B x ceil(T / 8) workgroups fit the chip in one
     * round, else 16), 8 / 16 = forced (A/B runs).  Both forms compute the same bits. */
    int32_t rb_tokens;
} idf_mdm_weights;

/* The token GEMM of the denoiser as a standalone op: C[M,N] = epi(A[M,K] . W[N,K]^T + bias) on the fp32 MFMA
 * (torch.nn.functional.linear semantics; model/diffusion_smpl.py:73-120 linear1/linear2/out_proj).  epi: 0 bias,
 * 1 bias + erf-GELU, 2 bias + resid (leading dimension ldc).  cfg 0 = shipped tile configuration.  K % 64 == 0, N % 16 == 0. */
int interdiff_gemm_f32(const float *A, int32_t lda, const float *W, const float *bias, const float *resid, float *C,
                       int32_t ldc, int32_t M, int32_t N, int32_t K, int32_t epi, int32_t cfg, void *stream);

/* per-sample constants derived from the memory `cond` (constant over all steps): for every
 * layer and clip the folded cross-attention operands
 *   G   [L][B][40][256]  (scores = x . G^T + g0, 40 = heads*MEM, pre-scaled by 1/sqrt(64))
 *   VWT [L][B][256][48]  (out = P . VW + out_bias, stored output-column-major, 40 padded to 48)
 *   g0  [L][B][40]
 * and, behind them, the same G / VW as split-f16 plane fragments for the row block's f16-MFMA form (csrc/denoiser.hip G_H2 / VW_H2: each (layer,
 * clip) matrix divided by the power of two that puts its largest magnitude in [2^13, 2^14), two f16 planes, fragment order) with the two powers
 * of two per (layer, clip).  Both forms are always folded; which one a forward reads follows tune[IDF_TUNE_FFN_MATH].  The layout is private to
 * the library: `memctx` is an opaque buffer of interdiff_mdm_memctx_floats(B) floats.
 * cond [MEM,B,256] (reference layout). */
/* The feed-forward block of one layer as a standalone op (what bench.py times for its roofline block; the denoiser launches the
 * same kernel): x2 [M,256] -> parts [IDF_FFN_SLICES][M][256] whose sum over the slabs is x2 + linear2(gelu(linear1(x2)))
 * (torch.nn.TransformerDecoderLayer._ff_block + residual; sublayers.py:331-341).  encoder != 0 selects enc_layer[layer].
 * Row tile by w->tune[IDF_TUNE_FFN] (see idf_mdm_weights). */
int interdiff_mdm_ffn(const idf_mdm_weights *w, int32_t layer, int32_t encoder, const float *x2, int32_t M, float *parts, void *stream);
/* floats of a memctx for B clips at memory length mem_len (1 .. IDF_MDM_MEM_MAX; 0 if out of range); interdiff_mdm_memctx_floats(B) is the mem_len = 10 case */
size_t interdiff_mdm_memctx_floats_for(int32_t B, int32_t mem_len);

size_t interdiff_mdm_memctx_floats(int32_t B);
size_t interdiff_mdm_workspace_bytes(int32_t B, int32_t T);
int interdiff_mdm_prepare_memory(const idf_mdm_weights *w, const float *cond, int32_t B,
                                 float *memctx, void *ws, size_t ws_bytes, void *stream);
/* x [B,1,C,T], ts int64 [B] -> x0 [B,1,C,T].
 * Size limits (IDF_E_INVAL beyond them): T <= max_T of the packed positional table (mdm.py pack_mdm_weights(max_T=): up to the 5000 rows of the
 * reference's PositionalEncoding).  Up to T = 208 the temporal self-attention of the two standard layers parks K and V of one (clip, head) in one CU's
 * LDS (2 x T x 68 floats + the score tile = 149 KiB at T = 208: the fast path, every BASELINE shape); longer clips take a K/V-tiled form of the same
 * attention (64-key tiles, running row maximum / sum; csrc/denoiser.hip self_attn_tiled_kernel) -- correct, not tuned.  Memory length: idf_mdm_weights.mem_len;
 * C <= 256 (any width: W_in's rows are packed zero-padded to a multiple of 4 floats, mdm.py; BASELINE config #1 has C = 106).  The reference itself is bounded only by PositionalEncoding(max_len=5000) (model/layers.py:11); its
 * datasets use T = 35 (eval_smpl_short.py:376-377) and BASELINE.json T = 100.  B is unbounded (clips are independent). */
int interdiff_mdm_forward(const idf_mdm_weights *w, const float *memctx, const float *x,
                          const int64_t *ts, int32_t B, int32_t T, float *x0,
                          void *ws, size_t ws_bytes, void *stream);

/* ------------------------------------------------------------------------------------
 * Conditioning path ("next" row of SURVEY.md §8(f)): MDM._get_embeddings (model/diffusion_smpl.py:195-223) minus
 * the dataset plumbing.
 *  interdiff_pointnet2_encode   replaces PointNet2Encoder.forward (model/layers.py:141-175) with one key point:
 *      obj_points [B,P,3] (P <= 2048) -> pc [B,256] = [key-point xyz | Linear(SA2 features)]; eval BatchNorm folded
 *      into the shared-MLP convolutions at pack time (weights [cout][cin] + bias per layer).
 *  interdiff_mdm_encode         replaces bodyEmbedding/objEmbedding + PositionalEmbedding + encoder:
 *      x_past [B,1,C,Tp] (the first Tp = past_len frames of the token tensor), pc [B,256] -> cond [Tp,B,256].
 * ---------------------------------------------------------------------------------- */
typedef struct {
    int64_t w[3], b[3];        /* offsets into the arena: folded conv*BN weight [c[l+1]][c[l]], bias [c[l+1]]        */
    int32_t c[4];              /* channel plan (input includes the 3 xyz channels)                                 */
} idf_pn_mlp;

typedef struct {
    const float *arena;
    idf_pn_mlp sa1[2];         /* radii 0.05 / 0.1, 16 / 32 samples: 4-16-16-32, 4-32-32-64                        */
    idf_pn_mlp sa2[2];         /* radii 0.1 / 0.2, 16 / 32 samples: 99-64-64-128, 99-64-96-128                     */
    int64_t lin_w, lin_b;      /* Linear [253][256], [253]                                                         */
} idf_pointnet2;

int interdiff_pointnet2_encode(const idf_pointnet2 *pn, const float *obj_points, int32_t B, int32_t P, float *out,
                               void *stream);
size_t interdiff_mdm_encode_workspace_bytes(int32_t B, int32_t Tp);
int interdiff_mdm_encode(const idf_mdm_weights *w, const float *pc, const float *x_past, int32_t B, int32_t Tp,
                         float *cond, void *ws, size_t ws_bytes, void *stream);

/* ------------------------------------------------------------------------------------
 * Sampler step   replaces p_mean_variance's inpainting + q_posterior mean and p_sample's
 * noise add (diffusion/gaussian_diffusion.py:307-311,374,532-547):
 *   x0   = mask ? gt : x0                      (mask [n] uint8, may be NULL)
 *   x    = c1*x0 + c2*x + sigma*eps            (sigma = 0 at t == 0)
 * eps comes from `noise` [n] when non-NULL, else from the counter-based generator
 * (Philox4x32-10 + Box-Muller) keyed by (seed, step_index, element).
 * `inpaint_only` != 0 applies just the first line (used before the correction hook).
 * ---------------------------------------------------------------------------------- */
int interdiff_inpaint(float *x0, const float *gt, const uint8_t *mask, int64_t n, void *stream);
int interdiff_posterior_step(float *x, const float *x0, const float *noise, int64_t n,
                             float c1, float c2, float sigma, uint64_t seed, uint64_t step_index,
                             void *stream);
int interdiff_randn(float *out, int64_t n, uint64_t seed, uint64_t step_index, void *stream);
/* The same two with an element offset: `elem0` (a multiple of 4, else IDF_E_INVAL) is the position of x[0] / out[0] inside the
 * tensor whose noise stream is being drawn.  The reference fills ONE randn_like tensor for the whole batch
 * (gaussian_diffusion.py:532); a rank that holds clips [first, first + B_local) of that batch passes
 * elem0 = first * C * T and draws exactly the elements the unsharded run would ("sliced from one global
 * tensor for parity with a 1-GPU run") -- sharded == unsharded bit for bit.  The un-suffixed entries are elem0 = 0. */
int interdiff_posterior_step_at(float *x, const float *x0, const float *noise, int64_t n,
                                float c1, float c2, float sigma, uint64_t seed, uint64_t step_index,
                                uint64_t elem0, void *stream);
int interdiff_randn_at(float *out, int64_t n, uint64_t seed, uint64_t step_index, uint64_t elem0, void *stream);
/* Graph-replayable form of the same update: all per-step scalars live in HBM, so one captured hipGraph of
 * [interdiff_mdm_forward -> interdiff_posterior_step_dev -> interdiff_sampler_advance] serves every plain step.
 *   state int64[4] = {t (current timestep), loop index (noise counter), seed, arrival counter (zero it once)};
 *   table  f32[steps][4] = {c1[t], c2[t], sigma[t] (0 at t == 0), t/1000};
 *   mask/gt may be NULL (x0 already inpainted).  With ts != NULL the launch also advances the state when its last
 *   workgroup retires: t -= 1, loop index += 1, ts[b] = max(t, 0) for b < B (interdiff_sampler_advance does only that). */
int interdiff_posterior_step_dev(float *x, const float *x0, const float *gt, const uint8_t *mask, int64_t n,
                                 const float *table, int64_t *state, int64_t *ts, int32_t B, void *stream);
int interdiff_sampler_advance(int64_t *state, int64_t *ts, int32_t B, void *stream);
/* One PLAIN reverse step (no denoised_fn hook) = interdiff_mdm_forward + interdiff_posterior_step_dev(ts != NULL) with the x0
 * prediction consumed inside the denoiser's last GEMM: x [B,1,C,T] is the sampler state, updated in place; ts, table, gt, mask as
 * above; state is int64[8] here: [0..3] as above, [4..5] scratch (this step's {t, loop index}, parked by one thread of layer 0's
 * QKV kernel, which also advances [0..1] and ts -- no arrival counter), [6] = element index of x[0] inside the whole sample's x
 * (a multiple of 4; 0 unless the clips of a sample are stepped as several independent chains, each with its own x slice, ts slice,
 * state, memctx and workspace: the noise of element e is then drawn at counter [6] + e, i.e. what the undivided batch would draw),
 * [7] reserved (0).  Same bits as the two-call form (the update arithmetic and
 * the noise counter are shared), and the two forms can alternate on one state.  Any T: with T % 4 == 0 a lane's four frames are one
 * aligned 16-byte access (x, gt 16-byte aligned, mask 4-byte aligned), other clip lengths (the reference's default T = 35) take a
 * per-row form of the same update.  Layer 0 must be a standard layer (IDF_E_INVAL otherwise: use the two-call form); replaces
 * gaussian_diffusion.py:425-461 (p_sample) for steps without a hook. */
int interdiff_mdm_forward_step(const idf_mdm_weights *w, const float *memctx, float *x, int64_t *ts, int32_t B, int32_t T,
                               const float *gt, const uint8_t *mask, const float *table, int64_t *state,
                               void *ws, size_t ws_bytes, void *stream);
/* interdiff_mdm_forward_step with flags that chain consecutive plain steps of ONE chain on ONE workspace (nothing may touch x, ts or the workspace between
 * the two calls): IDF_STEP_EMBED_NEXT -- this call's last launch also computes the NEXT step's embedding from the token rows it has just updated
 * (csrc/tail_h2.h: LN3 -> heads -> update -> embedding by one workgroup per 16 token rows); IDF_STEP_EMBED_READY -- the previous call was made with
 * IDF_STEP_EMBED_NEXT: this call starts at its QKV projection.  Results are bit-identical to unflagged calls.  Honoured only when
 * interdiff_mdm_step_chaining(w) is 1 (split arithmetic selected in tune[IDF_TUNE_FFN_MATH], token width 144, out_w_h2 / in_w_h2 packed); otherwise the
 * flags are ignored and every call runs its own embedding (same results).  Reference seam: one iteration of p_sample_loop_progressive
 * (gaussian_diffusion.py:640-681) -- the reference runs ~350 launches per iteration. */
#define IDF_STEP_EMBED_READY 1
#define IDF_STEP_EMBED_NEXT 2
int interdiff_mdm_forward_step_ex(const idf_mdm_weights *w, const float *memctx, float *x, int64_t *ts, int32_t B, int32_t T,
                                  const float *gt, const uint8_t *mask, const float *table, int64_t *state, void *ws,
                                  size_t ws_bytes, int32_t flags, void *stream);
int interdiff_mdm_step_chaining(const idf_mdm_weights *w);
/* Every kernel of the library that issues the f16 MFMA claims its CU for itself (all 160 KiB of LDS, the whole register file: DESIGN.md "exclusive CU");
 * since round 5 each launcher VERIFIES that on the device it launches on -- occupancy query == 1, static + dynamic LDS == 160 KiB, >= 256 registers
 * allocated -- and a kernel that does not pass runs as its fp32-MFMA counterpart instead (same results to rounding; the choice is per device and
 * process-stable, so every route of one process computes the same bits).  This entry forces the check for every such kernel on the current device,
 * writes one text line per kernel into buf (<= cap bytes, NUL-terminated) and returns how many do NOT pass (0 on MI355X), or a negative IDF_E_*. */
int interdiff_exclusive_cu_report(char *buf, int32_t cap);
/* DEBUG, tests only (round 6): from now on every f16-MFMA kernel whose name -- as the report above prints it -- contains one of the comma-separated `patterns` is
 * treated as NOT owning its CU, so its launcher takes the fp32 kernel: the fallbacks can be executed on a device where every claim holds.  NULL / "" clears the
 * list.  Process-wide state (the only such switch in the library; empty unless a test sets it); callers drop captured graphs, which bake the kernel choice in. */
int interdiff_debug_deny_exclusive(const char *patterns);

/* ------------------------------------------------------------------------------------
 * Correction predictor   replaces ObjProjector.sample (model/correction_smpl.py:79-138,
 * eval branch) with BatchNorm folded into the 1x1 convolutions at pack time
 * (interdiff_amd/objprojector.py: pack_objprojector).  arena layout: see that file.
 * obj_angles [T,B,6], obj_trans [T,B,3], markers [T,B,P,3] (P = 67), contact int32 [B,P]
 * -> out [T,B,9].
 * ---------------------------------------------------------------------------------- */
typedef struct {
    int32_t T, past_len, P, n_pre;
    const float *arena;
    int64_t dct_pad;           /* [n_pre][past_len]  DCT with the idx_pad frames folded in */
    int64_t dct;               /* [n_pre][T]                                             */
    int64_t idct;              /* [T][n_pre]                                             */
    int64_t hand_bonus;        /* [P] 0.5 on hand markers                                */
    int64_t layer[12];         /* 3 stacks x 4 layers; each: see pack_objprojector       */
    int32_t cin[12], cout[12];
} idf_objproj;

int interdiff_objprojector_sample(const idf_objproj *op, const float *obj_angles,
                                  const float *obj_trans, const float *markers,
                                  const int32_t *contact, int32_t B, float *out, void *stream);

/* ------------------------------------------------------------------------------------
 * denoised_fn, fused   replaces eval_smpl_short.py:84-130 for one gated step.
 * x0 [B,1,144,T] is updated in place.  hand_pose [T,B,90] (already idx-padded),
 * beta [T,B,10], obj_points [B,P,3], gt [B,1,144,T], markers_idx int32 [67].
 * Optional debug outputs (may be NULL): condition uint8 [B], contact int32 [B,67],
 * distance [B], loss [B].
 * ---------------------------------------------------------------------------------- */
typedef struct {
    const idf_smpl_model *smpl;
    const idf_objproj    *objproj;
    const int32_t *faces;      /* [F][3] */
    const int32_t *adj_ptr, *adj_face, *adj_corner;
    const int32_t *markers_idx;
    int32_t n_markers, n_points, past_len;
    int32_t tune;              /* A/B only, 0 in the product.  Bit 1 (value 2): the hook launches the one-launch predictor after the contact scan; default: the predictor's three
                                  stacks ride in the scan's launch as leading workgroups and a pick kernel follows the labels (csrc/correction.hip correction_impl; same bits) */
    /* Scan order of the exact nearest-vertex scans (all four or none; NULL = identity order: same results, no culling benefit).
     * vorder [V]: scan position -> vertex, a permutation that keeps blocks of 16 consecutive positions spatially compact under any
     * pose (the host uses the Morton order of the rest pose, interdiff_amd/correction.py scan_order); faces_scan [F][3] and
     * markers_scan [n_markers] are `faces` / `markers_idx` mapped to scan positions; adj_pair_scan [nnz][2]: for entry e of the
     * adjacency (adj_ptr / adj_face / adj_corner order) the scan positions (a, b) of the incident face's other two vertices such
     * that the face normal at the vertex is (a - v) x (b - v).  Results (indices included) never depend on the order. */
    const int32_t *vorder, *faces_scan, *markers_scan, *adj_pair_scan;
    const int32_t *adj_pair;   /* nullable [nnz][2]: adj_pair_scan's pairs as ORIGINAL vertex ids (kernels that read vertices from HBM) */
    const int32_t *vrank;      /* nullable [V]: vertex -> scan position, the inverse of vorder (round 6): the contact scan then reads a frame's vertices in their own order (coalesced) and scatters them into LDS; NULL = gather through vorder, same results */
} idf_correction_ctx;

size_t interdiff_correction_workspace_bytes(const idf_correction_ctx *c, int32_t B, int32_t T);
int interdiff_correction(const idf_correction_ctx *c, float *x0, const float *gt,
                         const float *hand_pose, const float *beta, const float *obj_points,
                         int32_t B, int32_t T, float blend_t /* t/1000 */,
                         uint8_t *condition, int32_t *contact, float *distance, float *loss,
                         void *ws, size_t ws_bytes, void *stream);
/* The same hook with its one per-call scalar read on the DEVICE: blend weight = table[state[0]][3] (the sampler's coefficient table
 * {c1, c2, sigma, t/1000} and state {t, ...} of interdiff_posterior_step_dev), no debug outputs.  Every launch parameter is then
 * independent of the timestep, so ONE captured hipGraph of a whole hook step -- denoiser forward, inpaint, this, posterior update --
 * serves all eleven corrected steps of a sample (t <= 500, t % 50 == 0: host-known, eval_smpl_short.py:85). */
int interdiff_correction_dev(const idf_correction_ctx *c, float *x0, const float *gt, const float *hand_pose,
                             const float *beta, const float *obj_points, int32_t B, int32_t T, const float *table,
                             const int64_t *state, void *ws, size_t ws_bytes, void *stream);

/* The nearest-vertex scan of the hook / the metrics on its own (tools.point2point_signed's object->human half, tools.py:45-76, fused
 * with the object transform eval_smpl_short.py:107): verts [T*B][V][3] (frame n = t*B + b), obj_points [B][P][3] canonical, objR [T*B][9]
 * row-major, objT [T*B][3] -> o2h [T*B][P] signed distance (nullable), idx int32 [T*B][P] nearest vertex, lowest index on ties
 * (nullable), stats uint64[IDF_CONTACT_STATS] (nullable) = {16-vertex blocks scored, blocks there are, box tests made} summed over
 * waves, the number of workgroups, and thread 0's clock cycles per phase summed over workgroups {records -> LDS, boxes + markers,
 * its wave's tasks, waiting for the other waves, reductions}. */
#define IDF_CONTACT_STATS 12
size_t interdiff_contact_nn_workspace_bytes(const idf_correction_ctx *c, int32_t B, int32_t T);
int interdiff_contact_nn(const idf_correction_ctx *c, const float *verts, const float *obj_points, const float *objR,
                         const float *objT, int32_t B, int32_t T, float *o2h, int32_t *idx, uint64_t *stats,
                         void *ws, size_t ws_bytes, void *stream);

/* Evaluation metrics (eval_smpl_short.py:24-81) over the T frames handed in (the caller slices the future frames,
 * :279); frame-major inputs: obj_pred/obj_gt [T,B,6] (axis-angle | translation), jtr/jtr_gt [T,B,J,3],
 * body_trans(_gt) [T,B,3], verts [T,B,V,3], obj_points [B,P,3] (canonical).  out6 [6][B] rows: global_mpjpe,
 * local_mpjpe, body_translation, obj_translation, obj_rot_error, penetrate. */
size_t interdiff_metrics_workspace_bytes(const idf_correction_ctx *c, int32_t B, int32_t T);
int interdiff_metrics(const idf_correction_ctx *c, const float *obj_pred, const float *jtr,
                      const float *body_trans, const float *obj_gt, const float *jtr_gt,
                      const float *body_trans_gt, const float *verts, const float *obj_points,
                      int32_t B, int32_t T, int32_t J, float *out6 /* [6][B] */,
                      void *ws, size_t ws_bytes, void *stream);

/* ------------------------------------------------------------------------------------
 * Physics post-optimisation ("next" row N4)   replaces optimization.py:19-173 (optimize):
 * Adam (lr 1e-3) over the rotation MATRICES of the 52 SMPL-H joints and of the object plus the two
 * translations of every frame, loss = penetration + regularisers + temporal smoothness + static-foot
 * term (calc_loss, :54-121), hand-written backward (no autograd).  B clips of T frames are optimised
 * side by side (frame n = b*T + t); the reference does one clip per call.
 *
 * Parameter row of a frame, IDF_OPT_NP floats: R[52][9] (global, 21 body, 30 hand joints; row-major
 * matrices) | body translation [3] | object translation [3] | object rotation [9].
 * Every pointer of idf_opt_state is DEVICE memory owned by the caller (interdiff_amd/optimize.py
 * allocates them as torch tensors): nothing is allocated inside the library.
 * ---------------------------------------------------------------------------------- */
#define IDF_OPT_NP 483
#define IDF_OPT_NLOSS 6      /* per-frame partials: collision, verts_reg, feet, reg (L1), reg_v (smoothness), unused */
typedef struct {
    const idf_correction_ctx *geo;   /* smpl model, faces, vertex->face adjacency (objproj / markers unused) */
    const float   *blendT;           /* [KB][K3P] : transpose of smpl->blend, K3P = 3V rounded up to 1024, zero padded */
    const int32_t *jv_ptr;           /* [J+1]  skinning weights by joint (CSR): entries of joint j are [jv_ptr[j], jv_ptr[j+1]) */
    const int32_t *jv_vtx;           /* [nnz]  vertex of the entry */
    const float   *jv_w;             /* [nnz]  weight of the entry */
    int32_t K3P, _pad;
} idf_opt_ctx;

typedef struct {
    int32_t B, T, P, max_iters;
    /* inputs */
    const float *betas;              /* [N][10] */
    const float *obj_points;         /* [B][P][3] canonical object points */
    /* optimiser state, [N][IDF_OPT_NP] each */
    float *param, *init, *grad, *m, *v, *best;
    /* per-iteration scratch */
    float *pose, *tr;                /* [N][156] axis-angle of param, [N][3] */
    float *verts, *vposed, *verts_gt, *gv;               /* [N][V][3] */
    float *jtr;                      /* [N][J][3] */
    float *pts, *y2x;                /* [N][P][3] */
    float *y2x_signed;               /* [N][P] signed distance of every object point to its nearest vertex */
    int32_t *yidx;                   /* [N][P] that vertex */
    int32_t *near;                   /* [N][V] 1 = some object point within 0.5 m (optimization.py:74-75) */
    float *dvposed;                  /* [N][K3P]  (columns >= 3V stay zero) */
    float *dA;                       /* [N][J][12] */
    float *dfeat;                    /* [K3P/1024][N][KB] split-K partials */
    float *gtr;                      /* [N][3] */
    float *lossf;                    /* [N][IDF_OPT_NLOSS] */
    float *loss;                     /* [B][4] total, collision, reg, reg_v of the last iteration (optimization.py:113-119) */
    float *loss_hist;                /* [max_iters][B][4] */
    float *best_loss;                /* [B] */
    int32_t *flag;                   /* [B] 1 = this iteration is the clip's best so far (after iteration 150, :147) */
    uint8_t *foot_static;            /* [N][2] frame t vs t+1, left / right foot (:47-52) */
    int32_t *foot_cnt;               /* [B][2] */
    int32_t *ctl;                    /* [4]: iteration number ii, Adam steps done, -, - (device side so that steps can be graph-captured) */
    void *smpl_ws; size_t smpl_ws_bytes;     /* interdiff_smpl_workspace_bytes(smpl, N) */
    /* scratch of the culled nearest-neighbour kernels (used when geo->vorder is set and all three are given; NULL = brute force):
     * porder int32 [B][P] (Morton order of every clip's object points), psort float [N][2048][4] (a frame's points in that order),
     * pbox float [N][32][2][4] (boxes of their 64-point patches) */
    int32_t *porder; float *psort, *pbox;
} idf_opt_state;

/* pose [N][156] axis-angle, trans / obj_angles / obj_trans [N][3] (optimization.py:20-32) -> param = init, zeroed moments,
 * verts_gt, static-foot flags; ctl = {first_iter, 0}. */
int interdiff_optimize_init(const idf_opt_ctx *c, const idf_opt_state *st, const float *pose, const float *trans,
                            const float *obj_angles, const float *obj_trans, int32_t first_iter, void *stream);
/* loss + gradient at the current param (grad, loss, lossf are left for inspection); no update. */
int interdiff_optimize_loss_grad(const idf_opt_ctx *c, const idf_opt_state *st, void *stream);
/* one iteration of the loop at optimization.py:139-165: loss_grad, Adam step, best-iterate bookkeeping, ctl advance. */
int interdiff_optimize_step(const idf_opt_ctx *c, const idf_opt_state *st, void *stream);
/* best iterate -> pose [N][156], trans, obj_angles, obj_trans [N][3] (optimization.py:150-172). */
int interdiff_optimize_finish(const idf_opt_ctx *c, const idf_opt_state *st, float *pose, float *trans, float *obj_angles,
                              float *obj_trans, void *stream);
/* HOST-side instance of the device inline that back-propagates through matrix_to_axis_angle + the SMPL Rodrigues for
 * n joints (R [n][9], g_out [n][9] -> g_in [n][9]); lets the CPU test suite check the derivative code without a GPU. */
int interdiff_debug_joint_map_vjp(const float *R, const float *g_out, float *g_in, int32_t n);
/* Diagnostic: n_wg one-wave workgroups each fill 6 KiB of LDS with a pattern and re-read it `spin` times; out[0] counts foreign writes seen,
 * out[4 + 4 k ..] = {workgroup, word, value found, pass} of the first 1000 (tools/lds_sentinel_probe.py; not used by the product path). */
int interdiff_debug_lds_sentinel(uint32_t *out, int32_t n_wg, int32_t spin, void *stream);
/* Diagnostic: the "aggressor" of the co-residency probes (DESIGN.md "exclusive CU"; tools/hook_stage_probe.py, tools/coresidency_repro.hip): `grid` workgroups of 256
 * threads that loop `iters` times over four v_mfma_f32_16x16x32_f16 on register operands and, with_loads != 0, one streaming 16-byte load per lane from
 * src[n_floats] (>= 4096 floats).  Small LDS / register needs on purpose, so that it shares CUs with kernels on other streams.  Never launched by the product path. */
int interdiff_debug_f16_aggressor(const float *src, size_t n_floats, float *sink, int32_t iters, int32_t grid, int32_t with_loads, void *stream);

/* ------------------------------------------------------------------------------------
 * Live per-kernel timing for bench.py's `roofline` block (not on the product path).
 * Between profile_begin and profile_end every kernel launch of the library is preceded by a
 * hipEventRecord on its stream; profile_end synchronises and attributes the time between
 * consecutive events to the kernel kind of the first (IDF_K_*).  ms/count arrays have
 * IDF_K_COUNT entries.
 * ---------------------------------------------------------------------------------- */
enum {
    IDF_K_EMBED = 0, IDF_K_GEMM_QKV, IDF_K_SELF_ATTN, IDF_K_GEMM_OUTPROJ, IDF_K_ROWBLOCK_QAN,
    IDF_K_ROWBLOCK_STD, IDF_K_FFN_FUSED, IDF_K_RESERVED7, IDF_K_GEMM_HEADS, IDF_K_MEM_PREP,
    IDF_K_INPAINT, IDF_K_POSTERIOR, IDF_K_CORR_PREPARE, IDF_K_SMPL_POSE, IDF_K_SMPL_BLEND_SKIN,
    IDF_K_CORR_CONTACT, IDF_K_CORR_REDUCE, IDF_K_OBJPROJ, IDF_K_CORR_BLEND, IDF_K_OTHER,
    IDF_K_COUNT
};
int interdiff_profile_begin(int32_t capacity);
int interdiff_profile_end(double *ms_per_kind, int64_t *count_per_kind);

/* indices into idf_mdm_weights.tune (value 0 = the shipped default) */
enum {
    IDF_TUNE_GEMM_EMBED = 0, IDF_TUNE_GEMM_QKV, IDF_TUNE_GEMM_OUTPROJ, IDF_TUNE_FFN, IDF_TUNE_FFN_MATH,
    IDF_TUNE_GEMM_HEADS, IDF_TUNE_CONTACT, IDF_TUNE_MISC, IDF_TUNE_COUNT
};

#ifdef __cplusplus
}
#endif
#endif /* INTERDIFF_HIP_H */
