// The two ends of a denoising step on the f16 matrix pipe, and both ends of CONSECUTIVE steps in one launch ("step tail", round 4):
//
//     heads:  x0 = LN3(u) . W_out^T + b_out            (model/diffusion_smpl.py:234-236 bodyFinalLinear | objFinalLinear)
//     update: x  <- inpaint, posterior mean, + sigma noise   (gaussian_diffusion.py:252-330,532-536; gemm.h epilogue_post: the same code)
//     embed:  u0 = x_tokens . W_in^T + b_in + temb[t] + pe   (diffusion_smpl.py:226-231 bodyEmbedding | objEmbedding, embed_timestep, PositionalEncoding)
//
// A plain step used to END with the last GEMM (7.0 us: 250 workgroups that each re-read their 32 rows of five slabs for one 32-column tile)
// and the next one to BEGIN with the embedding GEMM (6.1 us), both latency chains of a few kilobytes of work.  The token row a workgroup has just
// updated is everything the next step's embedding needs from it, so ONE workgroup owns 16 token rows across all 144 channels and does
// LN3 -> heads -> update -> embedding of the next step without leaving its CU (512 threads); the next step starts at its QKV projection.  MODE picks the
// part that runs, all from the same code so that every route computes the same bits:
//     0  embed only            (first step of a captured block; eager forward)
//     1  heads -> x0           (interdiff_mdm_forward: the eager route's denoiser call)
//     2  heads + update        (last step of a block)
//     3  heads + update + embedding of the next step
// Arithmetic: split-f16 products like ffn_h2.h (v = hi + lo' 2^-11, three v_mfma_f32_16x16x32_f16 per product, fp32 accumulate).  The heads' A operand
// is a LayerNorm output (range proved by the packer: mdm.py ln_h2_range_ok); the embedding's A operand is the sampler state itself, which has no
// a-priori range, so each token row is divided by the power of two of its largest magnitude before the split and the fp32 result multiplied back
// (the QKV kernel's rule).  Weights: pre-split plane fragments in the order a lane reads them (mdm.py pack_tail_h2):
//     W_out: [9 output tiles][8 K steps][2 planes][64 lanes][8 halves]    lane (li = channel in tile, kq): W_out[16 nt + li][32 s + 8 kq .. + 7]
//     W_in:  [16 output tiles][5 K steps][2 planes][64 lanes][8 halves]   lane (li, kq): W_in[16 nt + li][32 s + 8 kq .. + 7], zero past channel 143
// Token width 144 only (the SMPL tokens of BASELINE configs #2-#4); any other width keeps the fp32 kernels of gemm.h.
// Takes its CU for itself like every kernel that issues the f16 MFMA (ffn_h2.h "exclusive CU").
#pragma once
#include "common.h"
#include "gemm.h"
#include "ffn_h2.h"

namespace idf_tail_h2 {

using idf_ffn_h2::h8;
constexpr int D = IDF_MDM_D, CW = 144, TRW = 16, NTH = 512, NWV = NTH / 64;
constexpr int NTO = CW / 16, KSH = D / 32;            // heads: 9 output tiles, 8 K steps
constexpr int KE = 160, KSE = KE / 32, NTE = D / 16;    // embedding: K = 144 padded to 160 = 5 steps, 16 output tiles
constexpr int OUT_H2_FLOATS = NTO * KSH * 2 * 64 * 4, IN_H2_FLOATS = NTE * KSE * 2 * 64 * 4;      // 36864, 40960
constexpr int RHS = D + 8;                            // row stride (halves) of the LN3 row planes
constexpr int XTS = CW + 4;                           // row stride (floats) of the fp32 token tile
constexpr int XHS = KE + 8;                           // row stride (halves) of the token planes
constexpr int USS = D + 4;                            // row stride (floats) of the u0 staging tile

struct TailArgs {
    // heads
    const float *u_in;            // IDF_FFN_SLICES partial slabs of the last layer's output
    size_t pstride;
    const float *ln_w, *ln_b;     // norm3 of the last layer
    const float *wout, *out_b;
    float *x0;                    // MODE 1: [B][1][144][T]
    idf_gemm::Args post;          // MODE 2, 3: N = 144, M, T, post_x / post_gt / post_mask / post_table / post_state (gemm.h)
    // embedding
    const float *x_tok;           // MODE 0: x [B][144][T]
    const float *win, *in_b, *temb, *pe;
    const int64_t *ts;            // timestep per clip (MODE 3: already the NEXT step's, advanced by layer 0's QKV kernel of this step)
    int n_steps;
    float *u0;                    // [M][256]
    int M, T;
    int plain_ids;                // A/B only (tune[IDF_TUNE_MISC] == 7): workgroup id -> row tile as in round 4 (tile = id); 0 = XCD-affine (below)
};

template <int MODE, bool RAGGED>
__global__ __launch_bounds__(NTH) void step_tail_h2_kernel(const float *__restrict__ u_in, size_t u_pstride, const float *__restrict__ wout, int M, int T, int nwg,
                                                           int plain_ids, const TailArgs a) {
    // (the leading scalars repeat what the heads' first requests need -- slab rows, weight fragments, sizes: they arrive preloaded in SGPRs (build.py), the struct behind
    // them is read from the argument segment while those requests fly)
    constexpr bool HEADS = MODE != 0, EMBED = MODE == 0 || MODE == 3, POST = MODE >= 2;
    asm volatile("" ::: "v255");                         // exclusive CU: 2 waves per SIMD x 256 registers = the register file (+ the dynamic LDS the launcher adds)
    __shared__ __attribute__((aligned(16))) _Float16 rpl[HEADS ? 2 * TRW * RHS : 8];     // [hi | lo'][16][RHS]: LN3 rows
    __shared__ __attribute__((aligned(16))) float xt[EMBED ? TRW * XTS : 4];             // [16][XTS]: the token rows the embedding contracts (fp32)
    __shared__ __attribute__((aligned(16))) _Float16 xpl[EMBED ? 2 * TRW * XHS : 8];     // their planes, each row divided by sc[row]
    __shared__ float sc[TRW];
    __shared__ __attribute__((aligned(16))) float us[EMBED ? TRW * USS : 4];             // u0 tile on its way out
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
    // XCD-affine row tiles (round 5): workgroup id runs on XCD id % 8; giving an XCD CONSECUTIVE tiles puts the rows of clips {2x, 2x + 1} (B = 16, T = 100) on XCD x
    // like the row block, the attention, the feed-forward and the QKV kernels already do -- the five slabs this kernel sums were written by that XCD, and the u0 / x rows
    // it writes are read by that XCD's QKV workgroups next (any order is correct: every workgroup computes the same tile).
    const int wid = blockIdx.x, xq = nwg >> 3, xr = nwg & 7, xcd = wid & 7;
    const int tile = plain_ids ? wid : (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (wid >> 3);
    const int m0 = tile * TRW;
    const int rown = (wave & 3) * 4 + kq;                 // row passes (waves 0..3): one 16-lane group per token row
    const bool rowpass = wave < 4;

    if constexpr (HEADS) {
        // output tiles of the heads GEMM: wave w owns tile w, wave 0 tile 8 as well (the in-kernel noise of the update is ~500 VALU instructions per tile and lane:
        // with four waves of three / two tiles it was the longest thing in the launch)
        constexpr int NTW = 2;
        const int ntw = wave == 0 ? 2 : 1;
        auto tile_of = [&](int j) { return j == 0 ? wave : NTO - 1; };
        Row16Raw<IDF_FFN_SLICES> raw;
        if (rowpass) raw.request(u_in + (size_t)min(m0 + rown, M - 1) * D, li, u_pstride);
        idf_gemm::PostOperands<1, 1> po[NTW];
        if constexpr (POST) {
#pragma unroll
            for (int j = 0; j < NTW; ++j)
                if (j < ntw) idf_gemm::post_prefetch<1, 1, RAGGED>(a.post, po[j], m0 + kq * 4, tile_of(j) * 16 + li);
        }
        float4 wf[NTW][KSH][2];
#pragma unroll
        for (int j = 0; j < NTW; ++j)
            if (j < ntw) {
#pragma unroll
                for (int s = 0; s < KSH; ++s)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) wf[j][s][pl] = idf_gemm::ld4(wout + (size_t)(((tile_of(j) * KSH + s) * 2 + pl) * 64 + lane) * 4);
            }
        if (rowpass) {
            Row16 r;
            raw.reduce(r);
            ln_row16(r, a.ln_w, a.ln_b, li);
            idf_ffn_h2::row16_store_planes(r, rpl + rown * RHS, rpl + (TRW + rown) * RHS, li);
        }
        __syncthreads();
        f32x4 am[NTW], ac[NTW];
#pragma unroll
        for (int j = 0; j < NTW; ++j) am[j] = ac[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // every A fragment first (16 reads in flight together), then one MFMA chain per tile under ONE branch: with the guard inside the K loop every step was its own
        // basic block -- eight serial LDS round trips in front of the matrix pipe.  Same products in the same order per accumulator.
        h8 ah[KSH], al[KSH];
#pragma unroll
        for (int s = 0; s < KSH; ++s) {
            ah[s] = *reinterpret_cast<const h8 *>(rpl + li * RHS + 32 * s + 8 * kq);
            al[s] = *reinterpret_cast<const h8 *>(rpl + (TRW + li) * RHS + 32 * s + 8 * kq);
        }
#pragma unroll
        for (int j = 0; j < NTW; ++j)
            if (j < ntw) {
#pragma unroll
                for (int s = 0; s < KSH; ++s) {
                    IDF_H2_MFMA(am[j], ah[s], __builtin_bit_cast(h8, wf[j][s][0]));
                    IDF_H2_MFMA(ac[j], ah[s], __builtin_bit_cast(h8, wf[j][s][1]));
                    IDF_H2_MFMA(ac[j], al[s], __builtin_bit_cast(h8, wf[j][s][0]));
                }
            }
#pragma unroll
        for (int j = 0; j < NTW; ++j)
            if (j < ntw) {
                const int col = tile_of(j) * 16 + li, rbase = m0 + kq * 4;
                const float bv = a.out_b[col];
                f32x4 v;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = am[j][q] + ac[j][q] * idf_ffn_h2::LO_UNSCALE;
                if constexpr (POST) {
                    const f32x4 accs[1][1] = {{v}};
                    const float bvs[1] = {bv};
                    idf_gemm::epilogue_post<1, 1, RAGGED>(a.post, accs, bvs, po[j], rbase, col, EMBED ? xt + (kq * 4) * XTS + col : nullptr, XTS);
                } else {                                  // x0[b][col][t]: the lane's four rows are four consecutive frames of one clip when T % 4 == 0
                    if (!RAGGED && rbase + 3 < M) {
                        const int b = rbase / T, t = rbase - b * T;
                        *reinterpret_cast<float4 *>(a.x0 + ((size_t)b * CW + col) * T + t) = make_float4(v[0] + bv, v[1] + bv, v[2] + bv, v[3] + bv);
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int row = rbase + q;
                            if (row < M) {
                                const int bb = row / T;
                                a.x0[((size_t)bb * CW + col) * T + (row - bb * T)] = v[q] + bv;
                            }
                        }
                    }
                }
            }
    }

    if constexpr (EMBED) {
        if constexpr (MODE == 0) {                        // token rows from x [B][144][T]
            if constexpr (!RAGGED) {
#pragma unroll
                for (int it = 0; it < (CW * 4 + NTH - 1) / NTH; ++it) {
                    const int idx = tid + it * NTH, c = min(idx >> 2, CW - 1), rg = idx & 3;
                    const int rowb = min(m0 + 4 * rg, M - 4), b = rowb / T, t = rowb - b * T;      // (T % 4 == 0: M % 4 == 0, a group of four rows stays inside one clip)
                    const float4 v = idf_gemm::ld4(a.x_tok + ((size_t)b * CW + c) * T + t);
                    if (idx < CW * 4) {
                        float *d = xt + (4 * rg) * XTS + c;
                        d[0] = v.x; d[XTS] = v.y; d[2 * XTS] = v.z; d[3 * XTS] = v.w;
                    }
                }
            } else {
#pragma unroll
                for (int it = 0; it < (CW * TRW + NTH - 1) / NTH; ++it) {
                    const int idx = tid + it * NTH, c = min(idx >> 4, CW - 1), rl = idx & 15, row = min(m0 + rl, M - 1), b = row / T;
                    const float v = a.x_tok[((size_t)b * CW + c) * T + (row - b * T)];
                    if (idx < CW * TRW) xt[rl * XTS + c] = v;
                }
            }
        }
        // the embedding's weight fragments (wave w owns output tiles 2w, 2w+1) and the row-major addends of the store pass: requested now, used after two barriers
        constexpr int NTEW = NTE / NWV;
        float4 ef[NTEW][KSE][2];
#pragma unroll
        for (int j = 0; j < NTEW; ++j)
#pragma unroll
            for (int s = 0; s < KSE; ++s)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) ef[j][s][pl] = idf_gemm::ld4(a.win + (size_t)((((wave * NTEW + j) * KSE + s) * 2 + pl) * 64 + lane) * 4);
        constexpr int NST = TRW * (D / 4) / NTH;          // float4 stores per thread: 2
        float4 eb[NST], et[NST], ep[NST];
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            const int idx = tid + it * NTH, rl = idx >> 6, c4 = (idx & 63) << 2, row = min(m0 + rl, M - 1), b = row / T, t = row - b * T;
            int64_t step = a.ts[b];
            step = step < 0 ? 0 : (step >= a.n_steps ? a.n_steps - 1 : step);
            eb[it] = idf_gemm::ld4(a.in_b + c4);
            et[it] = idf_gemm::ld4(a.temb + (size_t)step * D + c4);
            ep[it] = idf_gemm::ld4(a.pe + (size_t)t * D + c4);
        }
        __syncthreads();                                 // token tile complete (MODE 3: written by the update above)
        if (rowpass) {   // power-of-two row scale (2^-e, e = exponent of the row's largest magnitude; an all-zero row keeps 1) and split: lane li owns the 4-float chunks li, 16 + li, 32 + li (< 36)
            float4 c[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) c[k] = li + 16 * k < CW / 4 ? *reinterpret_cast<const float4 *>(xt + rown * XTS + 4 * (li + 16 * k)) : make_float4(0.f, 0.f, 0.f, 0.f);
            float amax = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) amax = fmaxf(amax, fmaxf(fmaxf(__builtin_fabsf(c[k].x), __builtin_fabsf(c[k].y)), fmaxf(__builtin_fabsf(c[k].z), __builtin_fabsf(c[k].w))));
            amax = row16_max(amax);
            const int e = (amax > 0.f && amax < INFINITY) ? (int)((__builtin_bit_cast(uint32_t, amax) >> 23) & 0xff) - 126 : 0;
            const float dn = __builtin_bit_cast(float, (uint32_t)((127 - e) << 23)), up = __builtin_bit_cast(float, (uint32_t)((127 + e) << 23));
            _Float16 *xh = xpl + rown * XHS, *xl = xpl + (TRW + rown) * XHS;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int ch = li + 16 * k;
                if (ch < KE / 4) {                        // chunks 36 .. 39 (channels 144 .. 159) are the K padding: zero
                    uint2 h, l;
                    idf_ffn_h2::split4_pk(make_float4(c[k].x * dn, c[k].y * dn, c[k].z * dn, c[k].w * dn), h, l);
                    *reinterpret_cast<uint2 *>(xh + 4 * ch) = h;
                    *reinterpret_cast<uint2 *>(xl + 4 * ch) = l;
                }
            }
            if (li == 0) sc[rown] = up;
        }
        __syncthreads();
        f32x4 em[NTEW], ec[NTEW];
#pragma unroll
        for (int j = 0; j < NTEW; ++j) em[j] = ec[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KSE; ++s) {
            const h8 ah = *reinterpret_cast<const h8 *>(xpl + li * XHS + 32 * s + 8 * kq), al = *reinterpret_cast<const h8 *>(xpl + (TRW + li) * XHS + 32 * s + 8 * kq);
#pragma unroll
            for (int j = 0; j < NTEW; ++j) {
                IDF_H2_MFMA(em[j], ah, __builtin_bit_cast(h8, ef[j][s][0]));
                IDF_H2_MFMA(ec[j], ah, __builtin_bit_cast(h8, ef[j][s][1]));
                IDF_H2_MFMA(ec[j], al, __builtin_bit_cast(h8, ef[j][s][0]));
            }
        }
#pragma unroll
        for (int j = 0; j < NTEW; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) us[(kq * 4 + q) * USS + (wave * NTEW + j) * 16 + li] = (em[j][q] + ec[j][q] * idf_ffn_h2::LO_UNSCALE) * sc[kq * 4 + q];
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            const int idx = tid + it * NTH, rl = idx >> 6, c4 = (idx & 63) << 2;
            float4 v = *reinterpret_cast<const float4 *>(us + rl * USS + c4);
            v.x += eb[it].x; v.y += eb[it].y; v.z += eb[it].z; v.w += eb[it].w;
            v.x += et[it].x + ep[it].x; v.y += et[it].y + ep[it].y; v.z += et[it].z + ep[it].z; v.w += et[it].w + ep[it].w;
            if (m0 + rl < M) idf_store16_wt(a.u0 + (size_t)(m0 + rl) * D + c4, v);      // read next by the QKV kernel's workgroups on other XCDs: write through
        }
    }
}

// launch with the dynamic LDS that tops the kernel's static LDS up to the CU's 160 KiB (exclusive CU); per (kernel, device) opt-in
// (ta == nullptr: only the exclusive-CU verdict of this instantiation, common.h idf_exclusive_cu: dynamic LDS >= 0, or -1)
template <int MODE, bool RAGGED>
inline int launch_tail_one(hipStream_t s, const TailArgs *ta) {
    static idf_excl_cache excl;
    static const char *const names[4] = {"step_tail_h2_kernel<embed>", "step_tail_h2_kernel<heads>", "step_tail_h2_kernel<heads+update>", "step_tail_h2_kernel<heads+update+embed>"};
    const int dyn = idf_exclusive_cu(reinterpret_cast<const void *>(&step_tail_h2_kernel<MODE, RAGGED>), names[MODE], NTH, excl);
    if (!ta) return dyn;
    if (dyn < 0) return IDF_NOT_EXCLUSIVE;
    hipLaunchKernelGGL((step_tail_h2_kernel<MODE, RAGGED>), dim3((unsigned)idf_cdiv(ta->M, TRW)), dim3(NTH), (size_t)dyn, s, ta->u_in, ta->pstride, ta->wout, ta->M, ta->T,
                       (int)idf_cdiv(ta->M, TRW), ta->plain_ids, *ta);
    return IDF_OK;
}
inline int launch_tail_sel(hipStream_t s, int mode, bool ragged, const TailArgs *ta) {
    switch (mode * 2 + (ragged ? 1 : 0)) {
    case 0: return launch_tail_one<0, false>(s, ta);
    case 1: return launch_tail_one<0, true>(s, ta);
    case 2: return launch_tail_one<1, false>(s, ta);
    case 3: return launch_tail_one<1, true>(s, ta);
    case 4: return launch_tail_one<2, false>(s, ta);
    case 5: return launch_tail_one<2, true>(s, ta);
    case 6: return launch_tail_one<3, false>(s, ta);
    default: return launch_tail_one<3, true>(s, ta);
    }
}
inline int launch_tail(hipStream_t s, int mode, const TailArgs &ta) { return launch_tail_sel(s, mode, (ta.T & 3) != 0, &ta); }
// The step tail is usable on this device only if ALL FOUR parts of its raggedness class get their CU (a step's two ends, and the chained / unchained forms of
// consecutive steps, must take the same arithmetic): decided once, up front, by the forward (csrc/denoiser.hip) and by interdiff_mdm_step_chaining.
inline bool tail_exclusive_ok(bool ragged) {
    bool ok = true;
    for (int mode = 0; mode < 4; ++mode) ok = (launch_tail_sel(nullptr, mode, ragged, nullptr) >= 0) && ok;
    return ok;
}

}  // namespace idf_tail_h2
