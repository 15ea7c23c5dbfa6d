// Contact-frame correction predictor ObjProjector.sample (rows D1, D2 of SURVEY.md §8).
//
// Behaviour restated from model/correction_smpl.py:79-138 (eval branch) and the ST-GCN layer
// model/layers.py:339-345 / sublayers.py:415-419,511-516.  The reference launches ~150 tiny torch
// kernels per call; here ONE 16-wave workgroup per clip keeps every activation ([C<=32][n_pre][68] fp32) in
// the 160 KB LDS of its CU and walks all 12 ST-GCN layers without touching HBM:
//   - the 1x1 convolutions ([positions x cin] . [cin x cout]) and the per-coefficient adjacency product
//     ([channels x 68] . A_t[68 x 68]) run on the fp32 MFMA; the 10x10 temporal mix stays on the VALU,
//   - eval-mode BatchNorm is folded into the 1x1 convolutions on the host (pack_objprojector),
//   - the idx_pad frame repetition (future frames = last past frame) is folded into a
//     [n_pre x past_len] DCT matrix, so only the past markers enter the relative branch,
//   - the IDCT is evaluated only for the node that the contact rule selects.
// Layer block layout in the arena (floats), for a layer with cin/cout channels over `nodes` nodes
// (cinp/coutp = channels rounded up to 16, zero padded):
//   version 0 (stacks 0,1):  Tm[n_pre][n_pre] (+12 pad)
//   version 2 (stack 2):     Tm[nodes][n_pre][n_pre], AT[n_pre][80][80]  (A transposed: [t][w][v], zero padded)
//   then Wt[coutp][cinp], bt[coutp], Wr[coutp][cinp], br[coutp], prelu[1]
// (Device code in a header since round 6: the correction hook runs PART 1 as sixteen leading workgroups of its contact-scan launch, csrc/correction.hip.)
#pragma once
#include "common.h"

namespace idf_objproj_dev {


constexpr int NP = 10;                     // n_pre (DCT coefficients)
constexpr int MAXN = 68;                   // nodes: 67 markers + the object itself
constexpr int VP = 80;                     // nodes padded to 5 MFMA tiles (adjacency operand only)
constexpr int CH = 9;
constexpr int PLANE = NP * MAXN;           // one channel of the big buffers
constexpr int POOL_CH = 48;                // max(cin + cout) over the 9->32->16->32->9 stacks
constexpr int NTHR = 1024, NWAVE = NTHR / 64;

__device__ __forceinline__ int pad16(int x) { return (x + 15) & ~15; }
__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

struct LayerP {
    const float *Tm, *AT, *Wt, *bt, *Wr, *br;
    float prelu;
};

// arena block of one layer (pack_objprojector): Tm | (AT) | Wt[coutp][cinp] | bt[coutp] | Wr[coutp][cinp] | br[coutp] | prelu
__device__ __forceinline__ LayerP layer_params(const float *blk, int cin, int cout, int nodes, bool v2) {
    const int cinp = pad16(cin), coutp = pad16(cout);
    LayerP p;
    p.Tm = blk;
    blk += v2 ? nodes * NP * NP : NP * NP + 12;         // shared 10x10 block is padded to 112 floats
    p.AT = v2 ? blk : nullptr;
    if (v2) blk += NP * VP * VP;
    p.Wt = blk; blk += coutp * cinp;
    p.bt = blk; blk += coutp;
    p.Wr = blk; blk += coutp * cinp;
    p.br = blk; blk += coutp;
    p.prelu = blk[0];
    return p;
}

// 1x1 convolution over channel-major planes on the fp32 MFMA:
//   out[o][pos] = bias[o] + sum_c W[o][c] in[c][pos]      (ACC: += what is there, then PReLU)
// M = positions (16 per tile), N = output channels, K = input channels (zero-padded weights).
template <bool ACC>
__device__ __forceinline__ void conv1x1(const float *in, float *out, const float *W, const float *bias, int cin, int cout,
                                        int npos, float slope) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, kq = lane >> 4;
    const int cinp = pad16(cin), NT = pad16(cout) >> 4, MT = (npos + 15) >> 4;
    for (int item = wave; item < MT * NT; item += NWAVE) {
        const int mt = item / NT, nt = item - mt * NT;
        const int pos = min(mt * 16 + li, npos - 1);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int c0 = 4 * kq; c0 < cinp; c0 += 16) {
            const float4 w = ld4(W + (nt * 16 + li) * cinp + c0);
            const float a0 = c0 + 0 < cin ? in[(c0 + 0) * npos + pos] : 0.f;
            const float a1 = c0 + 1 < cin ? in[(c0 + 1) * npos + pos] : 0.f;
            const float a2 = c0 + 2 < cin ? in[(c0 + 2) * npos + pos] : 0.f;
            const float a3 = c0 + 3 < cin ? in[(c0 + 3) * npos + pos] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, w.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, w.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, w.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, w.w, acc, 0, 0, 0);
        }
        const int o = nt * 16 + li;
        if (o < cout) {
            const float bv = bias[o];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = mt * 16 + kq * 4 + r;
                if (q < npos) {
                    float v = acc[r] + bv;
                    if (ACC) {
                        v += out[o * npos + q];
                        v = v >= 0.f ? v : slope * v;
                    }
                    out[o * npos + q] = v;
                }
            }
        }
    }
}

// one ST-GCN layer on channel-major planes [c][k][node] with row stride `nodes`
__device__ inline void st_gcn_layer(float *in, float *out, const LayerP &p, int cin, int cout, int nodes, bool v2) {
    const int npos = NP * nodes, stride = npos;
    conv1x1<false>(in, out, p.Wr, p.br, cin, cout, npos, 0.f);      // residual branch (BN folded)
    __syncthreads();
    // temporal mixing, in place: y[q] = sum_t x[t] Tm[(v)][t][q]
    for (int i = threadIdx.x; i < cin * nodes; i += NTHR) {
        const int c = i / nodes, v = i - c * nodes;
        float *col = in + c * stride + v;
        const float *Tm = p.Tm + (v2 ? v * NP * NP : 0);
        float x[NP], y[NP];
#pragma unroll
        for (int t = 0; t < NP; ++t) { x[t] = col[t * nodes]; y[t] = 0.f; }
#pragma unroll
        for (int t = 0; t < NP; ++t)
#pragma unroll
            for (int q = 0; q < NP; q += 2) {
                const float2 tm = *reinterpret_cast<const float2 *>(Tm + t * NP + q);
                y[q] += x[t] * tm.x;
                y[q + 1] += x[t] * tm.y;
            }
#pragma unroll
        for (int q = 0; q < NP; ++q) col[q * nodes] = y[q];
    }
    __syncthreads();
    if (v2) {
        // spatial mixing on the MFMA, in place: per coefficient t, Y[c][w] = sum_v X[c][t][v] A[t][v][w].
        // One wave owns the whole row block (16 channels, one t): it pulls its X fragments into registers first,
        // so writing the result back over the same rows is safe without a barrier.  (nodes % 4 == 0 here.)
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, kq = lane >> 4;
        const int MT = pad16(cin) >> 4;
        for (int item = wave; item < MT * NP; item += NWAVE) {
            const int mt = item / NP, t = item - mt * NP, c = mt * 16 + li;
            float4 a[VP / 16];
#pragma unroll
            for (int s4 = 0; s4 < VP / 16; ++s4) {
                const int v0 = 16 * s4 + 4 * kq;
                a[s4] = (c < cin && v0 < nodes) ? ld4(in + (c * NP + t) * nodes + v0) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            f32x4 acc[VP / 16];
#pragma unroll
            for (int wt = 0; wt < VP / 16; ++wt) acc[wt] = f32x4{0.f, 0.f, 0.f, 0.f};
            const float *At = p.AT + (size_t)t * VP * VP;
#pragma unroll
            for (int s4 = 0; s4 < VP / 16; ++s4) {
                float4 bw[VP / 16];
#pragma unroll
                for (int wt = 0; wt < VP / 16; ++wt) bw[wt] = ld4(At + (wt * 16 + li) * VP + 16 * s4 + 4 * kq);
#pragma unroll
                for (int wt = 0; wt < VP / 16; ++wt) acc[wt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s4].x, bw[wt].x, acc[wt], 0, 0, 0);
#pragma unroll
                for (int wt = 0; wt < VP / 16; ++wt) acc[wt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s4].y, bw[wt].y, acc[wt], 0, 0, 0);
#pragma unroll
                for (int wt = 0; wt < VP / 16; ++wt) acc[wt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s4].z, bw[wt].z, acc[wt], 0, 0, 0);
#pragma unroll
                for (int wt = 0; wt < VP / 16; ++wt) acc[wt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s4].w, bw[wt].w, acc[wt], 0, 0, 0);
            }
#pragma unroll
            for (int wt = 0; wt < VP / 16; ++wt) {
                const int w = wt * 16 + li;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int cc = mt * 16 + kq * 4 + r;
                    if (cc < cin && w < nodes) in[(cc * NP + t) * nodes + w] = acc[wt][r];
                }
            }
        }
        __syncthreads();
    }
    conv1x1<true>(in, out, p.Wt, p.bt, cin, cout, npos, p.prelu);   // tcn (BN folded) + res, PReLU
    __syncthreads();
}

// run one 4-layer stack; input (9 ch) must already sit at pool[0 ..); returns pointer to the 9-ch output
__device__ inline float *run_stack(float *pool, const idf_objproj &op, const float *arena, int stack, int nodes) {
    const int plane = NP * nodes;
    float *start = pool, *end = pool + (POOL_CH - 32) * PLANE;   // 32-channel tensors live at the END
    float *cur = start;
    for (int l = 0; l < 4; ++l) {
        const int li = stack * 4 + l, cin = op.cin[li], cout = op.cout[li];
        float *nxt = (cur == start) ? end : start;
        const LayerP p = layer_params(arena + op.layer[li], cin, cout, nodes, stack == 2);
        st_gcn_layer(cur, nxt, p, cin, cout, nodes, stack == 2);
        cur = nxt;
    }
    (void)plane;
    return cur;
}

// PART 0: the whole predictor in one launch (interdiff_objprojector_sample).  The contact labels enter only at the very end (which node's IDCT is evaluated), so the correction
// hook runs it in two parts (round 6): PART 1 = everything up to the three stacks' output keep[9][n_pre][68], written to `keep_g` [B][9 * n_pre * 68] -- it needs the markers only and
// runs on a side stream BESIDE the contact scan --, PART 2 = node selection + IDCT from `keep_g`, after the scan's labels exist.  Same instructions on the same values: same bits.
// `sm`: OBJPROJ_LDS bytes of LDS, 16-byte aligned; `b`: the clip; all NTHR threads of the workgroup call it together.
template <int PART>
__device__ __forceinline__ void objproj_body(float *sm, const idf_objproj &op, const float *__restrict__ obj_angles,
                                             const float *__restrict__ obj_trans, const float *__restrict__ markers,
                                             const int32_t *__restrict__ contact, int B, int b, float *__restrict__ keep_g, float *__restrict__ out) {
    float *pool = sm;                                   // [POOL_CH][PLANE]
    float *keep = pool + POOL_CH * PLANE;               // [CH][NP][MAXN]: node 0 = object, nodes 1.. = markers
    float *small = keep + CH * PLANE;                   // scratch: [CH][NP] + misc
    __shared__ int pick_s;
    const int tid = threadIdx.x;
    const int T = op.T, past = op.past_len, P = op.P, P1 = P + 1;
    const float *ar = op.arena;
    const float *Dp = ar + op.dct_pad, *Df = ar + op.dct, *Di = ar + op.idct;

    if constexpr (PART != 2) {
    // ---- object DCT coefficients (idx_pad folded): og[c][k], c<9
    for (int i = tid; i < CH * NP; i += NTHR) {
        const int c = i / NP, k = i - c * NP;
        float s = 0.f;
        for (int t = 0; t < past; ++t) {
            const float v = c < 6 ? obj_angles[((size_t)t * B + b) * 6 + c] : obj_trans[((size_t)t * B + b) * 3 + (c - 6)];
            s += Dp[k * past + t] * v;
        }
        small[i] = s;
    }
    __syncthreads();
    // ---- relative branch input rel[c][k][p] -> pool START (stride P) and keep[.][.][1+p] (stride P1)
    for (int i = tid; i < CH * NP * P; i += NTHR) {
        const int c = i / (NP * P), r = i - c * NP * P, k = r / P, p = r - k * P;
        float v = small[c * NP + k];
        if (c >= 6) {
            float s = 0.f;
            for (int t = 0; t < past; ++t) s += Dp[k * past + t] * markers[(((size_t)t * B + b) * P + p) * 3 + (c - 6)];
            v -= s;
        }
        pool[c * NP * P + k * P + p] = v;
        keep[c * NP * P1 + k * P1 + 1 + p] = v;
    }
    __syncthreads();
    {
        float *o = run_stack(pool, op, ar, 0, P);
        // rel' = rel + stack(rel);  multi = [rel'[:6], rel'[6:] + DCT(markers over ALL frames)]
        for (int i = tid; i < CH * NP * P; i += NTHR) {
            const int c = i / (NP * P), r = i - c * NP * P, k = r / P, p = r - k * P;
            float v = keep[c * NP * P1 + k * P1 + 1 + p] + o[c * NP * P + k * P + p];
            if (c >= 6) {
                float s = 0.f;
                for (int t = 0; t < T; ++t) s += Df[k * T + t] * markers[(((size_t)t * B + b) * P + p) * 3 + (c - 6)];
                v += s;
            }
            keep[c * NP * P1 + k * P1 + 1 + p] = v;
        }
        __syncthreads();
    }
    // ---- object-only branch (1 node)
    for (int i = tid; i < CH * NP; i += NTHR) pool[i] = small[i];
    __syncthreads();
    {
        float *o = run_stack(pool, op, ar, 1, 1);
        for (int i = tid; i < CH * NP; i += NTHR) {
            const int c = i / NP, k = i - c * NP;
            keep[c * NP * P1 + k * P1] = small[i] + o[i];
        }
        __syncthreads();
    }
    // ---- joint branch over the 68 nodes
    for (int i = tid; i < CH * NP * P1; i += NTHR) pool[i] = keep[i];
    __syncthreads();
    {
        float *o = run_stack(pool, op, ar, 2, P1);
        for (int i = tid; i < CH * NP * P1; i += NTHR) keep[i] += o[i];
        __syncthreads();
    }
    }
    if constexpr (PART == 1) {
        for (int i = tid; i < CH * NP * P1; i += NTHR) keep_g[(size_t)b * (CH * NP * MAXN) + i] = keep[i];
        return;
    }
    if constexpr (PART == 2) {
        for (int i = tid; i < CH * NP * P1; i += NTHR) keep[i] = keep_g[(size_t)b * (CH * NP * MAXN) + i];
        __syncthreads();
    }
    // ---- node selection (correction_smpl.py:125-136): no contact -> node 0, else 1 + argmax(contact + hand bonus)
    if (tid == 0) {
        const float *bonus = ar + op.hand_bonus;
        long csum = 0;
        float best = -1.f;
        int bi = 0;
        for (int p = 0; p < P; ++p) {
            const int cv = contact[(size_t)b * P + p];
            csum += cv;
            const float sc = (float)cv + bonus[p];
            if (sc > best) { best = sc; bi = p; }
        }
        pick_s = csum > 0 ? 1 + bi : 0;
    }
    __syncthreads();
    const int pick = pick_s;
    for (int i = tid; i < T * CH; i += NTHR) {
        const int t = i / CH, c = i - t * CH;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NP; ++k) s += Di[t * NP + k] * keep[c * NP * P1 + k * P1 + pick];
        out[((size_t)t * B + b) * CH + c] = s;
    }
}

constexpr size_t OBJPROJ_LDS = ((size_t)POOL_CH * PLANE + (size_t)CH * PLANE + 128) * sizeof(float);

}  // namespace idf_objproj_dev
