// Contact-frame correction predictor ObjProjector.sample (rows D1, D2 of SURVEY.md §8).
//
// Behaviour restated from model/correction_smpl.py:79-138 (eval branch) and the ST-GCN layer
// model/layers.py:339-345 / sublayers.py:415-419,511-516.  The reference launches ~150 tiny torch
// kernels per call; here ONE workgroup per clip keeps every activation ([C<=32][n_pre][68] fp32) in
// the 160 KB LDS of its CU and walks all 12 ST-GCN layers without touching HBM:
//   - eval-mode BatchNorm is folded into the 1x1 convolutions on the host (pack_objprojector),
//   - the idx_pad frame repetition (future frames = last past frame) is folded into a
//     [n_pre x past_len] DCT matrix, so only the past markers enter the relative branch,
//   - the IDCT is evaluated only for the node that the contact rule selects.
// Layer block layout in the arena (floats), for a layer with cin/cout channels over `nodes` nodes:
//   version 0 (stacks 0,1):  Tm[n_pre][n_pre]
//   version 2 (stack 2):     Tm[nodes][n_pre][n_pre], Am[n_pre][nodes][nodes]
//   then Wt[cout][cin], bt[cout], Wr[cout][cin], br[cout], prelu[1]
#include "common.h"

namespace {

constexpr int NP = 10;                     // n_pre (DCT coefficients)
constexpr int MAXN = 68;                   // nodes: 67 markers + the object itself
constexpr int CH = 9;
constexpr int PLANE = NP * MAXN;           // one channel of the big buffers
constexpr int POOL_CH = 48;                // max(cin + cout) over the 9->32->16->32->9 stacks

struct LayerP {
    const float *Tm, *Am, *Wt, *bt, *Wr, *br;
    float prelu;
};

__device__ __forceinline__ LayerP layer_params(const float *blk, int cin, int cout, int nodes, bool v2) {
    LayerP p;
    p.Tm = blk;
    blk += v2 ? nodes * NP * NP : NP * NP;
    p.Am = v2 ? blk : nullptr;
    if (v2) blk += NP * nodes * nodes;
    p.Wt = blk; blk += cout * cin;
    p.bt = blk; blk += cout;
    p.Wr = blk; blk += cout * cin;
    p.br = blk; blk += cout;
    p.prelu = blk[0];
    return p;
}

// out[o][pos] = bias[o] + sum_c W[o][c] in[c][pos]  (ACC: add to what is there, then PReLU)
template <bool ACC>
__device__ __forceinline__ void conv1x1(const float *in, float *out, const float *W, const float *bias, int cin, int cout,
                                        int npos, int in_stride, int out_stride, float slope) {
    for (int pos = threadIdx.x; pos < npos; pos += 256) {
        float x[32];
#pragma unroll
        for (int c = 0; c < 32; ++c) x[c] = c < cin ? in[c * in_stride + pos] : 0.f;
        for (int o = 0; o < cout; ++o) {
            float s = bias[o];
            const float *w = W + o * cin;
#pragma unroll
            for (int c = 0; c < 32; ++c)
                if (c < cin) s += w[c] * x[c];
            if (ACC) {
                s += out[o * out_stride + pos];
                s = s >= 0.f ? s : slope * s;
            }
            out[o * out_stride + pos] = s;
        }
    }
}

// one ST-GCN layer on channel-major planes [c][k][node] with row stride `nodes`
__device__ void st_gcn_layer(float *in, float *out, const LayerP &p, int cin, int cout, int nodes, bool v2) {
    const int npos = NP * nodes, stride = npos;
    conv1x1<false>(in, out, p.Wr, p.br, cin, cout, npos, stride, stride, 0.f);      // residual branch (BN folded)
    __syncthreads();
    // temporal mixing, in place: y[q] = sum_t x[t] Tm[(v)][t][q]
    for (int i = threadIdx.x; i < cin * nodes; i += 256) {
        const int c = i / nodes, v = i - c * nodes;
        float *col = in + c * stride + v;
        const float *Tm = p.Tm + (v2 ? v * NP * NP : 0);
        float x[NP], y[NP];
#pragma unroll
        for (int t = 0; t < NP; ++t) { x[t] = col[t * nodes]; y[t] = 0.f; }
#pragma unroll
        for (int t = 0; t < NP; ++t)
#pragma unroll
            for (int q = 0; q < NP; ++q) y[q] += x[t] * Tm[t * NP + q];
#pragma unroll
        for (int q = 0; q < NP; ++q) col[q * nodes] = y[q];
    }
    __syncthreads();
    if (v2) {   // spatial mixing, in place per (c,t) row: y[w] = sum_v x[v] A[t][v][w]
        for (int i = threadIdx.x; i < cin * NP; i += 256) {
            const int t = i / cin, c = i - t * cin;                 // c fastest: a wave shares A[t] (broadcast loads)
            float *row = in + c * stride + t * nodes;
            const float *At = p.Am + (size_t)t * nodes * nodes;
            float x[MAXN];
#pragma unroll
            for (int v = 0; v < MAXN; ++v) x[v] = v < nodes ? row[v] : 0.f;
            for (int w = 0; w < nodes; ++w) {
                float s = 0.f;
#pragma unroll
                for (int v = 0; v < MAXN; ++v)
                    if (v < nodes) s += x[v] * At[v * nodes + w];
                row[w] = s;
            }
        }
        __syncthreads();
    }
    conv1x1<true>(in, out, p.Wt, p.bt, cin, cout, npos, stride, stride, p.prelu);   // tcn (BN folded) + res, PReLU
    __syncthreads();
}

// run one 4-layer stack; input (9 ch) must already sit at pool[0 ..); returns pointer to the 9-ch output
__device__ float *run_stack(float *pool, const idf_objproj &op, const float *arena, int stack, int nodes) {
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

__global__ __launch_bounds__(256) void objproj_kernel(const idf_objproj op, const float *__restrict__ obj_angles,
                                                      const float *__restrict__ obj_trans, const float *__restrict__ markers,
                                                      const int32_t *__restrict__ contact, int B, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *pool = sm;                                   // [POOL_CH][PLANE]
    float *keep = pool + POOL_CH * PLANE;               // [CH][NP][MAXN]: node 0 = object, nodes 1.. = markers
    float *small = keep + CH * PLANE;                   // scratch: [CH][NP] + misc
    __shared__ int pick_s;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int T = op.T, past = op.past_len, P = op.P, P1 = P + 1;
    const float *ar = op.arena;
    const float *Dp = ar + op.dct_pad, *Df = ar + op.dct, *Di = ar + op.idct;

    // ---- object DCT coefficients (idx_pad folded): og[c][k], c<9
    for (int i = tid; i < CH * NP; i += 256) {
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
    for (int i = tid; i < CH * NP * P; i += 256) {
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
        for (int i = tid; i < CH * NP * P; i += 256) {
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
    for (int i = tid; i < CH * NP; i += 256) pool[i] = small[i];
    __syncthreads();
    {
        float *o = run_stack(pool, op, ar, 1, 1);
        for (int i = tid; i < CH * NP; i += 256) {
            const int c = i / NP, k = i - c * NP;
            keep[c * NP * P1 + k * P1] = small[i] + o[i];
        }
        __syncthreads();
    }
    // ---- joint branch over the 68 nodes
    for (int i = tid; i < CH * NP * P1; i += 256) pool[i] = keep[i];
    __syncthreads();
    {
        float *o = run_stack(pool, op, ar, 2, P1);
        for (int i = tid; i < CH * NP * P1; i += 256) keep[i] += o[i];
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
    for (int i = tid; i < T * CH; i += 256) {
        const int t = i / CH, c = i - t * CH;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NP; ++k) s += Di[t * NP + k] * keep[c * NP * P1 + k * P1 + pick];
        out[((size_t)t * B + b) * CH + c] = s;
    }
}

}  // namespace

extern "C" int interdiff_objprojector_sample(const idf_objproj *op, const float *obj_angles, const float *obj_trans,
                                             const float *markers, const int32_t *contact, int32_t B, float *out,
                                             void *stream) {
    if (!op || !obj_angles || !obj_trans || !markers || !contact || !out || B <= 0) return IDF_E_INVAL;
    if (op->n_pre != NP || op->P + 1 != MAXN || op->past_len < 1 || op->past_len > op->T) return IDF_E_INVAL;
    for (int l = 0; l < 12; ++l)
        if (op->cin[l] > 32 || op->cout[l] > 32 || op->cin[l] + op->cout[l] > POOL_CH) return IDF_E_INVAL;
    const size_t lds = ((size_t)POOL_CH * PLANE + (size_t)CH * PLANE + 128) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(objproj_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return IDF_E_LAUNCH;
        attr_set = true;
    }
    idf_prof_mark(IDF_K_OBJPROJ, idf_stream(stream));
    hipLaunchKernelGGL(objproj_kernel, dim3(B), dim3(256), lds, idf_stream(stream), *op, obj_angles, obj_trans, markers, contact, B, out);
    idf_prof_mark(-1, idf_stream(stream));
    IDF_CHECK_LAUNCH();
    return IDF_OK;
}
