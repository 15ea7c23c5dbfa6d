// SMPL-H forward kinematics + linear blend skinning for gfx950 (rows B1-B3 of SURVEY.md §8).
//
// Behaviour restated from libsmpl/smplpytorch/pytorch/smpl_layer.py:72-175.  The reference runs
// ~1000 tiny torch kernels per call (Python loops over 52 joints) and materialises the per-vertex
// transforms th_T [N,4,4,6890] (705 MB at N=1600).  Here:
//   kernel 1 (one wave per frame): Rodrigues for all joints, joints = J_t + J_s.beta (the joint
//       regressor is folded into the shape basis on the host), the kinematic chain staged in LDS,
//       rest-pose removal; emits the blend-shape feature row [R_1..R_{J-1} - I | beta | 1] and the
//       per-joint 3x4 transforms A.
//   kernel 2 (32 frames x 64 vertices per workgroup): ONE fp32-MFMA GEMM feat[32,KB] x blend[KB,192]
//       gives v_posed (template, shape and pose blend shapes in one contraction: 95% of the FLOPs),
//       staged through LDS, then the <=S-bone skinning sum, the 3x4 apply and the translation in
//       the epilogue.  Nothing per-vertex except the final vertices ever reaches HBM.
#include "common.h"
#include "rot_math.h"
#include <algorithm>

// phase stamps exist only in tools/smpl_probe.hip (which defines the macro before including this file)
#ifndef IDF_SMPL_STAMP
#define IDF_SMPL_STAMP(i) do { } while (0)
#endif

namespace {

constexpr int MAXJ = 64;

__global__ __launch_bounds__(64) void smpl_pose_kernel(const idf_smpl_model m, const float *__restrict__ pose,
                                                       const float *__restrict__ betas, const float *__restrict__ trans,
                                                       float *__restrict__ feat, float *__restrict__ A,
                                                       float *__restrict__ jtr) {
    __shared__ float Rs[MAXJ * 9], Js[MAXJ * 3], Gs[MAXJ * 12];
    const int64_t n = blockIdx.x;
    const int j = threadIdx.x, J = m.J, nb = m.n_betas, KB = m.KB;
    const float *beta = betas + n * nb;
    if (j < J) {
        rot::rodrigues_smpl(pose + n * 3 * J + 3 * j, Rs + j * 9);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float s = m.jt[j * 3 + c];
            const float *jsr = m.js + (size_t)(j * 3 + c) * nb;
            for (int k = 0; k < nb; ++k) s += jsr[k] * beta[k];
            Js[j * 3 + c] = s;
        }
    }
    float *f = feat + n * KB;
    if (j >= 1 && j < J) {
#pragma unroll
        for (int e = 0; e < 9; ++e) f[(j - 1) * 9 + e] = Rs[j * 9 + e] - ((e == 0 || e == 4 || e == 8) ? 1.0f : 0.0f);
    }
    const int base = 9 * (J - 1);
    for (int k = j; k < KB - base; k += 64) f[base + k] = k < nb ? beta[k] : (k == nb ? 1.0f : 0.0f);
    __syncthreads();
    if (j < 12) {
        const int r = j >> 2, c = j & 3;
        Gs[j] = c < 3 ? Rs[r * 3 + c] : Js[r];
    }
    __syncthreads();
    for (int i = 1; i < J; ++i) {
        const int p = m.parents[i];
        if (j < 12) {
            const int r = j >> 2, c = j & 3;
            const float *gp = Gs + p * 12 + r * 4;
            float v;
            if (c < 3)
                v = gp[0] * Rs[i * 9 + c] + gp[1] * Rs[i * 9 + 3 + c] + gp[2] * Rs[i * 9 + 6 + c];
            else
                v = gp[0] * (Js[i * 3] - Js[p * 3]) + gp[1] * (Js[i * 3 + 1] - Js[p * 3 + 1]) +
                    gp[2] * (Js[i * 3 + 2] - Js[p * 3 + 2]) + gp[3];
            Gs[i * 12 + j] = v;
        }
        __syncthreads();
    }
    if (j < J) {
        const float *g = Gs + j * 12;
        const float jx = Js[j * 3], jy = Js[j * 3 + 1], jz = Js[j * 3 + 2];
        float *a = A + ((size_t)n * J + j) * 12;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            a[r * 4 + 0] = g[r * 4 + 0];
            a[r * 4 + 1] = g[r * 4 + 1];
            a[r * 4 + 2] = g[r * 4 + 2];
            a[r * 4 + 3] = g[r * 4 + 3] - (g[r * 4] * jx + g[r * 4 + 1] * jy + g[r * 4 + 2] * jz);
            jtr[((size_t)n * J + j) * 3 + r] = g[r * 4 + 3] + trans[n * 3 + r];
        }
    }
}

#ifndef IDF_SMPL_FT
#define IDF_SMPL_FT 32
#define IDF_SMPL_SF 16
#endif
constexpr int FT = IDF_SMPL_FT, VT = 64, NTC = 3 * VT;  // frames / vertices / coordinates per workgroup
constexpr int BT = 256;                                 // threads (4 waves x 48 coordinates); two workgroups share a CU (one's skinning phase beside the other's MFMA loop)
constexpr int FM = FT / 16;                             // 16-frame MFMA tiles per wave (each basis fragment feeds FM*4 MFMAs)
constexpr int STS = NTC + 1;                            // stage row stride
constexpr int SF = IDF_SMPL_SF;                         // frames per skinning sub-step (their joint transforms are staged in LDS)
#ifndef IDF_SMPL_TBF                                     // (tools/smpl_probe.hip rebuilds this file with other block shapes)
#define IDF_SMPL_TBF 10
#define IDF_SMPL_TBV 4
#define IDF_SMPL_FRAME_SLOW 1
#endif
constexpr int TBF = IDF_SMPL_TBF, TBV = IDF_SMPL_TBV;   // workgroup order: blocks of TBF frame tiles x TBV vertex tiles (L2 working set ~2.9 MiB)
constexpr bool FRAME_SLOW = IDF_SMPL_FRAME_SLOW;        // consecutive blocks walk the vertex blocks of one frame block (else the frame blocks of one vertex block)

// One workgroup = 32 frames x 64 vertices.  Phase 1: the blend-shape GEMM feat[32,KB] x blend[KB,192] on the fp32 MFMA; the feature
// tile sits in LDS, each wave streams the basis rows of its 48 output coordinates straight from global memory (three register sets,
// two k-groups in flight) and uses every fragment for both 16-frame tiles.  Phase 2: v_posed leaves the accumulators through LDS; for 16 frames at a time the 52 joint transforms of each frame are
// staged in LDS too (one contiguous 39-KiB copy instead of 4 x 3 scattered 16-byte gathers per (frame, vertex) from global memory:
// 2.1 GB -> 0.43 GB per call), every thread keeps its vertex's <= S skinning weights / bones in registers across all 32 frames,
// and the skinned vertices go back through LDS so that a frame's 64 vertices leave as one 768-byte run of 16-byte stores.
template <int NG>                                       // k-groups of 16: KB = 16 NG, a compile-time constant so that the k-loop unrolls
__global__ __launch_bounds__(BT) void smpl_blend_skin_kernel(const idf_smpl_model m, const float *__restrict__ feat,
                                                              const float *__restrict__ A, const float *__restrict__ trans,
                                                              int64_t N, float *__restrict__ verts,
                                                              float *__restrict__ v_posed) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    idf_args_now(m.V, m.J, m.S, m.blend, m.skin_idx, m.skin_w, feat, A, trans, N, verts, v_posed, gridDim.x);      // every argument into SGPRs now (common.h)
    constexpr int KB = 16 * NG, nq = KB / 4;
    const int V = m.V, J = m.J, S = m.S;
    float *As = sm, *stage = sm;                       // the stage image reuses the feature image once the contraction is done
    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, fh = tid >> 8, li = lane & 15, kq = lane >> 4;       // fh = 0 (one frame-tile set per workgroup)
    // Workgroup order, for L2 locality (speed only).  (1) XCD-affine: workgroup id runs on XCD id % 8, each with its own 4-MiB L2,
    // so the logical index is permuted to give one XCD CONSECUTIVE logical workgroups.  (2) The logical index walks blocks of
    // TBF frame tiles x TBV vertex tiles: the ~64 workgroups an XCD runs at a time then share TBV basis slices (4 x 368 KiB) and
    // TBF frame tiles' features + joint transforms (10 x 141 KiB) -- 2.9 MiB, resident.  Consecutive blocks keep the frame block
    // and move to the next vertex block.  Fabric-side reads per call (rocprofv3 FETCH_SIZE, tools/smpl_probe.hip rebuilt with
    // other shapes): 144 MB this way, 203 MB with the vertex block kept instead, 142-210 MB for 17x3 / 25x2 / 10x8 / 13x4 blocks
    // -- and 430-440 us for ALL of them: the 40-MB basis + 7 MB of per-frame operands live in the Infinity Cache and the kernel
    // is bound by its matrix and skinning phases, not by these reads.  (Frame-tile-major order cycled each XCD through all 7 MiB
    // of features + transforms for every vertex tile; round 1 additionally spread each basis slice over all eight L2s: 657 MB.)
    const int nft = (int)((N + FT - 1) / FT), nvt = (V + VT - 1) / VT, nfb = (nft + TBF - 1) / TBF;
    const int nwg = gridDim.x, id = blockIdx.x;
#ifdef IDF_SMPL_XF                                      // tools/smpl_probe.hip only: XCD x owns frame part x % XF, vertex part x / XF; frames fast
    constexpr int XF = IDF_SMPL_XF, XV = 8 / XF;
    const int nfp = (nft + XF - 1) / XF, nvp = (nvt + XV - 1) / XV, xcd = id & 7, l = id >> 3;
    (void)nwg; (void)nfb;
    const int fl = l % nfp, vpl = l / nfp;
    const int ftile = (xcd % XF) * nfp + fl, vtile = (xcd / XF) * nvp + vpl;
    if (vpl >= nvp) return;
#else
    const int xq = nwg >> 3, xr = nwg & 7, xcd = id & 7;
    const int lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (id >> 3);
    const int blk = lid / (TBF * TBV), wi = lid - blk * (TBF * TBV);
    const int nvb = (nvt + TBV - 1) / TBV;
    const int ftile = (FRAME_SLOW ? blk / nvb : blk % nfb) * TBF + wi % TBF, vtile = (FRAME_SLOW ? blk % nvb : blk / nfb) * TBV + wi / TBF;
#endif
    if (ftile >= nft || vtile >= nvt) return;          // ragged edge blocks (workgroup-uniform)
    const int64_t f0 = (int64_t)ftile * FT;
    const int v0 = vtile * VT;
    IDF_SMPL_STAMP(0);

    // feature tile -> LDS by asm DMA (common.h), row-major [FT][KB]: LDS cell p = 64 i + lane of instruction i is 16-byte position
    // p % nq of row p / nq and receives source chunk pos ^ ((row >> 1) & 7) -- the XOR keeps the MFMA A-fragment reads (16 rows x
    // one 16-B chunk) free of bank conflicts although the row stride (KB * 4 bytes, KB % 32 == 0) is a multiple of 128 bytes.
    // All of a wave's instructions are in flight together; rows past the batch end re-read the last frame (never stored).
    {
        const uint32_t as_lds = idf_lds_addr(As);
        const int ncell = FT * nq;
        for (int i = wave; i * 64 < ncell; i += BT / 64) {
            const int p = min(i * 64 + lane, ncell - 1), row = p / nq, pos = p - row * nq;
            idf_dma16_v(feat + (size_t)min(f0 + row, N - 1) * KB + ((pos ^ ((row >> 1) & 7)) << 2), as_lds + (uint32_t)(i * 1024));
        }
    }
    // this thread's vertex (fixed over all frames of the tile): skinning weights and bone offsets into a frame's transform block
    const int vl = tid & 63, vme = v0 + vl;
    float sw[4];
    int so[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {                      // clamped addresses, selects afterwards: eight independent loads, one trip to memory
        const size_t e = (size_t)min(vme, V - 1) * S + min(k, S - 1);
        sw[k] = m.skin_w[e];
        so[k] = m.skin_idx[e] * 12;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bool ok = k < S && vme < V;
        sw[k] = ok ? sw[k] : 0.f;
        so[k] = ok ? so[k] : 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's DMA (and the loads above) have landed ...
    __syncthreads();                                   // ... and so has everybody else's
    IDF_SMPL_STAMP(1);                                 // feature tile in LDS

    // the basis is stored in MFMA fragment order [vertex tile][wave][t][k-group][lane][4] (smpl.py pack_smpl_model; rows past 3V are
    // zero): a wave's load reads 1 KiB contiguous instead of gathering sixteen 64-byte pieces of sixteen rows
    const float *brow[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) brow[t] = m.blend + ((size_t)((vtile * 4 + wave) * 3 + t) * NG * 64 + lane) * 4;
    f32x4 acc[FM][3];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int t = 0; t < 3; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the basis fragments come straight from global memory (each is used by this wave only): three register sets keep the
    // loads of groups g+1 and g+2 in flight behind the 4*FM*3 MFMAs of group g
    // the basis fragments come straight from global memory (each is used by this wave only) through a ring of PD+1 register sets:
    // the loads of groups g+1 .. g+PD are in flight behind the MFMAs of group g.  The loop is fully unrolled (NG is a template
    // parameter): in a rolled loop the compiler's wait-count bookkeeping gives up at the back edge and drains the whole queue
    // (s_waitcnt vmcnt(0)) every iteration, whatever the prefetch depth.
    constexpr int PD = 3;
    float4 bq[PD + 1][3];
    auto ldg = [&](float4 (&dst)[3], int gi) {
#pragma unroll
        for (int t = 0; t < 3; ++t) dst[t] = *reinterpret_cast<const float4 *>(brow[t] + gi * 256);
    };
    auto mac = [&](const float4 (&bc)[3], int gi) {
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int row = (fh * FM + i) * 16 + li;
            const float4 a = *reinterpret_cast<const float4 *>(As + row * KB + (((gi * 4 + kq) ^ ((row >> 1) & 7)) << 2));
#pragma unroll
            for (int t = 0; t < 3; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bc[t].x, acc[i][t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 3; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bc[t].y, acc[i][t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 3; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bc[t].z, acc[i][t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 3; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bc[t].w, acc[i][t], 0, 0, 0);
        }
    };
#pragma unroll
    for (int g = 0; g < PD && g < NG; ++g) ldg(bq[g], g);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (g + PD < NG) ldg(bq[(g + PD) % (PD + 1)], g + PD);
        __builtin_amdgcn_sched_barrier(0);             // keep the prefetch HERE: left alone, the scheduler sinks every load to just before its use
        mac(bq[g % (PD + 1)], g);
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();                                   // every wave is done reading As
    IDF_SMPL_STAMP(2);                                 // blend-shape GEMM
    // rows of the basis past 3V (the last vertex tile) were read from row 0: their columns are never stored
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) stage[((fh * FM + i) * 16 + kq * 4 + r) * STS + (wave * 3 + t) * 16 + li] = acc[i][t][r];

    float *Asub = sm + ((FT * STS + 255) & ~255);      // [SF][J*12] joint transforms of the current 16 frames (1-KiB aligned: DMA target)
    float *ost = Asub + ((SF * J * 12 + 255) & ~255);  // [SF][NTC] skinned vertices of the current frames (Asub rounded up to whole KiB: its DMA copies whole KiB)
    float *trs = ost + SF * NTC;                       // [FT][3] translations of the tile's frames
    const int arow = J * 12;
    if (tid < FT * 3) trs[tid] = trans[min(f0 + tid / 3, N - 1) * 3 + tid % 3];     // consumed after the barriers below
    for (int sub = 0; sub < FT / SF; ++sub) {
        const int64_t fs = f0 + sub * SF;
        if (fs >= N) break;                            // (workgroup-uniform) nothing left to skin: the copy below must not run past A's slack
        __syncthreads();                               // stage complete (first pass) / previous sub-step's Asub and ost consumed
        // A is [N][J][12] contiguous: the sub-tile is one run of SF*J*12 floats = 39 x 1 KiB -> a linear DMA copy.  The workspace
        // carries one sub-tile of slack behind A (interdiff_smpl_workspace_bytes), so a sub-tile that runs past the last frame
        // stays in bounds (those frames are never stored).
        {
            const uint32_t asub_lds = idf_lds_addr(Asub);
            const float *src = idf_uniform_ptr(A + (size_t)fs * arow);
            for (int i = wave; i * 256 < SF * arow; i += BT / 64) idf_dma16_s(src + (size_t)i * 256, (uint32_t)(lane << 4), asub_lds + (uint32_t)(i * 1024));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        IDF_SMPL_STAMP(3 + 3 * sub);                   // joint transforms staged
#pragma unroll
        for (int k = 0; k < SF * VT / BT; ++k) {
            const int f = (tid >> 6) + (BT / 64) * k, fr = sub * SF + f;
            const float px = stage[fr * STS + 3 * vl], py = stage[fr * STS + 3 * vl + 1], pz = stage[fr * STS + 3 * vl + 2];
            float4 t0 = make_float4(0.f, 0.f, 0.f, 0.f), t1 = t0, t2 = t0;
            const float *Af = Asub + f * arow;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (b < S) {
                    const float w = sw[b];
                    const float4 *a = reinterpret_cast<const float4 *>(Af + so[b]);
                    const float4 a0 = a[0], a1 = a[1], a2 = a[2];
                    // explicit fmaf everywhere below: with implicit contraction the unrolled copies of this loop body were compiled
                    // to DIFFERENT fma / mul+add mixes, so a frame's vertices depended (by an ulp) on its position in the batch
                    t0.x = fmaf(w, a0.x, t0.x); t0.y = fmaf(w, a0.y, t0.y); t0.z = fmaf(w, a0.z, t0.z); t0.w = fmaf(w, a0.w, t0.w);
                    t1.x = fmaf(w, a1.x, t1.x); t1.y = fmaf(w, a1.y, t1.y); t1.z = fmaf(w, a1.z, t1.z); t1.w = fmaf(w, a1.w, t1.w);
                    t2.x = fmaf(w, a2.x, t2.x); t2.y = fmaf(w, a2.y, t2.y); t2.z = fmaf(w, a2.z, t2.z); t2.w = fmaf(w, a2.w, t2.w);
                }
            }
            float *o = ost + f * NTC + 3 * vl;
            o[0] = fmaf(t0.x, px, fmaf(t0.y, py, fmaf(t0.z, pz, t0.w))) + trs[fr * 3 + 0];
            o[1] = fmaf(t1.x, px, fmaf(t1.y, py, fmaf(t1.z, pz, t1.w))) + trs[fr * 3 + 1];
            o[2] = fmaf(t2.x, px, fmaf(t2.y, py, fmaf(t2.z, pz, t2.w))) + trs[fr * 3 + 2];
        }
        __syncthreads();
        IDF_SMPL_STAMP(4 + 3 * sub);                   // skinning
        // a frame's 64 vertices are 192 consecutive floats in verts: 16-byte stores (the row start 3*(n*V+v0) is a multiple of 4
        // floats only when n*V*3 is: handle the general case with 4-byte stores at the tile edge ... V*3 = 20670 is even but not
        // a multiple of 4, so rows are only 8-byte aligned: use 8-byte stores)
        const int nvalid = min(VT, V - v0) * 3;
        for (int idx = tid; idx < SF * (NTC / 2); idx += BT) {
            const int f = idx / (NTC / 2), c2 = (idx - f * (NTC / 2)) * 2;
            if (fs + f >= N || c2 >= nvalid) continue;
            const size_t go = ((size_t)(fs + f) * V + v0) * 3 + c2;
            *reinterpret_cast<float2 *>(verts + go) = *reinterpret_cast<const float2 *>(ost + f * NTC + c2);
            if (v_posed) *reinterpret_cast<float2 *>(v_posed + go) = make_float2(stage[(sub * SF + f) * STS + c2], stage[(sub * SF + f) * STS + c2 + 1]);
        }
        IDF_SMPL_STAMP(5 + 3 * sub);                   // stores issued
    }
}

}  // namespace

extern "C" size_t interdiff_smpl_workspace_bytes(const idf_smpl_model *m, int64_t N) {
    if (!m || N < 0) return 0;
    // feature rows [N][KB] | joint transforms [N][J][12] + one 16-frame sub-tile of slack (the skinning phase copies whole sub-tiles)
    return idf_align((size_t)N * m->KB * sizeof(float)) + idf_align(((size_t)N + SF) * m->J * 12 * sizeof(float));
}

extern "C" int interdiff_smpl_forward(const idf_smpl_model *m, const float *pose, const float *betas, const float *trans,
                                      int64_t N, float *verts, float *jtr, float *v_posed, void *ws, size_t ws_bytes,
                                      void *stream) {
    if (!m || !pose || !betas || !trans || !verts || !jtr || !ws || N < 0) return IDF_E_INVAL;
    if (m->J < 1 || m->J > MAXJ || m->KB % 16 != 0 || m->KB < 9 * (m->J - 1) + m->n_betas + 1 || m->S < 1) return IDF_E_INVAL;
    if (ws_bytes < interdiff_smpl_workspace_bytes(m, N)) return IDF_E_NOMEM;
    if (N == 0) return IDF_OK;
    hipStream_t s = idf_stream(stream);
    float *feat = reinterpret_cast<float *>(ws);
    float *A = reinterpret_cast<float *>(reinterpret_cast<char *>(ws) + idf_align((size_t)N * m->KB * sizeof(float)));
    idf_prof_mark(IDF_K_SMPL_POSE, s);
    hipLaunchKernelGGL(smpl_pose_kernel, dim3((unsigned)N), dim3(64), 0, s, *m, pose, betas, trans, feat, A, jtr);
    const size_t lds = std::max((size_t)FT * m->KB, (size_t)((FT * STS + 255) & ~255) + ((SF * m->J * 12 + 255) & ~255) + (size_t)SF * NTC + FT * 3) * sizeof(float);
    // KB % 32: the swizzled feature image; J*12*SF % 256: the joint transforms of 16 frames are whole KiB; <= 4 bones per vertex
    if (lds > 150 * 1024 || m->KB % 32 != 0 || (m->V * 3) % 2 != 0 || m->S > 4) return IDF_E_INVAL;     // <= 4 bones per vertex (SMPL / SMPL-H skinning)
    if (m->KB != 480) return IDF_E_INVAL;              // SMPL-H: 9 * 51 pose + 10 shape + 1 template = 470 -> 480 (the only basis width built)
    static std::atomic<uint64_t> lds_ok{0};
    if (lds > 64 * 1024 && idf_opt_in_lds(reinterpret_cast<const void *>(smpl_blend_skin_kernel<30>), 150 * 1024, lds_ok) != IDF_OK) return IDF_E_LAUNCH;
    idf_prof_mark(IDF_K_SMPL_BLEND_SKIN, s);
#ifdef IDF_SMPL_XF
    const unsigned nwg_launch = (unsigned)(8 * idf_cdiv(idf_cdiv(N, FT), IDF_SMPL_XF) * idf_cdiv(idf_cdiv(m->V, VT), 8 / IDF_SMPL_XF));
#else
    const unsigned nwg_launch = (unsigned)(idf_cdiv(idf_cdiv(N, FT), TBF) * idf_cdiv(idf_cdiv(m->V, VT), TBV) * TBF * TBV);
#endif
    hipLaunchKernelGGL(smpl_blend_skin_kernel<30>, dim3(nwg_launch), dim3(BT), lds, s, *m,
                       feat, A, trans, N, verts, v_posed);
    idf_prof_mark(-1, s);
    IDF_CHECK_LAUNCH();
    return IDF_OK;
}
