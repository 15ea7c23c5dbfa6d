// The physics-informed correction hook `denoised_fn` (row C1 of SURVEY.md §8), fused.
//
// Behaviour restated from eval_smpl_short.py:84-130.  The reference builds the object point cloud
// [T,B,2048,3], all vertex normals (3 scatter-adds), both nearest-neighbour directions, and TWO
// [T,B,2048,67] marker-distance tensors (2 x 878 MB at B=16,T=100, via 2.6 GB temporaries).  Only
// the object->human signed distance, the per-frame minimum marker distance and the per-marker
// contact flag are ever consumed, so here one workgroup per frame
//   - parks the frame's 6890 vertices in LDS (110 KB of the CU's 160 KB),
//   - transforms its share of the canonical object points on the fly (never stored),
//   - scans the vertices for the exact nearest neighbour (same arithmetic as geometry.hip),
//   - evaluates the vertex normal ONLY at the nearest vertices, from LDS, through the adjacency,
//   - reduces loss / min-distance / contact flags in LDS and writes a few floats per frame.
// Frame index n = t*B + b everywhere (the reference's .view(T*B, ...)).
#include "common.h"
#include "rot_math.h"
#include <float.h>

namespace {

constexpr int NBODY = 22;              // body joints predicted as rot6d (smpl_dim = 132)
constexpr int CTOK = 144;

// ---- K1: tokens -> SMPL pose (axis-angle), translation, object rotation/translation -----------------
// one thread per (frame, slot): slots 0..21 body joints, 22 = hands + trans copy, 23 = object
__global__ __launch_bounds__(256) void corr_prepare_kernel(const float *__restrict__ x0, const float *__restrict__ gt,
                                                           const float *__restrict__ hand_pose, int B, int T,
                                                           float *__restrict__ pose, float *__restrict__ trans,
                                                           float *__restrict__ objR, float *__restrict__ objT,
                                                           float *__restrict__ gt_angles, float *__restrict__ gt_trans) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t N = (int64_t)B * T;
    if (i >= N * 24) return;
    const int64_t n = i / 24;
    const int slot = (int)(i - n * 24), t = (int)(n / B), b = (int)(n - (int64_t)t * B);
    const float *xb = x0 + (size_t)b * CTOK * T + t;                   // channel c at xb[c*T]
    if (slot < NBODY) {
        float d[6], m[9], a[3];
#pragma unroll
        for (int k = 0; k < 6; ++k) d[k] = xb[(size_t)(slot * 6 + k) * T];
        rot::rot6d_to_matrix(d, m);
        rot::matrix_to_axis_angle(m, a);
        float *p = pose + n * 156 + slot * 3;
        p[0] = a[0]; p[1] = a[1]; p[2] = a[2];
    } else if (slot == 22) {
        for (int k = 0; k < 90; ++k) pose[n * 156 + 66 + k] = hand_pose[n * 90 + k];
#pragma unroll
        for (int k = 0; k < 3; ++k) trans[n * 3 + k] = xb[(size_t)(132 + k) * T];
    } else {
        float d[6], m[9];
#pragma unroll
        for (int k = 0; k < 6; ++k) d[k] = xb[(size_t)(135 + k) * T];
        rot::rot6d_to_matrix(d, m);
#pragma unroll
        for (int k = 0; k < 9; ++k) objR[n * 9 + k] = m[k];
        const float *gb = gt + (size_t)b * CTOK * T + t;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            objT[n * 3 + k] = xb[(size_t)(141 + k) * T];
            gt_trans[n * 3 + k] = gb[(size_t)(141 + k) * T];
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) gt_angles[n * 6 + k] = gb[(size_t)(135 + k) * T];
    }
}

__device__ __forceinline__ float dist2_exact(float qx, float qy, float qz, float rx, float ry, float rz) {
#pragma clang fp contract(off)
    const float dx = qx - rx, dy = qy - ry, dz = qz - rz;
    const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
    return (xx + yy) + zz;
}

__device__ __forceinline__ float3 f3(const float4 v) { return make_float3(v.x, v.y, v.z); }
__device__ __forceinline__ float3 sub3(float3 a, float3 b) { return make_float3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ float3 cross3(float3 a, float3 b) {
    return make_float3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

// ---- K4: per-frame contact analysis -------------------------------------------------------------------
// One workgroup (CT threads) per frame; every thread owns QP object points as QP/2 PACKED pairs: the exact-arithmetic
// distance (dx*dx + dy*dy) + dz*dz of a pair against the LDS-broadcast vertex is 3 v_pk_add + 3 v_pk_mul + 2 v_pk_add
// (no FMA contraction, so the argmin is bit-identical to geometry.hip and to the oracle).
// The scan itself only keeps the running MINIMUM (v_min3_f32 over vertex pairs); which vertex it was is settled per block of
// VB vertices -- one compare + two selects per point per BLOCK instead of per vertex -- and resolved afterwards by re-scoring
// the one winning block: the first vertex of the first block whose distance equals the minimum, i.e. exactly the
// lowest-index-wins rule of the brute force.  11 -> ~8.6 VALU lane-ops per (point, vertex) pair.
constexpr int MAXP = 2048;             // object points per frame handled by one workgroup (CT * QP)
constexpr int MAXM = 128;
constexpr int VB = 8;                  // vertices per index-bookkeeping block
typedef float v2f __attribute__((ext_vector_type(2)));

template <int CT, int QP>
__global__ __launch_bounds__(CT) void corr_contact_kernel(const float *__restrict__ verts, int V,
                                                          const float *__restrict__ obj_points, int P,
                                                          const float *__restrict__ objR, const float *__restrict__ objT,
                                                          const int32_t *__restrict__ faces, const int32_t *__restrict__ adj_ptr,
                                                          const int32_t *__restrict__ adj_face,
                                                          const int32_t *__restrict__ adj_corner,
                                                          const int32_t *__restrict__ markers_idx, int M, int B,
                                                          float *__restrict__ markers_out, float *__restrict__ loss_sum,
                                                          float *__restrict__ min_dist, int32_t *__restrict__ label,
                                                          float *__restrict__ o2h_out /* nullable [N][P] */,
                                                          int64_t nn_from /* frames below it skip the NN scan */) {
    extern __shared__ __attribute__((aligned(16))) float4 vs[];          // [V rounded up to VB, + one far-away block] then markers [MAXM]
    const int V4 = (V + VB - 1) / VB * VB;                                 // records are (x, y, z, z): {z, z} is a ready-made packed operand
    float4 *ms = vs + V4 + VB;
    __shared__ int flags[MAXM];
    __shared__ float red[CT];
    const int64_t n = blockIdx.x;
    const int b = (int)(n % B), tid = threadIdx.x;
    const float *vf = verts + (size_t)n * V * 3;
    for (int v = tid; v < V4 + VB; v += CT)                                // the extra block lets the scan prefetch one block past the end
        vs[v] = v < V ? make_float4(vf[3 * v], vf[3 * v + 1], vf[3 * v + 2], vf[3 * v + 2]) : make_float4(3e18f, 3e18f, 3e18f, 3e18f);
    if (tid < MAXM) flags[tid] = 0;
    __syncthreads();
    if (tid < M) {
        const float4 mk = vs[markers_idx[tid]];
        ms[tid] = mk;
        float *mo = markers_out + ((size_t)n * M + tid) * 3;
        mo[0] = mk.x; mo[1] = mk.y; mo[2] = mk.z;
    }
    float R[9], tr[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = objR[n * 9 + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) tr[k] = objT[n * 3 + k];
    float qx[QP], qy[QP], qz[QP];
    const float *op = obj_points + (size_t)b * P * 3;
#pragma unroll
    for (int k = 0; k < QP; ++k) {
        const int i = tid + CT * k;
        float px = 0.f, py = 0.f, pz = 0.f;
        if (i < P) { px = op[3 * i]; py = op[3 * i + 1]; pz = op[3 * i + 2]; }
        // matmul(points, R^T) + t  (eval_smpl_short.py:107)
        qx[k] = (px * R[0] + py * R[1] + pz * R[2]) + tr[0];
        qy[k] = (px * R[3] + py * R[4] + pz * R[5]) + tr[1];
        qz[k] = (px * R[6] + py * R[7] + pz * R[8]) + tr[2];
    }
    v2f QX[QP / 2], QY[QP / 2], QZ[QP / 2], best[QP / 2];
    int bi[QP];
#pragma unroll
    for (int k = 0; k < QP / 2; ++k) {
        QX[k] = v2f{qx[2 * k], qx[2 * k + 1]};
        QY[k] = v2f{qy[2 * k], qy[2 * k + 1]};
        QZ[k] = v2f{qz[2 * k], qz[2 * k + 1]};
        best[k] = v2f{FLT_MAX, FLT_MAX};
        bi[2 * k] = bi[2 * k + 1] = 0;
    }
    __syncthreads();
    // the signed object->human distance is only consumed on future frames (eval_smpl_short.py:121 slices
    // loss_dist_o[past_len:]); past frames only need the marker distances below
    const bool do_nn = n >= nn_from;
    if (do_nn) {
#pragma clang fp contract(off)
        int bblk[QP];                                    // first vertex of the block that last lowered the minimum
#pragma unroll
        for (int k = 0; k < QP; ++k) bblk[k] = 0;
        // software pipeline: the LDS records of the NEXT block are in flight while the current one is scored
        float4 cur[VB], nxt[VB];
#pragma unroll
        for (int u = 0; u < VB; ++u) cur[u] = vs[u];
        for (int v0 = 0; v0 < V4; v0 += VB) {
#pragma unroll
            for (int u = 0; u < VB; ++u) nxt[u] = vs[v0 + VB + u];
            v2f bm[QP / 2];
#pragma unroll
            for (int k = 0; k < QP / 2; ++k) bm[k] = v2f{FLT_MAX, FLT_MAX};
#pragma unroll
            for (int u = 0; u < VB; u += 2) {
                const float4 p = cur[u], r = cur[u + 1];
                const v2f PX = v2f{p.x, p.x}, PY = v2f{p.y, p.y}, PZ = v2f{p.z, p.w};
                const v2f RX = v2f{r.x, r.x}, RY = v2f{r.y, r.y}, RZ = v2f{r.z, r.w};
#pragma unroll
                for (int k = 0; k < QP / 2; ++k) {
                    const v2f dx = QX[k] - PX, dy = QY[k] - PY, dz = QZ[k] - PZ;
                    const v2f d2 = (dx * dx + dy * dy) + dz * dz;
                    const v2f ex = QX[k] - RX, ey = QY[k] - RY, ez = QZ[k] - RZ;
                    const v2f e2 = (ex * ex + ey * ey) + ez * ez;
                    bm[k].x = fminf(fminf(bm[k].x, d2.x), e2.x);          // v_min3_f32
                    bm[k].y = fminf(fminf(bm[k].y, d2.y), e2.y);
                }
            }
#pragma unroll
            for (int k = 0; k < QP / 2; ++k) {
                if (bm[k].x < best[k].x) { best[k].x = bm[k].x; bblk[2 * k] = v0; }
                if (bm[k].y < best[k].y) { best[k].y = bm[k].y; bblk[2 * k + 1] = v0; }
            }
#pragma unroll
            for (int u = 0; u < VB; ++u) cur[u] = nxt[u];
        }
        // resolve the index inside the winning block: lowest vertex whose (bit-identical) distance equals the minimum
#pragma unroll
        for (int k = 0; k < QP; ++k) {
            const float bk = (k & 1) ? best[k >> 1].y : best[k >> 1].x;
            int idx = bblk[k];
#pragma unroll
            for (int u = VB - 1; u >= 0; --u) {
                const float4 p = vs[bblk[k] + u];
                if (dist2_exact(qx[k], qy[k], qz[k], p.x, p.y, p.z) == bk) idx = bblk[k] + u;
            }
            bi[k] = idx;
        }
    }
    float loss = 0.f, mind = FLT_MAX;
#pragma unroll
    for (int k = 0; k < QP; ++k) {
        const int i = tid + CT * k;
        if (i >= P) continue;
        if (do_nn) {
        // normal of the nearest vertex (data/tools.py:4-40 restricted to one vertex), from LDS
        const int v = bi[k];
        float3 acc = make_float3(0.f, 0.f, 0.f);
        for (int e = adj_ptr[v]; e < adj_ptr[v + 1]; ++e) {
            const int f = adj_face[e], c = adj_corner[e];
            const float3 p0 = f3(vs[faces[3 * f]]), p1 = f3(vs[faces[3 * f + 1]]), p2 = f3(vs[faces[3 * f + 2]]);
            float3 nn;
            if (c == 1) nn = cross3(sub3(p2, p1), sub3(p0, p1));
            else if (c == 2) nn = cross3(sub3(p0, p2), sub3(p1, p2));
            else nn = cross3(sub3(p1, p0), sub3(p2, p0));
            acc.x += nn.x; acc.y += nn.y; acc.z += nn.z;
        }
        const float nl = fmaxf(sqrtf(acc.x * acc.x + acc.y * acc.y + acc.z * acc.z), 1e-6f);
        const float4 pv = vs[v];
        const float vx = qx[k] - pv.x, vy = qy[k] - pv.y, vz = qz[k] - pv.z;
        const float dt = (acc.x / nl) * vx + (acc.y / nl) * vy + (acc.z / nl) * vz;
        const float d = sqrtf(vx * vx + vy * vy + vz * vz);
        const float o2h = d * (dt > 0.f ? 1.f : (dt < 0.f ? -1.f : 0.f));
        if (o2h_out) o2h_out[(size_t)n * P + i] = o2h;
        if (o2h < 0.f) loss += fabsf(o2h) * 20.0f;                      // eval_smpl_short.py:113-119
        }
        for (int m = 0; m < M; ++m) {
            const float4 mk = ms[m];
            const float dx = mk.x - qx[k], dy = mk.y - qy[k], dz = mk.z - qz[k];
            const float dm = sqrtf(dx * dx + dy * dy + dz * dz);
            mind = fminf(mind, dm);
            if (dm < 0.02f) flags[m] = 1;                                // benign race: all writers store 1
        }
    }
    // deterministic block reductions
    red[tid] = loss;
    __syncthreads();
    for (int s = CT / 2; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    const float loss_total = red[0];
    __syncthreads();
    red[tid] = mind;
    __syncthreads();
    for (int s = CT / 2; s > 0; s >>= 1) {
        if (tid < s) red[tid] = fminf(red[tid], red[tid + s]);
        __syncthreads();
    }
    if (tid == 0) { loss_sum[n] = loss_total; min_dist[n] = red[0]; }
    if (tid < M) label[(size_t)n * M + tid] = flags[tid];
}

// (An fp32-MFMA "filter" variant of this scan -- g~ = |p|^2 - 2 q.p on v_mfma_f32_16x16x4, exact re-evaluation of
// the vertices inside the error band -- was built, verified bit-identical and measured on MI355X: its two sweeps run
// at 40% of the matrix pipe next to the VALU work that reads every product, 6.1 ms vs 5.0 ms per call for this kernel.
// The fp32 MFMA issues at the fp32 VALU rate on gfx950, so the filter cannot win)
int launch_contact(hipStream_t s, int64_t N, const float *verts, int V, const float *obj_points, int P, const float *objR,
                   const float *objT, const idf_correction_ctx *c, int B, float *markers, float *loss_sum, float *min_dist,
                   int32_t *label, float *o2h, int64_t nn_from) {
    const int M = c->n_markers;
    const size_t lds = ((size_t)((V + VB - 1) / VB * VB) + VB + MAXM) * sizeof(float4);
    if (lds > 160 * 1024 - 8192 || P > MAXP) return IDF_E_INVAL;
    // two shapes of the same scan: 16 waves x 2 points per thread (default) or 8 waves x 4 points (tune = 1: half the LDS broadcasts)
    if (c->tune == 1) {
        static std::atomic<uint64_t> lds_ok{0};
        if (idf_opt_in_lds(reinterpret_cast<const void *>(corr_contact_kernel<512, 4>), 160 * 1024 - 8192, lds_ok) != IDF_OK) return IDF_E_LAUNCH;
        hipLaunchKernelGGL((corr_contact_kernel<512, 4>), dim3((unsigned)N), dim3(512), lds, s, verts, V, obj_points, P, objR, objT, c->faces,
                           c->adj_ptr, c->adj_face, c->adj_corner, c->markers_idx, M, B, markers, loss_sum, min_dist, label, o2h, nn_from);
    } else {
        static std::atomic<uint64_t> lds_ok{0};
        if (idf_opt_in_lds(reinterpret_cast<const void *>(corr_contact_kernel<1024, 2>), 160 * 1024 - 8192, lds_ok) != IDF_OK) return IDF_E_LAUNCH;
        hipLaunchKernelGGL((corr_contact_kernel<1024, 2>), dim3((unsigned)N), dim3(1024), lds, s, verts, V, obj_points, P, objR, objT, c->faces,
                           c->adj_ptr, c->adj_face, c->adj_corner, c->markers_idx, M, B, markers, loss_sum, min_dist, label, o2h, nn_from);
    }
    return IDF_OK;
}

// ---- K5: per-clip decisions (eval_smpl_short.py:119-125) ---------------------------------------------
__global__ __launch_bounds__(128) void corr_reduce_kernel(const float *__restrict__ loss_sum, const float *__restrict__ min_dist,
                                                          const int32_t *__restrict__ label, int B, int T, int past, int P, int M,
                                                          uint8_t *__restrict__ condition, int32_t *__restrict__ contact,
                                                          float *__restrict__ distance_out, float *__restrict__ loss_out) {
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid < M) {
        int c = 0;
        for (int t = past; t < T; ++t) c += label[((size_t)t * B + b) * M + tid];
        contact[(size_t)b * M + tid] = c;
    }
    if (tid == 0) {
        float ls = 0.f, ds = 0.f;
        for (int t = past; t < T; ++t) ls += loss_sum[(size_t)t * B + b] / (float)P;      // mean over points, then frames
        for (int t = 0; t < T; ++t) ds += min_dist[(size_t)t * B + b];
        const float loss = ls / (float)(T - past), dist = ds / (float)T;
        condition[b] = !((loss < 0.002f) && (dist < 0.02f));
        if (distance_out) distance_out[b] = dist;
        if (loss_out) loss_out[b] = loss;
    }
}

// ---- K7: blend (eval_smpl_short.py:127-129): x = a*x + (1-a)*[body, proj] where condition ------------
__global__ __launch_bounds__(256) void corr_blend_kernel(float *__restrict__ x0, const float *__restrict__ proj,
                                                         const uint8_t *__restrict__ condition, int B, int T, float a) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * CTOK * T) return;
    const int b = (int)(i / (CTOK * T)), r = (int)(i - (int64_t)b * CTOK * T), c = r / T, t = r - c * T;
    if (!condition[b]) return;
    const float x = x0[i];
    const float other = c < 135 ? x : proj[((size_t)t * B + b) * 9 + (c - 135)];
    x0[i] = a * x + (1.0f - a) * other;
}

struct CorrWs {
    float *pose, *trans, *objR, *objT, *gt_angles, *gt_trans, *verts, *jtr, *markers, *loss_sum, *min_dist, *proj;
    int32_t *label, *contact;
    uint8_t *condition;
    void *smpl_ws;
    size_t smpl_ws_bytes, total;
};

CorrWs carve(const idf_correction_ctx *c, int B, int T, void *ws) {
    const int64_t N = (int64_t)B * T;
    const int V = c->smpl->V, J = c->smpl->J, M = c->n_markers;
    char *p = reinterpret_cast<char *>(ws);
    size_t off = 0;
    auto take = [&](size_t bytes) { char *r = p ? p + off : nullptr; off += idf_align(bytes); return r; };
    CorrWs w;
    w.pose = (float *)take(N * 156 * 4);
    w.trans = (float *)take(N * 3 * 4);
    w.objR = (float *)take(N * 9 * 4);
    w.objT = (float *)take(N * 3 * 4);
    w.gt_angles = (float *)take(N * 6 * 4);
    w.gt_trans = (float *)take(N * 3 * 4);
    w.verts = (float *)take((size_t)N * V * 3 * 4);
    w.jtr = (float *)take((size_t)N * J * 3 * 4);
    w.markers = (float *)take((size_t)N * M * 3 * 4);
    w.loss_sum = (float *)take(N * 4);
    w.min_dist = (float *)take(N * 4);
    w.proj = (float *)take(N * 9 * 4);
    w.label = (int32_t *)take((size_t)N * M * 4);
    w.contact = (int32_t *)take((size_t)B * M * 4);
    w.condition = (uint8_t *)take(B);
    w.smpl_ws_bytes = interdiff_smpl_workspace_bytes(c->smpl, N);
    w.smpl_ws = take(w.smpl_ws_bytes);
    w.total = off;
    return w;
}

}  // namespace

extern "C" size_t interdiff_correction_workspace_bytes(const idf_correction_ctx *c, int32_t B, int32_t T) {
    if (!c || !c->smpl || B <= 0 || T <= 0) return 0;
    return carve(c, B, T, nullptr).total;
}

extern "C" int interdiff_correction(const idf_correction_ctx *c, float *x0, const float *gt, const float *hand_pose,
                                    const float *beta, const float *obj_points, int32_t B, int32_t T, float blend_t,
                                    uint8_t *condition, int32_t *contact, float *distance, float *loss, void *ws,
                                    size_t ws_bytes, void *stream) {
    if (!c || !c->smpl || !c->objproj || !x0 || !gt || !hand_pose || !beta || !obj_points || !ws || B <= 0 || T <= 0) return IDF_E_INVAL;
    const int V = c->smpl->V, M = c->n_markers, P = c->n_points;
    if (c->smpl->J != 52 || c->smpl->n_betas != 10 || M > MAXM || M != c->objproj->P || P > MAXP || T != c->objproj->T ||
        c->past_len != c->objproj->past_len || c->past_len >= T)
        return IDF_E_INVAL;
    CorrWs w = carve(c, B, T, ws);
    if (ws_bytes < w.total) return IDF_E_NOMEM;
    hipStream_t s = idf_stream(stream);
    const int64_t N = (int64_t)B * T;
    idf_prof_mark(IDF_K_CORR_PREPARE, s);
    hipLaunchKernelGGL(corr_prepare_kernel, dim3((unsigned)idf_cdiv(N * 24, 256)), dim3(256), 0, s, x0, gt, hand_pose, B, T, w.pose,
                       w.trans, w.objR, w.objT, w.gt_angles, w.gt_trans);
    int rc = interdiff_smpl_forward(c->smpl, w.pose, beta, w.trans, N, w.verts, w.jtr, nullptr, w.smpl_ws, w.smpl_ws_bytes, stream);
    if (rc) return rc;
    idf_prof_mark(IDF_K_CORR_CONTACT, s);
    rc = launch_contact(s, N, w.verts, V, obj_points, P, w.objR, w.objT, c, B, w.markers, w.loss_sum, w.min_dist, w.label, nullptr,
                        (int64_t)c->past_len * B);
    if (rc) return rc;
    uint8_t *cond = condition ? condition : w.condition;
    int32_t *cont = contact ? contact : w.contact;
    idf_prof_mark(IDF_K_CORR_REDUCE, s);
    hipLaunchKernelGGL(corr_reduce_kernel, dim3(B), dim3(128), 0, s, w.loss_sum, w.min_dist, w.label, B, T, c->past_len, P, M, cond,
                       cont, distance, loss);
    rc = interdiff_objprojector_sample(c->objproj, w.gt_angles, w.gt_trans, w.markers, cont, B, w.proj, stream);
    if (rc) return rc;
    idf_prof_mark(IDF_K_CORR_BLEND, s);
    hipLaunchKernelGGL(corr_blend_kernel, dim3((unsigned)idf_cdiv((int64_t)B * CTOK * T, 256)), dim3(256), 0, s, x0, w.proj, cond, B, T,
                       blend_t);
    idf_prof_mark(-1, s);
    IDF_CHECK_LAUNCH();
    return IDF_OK;
}

// ---- evaluation metrics (row E1, eval_smpl_short.py:24-81) --------------------------------------------------------
namespace {

__global__ __launch_bounds__(256) void metrics_prepare_kernel(const float *__restrict__ obj_pred, int64_t N,
                                                              float *__restrict__ objR, float *__restrict__ objT) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float q[4], m[9];
    rot::axis_angle_to_quaternion(obj_pred + n * 6, q);
    rot::quaternion_to_matrix(q, m);
#pragma unroll
    for (int k = 0; k < 9; ++k) objR[n * 9 + k] = m[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) objT[n * 3 + k] = obj_pred[n * 6 + 3 + k];
}

__device__ __forceinline__ float block_sum(float v, float *red) {      // 256 threads, deterministic tree
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
}

// one workgroup per clip; out6 rows: global_mpjpe, local_mpjpe, body_translation, obj_translation, obj_rot_error, penetrate
__global__ __launch_bounds__(256) void metrics_reduce_kernel(const float *__restrict__ obj_pred, const float *__restrict__ jtr,
                                                             const float *__restrict__ body_trans, const float *__restrict__ obj_gt,
                                                             const float *__restrict__ jtr_gt, const float *__restrict__ body_trans_gt,
                                                             const float *__restrict__ o2h, int B, int T, int J, int P,
                                                             float *__restrict__ out6) {
    __shared__ float red[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    float g = 0.f, l = 0.f;
    for (int i = tid; i < T * J; i += 256) {
        const int t = i / J, j = i - t * J;
        const float *a = jtr + (((size_t)t * B + b) * J + j) * 3, *c = jtr_gt + (((size_t)t * B + b) * J + j) * 3;
        const float *a0 = jtr + ((size_t)t * B + b) * J * 3, *c0 = jtr_gt + ((size_t)t * B + b) * J * 3;
        const float dx = a[0] - c[0], dy = a[1] - c[1], dz = a[2] - c[2];
        g += sqrtf(dx * dx + dy * dy + dz * dz);
        const float ex = (a[0] - a0[0]) - (c[0] - c0[0]), ey = (a[1] - a0[1]) - (c[1] - c0[1]), ez = (a[2] - a0[2]) - (c[2] - c0[2]);
        l += sqrtf(ex * ex + ey * ey + ez * ez);
    }
    float bt = 0.f, ot = 0.f, rq = 0.f;
    for (int t = tid; t < T; t += 256) {
        const size_t n = (size_t)t * B + b;
        const float *x = body_trans + n * 3, *y = body_trans_gt + n * 3;
        bt += sqrtf((x[0] - y[0]) * (x[0] - y[0]) + (x[1] - y[1]) * (x[1] - y[1]) + (x[2] - y[2]) * (x[2] - y[2]));
        const float *p = obj_pred + n * 6, *q = obj_gt + n * 6;
        ot += sqrtf((p[3] - q[3]) * (p[3] - q[3]) + (p[4] - q[4]) * (p[4] - q[4]) + (p[5] - q[5]) * (p[5] - q[5]));
        float qa[4], qb[4];
        rot::axis_angle_to_quaternion(p, qa);
        rot::axis_angle_to_quaternion(q, qb);
        float e1 = 0.f, e2 = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) { e1 += fabsf(qa[k] - qb[k]); e2 += fabsf(qa[k] + qb[k]); }
        rq += fminf(e1, e2);
    }
    const float gs = block_sum(g, red), ls = block_sum(l, red), bs = block_sum(bt, red), os = block_sum(ot, red), rs = block_sum(rq, red);
    float pen = 0.f;
    for (int t = 0; t < T; ++t) {
        const float *row = o2h + ((size_t)t * B + b) * P;
        float c = 0.f;
        for (int i = tid; i < P; i += 256) c += row[i] < 0.f ? 1.f : 0.f;
        const float cs = block_sum(c, red);
        pen += cs / (float)P;
    }
    if (tid == 0) {
        out6[0 * B + b] = gs / (float)J / (float)T;
        out6[1 * B + b] = ls / (float)J / (float)T;
        out6[2 * B + b] = bs / (float)T;
        out6[3 * B + b] = os / (float)T;
        out6[4 * B + b] = rs / (float)T;
        out6[5 * B + b] = pen / (float)T;
    }
}

struct MetWs {
    float *objR, *objT, *o2h, *markers, *loss_sum, *min_dist;
    int32_t *label;
    size_t total;
};
MetWs carve_metrics(const idf_correction_ctx *c, int B, int T, void *ws) {
    const int64_t N = (int64_t)B * T;
    char *p = reinterpret_cast<char *>(ws);
    size_t off = 0;
    auto take = [&](size_t bytes) { char *r = p ? p + off : nullptr; off += idf_align(bytes); return r; };
    MetWs w;
    w.objR = (float *)take(N * 9 * 4);
    w.objT = (float *)take(N * 3 * 4);
    w.o2h = (float *)take((size_t)N * c->n_points * 4);
    w.markers = (float *)take((size_t)N * c->n_markers * 3 * 4);
    w.loss_sum = (float *)take(N * 4);
    w.min_dist = (float *)take(N * 4);
    w.label = (int32_t *)take((size_t)N * c->n_markers * 4);
    w.total = off;
    return w;
}

}  // namespace

extern "C" size_t interdiff_metrics_workspace_bytes(const idf_correction_ctx *c, int32_t B, int32_t T) {
    if (!c || B <= 0 || T <= 0) return 0;
    return carve_metrics(c, B, T, nullptr).total;
}

extern "C" int interdiff_metrics(const idf_correction_ctx *c, const float *obj_pred, const float *jtr, const float *body_trans,
                                 const float *obj_gt, const float *jtr_gt, const float *body_trans_gt, const float *verts,
                                 const float *obj_points, int32_t B, int32_t T, int32_t J, float *out6, void *ws, size_t ws_bytes,
                                 void *stream) {
    if (!c || !c->smpl || !obj_pred || !jtr || !body_trans || !obj_gt || !jtr_gt || !body_trans_gt || !verts || !obj_points || !out6 ||
        !ws || B <= 0 || T <= 0 || J <= 0)
        return IDF_E_INVAL;
    const int V = c->smpl->V, M = c->n_markers, P = c->n_points;
    if (M > MAXM || P > MAXP) return IDF_E_INVAL;
    MetWs w = carve_metrics(c, B, T, ws);
    if (ws_bytes < w.total) return IDF_E_NOMEM;
    hipStream_t s = idf_stream(stream);
    const int64_t N = (int64_t)B * T;
    idf_prof_mark(IDF_K_OTHER, s);
    hipLaunchKernelGGL(metrics_prepare_kernel, dim3((unsigned)idf_cdiv(N, 256)), dim3(256), 0, s, obj_pred, N, w.objR, w.objT);
    const int rc = launch_contact(s, N, verts, V, obj_points, P, w.objR, w.objT, c, B, w.markers, w.loss_sum, w.min_dist, w.label, w.o2h, 0);
    if (rc) return rc;
    hipLaunchKernelGGL(metrics_reduce_kernel, dim3(B), dim3(256), 0, s, obj_pred, jtr, body_trans, obj_gt, jtr_gt, body_trans_gt, w.o2h, B,
                       T, J, P, out6);
    idf_prof_mark(-1, s);
    IDF_CHECK_LAUNCH();
    return IDF_OK;
}
