// Physics post-optimisation for gfx950 ("next" row N4 of SURVEY.md §8(f)): optimization.py:19-173 of the reference.
//
// The reference runs torch autograd over ~2500 tiny kernels per Adam iteration (Python loops over 52 joints in the SMPL
// layer, a [T,P,V,3] distance tensor for the contact radius).  Here one iteration is 17 launches with a hand-written
// backward pass:
//   forward   param -> axis-angle (matrix_to_axis_angle), SMPL (smpl.hip), ONE nearest-neighbour scan for both the
//             point->vertex argmin and the vertex contact-radius mask, normals only at the nearest vertices;
//   loss      per frame: penetration term + vertex regulariser, d/dverts staged in LDS (the scatter onto nearest
//             vertices is an LDS atomic add), d/d object pose reduced in the same workgroup;
//   skinning^T  per vertex: d/dv_posed = (sum_s w A)^T g ; per (frame, joint): dA = sum_v w g (x) [v_posed;1] over the
//             joint's own vertex list (CSR by joint, deterministic);
//   blend^T   d/dfeat = d/dv_posed . blend : fp32-MFMA split-K GEMM against the transposed basis;
//   chain^T   per frame: kinematic chain backwards through LDS, static-foot term, then the 9x9 Jacobian of
//             rotation -> axis-angle -> SMPL Rodrigues by forward-mode duals (rot_dual.h);
//   update    regularisers + temporal smoothness (closed-form gradients), Adam, best-iterate bookkeeping on the device.
#include "common.h"
#include "rot_math.h"
#include "rot_dual.h"
#include <cfloat>

namespace {

constexpr int NP = IDF_OPT_NP, NL = IDF_OPT_NLOSS, NJ = 52, MAXJ = 64;
constexpr int OFF_TR = 468, OFF_OT = 471, OFF_OR = 474;
constexpr int FOOT0 = 10;                               // joints 10 / 11: left / right foot (optimization.py:47-48)
constexpr int KSLICE = 1024;                            // k per workgroup of the blend^T GEMM

__device__ __forceinline__ float sgn(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }

// ---- init ----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void opt_init_kernel(const float *__restrict__ pose, const float *__restrict__ trans,
                                                      const float *__restrict__ obj_angles, const float *__restrict__ obj_trans,
                                                      float *__restrict__ param, float *__restrict__ init, float *__restrict__ best) {
    const int64_t n = blockIdx.x;
    const int j = threadIdx.x;
    float out[9];
    int off = -1, cnt = 0;
    if (j < NJ || j == NJ + 2) {                        // optimization.py:28-31: pytorch3d axis_angle_to_matrix
        const float *a = j < NJ ? pose + n * 156 + 3 * j : obj_angles + n * 3;
        float q[4];
        rot::axis_angle_to_quaternion(a, q);
        rot::quaternion_to_matrix(q, out);
        off = j < NJ ? 9 * j : OFF_OR;
        cnt = 9;
    } else if (j == NJ || j == NJ + 1) {
        const float *a = (j == NJ ? trans : obj_trans) + n * 3;
        out[0] = a[0]; out[1] = a[1]; out[2] = a[2];
        off = j == NJ ? OFF_TR : OFF_OT;
        cnt = 3;
    }
    for (int e = 0; e < cnt; ++e) {
        const size_t i = (size_t)n * NP + off + e;
        param[i] = out[e]; init[i] = out[e]; best[i] = out[e];
    }
}

__global__ __launch_bounds__(64) void opt_setctl_kernel(int32_t *__restrict__ ctl, int32_t first_iter) {
    if (threadIdx.x < 4) ctl[threadIdx.x] = threadIdx.x == 0 ? first_iter : 0;
}

__global__ __launch_bounds__(64) void opt_static_kernel(const float *__restrict__ jtr, int T, int J, uint8_t *__restrict__ foot_static,
                                                        int32_t *__restrict__ foot_cnt, float *__restrict__ best_loss) {
    const int b = blockIdx.x, f = threadIdx.x;
    if (f >= 2) return;
    int cnt = 0;
    for (int t = 0; t < T; ++t) {
        uint8_t s = 0;
        if (t < T - 1) {
            const float *a = jtr + ((size_t)(b * T + t) * J + FOOT0 + f) * 3, *c = a + (size_t)J * 3;
            const float dx = c[0] - a[0], dz = c[2] - a[2];
            s = (sqrtf(dx * dx + dz * dz) + 1e-6f) < 0.008f;     // optimization.py:49-52
        }
        foot_static[(size_t)(b * T + t) * 2 + f] = s;
        cnt += s;
    }
    foot_cnt[b * 2 + f] = cnt;
    if (f == 0) best_loss[b] = 1e7f;                              // :122
}

// ---- forward glue --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void opt_unpack_kernel(const float *__restrict__ param, float *__restrict__ pose, float *__restrict__ tr) {
    const int64_t n = blockIdx.x;
    const int j = threadIdx.x;
    if (j < NJ) rot::matrix_to_axis_angle(param + n * NP + 9 * j, pose + n * 156 + 3 * j);      // optimization.py:56
    else if (j < NJ + 3) tr[n * 3 + (j - NJ)] = param[n * NP + OFF_TR + (j - NJ)];
}

__global__ __launch_bounds__(256) void opt_objpts_kernel(const float *__restrict__ param, const float *__restrict__ obj_points, int P,
                                                         int T, float *__restrict__ pts) {
    const int64_t n = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const float *R = param + n * NP + OFF_OR, *t = param + n * NP + OFF_OT;
    const float *x = obj_points + ((size_t)(n / T) * P + p) * 3;
    float *o = pts + ((size_t)n * P + p) * 3;                     // optimization.py:62
#pragma unroll
    for (int r = 0; r < 3; ++r) o[r] = (x[0] * R[r * 3] + x[1] * R[r * 3 + 1] + x[2] * R[r * 3 + 2]) + t[r];
}

// ---- both nearest-neighbour questions of calc_loss in ONE scan (optimization.py:64-65,74-75) ------------------------------
// point2point_signed needs, per object point, the nearest vertex (tools.py:45-50); the contact-radius mask needs, per
// vertex, whether ANY object point lies within 0.5 m.  Both read the same P x V distances, so one pass serves both: every
// thread owns two object points as a packed pair (exact (dx*dx + dy*dy) + dz*dz, no FMA contraction, lowest index wins:
// the argmin is bit-identical to geometry.hip and to the oracle), vertices stream through LDS in chunks as (x, y, z, z)
// records, and the per-vertex "some point is near" bit is a wave ballot folded into a 64-bit scalar mask (SALU work that
// overlaps the VALU of the other waves), OR-ed into LDS once per 64 vertices.
// As in correction.hip's contact scan the loop only keeps the running MINIMUM (v_min3_f32 over vertex pairs); which vertex
// it was is settled per block of 8 -- one compare + two selects per point per block -- and resolved after the scan by
// re-scoring the winning block (the lowest vertex whose bit-identical distance equals the minimum): 17 -> ~11.5 VALU
// instructions per vertex per thread.
constexpr int NN_T = 256, NN_RC = 1024, NN_VB = 8;
typedef float v2f __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(NN_T) void opt_nn_kernel(const float *__restrict__ pts, int P, const float *__restrict__ verts, int V,
                                                      int32_t *__restrict__ yidx, int32_t *__restrict__ near /* [N][V], zeroed */) {
    __shared__ __attribute__((aligned(16))) float4 rs[NN_RC + NN_VB];
    __shared__ unsigned nfw[NN_RC / 32];
    const int64_t n = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int i0 = blockIdx.x * NN_T * 2 + tid, i1 = i0 + NN_T;
    const float *pn = pts + (size_t)n * P * 3, *vn = verts + (size_t)n * V * 3;
    const float FAR = 3e18f;
    const v2f QX = v2f{i0 < P ? pn[3 * i0] : FAR, i1 < P ? pn[3 * i1] : FAR};
    const v2f QY = v2f{i0 < P ? pn[3 * i0 + 1] : FAR, i1 < P ? pn[3 * i1 + 1] : FAR};
    const v2f QZ = v2f{i0 < P ? pn[3 * i0 + 2] : FAR, i1 < P ? pn[3 * i1 + 2] : FAR};
    v2f best = v2f{FLT_MAX, FLT_MAX};
    int b0 = 0, b1 = 0;                                   // first vertex of the block that last lowered the minimum
    for (int c0 = 0; c0 < V; c0 += NN_RC) {
        __syncthreads();
        for (int j = tid; j < NN_RC + NN_VB; j += NN_T) {
            const int v = c0 + j;
            rs[j] = (v < V && j < NN_RC) ? make_float4(vn[3 * v], vn[3 * v + 1], vn[3 * v + 2], vn[3 * v + 2])
                                         : make_float4(-FAR, -FAR, -FAR, -FAR);
        }
        if (tid < NN_RC / 32) nfw[tid] = 0;
        __syncthreads();
        const int cn = (min(NN_RC, V - c0) + 63) & ~63;
        {
#pragma clang fp contract(off)
            // software pipeline with two register sets: the LDS records of the next block are in flight while the current
            // one is scored
            float4 cur[NN_VB], nxt[NN_VB];
#pragma unroll
            for (int u = 0; u < NN_VB; ++u) cur[u] = rs[u];
            for (int g = 0; g < cn; g += 64) {
                unsigned long long mask = 0;
#pragma unroll 1
                for (int u8 = 0; u8 < 64; u8 += NN_VB) {
#pragma unroll
                    for (int u = 0; u < NN_VB; ++u) nxt[u] = rs[g + u8 + NN_VB + u];          // rs has one pad block
                    v2f bm = v2f{FLT_MAX, FLT_MAX};
#pragma unroll
                    for (int u = 0; u < NN_VB; u += 2) {
                        const float4 p = cur[u], r = cur[u + 1];
                        const v2f dx = QX - v2f{p.x, p.x}, dy = QY - v2f{p.y, p.y}, dz = QZ - v2f{p.z, p.w};
                        const v2f d2 = (dx * dx + dy * dy) + dz * dz;
                        const v2f ex = QX - v2f{r.x, r.x}, ey = QY - v2f{r.y, r.y}, ez = QZ - v2f{r.z, r.w};
                        const v2f e2 = (ex * ex + ey * ey) + ez * ez;
                        bm.x = fminf(fminf(bm.x, d2.x), e2.x);                              // v_min3_f32
                        bm.y = fminf(fminf(bm.y, d2.y), e2.y);
                        // sqrt(d2) < 0.5 (optimization.py:75) <=> d2 < 0.25 up to the rounding of the last ulp
                        const unsigned long long bd = __builtin_amdgcn_ballot_w64(d2.x < 0.25f) | __builtin_amdgcn_ballot_w64(d2.y < 0.25f);
                        const unsigned long long be = __builtin_amdgcn_ballot_w64(e2.x < 0.25f) | __builtin_amdgcn_ballot_w64(e2.y < 0.25f);
                        unsigned bitd, bite;                              // (ballot != 0) on the scalar unit
                        asm volatile("s_cmp_lg_u64 %1, 0\n\ts_cselect_b32 %0, 1, 0" : "=s"(bitd) : "s"(bd) : "scc");
                        asm volatile("s_cmp_lg_u64 %1, 0\n\ts_cselect_b32 %0, 1, 0" : "=s"(bite) : "s"(be) : "scc");
                        mask |= ((unsigned long long)bitd << (u8 + u)) | ((unsigned long long)bite << (u8 + u + 1));
                    }
                    if (bm.x < best.x) { best.x = bm.x; b0 = c0 + g + u8; }
                    if (bm.y < best.y) { best.y = bm.y; b1 = c0 + g + u8; }
#pragma unroll
                    for (int u = 0; u < NN_VB; ++u) cur[u] = nxt[u];
                }
                if (lane == 0 && mask) {
                    atomicOr(&nfw[g >> 5], (unsigned)mask);
                    atomicOr(&nfw[(g >> 5) + 1], (unsigned)(mask >> 32));
                }
            }
        }
        __syncthreads();
        for (int j = tid; j < NN_RC; j += NN_T)
            if ((nfw[j >> 5] >> (j & 31)) & 1u) near[(size_t)n * V + c0 + j] = 1;      // several point blocks may store the same 1
    }
    // resolve the index inside the winning block: the lowest vertex whose distance is (at or, defensively, below) the minimum -- `<=`, not `==`:
    // should v_min3_f32 and this recompute ever differ by a bit, the true winner still matches instead of no lane at all
    {
#pragma clang fp contract(off)
        int r0 = b0, r1 = b1;
#pragma unroll
        for (int u = NN_VB - 1; u >= 0; --u) {
            const int v0 = min(b0 + u, V - 1), v1 = min(b1 + u, V - 1);
            const float dx0 = QX.x - vn[3 * v0], dy0 = QY.x - vn[3 * v0 + 1], dz0 = QZ.x - vn[3 * v0 + 2];
            const float dx1 = QX.y - vn[3 * v1], dy1 = QY.y - vn[3 * v1 + 1], dz1 = QZ.y - vn[3 * v1 + 2];
            if ((dx0 * dx0 + dy0 * dy0) + dz0 * dz0 <= best.x && b0 + u < V) r0 = b0 + u;
            if ((dx1 * dx1 + dy1 * dy1) + dz1 * dz1 <= best.y && b1 + u < V) r1 = b1 + u;
        }
        if (i0 < P) yidx[(size_t)n * P + i0] = r0;
        if (i1 < P) yidx[(size_t)n * P + i1] = r1;
    }
}

__device__ __forceinline__ float3 ld3(const float *p) { return make_float3(p[0], p[1], p[2]); }
__device__ __forceinline__ float3 sub3(float3 a, float3 b) { return make_float3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ float3 cross3(float3 a, float3 b) {
    return make_float3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

// vector to the nearest vertex, its norm, and the inside/outside sign from the vertex normal (tools.py:55-61) -- the normal
// (data/tools.py:4-40) is evaluated for the nearest vertices only, in the reference's accumulation order.
__global__ __launch_bounds__(256) void opt_signed_kernel(const float *__restrict__ pts, int P, const float *__restrict__ verts, int V,
                                                         const int32_t *__restrict__ yidx, const int32_t *__restrict__ faces,
                                                         const int32_t *__restrict__ adj_ptr, const int32_t *__restrict__ adj_face,
                                                         const int32_t *__restrict__ adj_corner,
                                                         const int32_t *__restrict__ adj_pair /* nullable [nnz][2]: the face's other two vertices (a, b): normal += (a - v) x (b - v) */,
                                                         float *__restrict__ y2x_signed, float *__restrict__ y2x) {
    const int64_t n = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float *vn = verts + (size_t)n * V * 3;
    const int v = yidx[(size_t)n * P + i];
    float3 acc = make_float3(0.f, 0.f, 0.f);
    if (adj_pair) {                                      // one 8-byte load per incident face instead of face id -> corner -> three vertex ids
        const float3 pv3 = ld3(vn + 3 * v);
        const int e0 = adj_ptr[v], e1 = adj_ptr[v + 1];
        for (int e = e0; e < e1; e += 4) {
            int2 ab[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) ab[u] = *reinterpret_cast<const int2 *>(adj_pair + 2 * (size_t)min(e + u, e1 - 1));
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (e + u < e1) {
                    const float3 nn = cross3(sub3(ld3(vn + 3 * ab[u].x), pv3), sub3(ld3(vn + 3 * ab[u].y), pv3));
                    acc.x += nn.x; acc.y += nn.y; acc.z += nn.z;
                }
            }
        }
    } else
    for (int e = adj_ptr[v]; e < adj_ptr[v + 1]; ++e) {
        const int f = adj_face[e], c = adj_corner[e];
        const float3 p0 = ld3(vn + 3 * faces[3 * f]), p1 = ld3(vn + 3 * faces[3 * f + 1]), p2 = ld3(vn + 3 * faces[3 * f + 2]);
        float3 nn;
        if (c == 1) nn = cross3(sub3(p2, p1), sub3(p0, p1));
        else if (c == 2) nn = cross3(sub3(p0, p2), sub3(p1, p2));
        else nn = cross3(sub3(p1, p0), sub3(p2, p0));
        acc.x += nn.x; acc.y += nn.y; acc.z += nn.z;
    }
    const float nl = fmaxf(sqrtf(acc.x * acc.x + acc.y * acc.y + acc.z * acc.z), 1e-6f);
    const float *q = pts + ((size_t)n * P + i) * 3, *pv = vn + 3 * v;
    const float vx = q[0] - pv[0], vy = q[1] - pv[1], vz = q[2] - pv[2];
    const float dt = (acc.x / nl) * vx + (acc.y / nl) * vy + (acc.z / nl) * vz;
    const float d = sqrtf(vx * vx + vy * vy + vz * vz);
    y2x_signed[(size_t)n * P + i] = d * (dt > 0.f ? 1.f : (dt < 0.f ? -1.f : 0.f));
    float *o = y2x + ((size_t)n * P + i) * 3;
    o[0] = vx; o[1] = vy; o[2] = vz;
}

// ---- loss on the geometry + its gradient with respect to vertices and object pose (optimization.py:66-79) ---------------
constexpr int LG_T = 512, LG_W = LG_T / 64, LG_R = 17;

__global__ __launch_bounds__(LG_T) void opt_lossgrad_kernel(const float *__restrict__ verts, const float *__restrict__ verts_gt, int V,
                                                            const float *__restrict__ y2x_signed, const int32_t *__restrict__ near,
                                                            const int32_t *__restrict__ yidx, const float *__restrict__ y2x,
                                                            const float *__restrict__ obj_points, int P, int T,
                                                            const int32_t *__restrict__ ctl, float *__restrict__ gv_out,
                                                            float *__restrict__ grad, float *__restrict__ gtr, float *__restrict__ lossf) {
    extern __shared__ float gvs[];                                // [3V] d loss / d verts of this frame
    __shared__ float red[LG_R][LG_W];
    const int64_t n = blockIdx.x;
    const int b = (int)(n / T), tid = threadIdx.x;
    const float invT = 1.0f / (float)T;
    const int ii = ctl[0];
    const float wcol = ii < 350 ? (float)(20.0 * ((double)ii / 350.0)) : 20.0f;          // :70, ratio = ii / 350
    const float *vn = verts + (size_t)n * V * 3, *vg = verts_gt + (size_t)n * V * 3;
    float acc[LG_R];
#pragma unroll
    for (int k = 0; k < LG_R; ++k) acc[k] = 0.f;
    for (int i = tid; i < 3 * V; i += LG_T) {
        const float wv = near[(size_t)n * V + i / 3] ? 0.f : 0.01f;                      // :72-76
        const float d = vn[i] - vg[i];
        gvs[i] = wv * invT * sgn(d);
        acc[16] += wv * fabsf(d);
    }
    __syncthreads();
    for (int p = tid; p < P; p += LG_T) {
        const float s = y2x_signed[(size_t)n * P + p];
        if (s < 0.f) {                                                                   // penetrating point, weight wcol (:69-70)
            const float *v = y2x + ((size_t)n * P + p) * 3;
            const float nrm = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
            const float k = wcol * invT / nrm;
            const float g[3] = {k * v[0], k * v[1], k * v[2]};
            const int yi = yidx[(size_t)n * P + p];
            const float *x = obj_points + ((size_t)b * P + p) * 3;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                atomicAdd(&gvs[3 * yi + r], -g[r]);
                acc[r * 3 + 0] += g[r] * x[0]; acc[r * 3 + 1] += g[r] * x[1]; acc[r * 3 + 2] += g[r] * x[2];
                acc[9 + r] += g[r];
            }
            acc[15] += wcol * fabsf(s);
        }
    }
    __syncthreads();
    float *go = gv_out + (size_t)n * V * 3;
    for (int i0 = tid * 3; i0 < 3 * V; i0 += LG_T * 3) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float g = gvs[i0 + c];
            go[i0 + c] = g;
            acc[12 + c] += g;
        }
    }
#pragma unroll
    for (int k = 0; k < LG_R; ++k) {
        const float s = wave_sum(acc[k]);
        if ((tid & 63) == 0) red[k][tid >> 6] = s;
    }
    __syncthreads();
    if (tid < LG_R) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < LG_W; ++w) s += red[tid][w];
        if (tid < 9) grad[n * NP + OFF_OR + tid] = s;
        else if (tid < 12) grad[n * NP + OFF_OT + (tid - 9)] = s;
        else if (tid < 15) gtr[n * 3 + (tid - 12)] = s;
        else lossf[n * NL + (tid - 15)] = s;                       // [0] collision, [1] verts_reg (both before the mean over T)
    }
}

// ---- skinning transposed -----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void opt_skin_vertex_kernel(const idf_smpl_model m, const float *__restrict__ A, const float *__restrict__ gv,
                                                              int K3P, float *__restrict__ dvposed) {
    const int64_t n = blockIdx.y;
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= m.V) return;
    float t[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float *An = A + (size_t)n * m.J * 12;
    for (int s = 0; s < m.S; ++s) {
        const float w = m.skin_w[(size_t)v * m.S + s];
        const float *a = An + m.skin_idx[(size_t)v * m.S + s] * 12;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) t[r * 3 + c] += w * a[r * 4 + c];
    }
    const float *g = gv + ((size_t)n * m.V + v) * 3;
    float *o = dvposed + (size_t)n * K3P + 3 * v;
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = t[c] * g[0] + t[3 + c] * g[1] + t[6 + c] * g[2];
}

__global__ __launch_bounds__(64) void opt_skin_joint_kernel(const int32_t *__restrict__ jv_ptr, const int32_t *__restrict__ jv_vtx,
                                                            const float *__restrict__ jv_w, const float *__restrict__ gv,
                                                            const float *__restrict__ vposed, int V, int J, float *__restrict__ dA) {
    const int j = blockIdx.x, lane = threadIdx.x;
    const int64_t n = blockIdx.y;
    float acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = 0.f;
    for (int e = jv_ptr[j] + lane; e < jv_ptr[j + 1]; e += 64) {
        const int v = jv_vtx[e];
        const float w = jv_w[e];
        const float *g = gv + ((size_t)n * V + v) * 3, *p = vposed + ((size_t)n * V + v) * 3;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float wg = w * g[r];
            acc[r * 4 + 0] += wg * p[0]; acc[r * 4 + 1] += wg * p[1]; acc[r * 4 + 2] += wg * p[2]; acc[r * 4 + 3] += wg;
        }
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const float s = wave_sum(acc[k]);
        if (lane == 0) dA[((size_t)n * J + j) * 12 + k] = s;
    }
}

// ---- blend^T : dfeat[s][n][k] = sum_{r in slice s} dvposed[n][r] * blendT[k][r]  (fp32 MFMA, 4 waves split the slice) ------
constexpr int DF_MT = 4;                                          // 16-frame tiles per workgroup

__global__ __launch_bounds__(256) void opt_dfeat_kernel(const float *__restrict__ dvp, const float *__restrict__ blendT, int M, int KB,
                                                        int K3P, float *__restrict__ part) {
    __shared__ f32x4 red[4][DF_MT][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
    const int c0 = blockIdx.x * 16, sl = blockIdx.y, m0 = blockIdx.z * 16 * DF_MT;
    const size_t kbase = (size_t)sl * KSLICE + wave * (KSLICE / 4) + kq * 4;
    const float *bp = blendT + (size_t)(c0 + li) * K3P + kbase;
    const float *ap[DF_MT];
#pragma unroll
    for (int i = 0; i < DF_MT; ++i) ap[i] = dvp + (size_t)min(m0 + i * 16 + li, M - 1) * K3P + kbase;
    f32x4 acc[DF_MT];
#pragma unroll
    for (int i = 0; i < DF_MT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < KSLICE / 4 / 16; ++s) {
        const float4 bv = *reinterpret_cast<const float4 *>(bp + s * 16);
#pragma unroll
        for (int i = 0; i < DF_MT; ++i) {
            const float4 av = *reinterpret_cast<const float4 *>(ap[i] + s * 16);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc[i], 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < DF_MT; ++i) red[wave][i][lane] = acc[i];
    __syncthreads();
    for (int i = wave; i < DF_MT; i += 4) {                       // wave w finishes frame tiles w, w+4
        const f32x4 a = red[0][i][lane], b = red[1][i][lane], c = red[2][i][lane], d = red[3][i][lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + i * 16 + kq * 4 + r;
            if (row < M) part[((size_t)sl * M + row) * KB + c0 + li] = (a[r] + b[r]) + (c[r] + d[r]);
        }
    }
}

// ---- kinematic chain transposed + rotation Jacobian -----------------------------------------------------------------------
__global__ __launch_bounds__(64) void opt_chain_bwd_kernel(const idf_smpl_model m, const float *__restrict__ pose, const float *__restrict__ betas,
                                                           const float *__restrict__ param, const float *__restrict__ dA,
                                                           const float *__restrict__ dfeat, int n_slices, int64_t N, int T,
                                                           const float *__restrict__ jtr, const uint8_t *__restrict__ foot_static,
                                                           const int32_t *__restrict__ foot_cnt, const float *__restrict__ gtr,
                                                           float *__restrict__ grad, float *__restrict__ lossf) {
    __shared__ float Rs[MAXJ * 9], Js[MAXJ * 3], Gs[MAXJ * 12], dG[MAXJ * 12], dR[MAXJ * 9], gfoot[8];
    const int64_t n = blockIdx.x;
    const int b = (int)(n / T), t = (int)(n % T);
    const int j = threadIdx.x, J = m.J, nb = m.n_betas, KB = m.KB;
    const float *beta = betas + n * nb;
    // forward quantities, as smpl_pose_kernel computes them
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
    // static-foot term (optimization.py:81-90,112): 1000 * mean over static frames and the two ground-plane coordinates
    if (j < 2) {
        const int cnt = foot_cnt[b * 2 + j];
        const float cf = cnt > 0 ? 1000.0f / (2.0f * (float)cnt) : 0.f;
        const float *cur = jtr + ((size_t)n * J + FOOT0 + j) * 3;
        float gx = 0.f, gz = 0.f, ls = 0.f;
        if (t < T - 1 && foot_static[n * 2 + j]) {
            const float *nx = cur + (size_t)J * 3;
            const float dx = nx[0] - cur[0], dz = nx[2] - cur[2];
            gx -= 2.f * cf * dx; gz -= 2.f * cf * dz;
            ls = cf * (dx * dx + dz * dz);
        }
        if (t > 0 && foot_static[(n - 1) * 2 + j]) {
            const float *pv = cur - (size_t)J * 3;
            gx += 2.f * cf * (cur[0] - pv[0]); gz += 2.f * cf * (cur[2] - pv[2]);
        }
        gfoot[j * 4 + 0] = gx; gfoot[j * 4 + 1] = 0.f; gfoot[j * 4 + 2] = gz; gfoot[j * 4 + 3] = ls;
    }
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
    // dG from dA (A = [G_R | G_t - G_R J], smpl_layer.py:135-142) and from d/djtr (jtr = G_t + trans)
    if (j < J) {
        const float *a = dA + ((size_t)n * J + j) * 12;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float at = a[r * 4 + 3];
            dG[j * 12 + r * 4 + 0] = a[r * 4 + 0] - at * Js[j * 3 + 0];
            dG[j * 12 + r * 4 + 1] = a[r * 4 + 1] - at * Js[j * 3 + 1];
            dG[j * 12 + r * 4 + 2] = a[r * 4 + 2] - at * Js[j * 3 + 2];
            dG[j * 12 + r * 4 + 3] = at + ((j == FOOT0 || j == FOOT0 + 1) ? gfoot[(j - FOOT0) * 4 + r] : 0.f);
        }
    }
    __syncthreads();
    // chain backwards: G_i = G_p [R_i | J_i - J_p]  (smpl_layer.py:121-130)
    for (int i = J - 1; i >= 1; --i) {
        const int p = m.parents[i];
        if (j < 12) {
            const int r = j >> 2, c = j & 3;
            const float *Dg = dG + i * 12, *Gp = Gs + p * 12;
            if (c < 3) {
                dR[i * 9 + r * 3 + c] = Gp[r] * Dg[c] + Gp[4 + r] * Dg[4 + c] + Gp[8 + r] * Dg[8 + c];
                dG[p * 12 + j] += (Dg[r * 4] * Rs[i * 9 + c * 3] + Dg[r * 4 + 1] * Rs[i * 9 + c * 3 + 1] + Dg[r * 4 + 2] * Rs[i * 9 + c * 3 + 2]) +
                                  Dg[r * 4 + 3] * (Js[i * 3 + c] - Js[p * 3 + c]);
            } else {
                dG[p * 12 + j] += Dg[r * 4 + 3];
            }
        }
        __syncthreads();
    }
    if (j < 9) dR[j] = dG[(j / 3) * 4 + (j % 3)];
    __syncthreads();
    if (j < J) {
        float g[9], gin[9];
#pragma unroll
        for (int e = 0; e < 9; ++e) g[e] = dR[j * 9 + e];
        if (j >= 1) {                                             // pose blend shapes: feat = R_j - I (smpl_layer.py:89-92,106)
            for (int s = 0; s < n_slices; ++s) {
                const float *df = dfeat + ((size_t)s * N + n) * KB + (j - 1) * 9;
#pragma unroll
                for (int e = 0; e < 9; ++e) g[e] += df[e];
            }
        }
        rotd::joint_map_vjp(param + n * NP + 9 * j, g, gin);
#pragma unroll
        for (int e = 0; e < 9; ++e) grad[n * NP + 9 * j + e] = gin[e];
    } else if (j < J + 3) {
        const int c = j - J;
        grad[n * NP + OFF_TR + c] = gtr[n * 3 + c] + gfoot[c] + gfoot[4 + c];
    } else if (j == J + 3) {
        lossf[n * NL + 2] = gfoot[3] + gfoot[7];
    }
}

// ---- regularisers + smoothness (optimization.py:92-111): closed-form gradient added to grad, loss partials -----------------
struct SegW { float a, c2, c1; };
__device__ __forceinline__ SegW seg_weights(int e, int T) {
    const float t0 = (float)T, t1 = (float)(T - 1), t2 = (float)(T - 2);
    if (e < 9) return {0.1f / (t0 * 9.f), 5.f / (t2 * 9.f), 5.f / (t1 * 9.f)};                 // global orientation
    if (e < 198) return {0.005f / (t0 * 3.f), 1000.f / (t2 * 3.f), 100.f / (t1 * 3.f)};       // body joints: .sum(2).sum(1) leaves [T,3] for the mean
    if (e < OFF_TR) return {0.f, 50.f / (t2 * 270.f), 50.f / (t1 * 270.f)};                  // hand joints
    if (e < OFF_OT) return {0.1f / (t0 * 3.f), 10.f / (t2 * 3.f), 10.f / (t1 * 3.f)};         // body translation
    if (e < OFF_OR) return {0.1f / (t0 * 3.f), 1000.f / (t2 * 3.f), 100.f / (t1 * 3.f)};      // object translation
    return {0.1f / (t0 * 9.f), 1000.f / (t2 * 9.f), 100.f / (t1 * 9.f)};                      // object rotation
}

__global__ __launch_bounds__(512) void opt_reg_kernel(const float *__restrict__ param, const float *__restrict__ init, int T,
                                                      float *__restrict__ grad, float *__restrict__ lossf) {
    __shared__ float red[2][8];
    const int64_t n = blockIdx.x;
    const int t = (int)(n % T), e = threadIdx.x;
    float lr = 0.f, lv = 0.f;
    if (e < NP) {
        const SegW w = seg_weights(e, T);
        const float *x = param + n * NP + e;
        auto at = [&](int dt) { return x[(ptrdiff_t)dt * NP]; };
        auto sd = [&](int tt) { return (tt >= 1 && tt <= T - 2) ? 2.f * at(tt - t) - at(tt - 1 - t) - at(tt + 1 - t) : 0.f; };
        auto fd = [&](int tt) { return (tt >= 0 && tt <= T - 2) ? at(tt + 1 - t) - at(tt - t) : 0.f; };
        const float d0 = x[0] - init[n * NP + e];
        const float s0 = sd(t), f0 = fd(t);
        const float g = w.a * sgn(d0) + 2.f * w.c2 * (2.f * s0 - sd(t - 1) - sd(t + 1)) + 2.f * w.c1 * (fd(t - 1) - f0);
        grad[n * NP + e] += g;
        lr = w.a * fabsf(d0);
        lv = w.c2 * s0 * s0 + w.c1 * f0 * f0;
    }
    lr = wave_sum(lr); lv = wave_sum(lv);
    if ((e & 63) == 0) { red[0][e >> 6] = lr; red[1][e >> 6] = lv; }
    __syncthreads();
    if (e < 2) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) s += red[e][w];
        lossf[n * NL + 3 + e] = s;
    }
}

// ---- torch.optim.Adam(lr=1e-3) defaults, single step --------------------------------------------------------------------
__global__ __launch_bounds__(256) void opt_adam_kernel(float *__restrict__ param, const float *__restrict__ grad, float *__restrict__ m,
                                                       float *__restrict__ v, int64_t total, const int32_t *__restrict__ ctl) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int step = ctl[1] + 1;
    const double bc1 = 1.0 - pow(0.9, (double)step), bc2 = 1.0 - pow(0.999, (double)step);
    const float step_size = (float)(1e-3 / bc1), bc2s = (float)sqrt(bc2);
    const float g = grad[i];
    const float mi = m[i] + (g - m[i]) * (float)(1.0 - 0.9);
    const float vi = v[i] * 0.999f + (float)(1.0 - 0.999) * g * g;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2s + 1e-8f;
    param[i] = param[i] - step_size * (mi / denom);
}

__global__ __launch_bounds__(64) void opt_loss_kernel(const float *__restrict__ lossf, int T, int B, const int32_t *__restrict__ ctl,
                                                      int max_iters, float *__restrict__ loss, float *__restrict__ loss_hist,
                                                      float *__restrict__ best_loss, int32_t *__restrict__ flag) {
    const int b = blockIdx.x, lane = threadIdx.x;
    float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int t = lane; t < T; t += 64)
#pragma unroll
        for (int k = 0; k < 5; ++k) s[k] += lossf[(size_t)(b * T + t) * NL + k];
#pragma unroll
    for (int k = 0; k < 5; ++k) s[k] = wave_sum(s[k]);
    if (lane == 0) {
        const float invT = 1.0f / (float)T;
        const float col = s[0] * invT, reg = s[3] + s[1] * invT, regv = s[4] + s[2];
        const float total = col + reg + regv;                      // optimization.py:105-110
        float *o = loss + b * 4;
        o[0] = total; o[1] = col; o[2] = reg; o[3] = regv;
        const int it = ctl[1];
        if (loss_hist && it < max_iters) {
            float *h = loss_hist + ((size_t)it * B + b) * 4;
            h[0] = total; h[1] = col; h[2] = reg; h[3] = regv;
        }
        if (flag) {
            const int better = ctl[0] > 150 && total < best_loss[b];   // :147
            if (better) best_loss[b] = total;
            flag[b] = better;
        }
    }
}

__global__ __launch_bounds__(256) void opt_save_kernel(const float *__restrict__ param, float *__restrict__ best, const int32_t *__restrict__ flag,
                                                       int T, int64_t total, int32_t *__restrict__ ctl) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < total && flag[(i / NP) / T]) best[i] = param[i];       // the parameters AFTER this iteration's step (:148-156)
    if (i == 0) { ctl[0] += 1; ctl[1] += 1; }
}

__global__ __launch_bounds__(64) void opt_finish_kernel(const float *__restrict__ best, float *__restrict__ pose, float *__restrict__ trans,
                                                        float *__restrict__ obj_angles, float *__restrict__ obj_trans) {
    const int64_t n = blockIdx.x;
    const int j = threadIdx.x;
    const float *p = best + n * NP;
    if (j < NJ) rot::matrix_to_axis_angle(p + 9 * j, pose + n * 156 + 3 * j);
    else if (j == NJ) rot::matrix_to_axis_angle(p + OFF_OR, obj_angles + n * 3);
    else if (j < NJ + 4) {
        const int c = j - NJ - 1;
        trans[n * 3 + c] = p[OFF_TR + c];
        obj_trans[n * 3 + c] = p[OFF_OT + c];
    }
}

bool valid(const idf_opt_ctx *c, const idf_opt_state *st) {
    if (!c || !st || !c->geo || !c->geo->smpl || !c->blendT || !c->jv_ptr || !c->jv_vtx || !c->jv_w) return false;
    const idf_smpl_model *m = c->geo->smpl;
    if (m->J != NJ || m->n_betas != 10 || m->KB % 16 != 0 || c->K3P % KSLICE != 0 || c->K3P < 3 * m->V) return false;
    if (st->B < 1 || st->T < 3 || st->P < 1 || (size_t)m->V * 3 * sizeof(float) > 150 * 1024) return false;
    const void *need[] = {st->betas, st->obj_points, st->param, st->init, st->grad, st->m, st->v, st->best, st->pose, st->tr, st->verts,
                          st->vposed, st->verts_gt, st->gv, st->jtr, st->pts, st->y2x, st->y2x_signed, st->near,
                          st->yidx, st->dvposed, st->dA, st->dfeat, st->gtr, st->lossf, st->loss, st->best_loss, st->flag,
                          st->foot_static, st->foot_cnt, st->ctl, st->smpl_ws};
    for (const void *p : need)
        if (!p) return false;
    return st->smpl_ws_bytes >= interdiff_smpl_workspace_bytes(m, (int64_t)st->B * st->T);
}

// loss + gradient at st->param; leaves grad / lossf / loss filled
int loss_grad(const idf_opt_ctx *c, const idf_opt_state *st, void *stream, bool bookkeeping) {
    const idf_smpl_model *m = c->geo->smpl;
    const int64_t N = (int64_t)st->B * st->T;
    const int V = m->V, J = m->J, P = st->P, T = st->T;
    hipStream_t s = idf_stream(stream);
    hipLaunchKernelGGL(opt_unpack_kernel, dim3((unsigned)N), dim3(64), 0, s, st->param, st->pose, st->tr);
    hipLaunchKernelGGL(opt_objpts_kernel, dim3((unsigned)idf_cdiv(P, 256), (unsigned)N), dim3(256), 0, s, st->param, st->obj_points, P, T, st->pts);
    int rc = interdiff_smpl_forward(m, st->pose, st->betas, st->tr, N, st->verts, st->jtr, st->vposed, st->smpl_ws, st->smpl_ws_bytes, stream);
    if (rc) return rc;
    if (c->geo->vorder && st->porder && st->psort && st->pbox) {
        // scan order given: the two questions are asked separately, each with its own exact cull (correction.hip) -- nearest vertex per
        // point by the hook's block-culled scan, "any point within 0.5 m" per vertex against the boxes of 64-point patches
        rc = idf_nn_scan_opt(s, N, T, st->verts, V, st->pts, st->obj_points, P, st->porder, c->geo, st->yidx);            // porder: from interdiff_optimize_init
        if (rc) return rc;
        rc = idf_near_mask_opt(s, N, T, st->verts, V, st->pts, P, st->porder, c->geo->vorder, st->psort, st->pbox, st->near);
        if (rc) return rc;
    } else {
        if (hipMemsetAsync(st->near, 0, (size_t)N * V * sizeof(int32_t), s) != hipSuccess) return IDF_E_LAUNCH;
        hipLaunchKernelGGL(opt_nn_kernel, dim3((unsigned)idf_cdiv(P, NN_T * 2), (unsigned)N), dim3(NN_T), 0, s, st->pts, P, st->verts, V, st->yidx,
                           st->near);
    }
    hipLaunchKernelGGL(opt_signed_kernel, dim3((unsigned)idf_cdiv(P, 256), (unsigned)N), dim3(256), 0, s, st->pts, P, st->verts, V, st->yidx,
                       c->geo->faces, c->geo->adj_ptr, c->geo->adj_face, c->geo->adj_corner, c->geo->adj_pair, st->y2x_signed, st->y2x);
    static std::atomic<uint64_t> lds_ok{0};
    if (idf_opt_in_lds(reinterpret_cast<const void *>(opt_lossgrad_kernel), 150 * 1024, lds_ok) != IDF_OK) return IDF_E_LAUNCH;
    hipLaunchKernelGGL(opt_lossgrad_kernel, dim3((unsigned)N), dim3(LG_T), (size_t)V * 3 * sizeof(float), s, st->verts, st->verts_gt, V,
                       st->y2x_signed, st->near, st->yidx, st->y2x, st->obj_points, P, T, st->ctl, st->gv, st->grad, st->gtr, st->lossf);
    // A [N][J][12] sits behind the feature rows in the SMPL workspace (smpl.hip: interdiff_smpl_forward)
    const float *A = reinterpret_cast<const float *>(reinterpret_cast<const char *>(st->smpl_ws) + idf_align((size_t)N * m->KB * sizeof(float)));
    hipLaunchKernelGGL(opt_skin_vertex_kernel, dim3((unsigned)idf_cdiv(V, 256), (unsigned)N), dim3(256), 0, s, *m, A, st->gv, c->K3P, st->dvposed);
    hipLaunchKernelGGL(opt_skin_joint_kernel, dim3((unsigned)J, (unsigned)N), dim3(64), 0, s, c->jv_ptr, c->jv_vtx, c->jv_w, st->gv, st->vposed, V,
                       J, st->dA);
    const int n_slices = c->K3P / KSLICE;
    hipLaunchKernelGGL(opt_dfeat_kernel, dim3((unsigned)(m->KB / 16), (unsigned)n_slices, (unsigned)idf_cdiv(N, 16 * DF_MT)), dim3(256), 0, s,
                       st->dvposed, c->blendT, (int)N, m->KB, c->K3P, st->dfeat);
    hipLaunchKernelGGL(opt_chain_bwd_kernel, dim3((unsigned)N), dim3(64), 0, s, *m, st->pose, st->betas, st->param, st->dA, st->dfeat, n_slices, N,
                       T, st->jtr, st->foot_static, st->foot_cnt, st->gtr, st->grad, st->lossf);
    hipLaunchKernelGGL(opt_reg_kernel, dim3((unsigned)N), dim3(512), 0, s, st->param, st->init, T, st->grad, st->lossf);
    hipLaunchKernelGGL(opt_loss_kernel, dim3((unsigned)st->B), dim3(64), 0, s, st->lossf, T, st->B, st->ctl, st->max_iters, st->loss,
                       bookkeeping ? st->loss_hist : nullptr, st->best_loss, bookkeeping ? st->flag : nullptr);
    IDF_CHECK_LAUNCH();
    return IDF_OK;
}

}  // namespace

extern "C" int interdiff_optimize_init(const idf_opt_ctx *c, const idf_opt_state *st, const float *pose, const float *trans,
                                       const float *obj_angles, const float *obj_trans, int32_t first_iter, void *stream) {
    if (!valid(c, st) || !pose || !trans || !obj_angles || !obj_trans) return IDF_E_INVAL;
    const idf_smpl_model *m = c->geo->smpl;
    const int64_t N = (int64_t)st->B * st->T;
    hipStream_t s = idf_stream(stream);
    hipLaunchKernelGGL(opt_init_kernel, dim3((unsigned)N), dim3(64), 0, s, pose, trans, obj_angles, obj_trans, st->param, st->init, st->best);
    const size_t pb = (size_t)N * NP * sizeof(float);
    if (hipMemsetAsync(st->m, 0, pb, s) != hipSuccess || hipMemsetAsync(st->v, 0, pb, s) != hipSuccess ||
        hipMemsetAsync(st->grad, 0, pb, s) != hipSuccess ||
        hipMemsetAsync(st->dvposed, 0, (size_t)N * c->K3P * sizeof(float), s) != hipSuccess ||
        hipMemsetAsync(st->flag, 0, (size_t)st->B * sizeof(int32_t), s) != hipSuccess)
        return IDF_E_LAUNCH;
    hipLaunchKernelGGL(opt_setctl_kernel, dim3(1), dim3(64), 0, s, st->ctl, first_iter);
    if (c->geo->vorder && st->porder && idf_point_order(s, st->obj_points, st->B, st->P, st->porder) != IDF_OK) return IDF_E_INVAL;      // the clips' canonical points never change
    // verts_gt / jtr_gt from the ORIGINAL axis-angle pose (optimization.py:43-45)
    const int rc = interdiff_smpl_forward(m, pose, st->betas, trans, N, st->verts_gt, st->jtr, nullptr, st->smpl_ws, st->smpl_ws_bytes, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(opt_static_kernel, dim3((unsigned)st->B), dim3(64), 0, s, st->jtr, st->T, m->J, st->foot_static, st->foot_cnt,
                       st->best_loss);
    IDF_CHECK_LAUNCH();
    return IDF_OK;
}

extern "C" int interdiff_optimize_loss_grad(const idf_opt_ctx *c, const idf_opt_state *st, void *stream) {
    if (!valid(c, st)) return IDF_E_INVAL;
    return loss_grad(c, st, stream, false);
}

extern "C" int interdiff_optimize_step(const idf_opt_ctx *c, const idf_opt_state *st, void *stream) {
    if (!valid(c, st)) return IDF_E_INVAL;
    const int rc = loss_grad(c, st, stream, true);
    if (rc) return rc;
    const int64_t total = (int64_t)st->B * st->T * NP;
    hipStream_t s = idf_stream(stream);
    hipLaunchKernelGGL(opt_adam_kernel, dim3((unsigned)idf_cdiv(total, 256)), dim3(256), 0, s, st->param, st->grad, st->m, st->v, total, st->ctl);
    hipLaunchKernelGGL(opt_save_kernel, dim3((unsigned)idf_cdiv(total, 256)), dim3(256), 0, s, st->param, st->best, st->flag, st->T, total, st->ctl);
    IDF_CHECK_LAUNCH();
    return IDF_OK;
}

extern "C" int interdiff_optimize_finish(const idf_opt_ctx *c, const idf_opt_state *st, float *pose, float *trans, float *obj_angles,
                                         float *obj_trans, void *stream) {
    if (!valid(c, st) || !pose || !trans || !obj_angles || !obj_trans) return IDF_E_INVAL;
    hipLaunchKernelGGL(opt_finish_kernel, dim3((unsigned)((int64_t)st->B * st->T)), dim3(64), 0, idf_stream(stream), st->best, pose, trans,
                       obj_angles, obj_trans);
    IDF_CHECK_LAUNCH();
    return IDF_OK;
}

extern "C" int interdiff_debug_joint_map_vjp(const float *R, const float *g_out, float *g_in, int32_t n) {
    if (!R || !g_out || !g_in || n < 0) return IDF_E_INVAL;
    for (int i = 0; i < n; ++i) rotd::joint_map_vjp(R + 9 * i, g_out + 9 * i, g_in + 9 * i);
    return IDF_OK;
}
