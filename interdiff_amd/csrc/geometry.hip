// Vertex normals + brute-force nearest neighbour + signed point-to-point distance
// (rows B4, B5 of SURVEY.md §8).
//
// vertex_normals restates data/tools.py:4-40 WITHOUT its three scatter-adds: a vertex->face
// adjacency (CSR, built once on the host in the reference's accumulation order: corner 1, corner 2,
// corner 0, ascending face index inside each) turns it into a deterministic gather.
// nn_argmin replaces the third-party chamfer_distance CUDA op (tools.py:45-47; only its indices are
// used): exact argmin of d2 = (dx*dx + dy*dy) + dz*dz, every operation individually rounded (no
// FMA contraction -> bit-identical to the CPU oracle), lowest index wins ties.
#include "common.h"
#include <float.h>

namespace {

__device__ __forceinline__ float3 ld3(const float *p) { return make_float3(p[0], p[1], p[2]); }
__device__ __forceinline__ float3 sub3(float3 a, float3 b) { return make_float3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ float3 cross3(float3 a, float3 b) {
    return make_float3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

// un-normalised area-weighted normal of vertex v (sum over incident faces, reference order)
__device__ __forceinline__ float3 vertex_normal_sum(const float *__restrict__ vf, const int32_t *__restrict__ faces,
                                                    const int32_t *__restrict__ adj_ptr, const int32_t *__restrict__ adj_face,
                                                    const int32_t *__restrict__ adj_corner, int v) {
    float3 acc = make_float3(0.f, 0.f, 0.f);
    for (int e = adj_ptr[v]; e < adj_ptr[v + 1]; ++e) {
        const int f = adj_face[e], c = adj_corner[e];
        const float3 p0 = ld3(vf + 3 * faces[3 * f]), p1 = ld3(vf + 3 * faces[3 * f + 1]), p2 = ld3(vf + 3 * faces[3 * f + 2]);
        float3 n;
        if (c == 1) n = cross3(sub3(p2, p1), sub3(p0, p1));
        else if (c == 2) n = cross3(sub3(p0, p2), sub3(p1, p2));
        else n = cross3(sub3(p1, p0), sub3(p2, p0));
        acc.x += n.x; acc.y += n.y; acc.z += n.z;
    }
    return acc;
}

__global__ __launch_bounds__(256) void vertex_normals_kernel(const float *__restrict__ verts, int V,
                                                             const int32_t *__restrict__ faces,
                                                             const int32_t *__restrict__ adj_ptr,
                                                             const int32_t *__restrict__ adj_face,
                                                             const int32_t *__restrict__ adj_corner,
                                                             float *__restrict__ normals) {
    const int v = blockIdx.x * 256 + threadIdx.x;
    const int64_t n = blockIdx.y;
    if (v >= V) return;
    const float3 a = vertex_normal_sum(verts + (size_t)n * V * 3, faces, adj_ptr, adj_face, adj_corner, v);
    const float nn = fmaxf(sqrtf(a.x * a.x + a.y * a.y + a.z * a.z), 1e-6f);          // F.normalize(eps=1e-6)
    float *o = normals + ((size_t)n * V + v) * 3;
    o[0] = a.x / nn; o[1] = a.y / nn; o[2] = a.z / nn;
}

// squared distance with every op rounded separately (matches the oracle bit for bit)
__device__ __forceinline__ float dist2_exact(float qx, float qy, float qz, float rx, float ry, float rz) {
#pragma clang fp contract(off)
    const float dx = qx - rx, dy = qy - ry, dz = qz - rz;
    const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
    return (xx + yy) + zz;
}

constexpr int QPT = 4;            // queries per thread
constexpr int RC = 1024;          // reference points per LDS chunk

__global__ __launch_bounds__(256) void nn_argmin_kernel(const float *__restrict__ q, int Pq, const float *__restrict__ r,
                                                        int Pr, int32_t *__restrict__ idx) {
    __shared__ float4 rs[RC];
    const int64_t n = blockIdx.y;
    const float *qn = q + (size_t)n * Pq * 3, *rn = r + (size_t)n * Pr * 3;
    const int q0 = blockIdx.x * 256 * QPT + threadIdx.x;
    float qx[QPT], qy[QPT], qz[QPT], best[QPT];
    int bi[QPT];
#pragma unroll
    for (int k = 0; k < QPT; ++k) {
        const int i = q0 + 256 * k;
        const bool ok = i < Pq;
        qx[k] = ok ? qn[3 * i] : 0.f; qy[k] = ok ? qn[3 * i + 1] : 0.f; qz[k] = ok ? qn[3 * i + 2] : 0.f;
        best[k] = FLT_MAX;
        bi[k] = 0;
    }
    for (int c0 = 0; c0 < Pr; c0 += RC) {
        const int cn = min(RC, Pr - c0);
        __syncthreads();
        for (int j = threadIdx.x; j < cn; j += 256) rs[j] = make_float4(rn[3 * (c0 + j)], rn[3 * (c0 + j) + 1], rn[3 * (c0 + j) + 2], 0.f);
        __syncthreads();
        for (int j = 0; j < cn; ++j) {
            const float4 p = rs[j];
#pragma unroll
            for (int k = 0; k < QPT; ++k) {
                const float d2 = dist2_exact(qx[k], qy[k], qz[k], p.x, p.y, p.z);
                if (d2 < best[k]) { best[k] = d2; bi[k] = c0 + j; }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < QPT; ++k) {
        const int i = q0 + 256 * k;
        if (i < Pq) idx[(size_t)n * Pq + i] = bi[k];
    }
}

// after the search: vector to the nearest point, its norm and (optionally) the sign from the
// normal at the nearest point  (tools.py:49-71)
__global__ __launch_bounds__(256) void p2p_finish_kernel(const float *__restrict__ a, int Pa, const float *__restrict__ b, int Pb,
                                                         const int32_t *__restrict__ idx, const float *__restrict__ b_normals,
                                                         float *__restrict__ signed_d, float *__restrict__ vec) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int64_t n = blockIdx.y;
    if (i >= Pa) return;
    const int j = idx[(size_t)n * Pa + i];
    const float *pa = a + ((size_t)n * Pa + i) * 3, *pb = b + ((size_t)n * Pb + j) * 3;
    const float vx = pa[0] - pb[0], vy = pa[1] - pb[1], vz = pa[2] - pb[2];
    float d = sqrtf(vx * vx + vy * vy + vz * vz);
    if (b_normals) {
        const float *nb = b_normals + ((size_t)n * Pb + j) * 3;
        const float dt = nb[0] * vx + nb[1] * vy + nb[2] * vz;
        d *= dt > 0.f ? 1.f : (dt < 0.f ? -1.f : 0.f);
    }
    signed_d[(size_t)n * Pa + i] = d;
    if (vec) {
        float *o = vec + ((size_t)n * Pa + i) * 3;
        o[0] = vx; o[1] = vy; o[2] = vz;
    }
}

}  // namespace

extern "C" int interdiff_vertex_normals(const float *verts, int64_t N, int32_t V, const int32_t *faces, const int32_t *adj_ptr,
                                        const int32_t *adj_face, const int32_t *adj_corner, float *normals, void *stream) {
    if (!verts || !faces || !adj_ptr || !adj_face || !adj_corner || !normals || N < 0 || V <= 0) return IDF_E_INVAL;
    if (N == 0) return IDF_OK;
    hipLaunchKernelGGL(vertex_normals_kernel, dim3((unsigned)idf_cdiv(V, 256), (unsigned)N), dim3(256), 0, idf_stream(stream), verts,
                       V, faces, adj_ptr, adj_face, adj_corner, normals);
    IDF_CHECK_LAUNCH();
    return IDF_OK;
}

extern "C" int interdiff_nn_argmin(const float *q, int32_t Pq, const float *r, int32_t Pr, int64_t N, int32_t *idx, void *stream) {
    if (!q || !r || !idx || Pq <= 0 || Pr <= 0 || N < 0) return IDF_E_INVAL;
    if (N == 0) return IDF_OK;
    hipLaunchKernelGGL(nn_argmin_kernel, dim3((unsigned)idf_cdiv(Pq, 256 * QPT), (unsigned)N), dim3(256), 0, idf_stream(stream), q, Pq,
                       r, Pr, idx);
    IDF_CHECK_LAUNCH();
    return IDF_OK;
}

extern "C" int interdiff_point2point_signed(const float *x, int32_t P1, const float *y, int32_t P2, int64_t N,
                                            const float *x_normals, const float *y_normals, float *y2x_signed,
                                            float *x2y_signed, int32_t *yidx, int32_t *xidx, float *y2x, float *x2y,
                                            void *stream) {
    if (!x || !y || !y2x_signed || !x2y_signed || !yidx || !xidx || P1 <= 0 || P2 <= 0 || N < 0) return IDF_E_INVAL;
    if (N == 0) return IDF_OK;
    hipStream_t s = idf_stream(stream);
    int rc = interdiff_nn_argmin(x, P1, y, P2, N, xidx, stream);          // nearest y for each x
    if (rc) return rc;
    rc = interdiff_nn_argmin(y, P2, x, P1, N, yidx, stream);              // nearest x for each y
    if (rc) return rc;
    hipLaunchKernelGGL(p2p_finish_kernel, dim3((unsigned)idf_cdiv(P2, 256), (unsigned)N), dim3(256), 0, s, y, P2, x, P1, yidx, x_normals,
                       y2x_signed, y2x);
    hipLaunchKernelGGL(p2p_finish_kernel, dim3((unsigned)idf_cdiv(P1, 256), (unsigned)N), dim3(256), 0, s, x, P1, y, P2, xidx, y_normals,
                       x2y_signed, x2y);
    IDF_CHECK_LAUNCH();
    return IDF_OK;
}
