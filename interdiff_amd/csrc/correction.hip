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
#include "objproj.h"
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

// ---- K3: scan order of a clip's object points ---------------------------------------------------------------
// porder[b][s] = index of the clip's s-th canonical object point in Morton order (10 bits per axis inside the clip's bounding box).
// The contact scan culls vertex blocks per WAVE; a wave that owns a compact patch of the object (128 consecutive points of this
// order) agrees on which blocks it needs.  A rigid transform keeps the patch compact, so the order is per clip, not per frame.
// One workgroup per clip: bounding box by a block reduction, 41-bit keys (code << 11 | index: unique, so the sort is stable by
// construction), bitonic sort of 2048 keys in LDS.  ~10 us per call for all clips.
constexpr int MAXP = 2048;             // object points per frame handled by one workgroup (CT * QP)
__global__ __launch_bounds__(1024) void corr_point_order_kernel(const float *__restrict__ obj_points, int P, int32_t *__restrict__ porder) {
    __shared__ unsigned long long keys[MAXP];
    __shared__ float red[6][1024];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float *op = obj_points + (size_t)b * P * 3;
    float p[2][3];
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int i = tid + 1024 * k;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            p[k][a] = i < P ? op[3 * i + a] : 0.f;
            if (i < P) { lo[a] = fminf(lo[a], p[k][a]); hi[a] = fmaxf(hi[a], p[k][a]); }
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) { red[a][tid] = lo[a]; red[3 + a][tid] = hi[a]; }
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (tid < s) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                red[a][tid] = fminf(red[a][tid], red[a][tid + s]);
                red[3 + a][tid] = fmaxf(red[3 + a][tid], red[3 + a][tid + s]);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int i = tid + 1024 * k;
        unsigned long long key = ~0ull;                                   // slots past P sort to the end
        if (i < P) {
            unsigned code = 0;
            unsigned q[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float ext = red[3 + a][0] - red[a][0];
                const float u = ext > 0.f ? (p[k][a] - red[a][0]) / ext : 0.f;
                q[a] = (unsigned)fminf(fmaxf(u * 1023.0f, 0.f), 1023.0f);      // NaN -> 0 (fmaxf returns the non-NaN operand)
            }
#pragma unroll
            for (int bit = 0; bit < 10; ++bit)
#pragma unroll
                for (int a = 0; a < 3; ++a) code |= ((q[a] >> bit) & 1u) << (3 * bit + a);
            key = ((unsigned long long)code << 11) | (unsigned)i;
        }
        keys[i] = key;
    }
    __syncthreads();
    for (int k = 2; k <= MAXP; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int l = 2 * j * (tid / j) + (tid % j), r = l + j;
            const unsigned long long x = keys[l], y = keys[r];
            const bool up = (l & k) == 0;
            if ((x > y) == up) { keys[l] = y; keys[r] = x; }
            __syncthreads();
        }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int i = tid + 1024 * k;
        if (i < P) porder[(size_t)b * P + i] = (int32_t)(keys[i] & 2047u);
    }
}

// ---- K4: per-frame contact analysis -------------------------------------------------------------------
// One workgroup (16 waves) per frame.  The frame's 2048 object points are cut into TASKS of 64 consecutive points of the clip's
// Morton order (K3) -- one point per lane -- and the waves pull tasks from an LDS counter: a patch that touches the body needs
// several times the vertex blocks of one that floats free, and a static split of 128 points per wave left the workgroup waiting
// for its slowest wave (tools/contact_probe.py, first version: thread 0 spent 220 k of 450 k cycles at the first barrier).
// Arithmetic is the exact (dx*dx + dy*dy) + dz*dz without FMA contraction, so the argmin is bit-identical to geometry.hip and to
// the oracle.  The fp32 VALU of gfx950 runs plain ops at 16 lanes per clock and PACKED ops (v_pk_add/mul_f32) at twice that, so
// every distance / box instruction works on a PAIR: two consecutive vertices (or two boxes) against the lane's one point --
// records are stored as pairs {x0 x1 y0 y1 | z0 z1 - -}, the point as {q, q}.  (Measured: one point per lane on plain ops cost
// 4.7 cycles per instruction per wave, exactly what a packed instruction costs.)
//
// EXACT block culling (round 3).  The vertex records sit in LDS in SCAN ORDER -- the Morton order of the REST pose, chosen at
// pack time (ctx->vorder): skinning is spatially smooth, so CB consecutive records stay a compact clump under any pose -- and
// each block of CB = 16 records gets its axis-aligned box, built in-kernel from the frame's posed vertices.  Per point:
//   seed    the distance to the FIRST record of every 4th block (108 real distances: a valid upper bound of the minimum), widened by the scan's margin;
//   scan    two levels: super-blocks of SB = 8 blocks (128 records) are tested first, their blocks only if some point needs them.
//           For a box: lower bound lb = |max(lo - q, q - hi, 0)|^2 -- fp32 subtraction is monotone, so each component is at most that of
//           any vertex inside the box.  A box is skipped iff lb > thr (STRICTLY) for every point of the WAVE (one ballot, one uniform
//           branch).  Rounds 3-5 scored with the exact distance E = (dx dx + dy dy) + dz dz and compared lb <= best bit for bit; round 6
//           RANKS with A = fma(dz, dz, fma(dy, dy, dx dx)) (7 instead of 9 instructions per record pair; |A / E - 1| <= 6 ulp-halves, both being
//           three roundings of the same non-negative sum) and carries a margin of 2^-18 (+1e-37) in every comparison that drops something:
//           thr = min(seed, smallest block minimum so far) x (1 + 2^-18).  An executed block is scored with v_min3_f32 over vertex pairs,
//           index bookkeeping once per block; a block whose minimum comes within the margin of the smallest one is remembered as the runner-up.
//   resolve the winning block -- and the runner-up, ~1e-4 of the points -- is re-scored with the EXACT E; among the records at the exact minimum
//           the LOWEST ORIGINAL vertex index wins.  A point with more than one runner-up (~1e-8) is rescanned over all records with E.  Exactly
//           the brute force's lowest-index-wins rule, independent of the scan order and of the ranking distance.  (A rescan costs its
//           workgroup as much as the whole frame: with every near-tie rescanned the kernel was 10 % SLOWER than the exact-ranked scan.)
// Then the normal of the nearest vertex from its incident faces (adjacency as (a, b) record pairs: ONE 8-byte load per face instead
// of face id -> corner -> three vertex ids), the signed distance, the marker distances.  The per-point loss goes to LDS by scan
// position and is reduced in a fixed tree: deterministic whatever wave scored the point.
constexpr int MAXM = 128;
// sqrtf(d2) < 0.02f  <=>  d2 < MARK_D2: the smallest float whose (correctly rounded) square root is >= 0.02f -- 0x39d1b716, one ulp below 0.02f * 0.02f
constexpr float MARK_D2 = 0.00039999996079131961f;
constexpr int CB = 16;                 // vertices per culling / index-bookkeeping block
constexpr int SB = 8;                  // blocks per super-block (128 vertices): its box is tested first, 8 block tests are skipped at once
constexpr int SEED_STRIDE = 4;         // the seed looks at the first record of every 4th block
constexpr int CT = 1024;               // threads per workgroup
constexpr int TASK = 64;               // points per task = one per lane
typedef float v2f __attribute__((ext_vector_type(2)));

// LDS image of the frame (float4 units).  Records: pair p = scan positions 2p, 2p+1 -> {x0 x1 y0 y1}, {z0 z1 - -}.  Boxes: pair j = boxes
// 2j, 2j+1 -> {lox0 lox1 loy0 loy1}, {loz0 loz1 hix0 hix1}, {hiy0 hiy1 hiz0 hiz1}.
struct ContactLds {
    int V4, nCB, nSB, nSeed;
    int rec, mark, box, sbox, seed, flags, red, pl, total;         // offsets in float4 units
    __host__ __device__ explicit ContactLds(int V) {
        V4 = (V + CB - 1) / CB * CB; nCB = V4 / CB; nSB = (nCB + SB - 1) / SB; nSeed = (nCB + SEED_STRIDE - 1) / SEED_STRIDE;
        rec = 0;
        mark = rec + V4;                             // V4 / 2 pairs x 2 float4
        box = mark + MAXM;
        sbox = box + 3 * (nSB * SB / 2);             // block boxes as pairs, padded with dummies to whole super-blocks
        seed = sbox + 3 * ((nSB + 2) / 2);
        flags = seed + 2 * ((nSeed + 1) / 2);        // [MAXM] int, [CT] float, [MAXP] float: in the dynamic region since round 6 (the launch that carries the predictor's
        red = flags + MAXM / 4;                      // workgroups shares ONE LDS size between both roles: statics of one role would sit on top of the other's 152 KB)
        pl = red + CT / 4;
        total = pl + MAXP / 4;
    }
};

__device__ __forceinline__ float3 lds_rec(const float4 *vs, int pos) {                // record at scan position pos
    const float *f = reinterpret_cast<const float *>(vs + 2 * (pos >> 1));
    const int h = pos & 1;
    return make_float3(f[h], f[2 + h], f[4 + h]);
}
__device__ __forceinline__ void lds_rec_store(float4 *vs, int pos, float x, float y, float z) {
    float *f = reinterpret_cast<float *>(vs + 2 * (pos >> 1));
    const int h = pos & 1;
    f[h] = x; f[2 + h] = y; f[4 + h] = z;
}
__device__ __forceinline__ void lds_box_store(float4 *bx, int k, float3 lo, float3 hi) {   // box k of a pair array
    float *f = reinterpret_cast<float *>(bx + 3 * (k >> 1));
    const int h = k & 1;
    f[h] = lo.x; f[2 + h] = lo.y; f[4 + h] = lo.z; f[6 + h] = hi.x; f[8 + h] = hi.y; f[10 + h] = hi.z;
}
__device__ __forceinline__ void lds_box_load(const float4 *bx, int k, float3 &lo, float3 &hi) {
    const float *f = reinterpret_cast<const float *>(bx + 3 * (k >> 1));
    const int h = k & 1;
    lo = make_float3(f[h], f[2 + h], f[4 + h]);
    hi = make_float3(f[6 + h], f[8 + h], f[10 + h]);
}

// OPT = the instantiation of the physics post-optimisation (optimize.hip, optimization.py:64-65): the object points come already
// transformed per frame (pts_frame [N][P][3]), frames are clip-major (clip = n / frames_per_clip) and the only output is the nearest
// vertex of every point -- no normals, markers or loss (the per-vertex contact-radius mask of :74-75 is its own kernel there).
// WITH_PRED (the correction hook, round 6): the launch carries `pred.nclips` extra LEADING workgroups that run the contact-frame predictor's three stacks (csrc/objproj.h PART 1: one
// clip each, 250 us, markers gathered from the posed vertices) beside the scan's N frame workgroups -- the predictor needs the markers only, the contact labels only pick which node's
// IDCT is evaluated afterwards (idf_objproj_pick).  Leading, so they are dispatched first; one launch, so nothing depends on a second queue (a side stream was tried first: its
// blocked barrier packet cost the caller's queue ~1 us per kernel launch for as long as the host ran ahead -- +27 us per step inside the sampler, profiles/r06_hook_overlap.txt).
struct PredArgs {
    idf_objproj op;
    const float *obj_angles, *obj_trans;      // [T][B][6], [T][B][3]
    const int32_t *markers_idx;               // [M] original vertex ids
    float *markers, *keep;                    // [N][M][3] (written here, frames time-major), [B][idf_objproj_keep_floats()]
    int nclips;
};
template <bool OPT, bool WITH_PRED = false>
__global__ __launch_bounds__(CT) void corr_contact_kernel(const float *__restrict__ verts, int V,
                                                          const float *__restrict__ obj_points, int P,
                                                          const int32_t *__restrict__ porder /* nullable [B][P]: scan position -> point */,
                                                          const float *__restrict__ objR, const float *__restrict__ objT,
                                                          const int32_t *__restrict__ faces /* identity order only */,
                                                          const int32_t *__restrict__ adj_ptr,
                                                          const int32_t *__restrict__ adj_face,
                                                          const int32_t *__restrict__ adj_corner,
                                                          const int32_t *__restrict__ adj_pair /* nullable [nnz][2]: per incident face the scan positions (a, b): normal += (a - v) x (b - v) */,
                                                          const int32_t *__restrict__ vorder /* nullable [V]: scan position -> vertex */,
                                                          const int32_t *__restrict__ vrank /* nullable [V]: vertex -> scan position (the inverse): the frame's vertices are then read in their own order */,
                                                          const int32_t *__restrict__ markers_pos /* scan positions */, int M, int B,
                                                          float *__restrict__ markers_out, float *__restrict__ loss_sum,
                                                          float *__restrict__ min_dist, int32_t *__restrict__ label,
                                                          float *__restrict__ o2h_out /* nullable [N][P] */,
                                                          int32_t *__restrict__ idx_out /* nullable [N][P]: nearest vertex */,
                                                          unsigned long long *__restrict__ stats /* nullable [IDF_CONTACT_STATS]: see interdiff_contact_nn */,
                                                          int64_t nn_from /* frames below it skip the NN scan */,
                                                          const float *__restrict__ pts_frame /* OPT: [N][P][3] */, int frames_per_clip /* OPT */,
                                                          const PredArgs pred) {
    extern __shared__ __attribute__((aligned(16))) float4 vs[];
    if constexpr (WITH_PRED) {
        if ((int)blockIdx.x < pred.nclips) {
            // predictor role: this clip's markers out of the posed vertices (what the scan's workgroups also write when asked to), then the three stacks
            const int b = blockIdx.x, Tn = pred.op.T;
            for (int i = threadIdx.x; i < Tn * M; i += CT) {
                const int t = i / M, m = i - t * M;
                const size_t nfr = (size_t)t * B + b;
                const float *v = verts + (nfr * V + pred.markers_idx[m]) * 3;
                float *o = pred.markers + (nfr * M + m) * 3;
                o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
            }
            __threadfence();
            __syncthreads();
            idf_objproj_dev::objproj_body<1>(reinterpret_cast<float *>(vs), pred.op, pred.obj_angles, pred.obj_trans, pred.markers, nullptr, B, b, pred.keep, nullptr);
            return;
        }
    }
    const ContactLds L(V);
    const int nCB = L.nCB, nSB = L.nSB;
    float4 *ms = vs + L.mark, *bb = vs + L.box, *sbb = vs + L.sbox, *sd = vs + L.seed;
    int *flags = reinterpret_cast<int *>(vs + L.flags);
    float *red = reinterpret_cast<float *>(vs + L.red);
    float *pl = reinterpret_cast<float *>(vs + L.pl);                       // per-point loss by scan position
    __shared__ int next_task;
    const int64_t n = WITH_PRED ? (int64_t)blockIdx.x - pred.nclips : (int64_t)blockIdx.x;
    const int b = OPT ? (int)(n / frames_per_clip) : (int)(n % B), tid = threadIdx.x;
    const float *vf = verts + (size_t)n * V * 3;
    const bool do_nn = n >= nn_from;
    // phase clocks of thread 0 (frames that scan only), summed into stats[3 + phase] when the caller asks for statistics
    long long t_prev = 0;
    int n_phase = 0;
    auto phase_done = [&]() {
        if (stats && tid == 0 && do_nn) {
            const long long now = (long long)__builtin_readcyclecounter();
            if (n_phase > 0) atomicAdd(stats + 3 + n_phase, (unsigned long long)(now - t_prev));
            t_prev = now;
        }
        ++n_phase;
    };
    phase_done();
    // (requested first, consumed behind the record gather: the object transform of the frame and this thread's marker position)
    float R[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f}, tr[3] = {0.f, 0.f, 0.f};
    if constexpr (!OPT) {
#pragma unroll
        for (int k = 0; k < 9; ++k) R[k] = objR[n * 9 + k];
#pragma unroll
        for (int k = 0; k < 3; ++k) tr[k] = objT[n * 3 + k];
    }
    const int my_marker_pos = (!OPT && tid < M) ? markers_pos[tid] : 0;
    // records -> LDS, four scan positions per thread and trip: all four permutation loads fly together, then all twelve coordinate loads (a gather from the skinning
    // kernel's output through two dependent round trips per TRIP instead of per position: 21 k -> ~8 k cycles of a ~350 k-cycle workgroup, tools/contact_probe.py).  The spare
    // words of a record pair carry the ORIGINAL vertex ids of its two records: the resolve step below reads them from LDS instead of gathering vorder[] per candidate.
    if (vorder && vrank) {
        // with the inverse permutation: vertices in THEIR order (a wave's 12-byte loads cover 768 contiguous bytes instead of 64 cache lines -- the gather below is bound by
        // line requests: 19 k cycles per workgroup), scattered into LDS by scan position
        for (int i0 = tid; i0 < V; i0 += 4 * CT) {
            typedef float f3v __attribute__((ext_vector_type(3), aligned(4)));
            int pos[4];
            f3v pv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = min(i0 + k * CT, V - 1);
                pos[k] = vrank[i];
                pv[k] = *reinterpret_cast<const f3v *>(vf + 3 * i);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (i0 + k * CT < V) {
                    lds_rec_store(vs, pos[k], pv[k].x, pv[k].y, pv[k].z);
                    reinterpret_cast<int *>(vs + 2 * (pos[k] >> 1) + 1)[2 + (pos[k] & 1)] = i0 + k * CT;
                }
            }
        }
        for (int v = V + tid; v < L.V4; v += CT) {                         // past the end: far away, no vertex
            lds_rec_store(vs, v, 3e18f, 3e18f, 3e18f);
            reinterpret_cast<int *>(vs + 2 * (v >> 1) + 1)[2 + (v & 1)] = 0x7fffffff;
        }
    } else
    for (int v0 = tid; v0 < L.V4; v0 += 4 * CT) {
        int o[4];
        float x[4], y[4], z[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int v = v0 + k * CT;
            o[k] = v < V ? (vorder ? vorder[v] : v) : -1;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            x[k] = y[k] = z[k] = 3e18f;                                       // past the end: far away
            if (o[k] >= 0) {                                                  // ONE 12-byte load per vertex (global_load_dwordx3): a scattered gather is bound by instructions x lanes, not bytes
                typedef float f3v __attribute__((ext_vector_type(3), aligned(4)));
                const f3v pv = *reinterpret_cast<const f3v *>(vf + 3 * o[k]);
                x[k] = pv.x; y[k] = pv.y; z[k] = pv.z;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int v = v0 + k * CT;
            if (v < L.V4) {
                lds_rec_store(vs, v, x[k], y[k], z[k]);
                reinterpret_cast<int *>(vs + 2 * (v >> 1) + 1)[2 + (v & 1)] = o[k] >= 0 ? o[k] : 0x7fffffff;
            }
        }
    }
    if (tid < MAXM) flags[tid] = 0;
    // gridDim.y workgroups share a frame's tasks (OPT: 320 frames on 256 CUs would otherwise take two full rounds)
    const int ntask_all = (P + TASK - 1) / TASK;
    const int task0 = (int)((long long)blockIdx.y * ntask_all / gridDim.y), task1 = (int)((long long)(blockIdx.y + 1) * ntask_all / gridDim.y);
    if (tid == 0) next_task = task0;
    for (int i = tid; i < MAXP; i += CT) pl[i] = 0.f;
    __syncthreads();
    phase_done();                                                          // 1: records in LDS
    if (!OPT && tid < ((M + 1) & ~1)) {                                    // markers as PAIRS, like the vertex records: pair p = markers 2p, 2p + 1 -> {x0 x1 y0 y1}, {z0 z1 - -}; a missing second marker sits far away
        float3 mk = make_float3(3e18f, 3e18f, 3e18f);
        if (tid < M) {
            mk = lds_rec(vs, my_marker_pos);
            if (markers_out) {                                             // (the hook's overlapped route gathers them in corr_markers_kernel, ahead of this kernel)
                float *mo = markers_out + ((size_t)n * M + tid) * 3;
                mo[0] = mk.x; mo[1] = mk.y; mo[2] = mk.z;
            }
        }
        lds_rec_store(ms, tid, mk.x, mk.y, mk.z);
    }
    if (do_nn) {
        // boxes: a wave takes the 64 record pairs of one super-block (8 blocks x 8 pairs), every lane the min / max of its pair; an 8-lane
        // reduction on the DPP path gives the block boxes, the wave reduction the super-block box.  (A first version -- one thread per
        // block reading its 16 records one by one, then one thread per super-block -- took 25 k cycles of a 390 k-cycle workgroup.)
#define IDF_DPP8(v, op, ctrl) v = op(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false)))
        const int lane_ = tid & 63, wave_ = tid >> 6, nsw = 2 * ((nSB + 2) / 2);
        for (int sb = wave_; sb < nsw; sb += CT / 64) {
            const int pr = sb * 64 + lane_;                                  // record pair
            float lx = FLT_MAX, ly = FLT_MAX, lz = FLT_MAX, hx = -FLT_MAX, hy = -FLT_MAX, hz = -FLT_MAX;
            if (2 * pr < V) {
                const float4 xy = vs[2 * pr], zz = vs[2 * pr + 1];
                const bool two = 2 * pr + 1 < V;
                lx = two ? fminf(xy.x, xy.y) : xy.x; hx = two ? fmaxf(xy.x, xy.y) : xy.x;
                ly = two ? fminf(xy.z, xy.w) : xy.z; hy = two ? fmaxf(xy.z, xy.w) : xy.z;
                lz = two ? fminf(zz.x, zz.y) : zz.x; hz = two ? fmaxf(zz.x, zz.y) : zz.x;
            }
            // quad xor 1, quad xor 2, half-row mirror: every lane of an 8-lane group ends with the group's value
            IDF_DPP8(lx, fminf, 0xB1); IDF_DPP8(lx, fminf, 0x4E); IDF_DPP8(lx, fminf, 0x141);
            IDF_DPP8(ly, fminf, 0xB1); IDF_DPP8(ly, fminf, 0x4E); IDF_DPP8(ly, fminf, 0x141);
            IDF_DPP8(lz, fminf, 0xB1); IDF_DPP8(lz, fminf, 0x4E); IDF_DPP8(lz, fminf, 0x141);
            IDF_DPP8(hx, fmaxf, 0xB1); IDF_DPP8(hx, fmaxf, 0x4E); IDF_DPP8(hx, fmaxf, 0x141);
            IDF_DPP8(hy, fmaxf, 0xB1); IDF_DPP8(hy, fmaxf, 0x4E); IDF_DPP8(hy, fmaxf, 0x141);
            IDF_DPP8(hz, fmaxf, 0xB1); IDF_DPP8(hz, fmaxf, 0x4E); IDF_DPP8(hz, fmaxf, 0x141);
            const int cb = sb * SB + (lane_ >> 3);
            if ((lane_ & 7) == 0 && cb < nSB * SB) lds_box_store(bb, cb, make_float3(lx, ly, lz), make_float3(hx, hy, hz));
            const float slx = -wave_max(-lx), sly = -wave_max(-ly), slz = -wave_max(-lz), shx = wave_max(hx), shy = wave_max(hy), shz = wave_max(hz);
            if (lane_ == 0) lds_box_store(sbb, sb, make_float3(slx, sly, slz), make_float3(shx, shy, shz));
        }
#undef IDF_DPP8
        for (int k = tid; k < 2 * ((L.nSeed + 1) / 2); k += CT) {           // seed records: the first record of every SEED_STRIDE-th block, as pairs
            const float3 p = lds_rec(vs, min(k * SEED_STRIDE, nCB - 1) * CB);
            lds_rec_store(sd, k, p.x, p.y, p.z);
        }
    }
    const float *op = OPT ? pts_frame + (size_t)n * P * 3 : obj_points + (size_t)b * P * 3;
    const int ln = tid & 63, ntask = task1;
    __syncthreads();
    phase_done();                                                          // 2: boxes, markers
    float mind2 = FLT_MAX;                                                  // squared distance to the nearest marker over this thread's points
    unsigned n_exec = 0, n_test = 0, n_tasks = 0;
    for (;;) {
        int task = 0;
        if (ln == 0) task = atomicAdd(&next_task, 1);
        task = __builtin_amdgcn_readfirstlane(task);
        if (task >= ntask) break;
        ++n_tasks;
        const int sp = task * TASK + ln;                                   // scan position of this lane's point
        const int i = sp < P ? (porder ? porder[(size_t)b * P + sp] : sp) : -1;
        const bool valid = i >= 0;
        float px = 0.f, py = 0.f, pz = 0.f;
        if (valid) { px = op[3 * i]; py = op[3 * i + 1]; pz = op[3 * i + 2]; }
        // matmul(points, R^T) + t  (eval_smpl_short.py:107); OPT: the caller's transformed points as they are
        const float qx = OPT ? px : (px * R[0] + py * R[1] + pz * R[2]) + tr[0];
        const float qy = OPT ? py : (px * R[3] + py * R[4] + pz * R[5]) + tr[1];
        const float qz = OPT ? pz : (px * R[6] + py * R[7] + pz * R[8]) + tr[2];
        // the signed object->human distance is only consumed on future frames (eval_smpl_short.py:121 slices
        // loss_dist_o[past_len:]); past frames only need the marker distances below
        if (do_nn) {
#pragma clang fp contract(off)
            const v2f QX = v2f{qx, qx}, QY = v2f{qy, qy}, QZ = v2f{qz, qz};
            // exact squared distances of the point to a record pair
            auto pair_d2 = [&](const float4 xy, const float4 zz) {
#pragma clang fp contract(off)
                const v2f dx = QX - v2f{xy.x, xy.y}, dy = QY - v2f{xy.z, xy.w}, dz = QZ - v2f{zz.x, zz.y};
                return (dx * dx + dy * dy) + dz * dz;
            };
            // The SCAN ranks vertices by a cheaper distance (round 6): A = fma(dz, dz, fma(dy, dy, dx * dx)) -- 7 instead of 9 instructions per record pair, 3 roundings like the
            // exact E = (dx * dx + dy * dy) + dz * dz over the same rounded differences, so |A / E - 1| <= 6 u (u = 2^-24; all terms non-negative, no cancellation).  It only FILTERS:
            // every comparison that could drop a vertex carries the margin SCAN_M = 1 + 2^-18 (64 u) plus 1e-37 for the subnormal range, the winning block is re-scored with E, and a
            // point whose runner-up block came within the margin is rescanned over all records with E -- the answer is the brute force's (lowest original index at the exact minimum).
            auto pair_a2 = [&](const float4 xy, const float2 zz) {
                const v2f dx = QX - v2f{xy.x, xy.y}, dy = QY - v2f{xy.z, xy.w}, dz = QZ - v2f{zz.x, zz.y};
                return __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
            };
            constexpr float SCAN_M = 1.000003814697265625f, SCAN_ABS = 1e-37f;
            // ---- seed: an upper bound of the minimum of A, widened by the margin: the cull threshold before any block has been scored
            float seed = FLT_MAX;
            for (int k = 0; k < (L.nSeed + 1) / 2; k += 4) {
                float4 xy[4]; float2 zz[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { xy[u] = sd[2 * min(k + u, (L.nSeed + 1) / 2 - 1)]; zz[u] = *reinterpret_cast<const float2 *>(&sd[2 * min(k + u, (L.nSeed + 1) / 2 - 1) + 1]); }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const v2f a2 = pair_a2(xy[u], zz[u]);
                    seed = fminf(fminf(seed, a2.x), a2.y);
                }
            }
            float thr = __builtin_fmaf(seed, SCAN_M, SCAN_ABS);      // boxes with lb > thr hold no vertex within the margin of the minimum
            float best = FLT_MAX, bthr = FLT_MAX;                    // smallest block minimum of A so far, and the same widened by the margin
            int tie = 0;                                   // runner-up blocks within the margin of the smallest: 0, 1 (bblk2), 2 = more than one
            int bblk = 0, bblk2 = 0;                                  // first record of the block that last lowered the minimum
            // lower bounds of the point to a PAIR of boxes
            auto pair_lb = [&](const float4 a, const float4 bq, const float4 c) {
#pragma clang fp contract(off)
                const v2f ax = v2f{a.x, a.y} - QX, bx = QX - v2f{bq.z, bq.w};
                const v2f ay = v2f{a.z, a.w} - QY, by = QY - v2f{c.x, c.y};
                const v2f az = v2f{bq.x, bq.y} - QZ, bz = QZ - v2f{c.z, c.w};
                const v2f ex = v2f{fmaxf(fmaxf(ax.x, bx.x), 0.f), fmaxf(fmaxf(ax.y, bx.y), 0.f)};       // v_max3_f32
                const v2f ey = v2f{fmaxf(fmaxf(ay.x, by.x), 0.f), fmaxf(fmaxf(ay.y, by.y), 0.f)};
                const v2f ez = v2f{fmaxf(fmaxf(az.x, bz.x), 0.f), fmaxf(fmaxf(az.y, bz.y), 0.f)};
                // ex <= |dx| of every vertex in the box (subtraction is monotone), so the REAL sum of squares is a lower bound of every vertex's; three roundings here (<= 3 u up)
                // against three in E and A: far inside the margin of thr
                return __builtin_elementwise_fma(ez, ez, __builtin_elementwise_fma(ey, ey, ex * ex));
            };
            auto score_block = [&](int cb) {
                ++n_exec;
                const int v0 = cb * CB;
                float4 xy[CB / 2]; float2 zz[CB / 2];
#pragma unroll
                for (int u = 0; u < CB / 2; ++u) { xy[u] = vs[v0 + 2 * u]; zz[u] = *reinterpret_cast<const float2 *>(&vs[v0 + 2 * u + 1]); }
                float bm = FLT_MAX;
#pragma unroll
                for (int u = 0; u < CB / 2; ++u) {
                    const v2f a2 = pair_a2(xy[u], zz[u]);
                    bm = fminf(fminf(bm, a2.x), a2.y);                          // v_min3_f32
                }
                // a new smallest block: the previous one stays a candidate iff it is within the margin of this one; otherwise a block within the margin of the smallest is one.
                // ONE runner-up block is remembered (bblk2); a second one (tie == 2: ~1e-8 of the points) sends the point to the rescan over all records.
                if (bm < best) {
                    const float w = __builtin_fmaf(bm, SCAN_M, SCAN_ABS);
                    if (best <= w) { tie = tie ? 2 : 1; bblk2 = bblk; } else tie = 0;
                    best = bm; bblk = v0; bthr = w; thr = fminf(thr, w);
                } else if (bm <= bthr) { tie = tie ? 2 : 1; bblk2 = v0; }
            };
            auto scan_super = [&](int sb) {                 // the blocks of super-block sb, two box tests per instruction
                const int cb0 = sb * SB;                    // (fetching 2 or 4 box pairs ahead of their tests: no difference, profiles/r06_contact_bound.txt)
#pragma unroll
                for (int j = 0; j < SB / 2; ++j) {
                    const int cb = cb0 + 2 * j;             // boxes past nCB are dummies (lo = +max: lb = inf, never needed)
                    const float4 *bx = bb + 3 * (cb >> 1);
                    const v2f lb = pair_lb(bx[0], bx[1], bx[2]);
                    n_test += 2;
                    if (__builtin_amdgcn_ballot_w64(valid && lb.x <= thr) != 0ull) score_block(cb);
                    if (__builtin_amdgcn_ballot_w64(valid && lb.y <= thr) != 0ull) score_block(cb + 1);     // thr may just have dropped
                }
            };
            if (!vorder) {
                // identity order: consecutive vertex ids are not neighbours, the boxes cull next to nothing -- score every block (the
                // brute force of rounds 1-2 in task form) instead of paying for the tests as well
                for (int cb = 0; cb < nCB; ++cb) score_block(cb);
            } else
            for (int sb = 0; sb < nSB; sb += 2) {
                const float4 *bx = sbb + 3 * (sb >> 1);
                const v2f lb = pair_lb(bx[0], bx[1], bx[2]);
                n_test += 2;
                if (__builtin_amdgcn_ballot_w64(valid && lb.x <= thr) != 0ull) scan_super(sb);              // wave-uniform branches
                if (__builtin_amdgcn_ballot_w64(valid && lb.y <= thr) != 0ull) scan_super(sb + 1);
            }
            // resolve with the EXACT distance: inside the winning block (no other block came within the margin), else over all records (~1e-5 of the points): the smallest E, and among
            // the records that attain it the lowest ORIGINAL index -- the brute force's rule, whatever the scan order
            int pos = bblk, org = 0x7fffffff;
            float ebest = FLT_MAX;
            auto resolve_pair = [&](int pr) {                 // record pair pr: E = dist2_exact, original ids from the pair's spare words (0x7fffffff past V; those records sit at 3e18)
                const float4 xy = vs[2 * pr], zz = vs[2 * pr + 1];
                const v2f d2 = pair_d2(xy, zz);
                const int o0 = __builtin_bit_cast(int, zz.z), o1 = __builtin_bit_cast(int, zz.w);
                if (d2.x < ebest || (d2.x == ebest && o0 < org)) { ebest = d2.x; org = o0; pos = 2 * pr; }
                if (d2.y < ebest || (d2.y == ebest && o1 < org)) { ebest = d2.y; org = o1; pos = 2 * pr + 1; }
            };
            if (tie < 2) {
#pragma unroll
                for (int u = 0; u < CB / 2; ++u) resolve_pair((bblk >> 1) + u);
                if (tie) {                                // ~1e-4 of the points: the runner-up block as well
#pragma unroll
                    for (int u = 0; u < CB / 2; ++u) resolve_pair((bblk2 >> 1) + u);
                }
            } else {
                for (int pr = 0; pr < (V + 1) / 2; ++pr) resolve_pair(pr);
            }
            pos = min(pos, V - 1);
            if (org == 0x7fffffff) org = vorder ? vorder[pos] : pos;      // NaN input: nothing compared equal
            if constexpr (OPT) {
                if (valid) idx_out[(size_t)n * P + i] = org;
            } else if (valid) {
                // normal of the nearest vertex (data/tools.py:4-40 restricted to one vertex), from LDS, in the reference's accumulation order
                const float3 v3 = lds_rec(vs, pos);
                float3 acc = make_float3(0.f, 0.f, 0.f);
                const int e0 = adj_ptr[org], e1 = adj_ptr[org + 1];
                if (adj_pair) {
                    for (int e = e0; e < e1; e += 4) {               // four incident faces per trip: their (a, b) loads fly together
                        int2 ab[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) ab[u] = *reinterpret_cast<const int2 *>(adj_pair + 2 * (size_t)min(e + u, e1 - 1));
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            if (e + u < e1) {
                                const float3 nn = cross3(sub3(lds_rec(vs, ab[u].x), v3), sub3(lds_rec(vs, ab[u].y), v3));
                                acc.x += nn.x; acc.y += nn.y; acc.z += nn.z;
                            }
                        }
                    }
                } else {
                    for (int e = e0; e < e1; ++e) {
                        const int f = adj_face[e], c = adj_corner[e];
                        const float3 p0 = lds_rec(vs, faces[3 * f]), p1 = lds_rec(vs, faces[3 * f + 1]), p2 = lds_rec(vs, faces[3 * f + 2]);
                        float3 nn;
                        if (c == 1) nn = cross3(sub3(p2, p1), sub3(p0, p1));
                        else if (c == 2) nn = cross3(sub3(p0, p2), sub3(p1, p2));
                        else nn = cross3(sub3(p1, p0), sub3(p2, p0));
                        acc.x += nn.x; acc.y += nn.y; acc.z += nn.z;
                    }
                }
                const float nl = fmaxf(sqrtf(acc.x * acc.x + acc.y * acc.y + acc.z * acc.z), 1e-6f);
                const float vx = qx - v3.x, vy = qy - v3.y, vz = qz - v3.z;
                const float dt = (acc.x / nl) * vx + (acc.y / nl) * vy + (acc.z / nl) * vz;
                const float d = sqrtf(vx * vx + vy * vy + vz * vz);
                const float o2h = d * (dt > 0.f ? 1.f : (dt < 0.f ? -1.f : 0.f));
                if (o2h_out) o2h_out[(size_t)n * P + i] = o2h;
                if (idx_out) idx_out[(size_t)n * P + i] = org;
                pl[sp] = o2h < 0.f ? fabsf(o2h) * 20.0f : 0.f;                  // eval_smpl_short.py:113-119
            }
        }
        if (!OPT && valid) {
            // Marker distances (eval_smpl_short.py:110-112: contact label = some object point within 2 cm of the marker; min over everything for the clip's condition),
            // two markers per packed instruction, WITHOUT a square root per pair: sqrtf is monotone and correctly rounded, so min_m sqrtf(d2_m) = sqrtf(min_m d2_m)
            // (one root per workgroup, below) and sqrtf(d2) < 0.02f <=> d2 < MARK_D2 -- the smallest float whose root is >= 0.02f (tests/test_abi_and_host.py re-derives it).
            // d2 keeps the roundings the scalar form compiled to: fma(dy, dy, dx * dx) + dz * dz.  (Round 6: the correctly-rounded root, its fix-up and an exec-mask write
            // per marker were 35 instructions x 67 markers per task, a quarter of the kernel.)
#pragma clang fp contract(off)
            const v2f QX = v2f{qx, qx}, QY = v2f{qy, qy}, QZ = v2f{qz, qz};
            for (int p2 = 0; p2 < (M + 1) / 2; ++p2) {
                const float4 xy = ms[2 * p2], zz = ms[2 * p2 + 1];
                const v2f dx = v2f{xy.x, xy.y} - QX, dy = v2f{xy.z, xy.w} - QY, dz = v2f{zz.x, zz.y} - QZ;
                const v2f d2 = __builtin_elementwise_fma(dy, dy, dx * dx) + dz * dz;
                mind2 = fminf(fminf(mind2, d2.x), d2.y);
                if (__builtin_amdgcn_ballot_w64(fminf(d2.x, d2.y) < MARK_D2) != 0ull) {      // wave-uniform, rare
                    if (d2.x < MARK_D2) flags[2 * p2] = 1;                     // benign race: all writers store 1
                    if (d2.y < MARK_D2) flags[2 * p2 + 1] = 1;                 // (the dummy of an odd count is infinitely far: never)
                }
            }
        }
    }
    if (stats && ln == 0 && do_nn) {
        atomicAdd(stats, (unsigned long long)n_exec);
        atomicAdd(stats + 1, (unsigned long long)nCB * n_tasks);
        atomicAdd(stats + 2, (unsigned long long)n_test);
    }
    phase_done();                                                          // 3: this wave's tasks
    if constexpr (OPT) return;                                             // every wave has written its points' indices: nothing to reduce
    __syncthreads();
    phase_done();                                                          // 4: waiting for the other waves
    // deterministic block reductions: the per-point losses by scan position in a fixed tree
    float loss = 0.f;
#pragma unroll
    for (int k = 0; k < MAXP / CT; ++k) loss += pl[tid + CT * k];
    red[tid] = loss;
    __syncthreads();
    for (int s = CT / 2; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    const float loss_total = red[0];
    __syncthreads();
    red[tid] = mind2;
    __syncthreads();
    for (int s = CT / 2; s > 0; s >>= 1) {
        if (tid < s) red[tid] = fminf(red[tid], red[tid + s]);
        __syncthreads();
    }
    if (tid == 0) { loss_sum[n] = loss_total; min_dist[n] = red[0] == FLT_MAX ? FLT_MAX : sqrtf(red[0]); }      // the one root of the frame (monotone: = the minimum of the roots)
    if (tid < M) label[(size_t)n * M + tid] = flags[tid];
    phase_done();                                                          // 5: reductions
    if (stats && tid == 0 && do_nn) atomicAdd(stats + 3, 1ull);            // workgroups counted
}

// (An fp32-MFMA "filter" variant of this scan -- g~ = |p|^2 - 2 q.p on v_mfma_f32_16x16x4, exact re-evaluation of
// the vertices inside the error band -- was built, verified bit-identical and measured on MI355X: its two sweeps run
// at 40% of the matrix pipe next to the VALU work that reads every product, 6.1 ms vs 5.0 ms per call for this kernel.
// The fp32 MFMA issues at the fp32 VALU rate on gfx950, so the filter cannot win)
size_t contact_lds_bytes(int V) { return (size_t)ContactLds(V).total * sizeof(float4); }

int launch_contact(hipStream_t s, int64_t N, const float *verts, int V, const float *obj_points, int P, int32_t *porder,
                   const float *objR, const float *objT, const idf_correction_ctx *c, int B, float *markers, float *loss_sum,
                   float *min_dist, int32_t *label, float *o2h, int32_t *idx, unsigned long long *stats, int64_t nn_from, const PredArgs *pred = nullptr) {
    const int M = c->n_markers;
    const size_t lds = contact_lds_bytes(V);
    if (lds > 160 * 1024 - 16384 || P > MAXP) return IDF_E_INVAL;
    // scan order: either all four of (vorder, faces_scan, markers_scan, adj_pair_scan) or none (identity order: same results, no culling benefit)
    const bool ordered = c->vorder != nullptr;
    if (ordered != (c->faces_scan != nullptr) || ordered != (c->markers_scan != nullptr) || ordered != (c->adj_pair_scan != nullptr)) return IDF_E_INVAL;
    const int32_t *faces = ordered ? c->faces_scan : c->faces, *mpos = ordered ? c->markers_scan : c->markers_idx;
    if (porder) hipLaunchKernelGGL(corr_point_order_kernel, dim3((unsigned)B), dim3(1024), 0, s, obj_points, P, porder);
    if (pred) {
        // one LDS size for both roles (the predictor's stacks need 152 KB, the scan less); the predictor's workgroups lead the grid
        const size_t lds2 = (pred->nclips > 0 && lds < idf_objproj_dev::OBJPROJ_LDS) ? idf_objproj_dev::OBJPROJ_LDS : lds;
        static std::atomic<uint64_t> lds_ok2{0};
        if (lds2 > 160 * 1024 - 2048) return IDF_E_INVAL;
        if (idf_opt_in_lds(reinterpret_cast<const void *>(corr_contact_kernel<false, true>), 160 * 1024 - 2048, lds_ok2) != IDF_OK) return IDF_E_LAUNCH;
        hipLaunchKernelGGL((corr_contact_kernel<false, true>), dim3((unsigned)(N + pred->nclips)), dim3(CT), lds2, s, verts, V, obj_points, P, porder, objR, objT, faces,
                           c->adj_ptr, c->adj_face, c->adj_corner, c->adj_pair_scan, c->vorder, c->vrank, mpos, M, B, markers, loss_sum, min_dist, label, o2h, idx,
                           stats, nn_from, nullptr, 0, *pred);
        return IDF_OK;
    }
    static std::atomic<uint64_t> lds_ok{0};
    if (idf_opt_in_lds(reinterpret_cast<const void *>(corr_contact_kernel<false>), 160 * 1024 - 16384, lds_ok) != IDF_OK) return IDF_E_LAUNCH;
    hipLaunchKernelGGL(corr_contact_kernel<false>, dim3((unsigned)N), dim3(CT), lds, s, verts, V, obj_points, P, porder, objR, objT, faces,
                       c->adj_ptr, c->adj_face, c->adj_corner, c->adj_pair_scan, c->vorder, c->vrank, mpos, M, B, markers, loss_sum, min_dist, label, o2h, idx,
                       stats, nn_from, nullptr, 0, PredArgs{});
    return IDF_OK;
}

// ---- K5: per-clip decisions (eval_smpl_short.py:119-125) ---------------------------------------------
__global__ __launch_bounds__(128) void corr_reduce_kernel(const float *__restrict__ loss_sum, const float *__restrict__ min_dist,
                                                          const int32_t *__restrict__ label, int B, int T, int past, int P, int M,
                                                          uint8_t *__restrict__ condition, int32_t *__restrict__ contact,
                                                          float *__restrict__ distance_out, float *__restrict__ loss_out) {
    // the per-frame values are fetched by all threads at once (one memory round trip instead of T dependent ones: the serial loop
    // cost 60 us per call) and summed by thread 0 from LDS in the reference's order, frame by frame: the same bits as before
    extern __shared__ float fr[];                         // [T] loss sums | [T] minimum distances
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int t = tid; t < T; t += 128) {
        fr[t] = loss_sum[(size_t)t * B + b];
        fr[T + t] = min_dist[(size_t)t * B + b];
    }
    if (tid < M) {
        int c = 0;
#pragma unroll 16
        for (int t = past; t < T; ++t) c += label[((size_t)t * B + b) * M + tid];           // independent loads: sixteen in flight
        contact[(size_t)b * M + tid] = c;
    }
    __syncthreads();
    if (tid == 0) {
        float ls = 0.f, ds = 0.f;
        for (int t = past; t < T; ++t) ls += fr[t] / (float)P;                             // mean over points, then frames
        for (int t = 0; t < T; ++t) ds += fr[T + t];
        const float loss = ls / (float)(T - past), dist = ds / (float)T;
        condition[b] = !((loss < 0.002f) && (dist < 0.02f));
        if (distance_out) distance_out[b] = dist;
        if (loss_out) loss_out[b] = loss;
    }
}

// ---- K7: blend (eval_smpl_short.py:127-129): x = a*x + (1-a)*[body, proj] where condition ------------
// a: the blend weight t / 1000 by value, or -- a_table != null -- read on the device as a_table[state[0] * 4 + 3] (the sampler's
// coefficient table and state: a captured hipGraph of a whole hook step then serves every timestep)
__global__ __launch_bounds__(256) void corr_blend_kernel(float *__restrict__ x0, const float *__restrict__ proj,
                                                         const uint8_t *__restrict__ condition, int B, int T, float a,
                                                         const float *__restrict__ a_table, const int64_t *__restrict__ a_state) {
    if (a_table) a = a_table[a_state[0] * 4 + 3];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * CTOK * T) return;
    const int b = (int)(i / (CTOK * T)), r = (int)(i - (int64_t)b * CTOK * T), c = r / T, t = r - c * T;
    if (!condition[b]) return;
    const float x = x0[i];
    const float other = c < 135 ? x : proj[((size_t)t * B + b) * 9 + (c - 135)];
    x0[i] = a * x + (1.0f - a) * other;
}

struct CorrWs {
    float *pose, *trans, *objR, *objT, *gt_angles, *gt_trans, *verts, *jtr, *markers, *loss_sum, *min_dist, *proj, *proj_keep;
    int32_t *label, *contact, *porder;
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
    w.porder = (int32_t *)take((size_t)B * c->n_points * 4);
    w.proj_keep = (float *)take((size_t)B * idf_objproj_keep_floats() * 4);
    w.smpl_ws_bytes = interdiff_smpl_workspace_bytes(c->smpl, N);
    w.smpl_ws = take(w.smpl_ws_bytes);
    w.total = off;
    return w;
}


// ---- the contact-radius mask of the post-optimisation (optimization.py:74-75): per vertex, does ANY object point lie within 0.5 m? ----
// The dual of the scan above: lanes are VERTICES, the frame's object points are cut into patches of 64 consecutive points of the
// clip's Morton order (compact under the frame's rigid transform) with their boxes; a patch is looked at only if its box comes
// closer than 0.5 m to some vertex of the wave that is still undecided (same monotone lower bound, so the cut is exact), and a
// wave leaves as soon as all its vertices have found a point.  K-A sorts the frame's points and boxes its patches, K-B answers.
constexpr int NPATCH = MAXP / 64;
__global__ __launch_bounds__(256) void opt_patch_kernel(const float *__restrict__ pts, int P, const int32_t *__restrict__ porder, int frames_per_clip,
                                                        float4 *__restrict__ psort /* [N][MAXP] */, float4 *__restrict__ pbox /* [N][NPATCH][2] */) {
    const int64_t n = blockIdx.x;
    const int b = (int)(n / frames_per_clip), tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *pn = pts + (size_t)n * P * 3;
    for (int k = wave; k < NPATCH; k += 4) {
        const int sp = 64 * k + lane;
        float x = 3e18f, y = 3e18f, z = 3e18f;                              // slots past P: far away (never within the radius)
        float lx = FLT_MAX, ly = FLT_MAX, lz = FLT_MAX, hx = -FLT_MAX, hy = -FLT_MAX, hz = -FLT_MAX;
        if (sp < P) {
            const int i = porder[(size_t)b * P + sp];
            x = pn[3 * i]; y = pn[3 * i + 1]; z = pn[3 * i + 2];
            lx = hx = x; ly = hy = y; lz = hz = z;
        }
        psort[(size_t)n * MAXP + sp] = make_float4(x, y, z, 0.f);
        lx = -wave_max(-lx); ly = -wave_max(-ly); lz = -wave_max(-lz);
        hx = wave_max(hx); hy = wave_max(hy); hz = wave_max(hz);
        if (lane == 0) {
            pbox[((size_t)n * NPATCH + k) * 2] = make_float4(lx, ly, lz, 0.f);
            pbox[((size_t)n * NPATCH + k) * 2 + 1] = make_float4(hx, hy, hz, 0.f);
        }
    }
}

__global__ __launch_bounds__(256) void opt_near_kernel(const float *__restrict__ verts, int V, int P, const float4 *__restrict__ psort,
                                                       const float4 *__restrict__ pbox, const int32_t *__restrict__ vorder,
                                                       int32_t *__restrict__ near) {
    __shared__ float4 ps[MAXP];                                            // the frame's points, sorted
    __shared__ float4 bx[NPATCH * 2];
    const int64_t n = blockIdx.y;
    // a wave takes 64 consecutive vertices of the SCAN order (a compact clump of the body): its lanes agree on which patches matter
    const int tid = threadIdx.x, pos = blockIdx.x * 256 + tid, vid = vorder[min(pos, V - 1)];
    for (int i = tid; i < MAXP; i += 256) ps[i] = psort[(size_t)n * MAXP + i];
    if (tid < NPATCH * 2) bx[tid] = pbox[(size_t)n * NPATCH * 2 + tid];
    const bool valid = pos < V;
    const float *vp = verts + ((size_t)n * V + vid) * 3;
    const float vx = vp[0], vy = vp[1], vz = vp[2];
    __syncthreads();
    const int npatch = (P + 63) / 64;
    bool hit = false;
    {
#pragma clang fp contract(off)
        const v2f VX = v2f{vx, vx}, VY = v2f{vy, vy}, VZ = v2f{vz, vz};
        for (int k = 0; k < npatch; ++k) {
            if (__builtin_amdgcn_ballot_w64(valid && !hit) == 0ull) break;               // every vertex of the wave is decided
            const float4 lo = bx[2 * k], hi = bx[2 * k + 1];
            const float ex = fmaxf(fmaxf(lo.x - vx, vx - hi.x), 0.f), ey = fmaxf(fmaxf(lo.y - vy, vy - hi.y), 0.f), ez = fmaxf(fmaxf(lo.z - vz, vz - hi.z), 0.f);
            const float xx = ex * ex, yy = ey * ey, zz = ez * ez;
            if (__builtin_amdgcn_ballot_w64(valid && !hit && (xx + yy) + zz < 0.25f) == 0ull) continue;      // no undecided vertex can have a point of this patch in range
            const float4 *pp = ps + 64 * k;
            for (int j0 = 0; j0 < 64; j0 += 16) {                        // sixteen points, then ask again whether anybody is still undecided
#pragma unroll
                for (int j = j0; j < j0 + 16; j += 2) {
                    const float4 p = pp[j], q = pp[j + 1];
                    // sqrt(d2) < 0.5 (optimization.py:75) <=> d2 < 0.25 up to the rounding of the last ulp; d = point - vertex as in the brute force
                    const v2f dx = v2f{p.x, q.x} - VX, dy = v2f{p.y, q.y} - VY, dz = v2f{p.z, q.z} - VZ;
                    const v2f d2 = (dx * dx + dy * dy) + dz * dz;
                    hit = hit || d2.x < 0.25f || d2.y < 0.25f;
                }
                if (__builtin_amdgcn_ballot_w64(valid && !hit) == 0ull) break;
            }
        }
    }
    if (valid) near[(size_t)n * V + vid] = hit ? 1 : 0;
}

}  // namespace

// The nearest-vertex half of the post-optimisation's scan (optimize.hip; declared in common.h): per transformed object point of
// every frame the nearest vertex.  Frames clip-major (n = b*T + t); `porder` [B][P] = idf_point_order of the clips' canonical points
// (they do not change over the iterations: computed once by interdiff_optimize_init).
int idf_point_order(hipStream_t s, const float *obj_points, int B, int P, int32_t *porder) {
    if (P > MAXP) return IDF_E_INVAL;
    hipLaunchKernelGGL(corr_point_order_kernel, dim3((unsigned)B), dim3(1024), 0, s, obj_points, P, porder);
    return IDF_OK;
}

int idf_nn_scan_opt(hipStream_t s, int64_t N, int frames_per_clip, const float *verts, int V, const float *pts_frame, const float *obj_points, int P,
                    const int32_t *porder, const idf_correction_ctx *c, int32_t *yidx) {
    const size_t lds = contact_lds_bytes(V);
    if (lds > 160 * 1024 - 16384 || P > MAXP || !c->vorder) return IDF_E_INVAL;
    const int B = (int)(N / frames_per_clip);
    static std::atomic<uint64_t> lds_ok{0};
    if (idf_opt_in_lds(reinterpret_cast<const void *>(corr_contact_kernel<true>), 160 * 1024 - 16384, lds_ok) != IDF_OK) return IDF_E_LAUNCH;
    hipLaunchKernelGGL(corr_contact_kernel<true>, dim3((unsigned)N, 2), dim3(CT), lds, s, verts, V, obj_points, P, porder, nullptr, nullptr, nullptr,
                       nullptr, nullptr, nullptr, nullptr, c->vorder, c->vrank, nullptr, 0, B, nullptr, nullptr, nullptr, nullptr, nullptr, yidx, nullptr, (int64_t)0,
                       pts_frame, frames_per_clip, PredArgs{});
    return IDF_OK;
}

// ... and the contact-radius mask: near[n][v] = 1 iff some object point of frame n lies within 0.5 m of vertex v.  `porder` as filled
// by idf_nn_scan_opt for the same clips; psort [N][2048] float4 and pbox [N][32][2] float4 are scratch.
int idf_near_mask_opt(hipStream_t s, int64_t N, int frames_per_clip, const float *verts, int V, const float *pts_frame, int P, const int32_t *porder,
                      const int32_t *vorder, float *psort, float *pbox, int32_t *near) {
    if (P > MAXP || !vorder) return IDF_E_INVAL;
    hipLaunchKernelGGL(opt_patch_kernel, dim3((unsigned)N), dim3(256), 0, s, pts_frame, P, porder, frames_per_clip, reinterpret_cast<float4 *>(psort),
                       reinterpret_cast<float4 *>(pbox));
    hipLaunchKernelGGL(opt_near_kernel, dim3((unsigned)idf_cdiv(V, 256), (unsigned)N), dim3(256), 0, s, verts, V, P, reinterpret_cast<const float4 *>(psort),
                       reinterpret_cast<const float4 *>(pbox), vorder, near);
    return IDF_OK;
}

namespace {

}  // namespace

extern "C" size_t interdiff_correction_workspace_bytes(const idf_correction_ctx *c, int32_t B, int32_t T) {
    if (!c || !c->smpl || B <= 0 || T <= 0) return 0;
    return carve(c, B, T, nullptr).total;
}

static int correction_impl(const idf_correction_ctx *c, float *x0, const float *gt, const float *hand_pose,
                           const float *beta, const float *obj_points, int32_t B, int32_t T, float blend_t, const float *blend_table,
                           const int64_t *blend_state, uint8_t *condition, int32_t *contact, float *distance, float *loss, void *ws,
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
    // Round 6: the predictor's three stacks need the markers only -- the contact labels pick which node's IDCT is evaluated at the very end -- so they ride in the contact scan's
    // launch as B leading workgroups (corr_contact_kernel<false, true>) and only the pick + IDCT kernel follows the labels.  ctx.tune & 2 (A/B, tests): scan, then the one-launch
    // predictor.  Same arithmetic either way: same bits.
    const bool fused = (c->tune & 2) == 0 && idf_objproj_check(c->objproj) == IDF_OK;
    idf_prof_mark(IDF_K_CORR_PREPARE, s);
    hipLaunchKernelGGL(corr_prepare_kernel, dim3((unsigned)idf_cdiv(N * 24, 256)), dim3(256), 0, s, x0, gt, hand_pose, B, T, w.pose,
                       w.trans, w.objR, w.objT, w.gt_angles, w.gt_trans);
    int rc = interdiff_smpl_forward(c->smpl, w.pose, beta, w.trans, N, w.verts, w.jtr, nullptr, w.smpl_ws, w.smpl_ws_bytes, stream);
    if (rc) return rc;
    idf_prof_mark(IDF_K_CORR_CONTACT, s);
    PredArgs pred{};
    if (fused) {
        pred.op = *c->objproj;
        pred.obj_angles = w.gt_angles; pred.obj_trans = w.gt_trans;
        pred.markers_idx = c->markers_idx; pred.markers = w.markers; pred.keep = w.proj_keep;
        pred.nclips = B;
    }
    // (the hook always launches the instantiation that CAN carry the predictor, with no predictor workgroups in the A/B form: two instantiations of the same source were seen to
    // differ in the last bit of a frame's loss -- the compiler contracts the normal / signed-distance expressions per instantiation)
    rc = launch_contact(s, N, w.verts, V, obj_points, P, w.porder, w.objR, w.objT, c, B, fused ? nullptr : w.markers, w.loss_sum, w.min_dist, w.label, nullptr,
                        nullptr, nullptr, (int64_t)c->past_len * B, &pred);
    if (rc) return rc;
    uint8_t *cond = condition ? condition : w.condition;
    int32_t *cont = contact ? contact : w.contact;
    idf_prof_mark(IDF_K_CORR_REDUCE, s);
    hipLaunchKernelGGL(corr_reduce_kernel, dim3(B), dim3(128), 2 * (size_t)T * sizeof(float), s, w.loss_sum, w.min_dist, w.label, B, T, c->past_len, P, M, cond,
                       cont, distance, loss);
    if (fused) {
        idf_prof_mark(IDF_K_OBJPROJ, s);
        rc = idf_objproj_pick(c->objproj, w.proj_keep, cont, B, w.proj, s);
    } else {
        rc = interdiff_objprojector_sample(c->objproj, w.gt_angles, w.gt_trans, w.markers, cont, B, w.proj, stream);
    }
    if (rc) return rc;
    idf_prof_mark(IDF_K_CORR_BLEND, s);
    hipLaunchKernelGGL(corr_blend_kernel, dim3((unsigned)idf_cdiv((int64_t)B * CTOK * T, 256)), dim3(256), 0, s, x0, w.proj, cond, B, T,
                       blend_t, blend_table, blend_state);
    idf_prof_mark(-1, s);
    IDF_CHECK_LAUNCH();
    return IDF_OK;
}

extern "C" int interdiff_correction(const idf_correction_ctx *c, float *x0, const float *gt, const float *hand_pose,
                                    const float *beta, const float *obj_points, int32_t B, int32_t T, float blend_t,
                                    uint8_t *condition, int32_t *contact, float *distance, float *loss, void *ws,
                                    size_t ws_bytes, void *stream) {
    return correction_impl(c, x0, gt, hand_pose, beta, obj_points, B, T, blend_t, nullptr, nullptr, condition, contact, distance, loss, ws, ws_bytes, stream);
}

extern "C" int interdiff_correction_dev(const idf_correction_ctx *c, float *x0, const float *gt, const float *hand_pose,
                                        const float *beta, const float *obj_points, int32_t B, int32_t T, const float *table,
                                        const int64_t *state, void *ws, size_t ws_bytes, void *stream) {
    if (!table || !state) return IDF_E_INVAL;
    return correction_impl(c, x0, gt, hand_pose, beta, obj_points, B, T, 0.f, table, state, nullptr, nullptr, nullptr, nullptr, ws, ws_bytes, stream);
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
    int32_t *label, *porder;
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
    w.porder = (int32_t *)take((size_t)B * c->n_points * 4);
    w.total = off;
    return w;
}

}  // namespace

// The contact scan on its own: signed object->human distance and nearest vertex of every object point of every frame
// (tools.point2point_signed's o2h half fused with the object transform, eval_smpl_short.py:107-112).  Frames n = t*B + b.
extern "C" size_t interdiff_contact_nn_workspace_bytes(const idf_correction_ctx *c, int32_t B, int32_t T) {
    if (!c || B <= 0 || T <= 0) return 0;
    return carve_metrics(c, B, T, nullptr).total;
}

extern "C" int interdiff_contact_nn(const idf_correction_ctx *c, const float *verts, const float *obj_points, const float *objR,
                                    const float *objT, int32_t B, int32_t T, float *o2h, int32_t *idx, uint64_t *stats, void *ws,
                                    size_t ws_bytes, void *stream) {
    if (!c || !c->smpl || !verts || !obj_points || !objR || !objT || !ws || B <= 0 || T <= 0 || (!o2h && !idx)) return IDF_E_INVAL;
    if (c->n_markers > MAXM || c->n_points > MAXP) return IDF_E_INVAL;
    MetWs w = carve_metrics(c, B, T, ws);
    if (ws_bytes < w.total) return IDF_E_NOMEM;
    hipStream_t s = idf_stream(stream);
    if (stats && hipMemsetAsync(stats, 0, IDF_CONTACT_STATS * sizeof(uint64_t), s) != hipSuccess) return IDF_E_LAUNCH;
    const int rc = launch_contact(s, (int64_t)B * T, verts, c->smpl->V, obj_points, c->n_points, w.porder, objR, objT, c, B, w.markers, w.loss_sum,
                                  w.min_dist, w.label, o2h ? o2h : w.o2h, idx, reinterpret_cast<unsigned long long *>(stats), 0);
    if (rc) return rc;
    IDF_CHECK_LAUNCH();
    return IDF_OK;
}

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
    const int rc = launch_contact(s, N, verts, V, obj_points, P, w.porder, w.objR, w.objT, c, B, w.markers, w.loss_sum, w.min_dist, w.label, w.o2h,
                                  nullptr, nullptr, 0);
    if (rc) return rc;
    hipLaunchKernelGGL(metrics_reduce_kernel, dim3(B), dim3(256), 0, s, obj_pred, jtr, body_trans, obj_gt, jtr_gt, body_trans_gt, w.o2h, B,
                       T, J, P, out6);
    idf_prof_mark(-1, s);
    IDF_CHECK_LAUNCH();
    return IDF_OK;
}
