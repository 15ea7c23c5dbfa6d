// The split-f16 feed-forward block as ONE software pipeline over hidden chunks (round 6) -- same contract, grid, slabs, arithmetic and BITS as
// ffn_h2.h's ffn_h2_kernel (which stays for the 16- and 64-row tiles), different schedule:
//
//     ffn_h2.h:   [linear1, 8 K steps] -> barrier -> [GELU + split: 3.4 k cycles of VALU with the matrix pipe and the DMA queue idle] -> [linear2, 7 K steps]
//     here:       for every chunk c of 32 hidden units:  linear1(c)  ->  GELU(c)  ->  linear2 += hid(c) . W2[:, c]      all three in flight at once
//
// A hidden unit's pre-activation needs all 256 input columns but only its own W1 row, and linear2's K step q needs only hidden units 32 q .. 32 q + 31: cut by
// hidden CHUNK instead of by phase, the GELU of chunk c runs while the matrix pipe works on chunk c + 1 (linear1) and chunk c - 1 (linear2) and while the weight
// stream of the chunks behind them is landing.  The waves specialise:
//   * P waves (0..3, one per SIMD): wave p owns the (hidden tile c * 2 + (p & 1), token tile p >> 1) unit of every chunk.  Its token tile's x2 planes are
//     REGISTER-resident for the whole kernel (8 K steps x 2 planes x 4 VGPRs = 64): linear1 reads only weight fragments from LDS.  8 K steps x 3 MFMAs, then
//     bias + GELU + split, then 2 x 8 bytes per lane into the chunk's hid planes (double-buffered, 4 KiB each).
//   * Q waves (4..7): wave q owns output tiles 4 q .. 4 q + 3 x both token tiles (16 accumulators of 4) and runs linear2 over the chunks as they appear.
// The two waves of a SIMD are one P and one Q wave: P's VALU phase (GELU) and Q's MFMAs share the SIMD's issue slots instead of taking turns in lock step.
// The x2 planes no longer live in LDS past the prologue, so the ring is 8 slots of 16 KiB = 128 KiB (six half-steps = 96 KiB in flight behind the two being read).
//
// Stream order (mdm.py pack_ffn_h2f: a permutation of pack_ffn_h2's 1-KiB fragments): per slice 28 HALF-STEPS in consumption order
//     a0 a1 a2 a3 b0 a4 b1 a5 b2 ... a13 b10 b11 b12 b13
//   a(2 c + h) = linear1 weights of chunk c, K steps 4 h .. 4 h + 3:  [4 K steps][2 hidden tiles][2 planes][1 KiB]   (16 KiB; chunk 6 has ONE hidden tile: 8 KiB)
//   b(2 c + h) = linear2 weights of chunk c (= K step c of phase 2):  [4 Q waves][2 output tiles 4 q + 2 h + i][2 planes][1 KiB]   (16 KiB)
// Half-tick t (one barrier each, 18 of them): P reads a(t) and multiplies a(t - 1); Q reads b(t - 3) (+ the hid planes of chunk (t - 3) / 2 when t - 3 is even) and
// multiplies b(t - 4).  Fragments are always read one half-tick ahead of their MFMAs (register double buffer), so a slot is free again at the next barrier.
//
// BIT-IDENTICAL to ffn_h2_kernel: every accumulator sees the same MFMAs on the same fragments in the same order (no K split across waves, chunks in order, per
// K step main, hi x lo', lo' x hi), GELU and splits are the same inline functions -- tests/test_hip_parity.py compares the three row tiles bit for bit.
#pragma once
#include "ffn_h2.h"
#include <utility>

namespace idf_ffn_h2f {
using namespace idf_ffn_h2;

constexpr int NR = 28;                               // ring half-steps per slice
constexpr int NSLOT = 8, SLOTB = 16384;              // ring slots / bytes of a slot
constexpr int NTICK = 18;                            // half-ticks
constexpr int HIDB = 4096;                           // bytes of one hid-plane chunk buffer: [2 token tiles][2 planes][16 rows][32 halves]

__host__ __device__ constexpr int a_ring(int j) { return j < 3 ? j : 2 * j - 3; }          // ring index of a(j), j = 0..13
__host__ __device__ constexpr int b_ring(int j) { return j <= 10 ? 2 * j + 4 : j + 14; }   // ring index of b(j), j = 0..13
__host__ __device__ constexpr int step_bytes(int r) { return (r == 21 || r == 23) ? 8192 : 16384; }      // a12 / a13: one hidden tile
__host__ __device__ constexpr int step_off(int r) {
    int o = 0;
    for (int i = 0; i < r; ++i) o += step_bytes(i);
    return o;
}
static_assert(step_off(NR) == SLICE_BYTES, "the half-steps are the slice");
__host__ __device__ constexpr int my_ins(int r) { return r >= 0 && r < NR ? step_bytes(r) / 8192 : 0; }   // DMA instructions PER WAVE of a half-step (8 waves x 1 KiB each)
// last ring index issued once the issue of half-tick t is out (t = -1: the prologue) / last ring index that must have landed at the barrier that opens half-tick t
__host__ __device__ constexpr int issued(int t) { return t < 0 ? 5 : (t == 0 ? 5 : (t == 1 ? 8 : (t == 2 ? 9 : (t == 3 ? 10 : (2 * t + 4 > NR - 1 ? NR - 1 : 2 * t + 4))))); }
__host__ __device__ constexpr int needed(int t) { return t < 3 ? t : (t <= 13 ? 2 * t - 2 : (t <= 16 ? t + 11 : NR - 1)); }
__host__ __device__ constexpr int flying(int t) {    // this wave's DMA instructions that may still be in flight at the barrier that opens half-tick t
    int n = 0;
    for (int r = needed(t) + 1; r <= issued(t - 1); ++r) n += my_ins(r);
    return n;
}

// compile-time loop: the body sees its index as a constant expression (every ring index, slot, wait count and buffer below is an immediate)
template <class F, int... T>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, T...>) { (f(std::integral_constant<int, T>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// MODE 0: product; 3: half-tick stamps of thread 0 (P wave 0) and thread 256 (Q wave 4) behind the slabs (tools/ffn_h2f_probe.hip)
template <int MODE = 0>
__global__ __launch_bounds__(NT) void ffn_h2f_kernel(const float *__restrict__ x2, int M, int nwg, const float *__restrict__ pack,
                                                      const float *__restrict__ b1p, const float *__restrict__ b2,
                                                      float *__restrict__ parts, int order) {
    constexpr int BM = 32;
    extern __shared__ __attribute__((aligned(1024))) float smem[];
    asm volatile("" ::: "v255");                       // the whole register file: EXCLUSIVE CU (ffn_h2.h)
    float *ring = smem;                                // 8 x 16 KiB
    float *Xs = smem + 6 * (SLOTB / 4);                // x2 rows / planes of the prologue: over slots 6, 7 (their first fills are issued behind barrier 1)
    float *hid = smem + NSLOT * (SLOTB / 4);           // 2 x 4 KiB
    float *biasL = hid + 2 * (HIDB / 4);               // 4 x 1 KiB: every P wave's private copy of the slice's linear1 bias

    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool isP = wave < 4;
    long long *stamps = nullptr;
    int n_stamp = 0;
    if constexpr (MODE == 3 || MODE == 4) stamps = reinterpret_cast<long long *>(parts + (size_t)NSL * M * D) + (size_t)blockIdx.x * 64 + (tid == 256 ? 32 : 0);
    auto substamp = [&](auto tc) {                     // MODE 4: inside half-ticks 6 (P: GELU) and 7: barrier passed / DMA issued / fragment reads issued / compute done
        if constexpr (MODE == 4 && (decltype(tc)::value == 6 || decltype(tc)::value == 7)) {
            if (tid == 0 || tid == 256) stamps[n_stamp] = __builtin_readcyclecounter();
            ++n_stamp;
        }
    };
    auto stamp = [&]() {
        if constexpr (MODE == 3) {
            if (tid == 0 || tid == 256) stamps[n_stamp] = __builtin_readcyclecounter();
            ++n_stamp;
        }
    };
    stamp();
    // workgroup -> (M tile, slice): XCD-affine, M-tile-major (ffn_h2.h order 0)
    const int id = blockIdx.x;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = id & 7;
    const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (id >> 3);
    const int sl = wg % NSL, mt = wg / NSL, m0 = mt * BM;
    (void)order;
    const float *stream = idf_uniform_ptr(pack + (size_t)sl * SLICE_FLOATS);
    const uint32_t lane16 = lane << 4;
    const uint32_t vsrc = (uint32_t)(wave * 1024) + lane16;
    const uint32_t sdst = idf_lds_addr(ring) + (uint32_t)(wave * 1024);

    auto issue = [&](auto rc) {                        // this wave's instructions of ring half-step r: instruction wave (+ 8) copies stream bytes [off + 1024 i, + 1024) to slot r % 8
        constexpr int r = decltype(rc)::value;
        constexpr uint32_t so = (uint32_t)step_off(r), dof = (uint32_t)((r % NSLOT) * SLOTB);
        idf_dma16_s(stream, vsrc + so, sdst + dof);
        if constexpr (step_bytes(r) == 16384) idf_dma16_s(stream, vsrc + so + 8192u, sdst + dof + 8192u);
    };
    // opens half-tick t: this wave's share of everything read in t has landed, every wave is done reading what it read in t - 1, then the freed slots are refilled
    auto tick_begin = [&](auto tc) {
        constexpr int t = decltype(tc)::value;
        if constexpr (t == 0 || needed(t) != needed(t - 1)) {
            constexpr int fl = flying(t);
            wait_vmcnt(fl);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0): my fragment reads of t - 1 (and my hid writes) are done
        __builtin_amdgcn_s_barrier();
        stamp();
        substamp(tc);
        constexpr int r0 = issued(t - 1) + 1, nr = issued(t) - issued(t - 1);
        static_for<nr>([&](auto k) { issue(std::integral_constant<int, r0 + decltype(k)::value>{}); });
        substamp(tc);
    };

    // slab 0 carries the residual and the output bias.  Plain loads, the OLDEST vector-memory operations of the wave (memory returns in order: the hand-counted waits
    // below only count what is younger than a DMA), unconditional so that no register copy -- and no wait -- sits at a join: the other slices read b2 four times over
    constexpr int NST = BM * (D / 4) / NT;             // float4 stores per thread
    float4 xres[NST];
    const float4 bres = *reinterpret_cast<const float4 *>(b2 + ((tid & 63) << 2));
#pragma unroll
    for (int it = 0; it < NST; ++it) {
        const float *rp = x2 + (size_t)min(m0 + (tid >> 6) + it * NW, M - 1) * D;
        xres[it] = *reinterpret_cast<const float4 *>((sl == 0 ? rp : b2) + ((tid & 63) << 2));
    }
    // ---- prologue: (P waves) the bias slice, the x2 rows (fp32, row r at r KiB), ring half-steps 0..5
    const uint32_t xs_lds = idf_lds_addr(Xs);
    if (isP) idf_dma16_s(idf_uniform_ptr(b1p + sl * HS), lane16, idf_lds_addr(biasL) + (uint32_t)(wave * 1024));       // 1 KiB: 208 floats + what follows (b1p carries 256 spare floats)
#pragma unroll
    for (int j = 0; j < BM / NW; ++j) {
        const int i = wave + NW * j;
        idf_dma16_s(idf_uniform_ptr(x2 + (size_t)min(m0 + i, M - 1) * D), lane16, xs_lds + (uint32_t)(i * 1024));
    }
    static_for<issued(-1) + 1>([&](auto k) { issue(k); });
    wait_vmcnt(2 * (issued(-1) + 1));                  // the rows (and the bias) are older than the twelve instructions just issued
    // split the rows this wave fetched, in place: lane l holds k = 4l .. 4l+3 of row r -> chunk l >> 1, half (l & 1)   (ffn_h2.h)
#pragma unroll
    for (int j = 0; j < BM / NW; ++j) {
        const int r = wave + NW * j;
        const float4 v = *reinterpret_cast<const float4 *>(Xs + r * 256 + lane * 4);
        uint2 hi, lo;
        split4_pk(v, hi, lo);
        float *dst = Xs + r * 256 + ((((lane >> 1) ^ (r & 15)) << 2)) + ((lane & 1) << 1);
        *reinterpret_cast<uint2 *>(dst) = hi;
        *reinterpret_cast<uint2 *>(dst + 128) = lo;
    }
    stamp();                                           // 1: rows fetched and split

    auto ld8 = [&](const float *p) { return __builtin_bit_cast(h8, *reinterpret_cast<const uint4 *>(p)); };
    float *Cs = ring;                                  // output staging [32][CSS] over slots 0..2 (every ring read is done at barrier 17)

    if (isP) {
        // ---- P wave: linear1 + GELU of unit (hidden tile 2 c + hp, token tile tp) of every chunk c
        const int tp = wave >> 1, hp = wave & 1;
        const int e = g ^ n;
        h8 xh[KS1], xl[KS1];                           // the token tile's planes, all 8 K steps: register-resident
        h8 fa[2][8];                                   // [buffer][4 K steps x (hi, lo')] of the half-step read last
        f32x4 aM = {0.f, 0.f, 0.f, 0.f}, aC = {0.f, 0.f, 0.f, 0.f};
        const float *bias_l = biasL + wave * 256 + 16 * hp + 4 * g;
        static_for<NTICK>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            tick_begin(tc);
            if constexpr (t == 0) {
#pragma unroll
                for (int s = 0; s < KS1; ++s) {
                    const float *row = Xs + (16 * tp + n) * 256 + ((e ^ (4 * s)) << 2);
                    xh[s] = ld8(row);
                    xl[s] = ld8(row + 128);
                }
            }
            if constexpr (t < 14) {                    // read a(t)
                constexpr int c = t >> 1;
                const float *sb = ring + (a_ring(t) % NSLOT) * (SLOTB / 4) + lane * 4;
                if constexpr (c < 6) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        fa[t & 1][2 * s] = ld8(sb + (2 * s + hp) * 512);
                        fa[t & 1][2 * s + 1] = ld8(sb + (2 * s + hp) * 512 + 256);
                    }
                } else if (hp == 0) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        fa[t & 1][2 * s] = ld8(sb + s * 512);
                        fa[t & 1][2 * s + 1] = ld8(sb + s * 512 + 256);
                    }
                }
            }
            substamp(tc);
            if constexpr (t >= 1 && t <= 14) {         // multiply a(t - 1); behind its second half: GELU of the chunk
                constexpr int c = (t - 1) >> 1, h = (t - 1) & 1;
                if (c < 6 || hp == 0) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        IDF_H2_MFMA(aM, fa[(t - 1) & 1][2 * s], xh[4 * h + s]);
                        IDF_H2_MFMA(aC, fa[(t - 1) & 1][2 * s], xl[4 * h + s]);
                        IDF_H2_MFMA(aC, fa[(t - 1) & 1][2 * s + 1], xh[4 * h + s]);
                    }
                }
                if constexpr (h == 1) {
                    // hid planes of a chunk in the READER's lane order, [token tile][plane][K group 0..3][row n][8 halves]: hidden 16 hp + 4 g .. + 3 of the chunk = K group 2 hp + (g >> 1), halves 4 (g & 1) .. + 3
                    float *dst = hid + (c & 1) * (HIDB / 4) + tp * 512 + (2 * hp + (g >> 1)) * 64 + n * 4 + (g & 1) * 2;
                    if (c < 6 || hp == 0) {
                        const float4 bv = *reinterpret_cast<const float4 *>(bias_l + 32 * c);
                        const f2 m01 = {aM[0], aM[1]}, m23 = {aM[2], aM[3]};
                        const f2 c01 = {aC[0], aC[1]}, c23 = {aC[2], aC[3]};
                        const f2 b01 = {bv.x, bv.y}, b23 = {bv.z, bv.w};
                        const f2 g01 = gelu_fast2(m01 + c01 * LO_UNSCALE + b01), g23 = gelu_fast2(m23 + c23 * LO_UNSCALE + b23);
                        const float4 v = make_float4(g01.x, g01.y, g23.x, g23.y);
                        uint2 hi, lo;
                        split4_pk(v, hi, lo);
                        *reinterpret_cast<uint2 *>(dst) = hi;
                        *reinterpret_cast<uint2 *>(dst + 256) = lo;
                    } else {                           // hidden units 208..223 of the slice do not exist: zero planes (the weights there are zero too, but 0 x stale bits may be NaN)
                        *reinterpret_cast<uint2 *>(dst) = uint2{0u, 0u};
                        *reinterpret_cast<uint2 *>(dst + 256) = uint2{0u, 0u};
                    }
                    aM = aC = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            substamp(tc);
        });
    } else {
        // ---- Q wave: linear2, output tiles 4 q + 2 h + i x both token tiles
        const int q = wave - 4;
        f32x4 oM[2][2][2], oC[2][2][2];                // [half h][tile i][token tile]
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int t = 0; t < 2; ++t) oM[h][i][t] = oC[h][i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        h8 fb[2][4];                                   // [buffer][tile i x (hi, lo')]
        h8 hx[2][4];                                   // [chunk & 1][token tile x (hi, lo')]
        static_for<NTICK>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            tick_begin(tc);
            if constexpr (t >= 3 && t <= 16) {         // read b(t - 3) (+ the chunk's hid planes in front of its first half)
                constexpr int j = t - 3, c = j >> 1;
                const float *sb = ring + (b_ring(j) % NSLOT) * (SLOTB / 4) + q * 1024 + lane * 4;
#pragma unroll
                for (int k = 0; k < 4; ++k) fb[t & 1][k] = ld8(sb + k * 256);
                if constexpr ((j & 1) == 0) {
                    const float *hb = hid + (c & 1) * (HIDB / 4) + lane * 4;
#pragma unroll
                    for (int k = 0; k < 4; ++k) hx[c & 1][k] = ld8(hb + k * 256);
                }
            }
            substamp(tc);
            if constexpr (t >= 4) {                    // multiply b(t - 4)
                constexpr int j = t - 4, c = j >> 1, h = j & 1;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) IDF_H2_MFMA(oM[h][i][tt], fb[(t - 1) & 1][2 * i], hx[c & 1][2 * tt]);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) IDF_H2_MFMA(oC[h][i][tt], fb[(t - 1) & 1][2 * i], hx[c & 1][2 * tt + 1]);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) IDF_H2_MFMA(oC[h][i][tt], fb[(t - 1) & 1][2 * i + 1], hx[c & 1][2 * tt]);
            }
            substamp(tc);
        });
        // the partial tile leaves through LDS as 16-byte row stores
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    float4 v;
                    v.x = oM[h][i][tt][0] + oC[h][i][tt][0] * LO_UNSCALE;
                    v.y = oM[h][i][tt][1] + oC[h][i][tt][1] * LO_UNSCALE;
                    v.z = oM[h][i][tt][2] + oC[h][i][tt][2] * LO_UNSCALE;
                    v.w = oM[h][i][tt][3] + oC[h][i][tt][3] * LO_UNSCALE;
                    *reinterpret_cast<float4 *>(Cs + (16 * tt + n) * CSS + (4 * q + 2 * h + i) * 16 + 4 * g) = v;
                }
    }
    stamp();                                           // 20: tile staged
    __syncthreads();
    float *out = parts + (size_t)sl * M * D;
#pragma unroll
    for (int it = 0; it < NST; ++it) {
        const int row = (tid >> 6) + it * NW, c4 = (tid & 63) << 2, gr = m0 + row;
        if (gr >= M) continue;
        float4 v = *reinterpret_cast<const float4 *>(Cs + row * CSS + c4);
        if (sl == 0) {
            const float4 x = xres[it];
            v.x += x.x + bres.x; v.y += x.y + bres.y; v.z += x.z + bres.z; v.w += x.w + bres.w;
        }
        idf_store16_wt(out + (size_t)gr * D + c4, v);
    }
    stamp();                                           // 21: stores issued
}

inline int launch_ffn_h2f(hipStream_t s, const float *x2, int M, const float *pack, const float *b1p, const float *b2, float *parts) {
    static idf_excl_cache excl;
    const int dyn = idf_exclusive_cu(reinterpret_cast<const void *>(&ffn_h2f_kernel<0>), "ffn_h2f_kernel<32 rows>", NT, excl);
    if (dyn != LDS_REQUEST) return IDF_NOT_EXCLUSIVE;
    hipLaunchKernelGGL((ffn_h2f_kernel<0>), dim3((unsigned)(idf_cdiv(M, 32) * NSL)), dim3(NT), LDS_REQUEST, s, x2, M, (int)(idf_cdiv(M, 32) * NSL), pack, b1p, b2, parts, 0);
    return IDF_OK;
}
}  // namespace idf_ffn_h2f
