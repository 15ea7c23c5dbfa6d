// The split-f16 feed-forward block as ONE software pipeline over hidden chunks, on SIXTEEN role-specialised waves (round 6) -- same contract, grid, slabs,
// arithmetic and BITS as ffn_h2.h's ffn_h2_kernel (which stays for the 16- and 64-row tiles), different schedule:
//
//     ffn_h2.h:   [linear1, 8 K steps] -> barrier -> [GELU + split: 3.4 k cycles of VALU with the matrix pipe and the DMA queue idle] -> [linear2, 7 K steps]
//     here:       for every chunk c of 32 hidden units:  linear1(c)  ->  GELU(c)  ->  linear2 += hid(c) . W2[:, c]      all three in flight at once
//
// A hidden unit's pre-activation needs all 256 input columns but only its own W1 row, and linear2's K step q needs only hidden units 32 q .. 32 q + 31: cut by
// hidden CHUNK instead of by phase, the GELU of chunk c runs while the matrix pipe works on chunk c + 1 (linear1) and chunk c - 1 (linear2) and while the weight
// stream of the chunks behind them is landing.  What a K step of ffn_h2.h costs is the weight stream's 1-KiB LDS-DMA pieces going through the CU's vector-memory
// request path (~17-20 cycles apiece for the whole CU, tools/ffn_h2f_probe.hip), and a wave that issues one SITS in its issue slot until the path takes it -- with
// every wave issuing its share behind each barrier, every wave lost 400-550 cycles per step before it could touch its fragments (round 6's first version of this
// kernel, eight waves: profiles/r06_ffn_h2f_probe_v1.txt).  So the roles are separated, four waves per SIMD at 128 registers each (the whole register file):
//   * L waves (12..15): the LOADERS.  They issue every piece of the weight stream (four per half-step each) and own every vmcnt wait; being parked in the issue
//     slot is all they do.
//   * P waves (0..7): linear1 + GELU.  Wave p owns unit u = p & 3 -- (hidden tile 2 c + (u & 1), token tile u >> 1) -- of the chunks c = p >> 2 (mod 2): two
//     half-ticks of 4 K steps x 3 MFMAs, then two half-ticks of bias + GELU + split (a pair of values each) and 2 x 8 bytes per lane into the chunk's hid planes
//     (double-buffered, 4 KiB): the two P waves of a SIMD alternate, so that one's VALU phase sits beside the other's (and the Q wave's) MFMAs.
//   * Q waves (8..11): linear2.  Wave q owns output tiles 4 q .. 4 q + 3 x both token tiles (16 accumulators of 4) and multiplies the chunks as they appear.
// The x2 planes stay in LDS (32 KiB) and are re-read per chunk (LDS reads are 256 B/clk; the request path is what is scarce); the ring is 6 slots of 16 KiB.
//
// Stream order (mdm.py pack_ffn_h2f: a permutation of pack_ffn_h2's 1-KiB fragments): per slice 28 HALF-STEPS in consumption order
//     a0 a1 a2 a3 a4 b0 a5 b1 a6 b2 ... a13 b9 b10 b11 b12 b13
//   a(2 c + h) = linear1 weights of chunk c, K steps 4 h .. 4 h + 3:  [4 K steps][2 hidden tiles][2 planes][1 KiB]   (16 KiB; chunk 6 has ONE hidden tile: 8 KiB)
//   b(2 c + h) = linear2 weights of chunk c (= K step c of phase 2):  [4 Q waves][2 output tiles 4 q + 2 h + i][2 planes][1 KiB]   (16 KiB)
// Half-tick t (one barrier each, 18 of them): P reads and multiplies a(t); the GELU of chunk c runs in half-ticks 2 c + 2 and 2 c + 3, one PAIR of a lane's four values in each
// (a wave's GELU of a pair is a ~500-cycle dependent chain of ~35 instructions: latency, not VALU throughput -- in one piece it was the longest thing between two barriers); Q reads
// b(t - 4) (+ the hid planes of chunk (t - 4) / 2 when t is even) and multiplies it.  A slot read in half-tick t is refilled behind the barrier that opens t + 1.
//
// BIT-IDENTICAL to ffn_h2_kernel: every accumulator sees the same MFMAs on the same fragments in the same order (no K split across waves, chunks in order, per
// K step main, hi x lo', lo' x hi), GELU and splits are the same inline functions -- tests/test_hip_parity.py compares the three row tiles bit for bit.
#pragma once
#include "ffn_h2.h"
#include <utility>

namespace idf_ffn_h2f {
using namespace idf_ffn_h2;

// gelu_fast2 + the pair's f16 split, one value at a time on PLAIN fp32 instructions with every fused multiply-add written out: the same roundings, in the same places, as
// the packed form the compiler makes of `gelu_fast2(m + c * LO_UNSCALE + b)` followed by split4_pk (v_pk_fma_f32 = two v_fma_f32; the last product is rounded once for the hi
// plane and kept unrounded inside the fma that takes the residual).  For code that runs BESIDE the matrix pipe: next to MFMAs a packed fp32 instruction costs a wave 20-30 cycles
// where a plain one costs 4-5 (MI355X_MICROARCH.md; ffn_h2f.h's GELU of a pair took ~850 cycles in the packed form).  The empty asm keeps the SLP vectoriser from re-packing two
// lanes' chains.  Bit-identical to the packed form: tools/ffn_h2f_probe.hip, tests/test_hip_parity.py (the three row tiles of the feed-forward block agree bit for bit).
#pragma clang fp contract(off)
#define IDF_OPAQUE(v) asm("" : "+v"(v))               /* (not volatile: free to move; it only hides the value's origin from the vectoriser) */
__device__ __forceinline__ void gelu_split1_plain(float m, float c, float b, _Float16 &hi, _Float16 &lo) {
    float x = __builtin_fmaf(c, LO_UNSCALE, m) + b;
    IDF_OPAQUE(x);
    float z = __builtin_fabsf(x) * 0.70710678118654752440f;
    IDF_OPAQUE(z);
    float d = __builtin_fmaf(z, 0.3275911f, 1.0f);
    IDF_OPAQUE(d);
    float t = __builtin_amdgcn_rcpf(d);
    IDF_OPAQUE(t);
    float p = __builtin_fmaf(t, 1.061405429f, -1.453152027f);
    IDF_OPAQUE(p);
    p = __builtin_fmaf(t, p, 1.421413741f);
    IDF_OPAQUE(p);
    p = __builtin_fmaf(t, p, -0.284496736f);
    IDF_OPAQUE(p);
    p = __builtin_fmaf(t, p, 0.254829592f);
    IDF_OPAQUE(p);
    p = t * p;
    IDF_OPAQUE(p);
    float zz = z * -z;
    IDF_OPAQUE(zz);
    float ex = __builtin_amdgcn_exp2f(zz * 1.44269504088896340736f);
    IDF_OPAQUE(ex);
    float er = __builtin_fmaf(-ex, p, 1.0f);
    IDF_OPAQUE(er);
    float hx = x * 0.5f, s1 = copysignf(er, x) + 1.0f;
    IDF_OPAQUE(hx);
    IDF_OPAQUE(s1);
    float gv = hx * s1;
    IDF_OPAQUE(gv);
    hi = (_Float16)gv;
    float r = __builtin_fmaf(hx, s1, -(float)hi);
    IDF_OPAQUE(r);
    lo = (_Float16)(r * LO_SCALE);
}
#pragma clang fp contract(fast)


constexpr int FNW = 16, FNT = FNW * 64;              // waves / threads
constexpr int NR = 28;                               // ring half-steps per slice
constexpr int NSLOT = 6, SLOTB = 16384;              // ring slots / bytes of a slot
constexpr int NTICK = 17;                            // half-ticks
constexpr int HIDB = 4096;                           // bytes of one hid-plane chunk buffer: [2 token tiles][2 planes][4 K groups][16 rows][8 halves]
constexpr int FBM = 32;                              // rows per workgroup

__host__ __device__ constexpr int a_ring(int j) { return j < 3 ? j : 2 * j - 3; }          // ring index of a(j), j = 0..13
__host__ __device__ constexpr int b_ring(int j) { return j <= 10 ? 2 * j + 4 : j + 14; }   // ring index of b(j), j = 0..13
__host__ __device__ constexpr int step_bytes(int r) { return (r == a_ring(12) || r == a_ring(13)) ? 8192 : 16384; }      // a12 / a13: one hidden tile
__host__ __device__ constexpr int step_off(int r) {
    int o = 0;
    for (int i = 0; i < r; ++i) o += step_bytes(i);
    return o;
}
static_assert(step_off(NR) == SLICE_BYTES, "the half-steps are the slice");
constexpr int NFILL = 6;                             // the first six half-steps (the ring's first fill) are issued by the Q and L waves, two pieces each, while the P waves fetch the rows; the loaders carry the stream from there
// a wave's DMA instructions of a half-step: role 0 = P, 1 = Q, 2 = L
__host__ __device__ constexpr int my_ins(int r, int role) { return r < 0 || r >= NR ? 0 : (r < NFILL ? (role == 0 ? 0 : 2) : (role == 2 ? step_bytes(r) / 4096 : 0)); }
// last ring index issued once the issue of half-tick t is out (t <= 0: the prologue fills the six slots) / last ring index that must have landed at the barrier that opens half-tick t
__host__ __device__ constexpr int issued(int t) { return t <= 0 ? NFILL - 1 : (t <= 3 ? 5 + t : (2 * t + 2 > NR - 1 ? NR - 1 : 2 * t + 2)); }
__host__ __device__ constexpr int needed(int t) { return t < 3 ? t : (t <= 13 ? 2 * t - 2 : (t <= 16 ? t + 11 : NR - 1)); }
__host__ __device__ constexpr int flying(int t, int role) {    // a wave's DMA instructions that may still be in flight at the barrier that opens half-tick t
    int n = 0;
    for (int r = needed(t) + 1; r <= issued(t - 1); ++r) n += my_ins(r, role);
    return n;
}
__host__ __device__ constexpr bool ring_ok() {       // a slot is refilled only after the half-step it held was read, and nothing is waited for before it is issued
    for (int t = 0; t < NTICK; ++t) {
        if (needed(t) > issued(t - 1)) return false;
        for (int r = issued(t - 1) + 1; r <= issued(t); ++r) {
            const int old = r - NSLOT;                // the half-step this one overwrites: read in a half-tick <= t - 1 ?
            if (old < 0) continue;
            bool read_before = false;
            for (int u = 0; u < t; ++u) read_before = read_before || needed(u) >= old;
            if (!read_before) return false;
        }
    }
    return issued(NTICK - 1) == NR - 1;
}
static_assert(ring_ok(), "ring schedule");

// compile-time loop: the body sees its index as a constant expression (every ring index, slot, wait count and buffer below is an immediate)
template <class F, int... T>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, T...>) { (f(std::integral_constant<int, T>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }
template <int N>
__device__ __forceinline__ void wait_vmcnt_imm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// MODE 0: product; 3: half-tick stamps of threads 0 (P wave 0), 512 (Q wave 8), 768 (L wave 12) behind the slabs (tools/ffn_h2f_probe.hip)
template <int MODE = 0>
__global__ __launch_bounds__(FNT) void ffn_h2f_kernel(const float *__restrict__ x2, int M, int nwg, const float *__restrict__ pack,
                                                       const float *__restrict__ b1p, const float *__restrict__ b2,
                                                       float *__restrict__ parts, int order) {
    extern __shared__ __attribute__((aligned(1024))) float smem[];
    asm volatile("" ::: "v127");                       // the whole register file (4 waves per SIMD x 128): EXCLUSIVE CU (ffn_h2.h)
    float *Xs = smem;                                  // x2 planes: row r at r KiB = [hi 512 B | lo' 512 B], 16-byte chunk t at position t ^ (r & 15)
    float *ring = smem + FBM * 256;                    // 6 x 16 KiB
    float *hid = ring + NSLOT * (SLOTB / 4);           // 2 x 4 KiB
    float *biasL = hid + 2 * (HIDB / 4);               // 8 x 1 KiB: every P wave's private copy of the slice's linear1 bias

    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool isP = wave < 8, isQ = wave >= 8 && wave < 12, isL = wave >= 12;
    long long *stamps = nullptr;
    int n_stamp = 0;
    if constexpr (MODE == 3) stamps = reinterpret_cast<long long *>(parts + (size_t)NSL * M * D) + (size_t)blockIdx.x * 96 + (tid == 512 ? 32 : (tid == 768 ? 64 : 0));
    if constexpr (MODE == 4) stamps = reinterpret_cast<long long *>(parts + (size_t)NSL * M * D) + (size_t)blockIdx.x * 96 + (tid >> 8) * 24;
    auto substamp = [&](auto tc) {                     // MODE 4: inside half-ticks 8..11: the barrier passed / (L) DMA issued / (P, Q) work done
        if constexpr (MODE == 4 && decltype(tc)::value >= 8 && decltype(tc)::value <= 11) {
            if (tid == 0 || tid == 256 || tid == 512 || tid == 768) stamps[n_stamp] = __builtin_readcyclecounter();
            ++n_stamp;
        }
    };
    auto stamp = [&]() {
        if constexpr (MODE == 3) {
            if (tid == 0 || tid == 512 || tid == 768) stamps[n_stamp] = __builtin_readcyclecounter();
            ++n_stamp;
        }
    };
    stamp();
    // workgroup -> (M tile, slice): XCD-affine, M-tile-major (ffn_h2.h order 0)
    const int id = blockIdx.x;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = id & 7;
    const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (id >> 3);
    const int sl = wg % NSL, mt = wg / NSL, m0 = mt * FBM;
    (void)order;
    const float *stream = idf_uniform_ptr(pack + (size_t)sl * SLICE_FLOATS);
    const uint32_t lane16 = lane << 4;
    const int lw = wave & 3;                           // loader index (L waves)
    const uint32_t vsrc = (uint32_t)(lw * 1024) + lane16;
    const uint32_t sdst = idf_lds_addr(ring) + (uint32_t)(lw * 1024);

    auto issue = [&](auto rc) {                        // a loader wave's instructions of ring half-step r: instruction lw + 4 j copies stream bytes [off + 1024 i, + 1024) to slot r % 6
        constexpr int r = decltype(rc)::value;
        constexpr uint32_t so = (uint32_t)step_off(r), dof = (uint32_t)((r % NSLOT) * SLOTB);
#pragma unroll
        for (int j = 0; j < my_ins(r, 2); ++j) idf_dma16_s(stream, vsrc + so + 4096u * j, sdst + dof + 4096u * j);
    };
    auto fill = [&]() {                                // a Q / L wave's two pieces of each of the first NFILL half-steps: pieces w8 and w8 + 8 (w8 = wave - 8)
#pragma unroll
        for (int r = 0; r < NFILL; ++r)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                idf_dma16_s(stream, (uint32_t)((wave - 8 + 8 * j) * 1024) + lane16 + (uint32_t)step_off(r), idf_lds_addr(ring) + (uint32_t)((wave - 8 + 8 * j) * 1024 + (r % NSLOT) * SLOTB));
    };
    // opens half-tick t.  Loaders: their share of everything read in t has landed; then the slots read in t - 1 are refilled.  Everyone: done reading what was read in t - 1.
    auto tick_begin = [&](auto tc, auto rolec) {
        constexpr int t = decltype(tc)::value, role = decltype(rolec)::value;
        constexpr bool L = role == 2;
        if constexpr ((t == 0 || needed(t) != needed(t - 1)) && (L || (role == 1 && needed(t - 1) < NFILL - 1))) wait_vmcnt_imm<flying(t, role)>();
        __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0): my fragment reads of t - 1 (and my hid writes) are done
        __builtin_amdgcn_s_barrier();
        stamp();
        substamp(tc);
        if constexpr (L && t > 0) {
            constexpr int r0 = issued(t - 1) + 1, nr = issued(t) - issued(t - 1);
            static_for<nr>([&](auto k) { issue(std::integral_constant<int, r0 + decltype(k)::value>{}); });
        }
        if constexpr (L) substamp(tc);
    };
    constexpr std::integral_constant<int, 0> ROLE_P{};
    constexpr std::integral_constant<int, 1> ROLE_Q{};
    constexpr std::integral_constant<int, 2> ROLE_L{};

    // The partial tile's 32 rows leave through the twelve P and L waves (the Q waves hold the 16 accumulator tiles and have no registers to spare): storing wave sw takes
    // rows sw, sw + 12, sw + 24.  Slab 0 carries the residual and the output bias: plain loads at kernel entry, the OLDEST vector-memory operations of the wave (memory
    // returns in order: the hand-counted waits only count what is younger than a DMA), unconditional in the slice so that no wait sits at a join -- the other slices read b2.
    constexpr int NST = 3, NSW = 12;
    const int sw = wave < 8 ? wave : wave - 4;
    float4 xres[NST], bres;
    auto residual_loads = [&]() {
        bres = *reinterpret_cast<const float4 *>(b2 + (lane << 2));
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            const float *rp = x2 + (size_t)min(m0 + min(sw + it * NSW, FBM - 1), M - 1) * D;
            const float *src = sl == 0 ? rp : b2;
            asm("" : "+s"(src));                       // (opaque: otherwise the compiler sees the other slices' loads as copies of bres and parks a vmcnt(0) + register copies in front of the prologue)
            xres[it] = *reinterpret_cast<const float4 *>(src + (lane << 2));
        }
    };
    // (P waves: the first to start) four x2 rows each (fp32, row r at r KiB) by DMA, split in place by the wave that fetched them: lane l holds k = 4l .. 4l+3 of row r -> chunk l >> 1, half (l & 1)   (ffn_h2.h)
    const uint32_t xs_lds = idf_lds_addr(Xs);
    auto fetch_rows = [&]() {
#pragma unroll
        for (int j = 0; j < FBM / 8; ++j) {
            const int i = wave + 8 * j;
            idf_dma16_s(idf_uniform_ptr(x2 + (size_t)min(m0 + i, M - 1) * D), lane16, xs_lds + (uint32_t)(i * 1024));
        }
    };
    auto split_rows = [&]() {
#pragma unroll
        for (int j = 0; j < FBM / 8; ++j) {
            const int r = wave + 8 * j;
            const float4 v = *reinterpret_cast<const float4 *>(Xs + r * 256 + lane * 4);
            uint2 hi, lo;
            split4_pk(v, hi, lo);
            float *dst = Xs + r * 256 + ((((lane >> 1) ^ (r & 15)) << 2)) + ((lane & 1) << 1);
            *reinterpret_cast<uint2 *>(dst) = hi;
            *reinterpret_cast<uint2 *>(dst + 128) = lo;
        }
        stamp();                                       // 1: rows fetched and split
    };
    float *Cs = smem;                                  // output staging [32][CSS] over the x2 planes and the head of slot 0 (last read in half-tick 13)
    auto store_rows = [&]() {
        float *out = parts + (size_t)sl * M * D;
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            const int row = sw + it * NSW, c4 = lane << 2, gr = m0 + row;
            if (row >= FBM || gr >= M) continue;
            float4 v = *reinterpret_cast<const float4 *>(Cs + row * CSS + c4);
            if (sl == 0) {
                const float4 x = xres[it];
                v.x += x.x + bres.x; v.y += x.y + bres.y; v.z += x.z + bres.z; v.w += x.w + bres.w;
            }
            idf_store16_wt(out + (size_t)gr * D + c4, v);
        }
        stamp();                                       // 21: stores issued
    };
    auto ld8 = [&](const float *p) { return __builtin_bit_cast(h8, *reinterpret_cast<const uint4 *>(p)); };

    if (isP) {
        // ---- P wave: linear1 + GELU of unit (hidden tile 2 c + hp, token tile tp) of the chunks c = cp (mod 2)
        residual_loads();
        idf_dma16_s(idf_uniform_ptr(b1p + sl * HS), lane16, idf_lds_addr(biasL) + (uint32_t)(wave * 1024));       // the slice's linear1 bias, 1 KiB: 208 floats + what follows (b1p carries 256 spare floats)
        fetch_rows();
        wait_vmcnt_imm<0>();
        split_rows();
        const int u = wave & 3, tp = u >> 1, hp = u & 1, cp = wave >> 2;
        const int e = g ^ n;
        f32x4 aM = {0.f, 0.f, 0.f, 0.f}, aC = {0.f, 0.f, 0.f, 0.f};
        const float *bias_l = biasL + wave * 256 + 16 * hp + 4 * g;
        const float *xrow = Xs + (16 * tp + n) * 256;
        static_for<NTICK>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            tick_begin(tc, ROLE_P);
            if constexpr (t < 14) {                    // a(t): chunk t / 2, K steps 4 h .. 4 h + 3
                constexpr int c = t >> 1, h = t & 1;
                if (cp == (c & 1) && (c < 6 || hp == 0)) {
                    const float *sb = ring + (a_ring(t) % NSLOT) * (SLOTB / 4) + lane * 4 + (c < 6 ? hp * 512 : 0);
                    constexpr int ks = c < 6 ? 1024 : 512;         // floats from one K step's fragments to the next
                    h8 wh[4], wl[4], xh[4], xl[4];
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        wh[s] = ld8(sb + s * ks);
                        wl[s] = ld8(sb + s * ks + 256);
                        const float *row = xrow + ((e ^ (4 * (4 * h + s))) << 2);
                        xh[s] = ld8(row);
                        xl[s] = ld8(row + 128);
                    }
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        IDF_H2_MFMA(aM, wh[s], xh[s]);
                        IDF_H2_MFMA(aC, wh[s], xl[s]);
                        IDF_H2_MFMA(aC, wl[s], xh[s]);
                    }
                }
            }
            if constexpr (t >= 2 && t <= 14 && (t & 1) == 0) {     // GELU of chunk (t - 2) / 2: a lane's four values as four independent chains of plain fp32 instructions (ffn_h2.h gelu_split1_plain)
                constexpr int c = (t - 2) >> 1;
                if (cp == (c & 1)) {
                    // hid planes of a chunk in the READER's lane order, [token tile][plane][K group 0..3][row n][8 halves]: hidden 16 hp + 4 g .. + 3 of the chunk = K group 2 hp + (g >> 1), halves 4 (g & 1) .. + 3
                    float *dst = hid + (c & 1) * (HIDB / 4) + tp * 512 + (2 * hp + (g >> 1)) * 64 + n * 4 + (g & 1) * 2;
                    uint2 ghi = {0u, 0u}, glo = {0u, 0u};      // (hidden units 208..223 of the slice do not exist: zero planes -- the weights there are zero too, but 0 x stale bits may be NaN)
                    if (c < 6 || hp == 0) {
                        const float4 bv = *reinterpret_cast<const float4 *>(bias_l + 32 * c);
                        __builtin_amdgcn_s_setprio(3);
                        _Float16 h0, l0, h1, l1, h2, l2, h3, l3;
                        gelu_split1_plain(aM[0], aC[0], bv.x, h0, l0);
                        gelu_split1_plain(aM[1], aC[1], bv.y, h1, l1);
                        gelu_split1_plain(aM[2], aC[2], bv.z, h2, l2);
                        gelu_split1_plain(aM[3], aC[3], bv.w, h3, l3);
                        ghi = uint2{__builtin_bit_cast(unsigned, (h2v{h0, h1})), __builtin_bit_cast(unsigned, (h2v{h2, h3}))};
                        glo = uint2{__builtin_bit_cast(unsigned, (h2v{l0, l1})), __builtin_bit_cast(unsigned, (h2v{l2, l3}))};
                        __builtin_amdgcn_s_setprio(0);
                    }
                    *reinterpret_cast<uint2 *>(dst) = ghi;
                    *reinterpret_cast<uint2 *>(dst + 256) = glo;
                    aM = aC = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            substamp(tc);
        });
        stamp();                                       // 19
        __syncthreads();
        store_rows();
    } else if (isQ) {
        // ---- Q wave: linear2, output tiles 4 q + 2 h + i x both token tiles
        fill();
        stamp();
        const int q = wave - 8;
        f32x4 oM[2][2][2], oC[2][2][2];                // [half h][tile i][token tile]
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int t = 0; t < 2; ++t) oM[h][i][t] = oC[h][i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        h8 hx[4];                                      // the chunk's hid planes: [token tile x (hi, lo')], read in front of the chunk's first half
        static_for<NTICK>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            tick_begin(tc, ROLE_Q);
            if constexpr (t >= 3) {                    // b(t - 3)
                constexpr int j = t - 3, c = j >> 1, h = j & 1;
                const float *sb = ring + (b_ring(j) % NSLOT) * (SLOTB / 4) + q * 1024 + lane * 4;
                h8 fb[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) fb[k] = ld8(sb + k * 256);
                if constexpr (h == 0) {
                    const float *hb = hid + (c & 1) * (HIDB / 4) + lane * 4;
#pragma unroll
                    for (int k = 0; k < 4; ++k) hx[k] = ld8(hb + k * 256);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) IDF_H2_MFMA(oM[h][i][tt], fb[2 * i], hx[2 * tt]);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) IDF_H2_MFMA(oC[h][i][tt], fb[2 * i], hx[2 * tt + 1]);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) IDF_H2_MFMA(oC[h][i][tt], fb[2 * i + 1], hx[2 * tt]);
            }
            substamp(tc);
        });
        // the partial tile goes to the staging area (every read of the x2 planes and of slot 0 was over at barrier 15)
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
        stamp();                                       // 20: tile staged
        __syncthreads();
    } else {
        // ---- L wave: nothing but the weight stream (and its share of the rows at both ends)
        residual_loads();
        fill();
        stamp();
        static_for<NTICK>([&](auto tc) { tick_begin(tc, ROLE_L); });
        stamp();                                       // 19
        __syncthreads();
        store_rows();
    }
}

inline int launch_ffn_h2f(hipStream_t s, const float *x2, int M, const float *pack, const float *b1p, const float *b2, float *parts) {
    static idf_excl_cache excl;
    const int dyn = idf_exclusive_cu(reinterpret_cast<const void *>(&ffn_h2f_kernel<0>), "ffn_h2f_kernel<32 rows>", FNT, excl);
    if (dyn != LDS_REQUEST) return IDF_NOT_EXCLUSIVE;
    hipLaunchKernelGGL((ffn_h2f_kernel<0>), dim3((unsigned)(idf_cdiv(M, FBM) * NSL)), dim3(FNT), LDS_REQUEST, s, x2, M, (int)(idf_cdiv(M, FBM) * NSL), pack, b1p, b2, parts, 0);
    return IDF_OK;
}
}  // namespace idf_ffn_h2f
