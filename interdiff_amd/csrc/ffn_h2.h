// Fused feed-forward block on the f16 matrix pipe with fp32-grade results ("split-f16", round 4):
//
//     u3 = x2 + gelu(x2 . W1^T + b1) . W2^T + b2          same contract, grid, slabs and epilogue as csrc/ffn.h
//
// gfx950 has no TF32 path: an fp32-input MFMA runs at the vector rate (157 TFLOP/s), 1/16 of the f16 / bf16 rate, and the exact-fp32
// kernel of ffn.h spends 26.6 k of its 40 k cycles per workgroup issuing it.  Here every fp32 operand v is split ONCE into two
// half-precision planes
//         hi  = f16(v)                      (0 when |v| < 2^-14: no subnormal ever enters the hi plane)
//         lo' = f16((v - f32(hi)) * 2^11)   (the residual, exact in fp32, scaled back into the normal range)
// so that v = hi + lo' 2^-11 up to 2^-23 |v| (an f16 keeps 11 significant bits, the residual has at most 12 + sign), and
//         x . w  =  xh wh  +  2^-11 (xh wl' + xl' wh)  +  O(2^-22 |x w|)
// is THREE v_mfma_f32_16x16x32_f16 per 32 k instead of EIGHT v_mfma_f32_16x16x4_f32 at 1/16 of the rate: 1/43 of the matrix-pipe
// time.  Products of halves are exact in fp32, both sums are accumulated in fp32 (a main and a correction accumulator per tile; the
// 2^-11 is applied once, in fp32, when the tile leaves).  A bf16 split would need three planes (8 bits each), six products and
// 6 bytes per weight in the stream; the f16 split needs two planes, three products and the SAME 4 bytes per weight as fp32 -- the
// stream, which is what bounds this kernel now, does not grow.  The price of f16 is its range: |v| must stay below 65504.  The host
// packer proves that at pack time from the weights (mdm.py ffn_h2_range_ok: LayerNorm bounds the input rows, |W1| row sums bound the
// hidden activations) and leaves ffn_pack_h2 = 0 -- this kernel unreachable -- for a layer where it cannot.
//
// Everything else follows ffn.h: grid = ceil(M / BM) x 5 hidden slices of 208 units, 512 threads; the slice's weights arrive as ONE
// contiguous pre-packed stream by inline-asm LDS-DMA through a ring of 32-KiB slots with hand-counted vmcnt waits; the partial
// [BM,256] tile goes to slab `slice` (slab 0 + residual + b2).  What differs:
//  * a ring slot holds one K = 32 step: [tile][plane][lane][8 halves] -- every (tile, plane) fragment is 1 KiB contiguous in lane
//    order, i.e. exactly what one ds_read_b128 of a wave fetches, conflict-free without a swizzle (mdm.py pack_ffn_h2);
//    phase 1: 13 hidden tiles x 2 planes = 26 KiB per step, 8 steps; phase 2: 16 output tiles x 2 planes = 32 KiB, 7 steps
//    (K = 208 padded to 224 with zero weights): 432 KiB per slice;
//  * the WEIGHTS are the MFMA's A operand and the token rows its B operand (the two layouts are mirror images, the swap is free):
//    D[i][n] then has lane n = token and registers i = 4 consecutive hidden units / output columns, so the GELU phase writes 8-byte
//    pieces of the hid planes and the epilogue 16-byte pieces of the output tile (ffn.h: scalar LDS stores);
//  * x2 rows land by DMA as fp32 and are split IN PLACE by the wave that fetched them (row r: hi plane at r KiB, lo' plane 512 B
//    behind it, 16-byte chunk t at position t ^ (r & 15)); the hid planes overwrite them in the same image;
//  * no K split across waves anywhere (ffn.h's 32-row kernel shares one column tile between two waves): every output element is
//    summed by one accumulator pair over the steps in order, whatever the row tile -- the 16-, 32- and 64-row instantiations are
//    BIT-IDENTICAL, so chains and shards of a batch may pick their tile freely (diffusion.py, dist.py);
//  * one barrier per K step; the fragments of step P are fetched right behind the barrier that publishes them and the MFMAs of step
//    P - 1 run while they land (register double buffer).
#pragma once
#include "common.h"
#include "philox.h"
#include "ffn.h"

namespace idf_ffn_h2 {

constexpr int D = IDF_MDM_D, FF = IDF_MDM_FF;
constexpr int NSL = IDF_FFN_SLICES;                 // hidden slices = partial slabs (5)
constexpr int NW = 8, NT = NW * 64;
constexpr int HS = 208, NH1 = HS / 16;              // hidden units / hidden tiles of a slice
constexpr int NO2 = D / 16;                         // output column tiles
constexpr int KS1 = D / 32, KS2 = (HS + 31) / 32;   // K steps of 32: 8 (phase 1), 7 (phase 2, 208 -> 224)
constexpr int NPAIR = KS1 + KS2;                    // ring fills ("pairs" in ffn.h's vocabulary: one K step each here)
constexpr int P1B = NH1 * 2 * 1024, P2B = NO2 * 2 * 1024;      // bytes of a phase-1 / phase-2 step: 26 KiB / 32 KiB
constexpr int SLICE_BYTES = KS1 * P1B + KS2 * P2B;  // 442368 = 432 KiB of stream per slice
constexpr int SLICE_FLOATS = SLICE_BYTES / 4;
constexpr int SLOT = 32768;                         // bytes of a ring slot
constexpr int CSS = D + 4;                          // row stride (floats) of the output staging tile
constexpr float LO_SCALE = 2048.0f, LO_UNSCALE = 1.0f / 2048.0f;

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

__host__ __device__ constexpr int step_off(int P) { return P < KS1 ? P * P1B : KS1 * P1B + (P - KS1) * P2B; }       // bytes into the slice stream
__host__ __device__ constexpr int step_ins(int P) { return P >= NPAIR ? 0 : (P < KS1 ? P1B / 1024 : P2B / 1024); }  // 1-KiB DMA instructions: 26 / 32

__device__ __forceinline__ void wait_vmcnt(int n) {          // n is a constant after unrolling
    switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    }
}

// v = hi + lo' / 2048 (see the header).  Both conversions round to nearest even (v_cvt_f16_f32).
__device__ __forceinline__ void split1(float v, _Float16 &hi, _Float16 &lo) {
    _Float16 h = (_Float16)v;
    if (__builtin_fabsf(v) < 6.103515625e-05f) h = (_Float16)0.0f;
    hi = h;
    lo = (_Float16)((v - (float)h) * LO_SCALE);
}
__device__ __forceinline__ void split4(const float4 v, uint2 &hi, uint2 &lo) {
    _Float16 a0, a1, a2, a3, b0, b1, b2, b3;
    split1(v.x, a0, b0);
    split1(v.y, a1, b1);
    split1(v.z, a2, b2);
    split1(v.w, a3, b3);
    hi = __builtin_bit_cast(uint2, (h4{a0, a1, a2, a3}));
    lo = __builtin_bit_cast(uint2, (h4{b0, b1, b2, b3}));
}

// The same split WITHOUT the hi plane's flush rule, on packed instructions (v_cvt_pk_f16_f32, v_pk_add_f32, v_pk_mul_f32: 3 VALU instructions per
// element instead of 8).  gfx950's f16 MFMA honours subnormal inputs (tools/mfma_f16_subnormal_probe.hip: 2^-24 x 2^10 comes out exact), so a subnormal
// hi is as good as a flushed one.  Every IN-KERNEL split of activations uses this form (x2 rows and hidden activations of the feed-forward kernel, the
// QKV kernel's scaled rows, the row block's A operands in csrc/denoiser.hip); the pre-packed WEIGHT planes keep the rule (mdm.py split_f16 -- either is exact
// to 2^-22, the two only differ below 2^-14).  tests/ffn_emulator.py restates both.
typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split4_pk(const float4 v, uint2 &hi, uint2 &lo) {
    const f2 a = {v.x, v.y}, b = {v.z, v.w};
    const h2v ha = __builtin_convertvector(a, h2v), hb = __builtin_convertvector(b, h2v);
    const f2 ra = (a - __builtin_convertvector(ha, f2)) * LO_SCALE, rb = (b - __builtin_convertvector(hb, f2)) * LO_SCALE;
    const h2v la = __builtin_convertvector(ra, h2v), lb = __builtin_convertvector(rb, h2v);
    hi = uint2{__builtin_bit_cast(unsigned, ha), __builtin_bit_cast(unsigned, hb)};
    lo = uint2{__builtin_bit_cast(unsigned, la), __builtin_bit_cast(unsigned, lb)};
}
__device__ __forceinline__ void split1_nf(float v, _Float16 &hi, _Float16 &lo) {
    hi = (_Float16)v;
    lo = (_Float16)((v - (float)hi) * LO_SCALE);
}

// the two f16 planes of a token row (split4_pk) for a kernel that keeps rows one per 16-lane group (common.h Row16): lane l16 owns the 4-float chunks
// {l16, 16 + l16, 32 + l16, 48 + l16}; hi_row / lo_row point at the row's first half in each plane
__device__ __forceinline__ void row16_store_planes(const Row16 &r, _Float16 *hi_row, _Float16 *lo_row, int l16) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint2 h, l;
        split4_pk(r.c[i], h, l);
        *reinterpret_cast<uint2 *>(hi_row + (i * 16 + l16) * 4) = h;
        *reinterpret_cast<uint2 *>(lo_row + (i * 16 + l16) * 4) = l;
    }
}

// gelu_fast (common.h) on a PAIR of values: packed fp32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) cost a wave what the
// plain ones cost, so the polynomial runs at half the issue slots; rcp / exp stay per element.  Same operations in the same order as gelu_fast.
__device__ __forceinline__ f2 gelu_fast2(f2 x) {
    const f2 ax = {__builtin_fabsf(x.x), __builtin_fabsf(x.y)};
    const f2 z = ax * 0.70710678118654752440f;
    const f2 d = 1.0f + 0.3275911f * z;
    const f2 t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    const f2 p = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
    const f2 zz = -(z * z);
    const f2 ex = {__expf(zz.x), __expf(zz.y)};
    const f2 er = 1.0f - p * ex;
    const f2 se = {copysignf(er.x, x.x), copysignf(er.y, x.y)};
    return 0.5f * x * (1.0f + se);
}

#define IDF_H2_MFMA(acc, a, b) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0)

// Sixteen consecutive floats at a wave-uniform address into SGPRs through the scalar cache, waited for inside the block.  As inline asm on purpose: a plain load
// here would be a vector (flat) load whose compiler-inserted wait is `vmcnt(0)` -- the pass cannot see the LDS-DMA stream that is in flight around it.
typedef float f4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void idf_sload16(const float *p, f4s &a, f4s &b, f4s &c, f4s &d) {
    p = idf_uniform_ptr(p);
    asm volatile("s_load_dwordx4 %0, %4, 0x0\n\ts_load_dwordx4 %1, %4, 0x10\n\ts_load_dwordx4 %2, %4, 0x20\n\ts_load_dwordx4 %3, %4, 0x30\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(a), "=&s"(b), "=&s"(c), "=&s"(d)
                 : "s"(p)
                 : "memory");
}

// TT = token tiles of 16 rows per workgroup (1, 2, 4 -> BM = 16, 32, 64); S = ring slots (4; 2 for TT = 4, where the planes take 64 KiB).
// LDS: BM KiB of planes + S x 32 KiB ring = 144 / 160 / 128 KiB (the launcher asks for all 160 either way: exclusive CU).
// MODE 0 is the product kernel; 1 = no MFMAs, 2 = no DMA after the prologue, 3 = phase stamps of thread 0 behind the slabs, 4 = no slab stores
// (tools/ffn_h2_probe.hip only; `if constexpr` keeps every trace of them out of MODE 0).
// LW = LOADER waves (round 6; 0 or 8): with 8, the workgroup is sixteen waves at 128 registers each (still the whole register file) and the weight ring is fed by waves 8..15 alone --
// they issue every LDS-DMA piece of a step (three or four each) and own every vmcnt wait -- while waves 0..7 compute exactly as before without ever touching the vector-memory issue:
// a wave that issues a 1-KiB piece SITS in its issue slot until the CU's request path takes it (60-100 cycles apiece with eight waves asking), which is what a K step cost beyond its
// 192-384 MFMA cycles (tools/experiments/ffn_h2f.h measured it).  Same instructions per accumulator: same bits.
template <int TT, int S, int MODE = 0, int LW = 0>
__global__ __launch_bounds__(NT + 64 * LW) void ffn_h2_kernel(const float *__restrict__ x2, int M, int nwg, const float *__restrict__ pack,
                                                     const float *__restrict__ b1p, const float *__restrict__ b2,
                                                     float *__restrict__ parts, int order) {
    constexpr int BM = 16 * TT;
    static_assert((TT == 1 || TT == 2 || TT == 4) && (S == 2 || S == 3 || S == 4), "geometry");
    static_assert(BM * 1024 + S * SLOT <= 160 * 1024, "LDS");
    static_assert(TT == 4 ? BM * CSS * 4 <= BM * 1024 + S * SLOT : BM * CSS * 4 <= 2 * SLOT, "output staging");
    extern __shared__ __attribute__((aligned(1024))) float smem[];
    static_assert(LW == 0 || LW == 8, "loader waves");
    if constexpr (LW == 0) asm volatile("" ::: "v255");            // the whole register file: see EXCLUSIVE CU below
    else asm volatile("" ::: "v127");                              // (sixteen waves: four per SIMD x 128)
    constexpr int NTH = NT + 64 * LW;                              // threads of the workgroup
    float *Xs = smem;                                              // planes: row r at r KiB = [hi 512 B | lo' 512 B]
    float *ring = smem + BM * 256;                  // (the slice's linear1 bias comes through the scalar cache: with four ring slots the planes and the ring are the whole 160 KiB)
    // (all 13 argument dwords -- the grid size among them, instead of gridDim.x from the hidden block -- arrive preloaded in SGPRs: build.py; no argument-segment read before the first DMA)

    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = LW > 0 && wave >= NW;                      // waves 8..15: the weight stream and nothing else
    const bool issues = LW == 0 || loader;                         // this wave issues (and waits for) ring pieces
    const int iw = LW > 0 ? (wave & 7) : wave;                     // its index among the issuing waves
    long long *stamps = nullptr;
    int n_stamp = 0;
    if constexpr (MODE == 3) stamps = reinterpret_cast<long long *>(parts + (size_t)NSL * M * D) + (size_t)blockIdx.x * 32;
    auto stamp = [&]() {
        if constexpr (MODE == 3) {
            if (tid == 0) stamps[n_stamp] = __builtin_readcyclecounter();
            ++n_stamp;
        }
    };
    stamp();
    // Workgroup order.  order 0 (shipped): M-tile-major over XCD-AFFINE logical ids (ffn.h xcd_affine_tile: all five slices of an M tile on one
    // XCD -- its x2 rows cross the fabric once, its slabs are written from one XCD).  A/B only (tools/ffn_h2_ab.py): order 1 = slice-major
    // affine ids (an XCD streams one or two of the five 432-KiB weight streams: 13 instead of 40 slice loads per launch), order 2 = plain ids
    // (rounds 1-3: the five slices of a tile on five XCDs).  One process (profiles/r04_ffn_split_f16_ab.txt): bursts 10.6 / 10.5 / 12.1 us,
    // denoiser forward 212 / 234-238 / 226-230 us, whole samples with correction 0.2363 / 0.2446 / 0.2437 ms per step for orders 0 / 1 / 2.
    const int id = blockIdx.x, nmt = nwg / NSL;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = id & 7;
    const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (id >> 3);
    // order 0: ffn.h's M-tile-major ids; 1: slice-major over XCD-affine ids; 2: M-tile-major over XCD-affine ids (an XCD holds ALL slices of its M tiles: x2 rows
    // cross the fabric once instead of five times, every XCD still streams all five slices)   (A/B: tools/ffn_h2_ab.py)
    const int sl = order == 1 ? wg / nmt : (order == 2 ? id % NSL : wg % NSL), mt = order == 1 ? wg - sl * nmt : (order == 2 ? id / NSL : wg / NSL), m0 = mt * BM;
    const float *stream = idf_uniform_ptr(pack + (size_t)sl * SLICE_FLOATS);
    const uint32_t lane16 = lane << 4;
    const uint32_t vsrc = (uint32_t)(iw * 1024) + lane16;                   // this lane's 16 B inside a step: instruction iw + 8 j adds 8192 j
    const uint32_t sdst = idf_lds_addr(ring) + (uint32_t)(iw * 1024);       // LDS byte address of this wave's first instruction in slot 0
    const bool lt2 = iw < 2;                                                // 26 = 3 * 8 + 2: issuing waves 0 and 1 issue a 4th instruction for a phase-1 step

    auto issue_step = [&](int P) {                    // DMA instructions of step P: instruction i = iw + 8 j copies stream bytes [step_off + 1024 i, +1024) to slot P % S
        if (P >= NPAIR || !issues) return;
        if constexpr (MODE == 2) { if (P >= S - 1) return; }
        const int nins = step_ins(P);
        const uint32_t so = (uint32_t)step_off(P), dof = (uint32_t)((P % S) * SLOT);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (8 * j + 7 < nins) idf_dma16_s(stream, vsrc + so + 8192u * j, sdst + dof + 8192u * j);
            else if (8 * j < nins && lt2) idf_dma16_s(stream, vsrc + so + 8192u * j, sdst + dof + 8192u * j);
        }
    };
    // wait until this wave's DMAs of step P have landed: the steps P + 1 .. P + S - 2 issued behind them may keep flying (S = 3: one step, S = 4: two)
    auto wait_step = [&](int P) {
        int flying = 0, ragged = 0;                   // this wave's instructions of the younger steps P + 1 .. P + S - 2 (constants after unrolling)
#pragma unroll
        for (int Q = P + 1; Q <= P + S - 2; ++Q) {
            bool live = Q < NPAIR;
            if constexpr (MODE == 2) live = live && Q < S - 1;
            if (live) {
                flying += step_ins(Q) / 8;
                ragged += step_ins(Q) % 8 != 0 ? 1 : 0;
            }
        }
        if (!issues) return;
        if (lt2) wait_vmcnt(flying + ragged);
        else wait_vmcnt(flying);
    };

    // ---- prologue: bias slice, x2 rows (fp32, row r at r KiB, linear), the first S - 1 steps
    const uint32_t xs_lds = idf_lds_addr(Xs);
    if (!loader) {
#pragma unroll
        for (int j = 0; j < BM / NW; ++j) {
            const int i = wave + NW * j;
            idf_dma16_s(idf_uniform_ptr(x2 + (size_t)min(m0 + i, M - 1) * D), lane16, xs_lds + (uint32_t)(i * 1024));
        }
    }
#pragma unroll
    for (int P = 0; P < S - 1; ++P) issue_step(P);
    // linear1 bias of this wave's hidden tiles (phase-1 ownership below): sixteen floats each, through the scalar cache,
    // behind the DMA issue (the wave is about to wait for its rows anyway); b1p carries 256 spare floats, so the second tile's address is valid for every wave
    // Phase-1 ownership.  With loader waves (TT = 2, LW = 8) the 26 (hidden tile, token tile) UNITS are spread 4 + 3 | 4 + 3 | 3 + 3 | 3 + 3 over the SIMDs' computing wave pairs
    // (w, w + 4): every wave owns hidden tile ta with both token tiles and tile tb with both (waves 0, 1) or with ONE token tile (waves 2..7: tiles 5, 9 and 12 are shared by two
    // waves, odd waves take token tile 1).  Whole tiles per wave (two for waves 0..4, one for 5..7) put 8 units on SIMD 0 and 6 on the others, and SIMD 0 sets the length of the
    // VALU-bound GELU phase (~430 cycles per unit and SIMD).  In the eight-wave form this balance measured WORSE (the waves that refill the ring after their matrix work must stay the
    // light ones: profiles/r06_ffn_balanced_units_not_adopted.txt); with the DMA issue on the loader waves that constraint is gone.  A unit's arithmetic is what it was: same bits.
    constexpr bool BAL = TT == 2 && LW == 8;
    const int ta = BAL ? (int)((0x7BA86420u >> (4 * (wave & 7))) & 15u) : (wave < 5 ? 2 * wave : 5 + wave);
    const int tb = BAL ? (int)((0xCC995531u >> (4 * (wave & 7))) & 15u) : ta + 1;
    const bool has_b = BAL || wave < 5;                                    // this wave multiplies for a second hidden tile (accumulator slot 1)
    const bool full_b = BAL ? wave < 2 : wave < 5;                         // ... with every token tile
    const bool sel1 = BAL && wave >= 2 && (wave & 1);                      // ... with ONE token tile: tile 1 (odd waves) or 0, kept in accumulator slot [1][0]
    f4s bias_s[2][4];
    if (!loader) {
        idf_sload16(b1p + sl * HS + ta * 16, bias_s[0][0], bias_s[0][1], bias_s[0][2], bias_s[0][3]);
        idf_sload16(b1p + sl * HS + tb * 16, bias_s[1][0], bias_s[1][1], bias_s[1][2], bias_s[1][3]);
    }
    {   // the x2 rows are older than this wave's share of the steps just issued
        int younger = 0;
#pragma unroll
        for (int P = 0; P < S - 1; ++P) younger += step_ins(P) / 8;         // + 1 for waves 0, 1 per ragged step
        if constexpr (LW > 0) {
            if (!loader) wait_vmcnt(0);                                     // (a computing wave issued nothing but its rows)
        } else {
            if (lt2) wait_vmcnt(younger + (S - 1));
            else wait_vmcnt(younger);
        }
    }
    // split the rows this wave fetched, in place: lane l holds k = 4l .. 4l+3 of row r -> chunk l >> 1, half (l & 1)
    if (!loader) {
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
    }

    // ---- tile maps.  Phase 1: above (ta, tb, has_b, full_b, sel1).  Phase 2: 16 output tiles, wave w owns 2w, 2w+1 with every token tile.
    const bool two1 = has_b;
    const int o0 = 2 * wave;
    f32x4 accM[2][TT], accC[2][TT];                   // phase 1: [0] = tile ta, [1] = tile tb (balanced map, one token tile only: slot [1][0] holds token tile sel1)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int t = 0; t < TT; ++t) accM[a][t] = accC[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    struct Frags {
        h8 wh[2], wl[2], xh[TT], xl[TT];
    };
    Frags F[2];
    const int e = g ^ n;                                                    // token-plane chunk (4 s + g) of row n sits at position (e ^ 4 s)
    auto ld8 = [&](const float *p) { return __builtin_bit_cast(h8, *reinterpret_cast<const uint4 *>(p)); };
    auto read_x = [&](int s, Frags &f) {              // B operand: planes of the token tiles, K step s (x2 planes in phase 1, hid planes in phase 2)
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const float *row = Xs + (16 * t + n) * 256 + ((e ^ (4 * s)) << 2);
            f.xh[t] = ld8(row);
            f.xl[t] = ld8(row + 128);
        }
    };
    auto read1 = [&](int s, Frags &f) {               // A operand: hidden tiles ta (, tb) of step s
        const float *sb = ring + (s % S) * (SLOT / 4) + lane * 4;
        f.wh[0] = ld8(sb + ta * 512);
        f.wl[0] = ld8(sb + ta * 512 + 256);
        if (two1) {
            f.wh[1] = ld8(sb + tb * 512);
            f.wl[1] = ld8(sb + tb * 512 + 256);
        }
        read_x(s, f);
    };
    auto read2 = [&](int q, Frags &f) {               // A operand: output tiles o0, o0 + 1 of phase-2 step q
        const float *sb = ring + ((KS1 + q) % S) * (SLOT / 4) + o0 * 512 + lane * 4;
        f.wh[0] = ld8(sb);
        f.wl[0] = ld8(sb + 256);
        f.wh[1] = ld8(sb + 512);
        f.wl[1] = ld8(sb + 768);
        read_x(q, f);
    };
    auto mma = [&](const Frags &f, bool both) {
        if constexpr (MODE == 1) {
            asm volatile("" ::"v"(f.wh[0]), "v"(f.wl[0]), "v"(f.wh[1]), "v"(f.wl[1]), "v"(f.xh[0]), "v"(f.xl[0]));
            return;
        }
#pragma unroll
        for (int t = 0; t < TT; ++t) IDF_H2_MFMA(accM[0][t], f.wh[0], f.xh[t]);
        if (both) {
#pragma unroll
            for (int t = 0; t < TT; ++t) IDF_H2_MFMA(accM[1][t], f.wh[1], f.xh[t]);
        }
#pragma unroll
        for (int t = 0; t < TT; ++t) IDF_H2_MFMA(accC[0][t], f.wh[0], f.xl[t]);
        if (both) {
#pragma unroll
            for (int t = 0; t < TT; ++t) IDF_H2_MFMA(accC[1][t], f.wh[1], f.xl[t]);
        }
#pragma unroll
        for (int t = 0; t < TT; ++t) IDF_H2_MFMA(accC[0][t], f.wl[0], f.xh[t]);
        if (both) {
#pragma unroll
            for (int t = 0; t < TT; ++t) IDF_H2_MFMA(accC[1][t], f.wl[1], f.xh[t]);
        }
    };
    // phase 1 of the balanced map: tile ta x both token tiles, tile tb x (both | the one token tile sel1, in accumulator slot [1][0]).  Per accumulator the same three MFMAs per K step, in the same order.
    auto mma1 = [&](const Frags &f) {
        if constexpr (!BAL) {
            mma(f, two1);
        } else {
            const h8 xbh = sel1 ? f.xh[TT - 1] : f.xh[0], xbl = sel1 ? f.xl[TT - 1] : f.xl[0];
#pragma unroll
            for (int t = 0; t < TT; ++t) IDF_H2_MFMA(accM[0][t], f.wh[0], f.xh[t]);
            IDF_H2_MFMA(accM[1][0], f.wh[1], xbh);
            if (full_b) IDF_H2_MFMA(accM[1][TT - 1], f.wh[1], f.xh[TT - 1]);
#pragma unroll
            for (int t = 0; t < TT; ++t) IDF_H2_MFMA(accC[0][t], f.wh[0], f.xl[t]);
            IDF_H2_MFMA(accC[1][0], f.wh[1], xbl);
            if (full_b) IDF_H2_MFMA(accC[1][TT - 1], f.wh[1], f.xl[TT - 1]);
#pragma unroll
            for (int t = 0; t < TT; ++t) IDF_H2_MFMA(accC[0][t], f.wl[0], f.xh[t]);
            IDF_H2_MFMA(accC[1][0], f.wl[1], xbh);
            if (full_b) IDF_H2_MFMA(accC[1][TT - 1], f.wl[1], f.xh[TT - 1]);
        }
    };
    auto publish = [&](int P) {                       // step P has landed for every wave, and every wave is done with step P - 1's slot
        wait_step(P);
        __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0) as a builtin: the compiler then KNOWS the LDS queue is empty
        __builtin_amdgcn_s_barrier();
    };

    // ---- phase 1: hid^T[hidden][token] = W1[slice] . x2^T
    // The DMA issue is what a K step costs now (26 - 32 instructions per step at the CU's ~20 cycles apiece; the MFMAs of a step are 192
    // cycles per wave), and a wave sits in its own issue: the two waves of a SIMD (w, w + 4) therefore take turns -- waves 0..3 refill
    // the ring BEFORE their fragment reads and MFMAs, waves 4..7 AFTER theirs -- so that one wave's issue runs beside the other's matrix work.
    const bool early = LW > 0 || S == 2 || wave < NW / 2;       // (a two-slot ring has no slack for the late group: its refill would land just before the wait for it; loader waves: nothing to wait behind)
    stamp();                                          // 1: x2 rows fetched and split
#pragma unroll
    for (int P = 0; P < KS1; ++P) {
        publish(P);                                   // (P = 0: also publishes the planes)
        stamp();                                      // 2 + P: step P published
        if (early) issue_step(P + S - 1);             // into the slot of step P - 1
        if (!loader) {
            read1(P, F[P & 1]);
            if (P > 0) mma1(F[(P - 1) & 1]);
        }
        if (!early) issue_step(P + S - 1);
    }
    if (!loader) mma1(F[(KS1 - 1) & 1]);

    // ---- hid = gelu(acc + b1), split, over the x2 planes (every wave is past its last read of them behind the next barrier)
    publish(KS1);                                     // first phase-2 step has landed too
    stamp();                                          // 10: phase 1 done
    issue_step(KS1 + S - 1);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        if (!loader && (a == 0 || two1)) {
            const f4s bsel = g == 0 ? bias_s[a][0] : (g == 1 ? bias_s[a][1] : (g == 2 ? bias_s[a][2] : bias_s[a][3]));      // lane group g takes hidden units 4 g .. 4 g + 3 of the tile
            const float4 bv = make_float4(bsel[0], bsel[1], bsel[2], bsel[3]);
            const int chunk = 2 * (a ? tb : ta) + (g >> 1);
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                if (BAL && a == 1 && t > 0 && !full_b) continue;               // tile tb of waves 2..7: the one unit in slot [1][0]
                const int tt = (BAL && a == 1 && t == 0 && sel1) ? TT - 1 : t;  // ... which is token tile sel1
                const f2 m01 = {accM[a][t][0], accM[a][t][1]}, m23 = {accM[a][t][2], accM[a][t][3]};
                const f2 c01 = {accC[a][t][0], accC[a][t][1]}, c23 = {accC[a][t][2], accC[a][t][3]};
                const f2 b01 = {bv.x, bv.y}, b23 = {bv.z, bv.w};
                const f2 g01 = gelu_fast2(m01 + c01 * LO_UNSCALE + b01), g23 = gelu_fast2(m23 + c23 * LO_UNSCALE + b23);
                const float4 v = make_float4(g01.x, g01.y, g23.x, g23.y);
                uint2 hi, lo;
                split4_pk(v, hi, lo);
                float *dst = Xs + (16 * tt + n) * 256 + ((chunk ^ n) << 2) + ((g & 1) << 1);
                *reinterpret_cast<uint2 *>(dst) = hi;
                *reinterpret_cast<uint2 *>(dst + 128) = lo;
            }
        }
    }
    // K padding 208 -> 224: chunks 26, 27 of every row, both planes, are zero (the weights there are zero too, but 0 x stale bits may be NaN)
    if (tid < BM * 4) {
        const int r = tid >> 2, pl = (tid >> 1) & 1, ch = 26 + (tid & 1);
        *reinterpret_cast<uint4 *>(Xs + r * 256 + pl * 128 + ((ch ^ (r & 15)) << 2)) = uint4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int t = 0; t < TT; ++t) accM[a][t] = accC[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();                     // hid planes visible
    stamp();                                          // 11: GELU phase done

    // ---- phase 2: part^T[out][token] = W2[:, slice] . hid^T
    if (!loader) read2(0, F[0]);
    constexpr int NST = BM * (D / 4) / NTH, NWT = NTH / 64;            // float4 stores per thread; waves of the workgroup
    float4 xres[NST], bres = float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 1; q < KS2; ++q) {
        publish(KS1 + q);
        stamp();                                      // 11 + q
        if (early) issue_step(KS1 + q + S - 1);
        if (q == KS2 - 1 && sl == 0) {                // slab 0 carries the residual and the output bias: plain loads, younger than every DMA (the wait above was vmcnt(0))
            bres = *reinterpret_cast<const float4 *>(b2 + ((tid & 63) << 2));
#pragma unroll
            for (int it = 0; it < NST; ++it)
                xres[it] = *reinterpret_cast<const float4 *>(x2 + (size_t)min(m0 + (tid >> 6) + it * NWT, M - 1) * D + ((tid & 63) << 2));
        }
        if (!loader) {
            read2(q, F[q & 1]);
            mma(F[(q - 1) & 1], true);
        }
        if (!early) issue_step(KS1 + q + S - 1);
    }
    if (!loader) mma(F[(KS2 - 1) & 1], true);

    // ---- partial tile leaves through LDS as 16-byte row stores.  TT <= 2: staged in ring slots 0 / 1 (the last step, still being read by
    // slower waves, sits in slot (NPAIR - 1) % 3 = 2); TT = 4: over the planes and slot 0, once every wave is done reading them.
    float *Cs = TT == 4 ? smem : ring;
    if constexpr (TT == 4) {
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
    }
    if (!loader)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            float4 v;
            v.x = accM[a][t][0] + accC[a][t][0] * LO_UNSCALE;
            v.y = accM[a][t][1] + accC[a][t][1] * LO_UNSCALE;
            v.z = accM[a][t][2] + accC[a][t][2] * LO_UNSCALE;
            v.w = accM[a][t][3] + accC[a][t][3] * LO_UNSCALE;
            *reinterpret_cast<float4 *>(Cs + (16 * t + n) * CSS + (o0 + a) * 16 + 4 * g) = v;
        }
    stamp();                                          // 18: last MFMAs issued, tile staged
    __syncthreads();
    float *out = parts + (size_t)sl * M * D;
#pragma unroll
    for (int it = 0; it < NST; ++it) {
        const int row = (tid >> 6) + it * NWT, c4 = (tid & 63) << 2, gr = m0 + row;
        if (gr >= M) continue;
        float4 v = *reinterpret_cast<const float4 *>(Cs + row * CSS + c4);
        if (sl == 0) {
            const float4 x = xres[it];
            v.x += x.x + bres.x; v.y += x.y + bres.y; v.z += x.z + bres.z; v.w += x.w + bres.w;
        }
        if constexpr (MODE == 4) { if (v.x == 12345.678f) idf_store16_wt(out + (size_t)gr * D + c4, v); }      // ablation: no slab stores (never true)
        else idf_store16_wt(out + (size_t)gr * D + c4, v);          // the slabs are read next by other XCDs: write through (common.h)
    }
    stamp();                                          // 19: stores issued
}

// EXCLUSIVE CU.  While this kernel ran beside OTHER kernels' workgroups on the same CU (two streams: the staggered-chains option of
// diffusion.py, or any second queue), the other kernel occasionally computed wrong values -- reproducer tools/hook_stage_probe.py: the
// one-wave SMPL pose kernel (6 KiB of LDS) got joints 50 / 51 wrong in ~40 % of the runs, only with this kernel as the neighbour (never
// with the fp32 kernel of ffn.h, never with this kernel's no-MFMA or no-DMA ablations), while an LDS sentinel beside it (tools/
// lds_sentinel_probe.py) saw no foreign LDS write.  The mechanism is not established.  The kernel therefore takes its CU for itself:
// it asks for the whole 160 KiB of LDS and for 256 VGPRs per wave (2 waves per SIMD x 256 = the register file), so that no other
// workgroup can be resident next to it -- 0 differences in the same probe.  It costs nothing: the grid is one workgroup per CU by design.
// Since round 5 the launchers VERIFY the claim (common.h idf_exclusive_cu: occupancy query == 1, LDS == 160 KiB, >= 256 registers allocated) and hand back
// IDF_NOT_EXCLUSIVE where it does not hold; the caller then runs the fp32 kernel of ffn.h.
constexpr int LDS_REQUEST = 160 * 1024;
#ifndef IDF_FFN_H2_SLOTS
#define IDF_FFN_H2_SLOTS 4
#endif
constexpr int FFN_H2_SLOTS = IDF_FFN_H2_SLOTS;            // ring slots of the 16- and 32-row kernels: 4 = planes + ring fill the CU's LDS exactly (3: rounds 4a; -D for A/B)
template <int TT, int S, int LW = 0>
inline int launch_h2_tt(hipStream_t s, const float *x2, int M, const float *pack, const float *b1p, const float *b2, float *parts, int order) {
    constexpr int BM = 16 * TT;
    static_assert(BM * 1024 + S * SLOT <= LDS_REQUEST, "LDS");
    static idf_excl_cache excl;
    const int dyn = idf_exclusive_cu(reinterpret_cast<const void *>(&ffn_h2_kernel<TT, S, 0, LW>),
                                     LW ? (TT == 1 ? "ffn_h2_kernel<16 rows, loader waves>" : (TT == 2 ? "ffn_h2_kernel<32 rows, loader waves>" : "ffn_h2_kernel<64 rows, loader waves>"))
                                        : (TT == 1 ? "ffn_h2_kernel<16 rows>" : (TT == 2 ? "ffn_h2_kernel<32 rows>" : "ffn_h2_kernel<64 rows>")), NT + 64 * LW, excl);
    if (dyn != LDS_REQUEST) return IDF_NOT_EXCLUSIVE;            // (the kernel has no static LDS: its dynamic request IS the CU's 160 KiB)
    hipLaunchKernelGGL((ffn_h2_kernel<TT, S, 0, LW>), dim3((unsigned)(idf_cdiv(M, BM) * NSL)), dim3(NT + 64 * LW), LDS_REQUEST, s, x2, M, (int)(idf_cdiv(M, BM) * NSL), pack, b1p, b2, parts, order);
    return IDF_OK;
}
// rows: 16 / 32 / 64 = the M tile (csrc/ffn.h ffn_tile_for_rows picks it from the launch's rows when 0); all three produce the same bits
// order 3 (A/B only): XCD-affine M-tile-major ids like order 0, with the THREE-slot ring of the round's first builds
inline int launch_ffn_h2(hipStream_t s, const float *x2, int M, const float *pack, const float *b1p, const float *b2, float *parts, int rows, int order) {
    if (order == 3) {
        if (rows == 16) return launch_h2_tt<1, 3>(s, x2, M, pack, b1p, b2, parts, 0);
        if (rows == 64) return launch_h2_tt<4, 2>(s, x2, M, pack, b1p, b2, parts, 0);
        return launch_h2_tt<2, 3>(s, x2, M, pack, b1p, b2, parts, 0);
    }
    if (rows == 16) return launch_h2_tt<1, FFN_H2_SLOTS>(s, x2, M, pack, b1p, b2, parts, order);
    if (rows == 64) return launch_h2_tt<4, 2>(s, x2, M, pack, b1p, b2, parts, order);
    // 32 rows: sixteen waves, eight of them loaders (round 6: -5.5 % per launch back to back, 10.5 -> 9.9 us: shorter prologue, 16 waves on the slab stores, the GELU phase with the ring kept full;
    // the K steps themselves do not move: they run at the ~45 B/clk a CU gets from its XCD's L2 -- profiles/r06_ffn_loader_waves.txt).  The 16-row tile gains 1 %, the 64-row tile does not fit
    // 128 registers: both keep eight waves.  order 4 (A/B only, tune[IDF_TUNE_MISC] = 10): the eight-wave form of rounds 4-5.
    if (order == 4) return launch_h2_tt<2, FFN_H2_SLOTS>(s, x2, M, pack, b1p, b2, parts, 0);
    return launch_h2_tt<2, FFN_H2_SLOTS, 8>(s, x2, M, pack, b1p, b2, parts, order);
}

// ------------------------------------------------------------------------------------------------------------------------------
// LayerNorm + linear (the QKV projection of the two standard layers) on the same arithmetic: C[M,N] = LN(sum of NP slabs of A) . W^T + bias,
// ffn.h's ln_linear_kernel with the contraction as three f16 MFMAs per product.  Grid, row fetch, LayerNorm, the 160-column slices, the
// 20-KiB ring steps, the DMA shares of the waves (three instructions for waves 0..3, two for 4..7), the residual copy (xn_out) and the
// sampler bookkeeping are ffn.h's.  What differs:
//  * a ring step is one K = 32 step [10 tiles][2 planes][64 lanes][8 halves] (mdm.py pack_linear160_h2), 8 steps per slice -- the same 160 KiB;
//  * the normalised rows are split into f16 planes in the layout of the feed-forward kernel above -- but ROW-SCALED: layer 0's input is the
//    embedding output, not a LayerNorm output, so no bound on it can be proved from the weights; every row is multiplied by the power of two
//    2^-e (e = exponent of the row's largest magnitude: exact) before the split and its outputs by 2^e afterwards (exact again), so the hi
//    plane lives in [0.5, 1) x sign whatever the input's scale and nothing can leave f16's range;
//  * weights are the A operand, tokens the B operand: D has lane = token, registers = 4 consecutive output columns (16-byte staging writes);
//  * tile map: 2 token tiles x 10 column tiles; every wave covers both token tiles for its column tiles: waves 0, 1 own two (2w, 2w+1), waves 2..7 one (w + 2).
constexpr int QCT = 10, QHS = QCT * 16;                  // column tiles / columns per workgroup (ffn.h LCT, LHS)
constexpr int QSTEP = QCT * 2 * 1024;                    // bytes of a K step: 20 KiB
constexpr int QSLICE_FLOATS = 8 * QSTEP / 4;             // 40960 floats per 160-column slice
constexpr int QBM = 32;

// PLANES (the QKV projection in front of the split-f16 self-attention, attn_h2.h): the output leaves as the f16 plane pair the attention contracts -- per token row and 64-column
// group (q / k / v x head) [hi 64 halves | lo' 64 halves] at planes_out + (row * 12 + group) * 128 halves -- instead of fp32 rows, so that the four query tiles of a (clip, head)
// fetch planes instead of each splitting K and V again.  The power of two a row is divided by needs no look at the OUTPUT: the kernel already divides every INPUT row by its own
// 2^e (Sc[row]; |x'| < 1), so |v_c| <= 2^e ||W_c||_1 + |b_c| -- a bound every slice of the launch computes identically from three static L1 norms and three bias maxima per layer
// (the six floats behind the packed stream: mdm.py qkv_bounds).  scales_out[row][4] = the three 2^(E - 15) the attention multiplies back (q, k, v; written by slice 0).
// LW = loader waves (round 6; 0 or 8), as in ffn_h2_kernel: with 8 the workgroup is sixteen waves at 128 registers; every wave takes two of the 32 rows through the slab sum / LayerNorm /
// split prologue (a row is one wave's latency chain: four rows per wave were four chains in a row), waves 8..15 then feed the ring, waves 0..7 multiply, all sixteen store.  Same bits.
template <int NP, bool PLANES = false, int LW = 0>
__global__ __launch_bounds__(NT + 64 * LW) void ln_linear_h2_kernel(const float *__restrict__ A, size_t a_pstride, int M, int nwg, const float *__restrict__ pack,
                                                           const float *__restrict__ lnw, const float *__restrict__ lnb, int nsl_grid, int step_B,
                                                           const float *__restrict__ bias, float *__restrict__ C, int ldc, int N,
                                                           float *__restrict__ xn_out, int64_t *__restrict__ step_state,
                                                           int64_t *__restrict__ step_ts, float *__restrict__ planes_out, float *__restrict__ scales_out) {
    // (argument order: the first 14 dwords -- what the weight stream and the row requests need -- arrive preloaded in SGPRs: build.py)
    extern __shared__ __attribute__((aligned(1024))) float smem[];
    static_assert(LW == 0 || LW == 8, "loader waves");
    if constexpr (LW == 0) asm volatile("" ::: "v255");            // exclusive CU, like the feed-forward kernel (see launch_h2_tt)
    else asm volatile("" ::: "v127");
    constexpr int NTH = NT + 64 * LW, NWT = NTH / 64;
    float *Xs = smem, *ring = smem + QBM * 256, *Sc = ring + 3 * (QSTEP / 4), *Sd = Sc + QBM;      // Sc [32]: 2^e of every row; Sd [32][8] (PLANES): the row's q / k / v dividers and their inverses
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = LW > 0 && wave >= NW, issues = LW == 0 || loader;
    const int iw = LW > 0 ? (wave & 7) : wave;          // index among the waves that feed the ring
    int mt, sl;
    idf_ffn::xcd_affine_tile(nwg, blockIdx.x, nsl_grid, mt, sl);
    const int m0 = mt * QBM, n0 = sl * QHS;
    const float *stream = idf_uniform_ptr(pack + (size_t)sl * QSLICE_FLOATS);
    const uint32_t vsrc = (uint32_t)(iw * 1024) + (uint32_t)(lane << 4);
    const uint32_t sdst = idf_lds_addr(ring) + (uint32_t)(iw * 1024);
    const bool low = iw < NW / 2;                         // issuing waves 0..3: a third DMA instruction per step
    auto issue_step = [&](int P) {
        if (P >= 8 || !issues) return;
        const uint32_t so = (uint32_t)(P * QSTEP), dof = (uint32_t)((P % 3) * QSTEP);
        idf_dma16_s(stream, vsrc + so, sdst + dof);
        idf_dma16_s(stream, vsrc + so + 8192u, sdst + dof + 8192u);
        if (low) idf_dma16_s(stream, vsrc + so + 16384u, sdst + dof + 16384u);
    };
    auto wait_one_step_flying = [&]() {
        if (low) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    };
    issue_step(0);
    issue_step(1);
    float bnd6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};     // PLANES: {||W||_1 max of the q, k, v rows; max |b| of q, k, v}, behind the stream; requested now, used in the epilogue
    if constexpr (PLANES) {
        const float *bnd = pack + (size_t)nsl_grid * QSLICE_FLOATS;
#pragma unroll
        for (int i = 0; i < 6; ++i) bnd6[i] = bnd[i];
    }
    {   // rows: slab sum, LayerNorm (null lnw: layer 0 takes the embedding as it is), residual copy, row scale, split into the plane image
        const float4 gw = lnw ? *reinterpret_cast<const float4 *>(lnw + lane * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 gb = lnw ? *reinterpret_cast<const float4 *>(lnb + lane * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 v[QBM / NWT];
#pragma unroll
        for (int i = 0; i < QBM / NWT; ++i) v[i] = ld4_sum<NP>(A + (size_t)min(m0 + wave + NWT * i, M - 1) * D + lane * 4, a_pstride);
        idf_args_now(bias, C, ldc, N, xn_out, step_state, step_ts);      // the rest of the argument segment, behind the requests
        if (step_state && blockIdx.x == 0 && threadIdx.x == 0) sampler_prepare_step(step_state, step_ts, step_B);
#pragma unroll
        for (int i = 0; i < QBM / NWT; ++i) {
            const int row = wave + NWT * i;
            float4 x = v[i];
            if (lnw) {
                float mean, rstd;
                ln_row_stats(x, mean, rstd);
                x.x = (x.x - mean) * rstd * gw.x + gb.x;
                x.y = (x.y - mean) * rstd * gw.y + gb.y;
                x.z = (x.z - mean) * rstd * gw.z + gb.z;
                x.w = (x.w - mean) * rstd * gw.w + gb.w;
            }
            if (xn_out && sl == 0 && m0 + row < M) idf_store16_wt(xn_out + (size_t)(m0 + row) * D + lane * 4, x);
            // power-of-two row scale: 2^-e with e the exponent of the row's largest magnitude (frexp form: |x| 2^-e in [0.5, 1)); an all-zero row keeps scale 1
            const float amax = wave_max(fmaxf(fmaxf(__builtin_fabsf(x.x), __builtin_fabsf(x.y)), fmaxf(__builtin_fabsf(x.z), __builtin_fabsf(x.w))));
            const int e = amax > 0.f ? (int)((__builtin_bit_cast(uint32_t, amax) >> 23) & 0xff) - 126 : 0;
            const float dn = __builtin_bit_cast(float, (uint32_t)((127 - e) << 23)), up = __builtin_bit_cast(float, (uint32_t)((127 + e) << 23));
            x.x *= dn; x.y *= dn; x.z *= dn; x.w *= dn;
            uint2 hi, lo;
            split4_pk(x, hi, lo);
            float *dst = Xs + row * 256 + ((((lane >> 1) ^ (row & 15)) << 2)) + ((lane & 1) << 1);
            *reinterpret_cast<uint2 *>(dst) = hi;
            *reinterpret_cast<uint2 *>(dst + 128) = lo;
            if (lane == 0) Sc[row] = up;
        }
    }
    wait_one_step_flying();                              // step 0 (and everything older) has landed; step 1 may fly
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();

    const bool two = wave < 2;
    const int c0 = two ? 2 * wave : wave + 2;
    f32x4 accM[2][2], accC[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int t = 0; t < 2; ++t) accM[a][t] = accC[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    struct Frags {
        h8 wh[2], wl[2], xh[2], xl[2];
    };
    Frags F[2];
    const int e_ = g ^ n;
    auto ld8 = [&](const float *p) { return __builtin_bit_cast(h8, *reinterpret_cast<const uint4 *>(p)); };
    auto rd = [&](int s, Frags &f) {
        const float *sb = ring + (s % 3) * (QSTEP / 4) + c0 * 512 + lane * 4;
        f.wh[0] = ld8(sb);
        f.wl[0] = ld8(sb + 256);
        if (two) {
            f.wh[1] = ld8(sb + 512);
            f.wl[1] = ld8(sb + 768);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float *row = Xs + (16 * t + n) * 256 + ((e_ ^ (4 * s)) << 2);
            f.xh[t] = ld8(row);
            f.xl[t] = ld8(row + 128);
        }
    };
    auto mma = [&](const Frags &f) {
#pragma unroll
        for (int t = 0; t < 2; ++t) IDF_H2_MFMA(accM[0][t], f.wh[0], f.xh[t]);
        if (two) {
#pragma unroll
            for (int t = 0; t < 2; ++t) IDF_H2_MFMA(accM[1][t], f.wh[1], f.xh[t]);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) IDF_H2_MFMA(accC[0][t], f.wh[0], f.xl[t]);
        if (two) {
#pragma unroll
            for (int t = 0; t < 2; ++t) IDF_H2_MFMA(accC[1][t], f.wh[1], f.xl[t]);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) IDF_H2_MFMA(accC[0][t], f.wl[0], f.xh[t]);
        if (two) {
#pragma unroll
            for (int t = 0; t < 2; ++t) IDF_H2_MFMA(accC[1][t], f.wl[1], f.xh[t]);
        }
    };
#pragma unroll
    for (int P = 0; P < 8; ++P) {
        if (P > 0) {                                     // step P has landed for every wave, and every wave is done with step P - 1's slot
            if (P + 1 < 8) wait_one_step_flying();
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
        }
        issue_step(P + 2);
        if (!loader) {
            rd(P, F[P & 1]);
            if (P > 0) mma(F[(P - 1) & 1]);
        }
    }
    if (!loader) mma(F[1]);
    // epilogue: x 2^e of the row, + bias, through LDS (over the ring, once every wave is done reading it), 16-byte row stores (write-through)
    constexpr int QCS = QHS + 4;
    float *Cs = ring;
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        if (!loader && (a == 0 || two)) {
            const int col = (c0 + a) * 16 + 4 * g;
            float4 bv;
            bv.x = bias[min(n0 + col, N - 1)]; bv.y = bias[min(n0 + col + 1, N - 1)]; bv.z = bias[min(n0 + col + 2, N - 1)]; bv.w = bias[min(n0 + col + 3, N - 1)];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float up = Sc[16 * t + n];
                float4 v;
                v.x = (accM[a][t][0] + accC[a][t][0] * LO_UNSCALE) * up + bv.x;
                v.y = (accM[a][t][1] + accC[a][t][1] * LO_UNSCALE) * up + bv.y;
                v.z = (accM[a][t][2] + accC[a][t][2] * LO_UNSCALE) * up + bv.z;
                v.w = (accM[a][t][3] + accC[a][t][3] * LO_UNSCALE) * up + bv.w;
                *reinterpret_cast<float4 *>(Cs + (16 * t + n) * QCS + col) = v;
            }
        }
    }
    if constexpr (PLANES) {
        // the three powers of two a row's q / k / v outputs are divided by depend on the ROW alone (its input scale and the layer's static bounds): derived ONCE per row here, by the
        // row's own thread, next to the staging writes (round 5 derived all three in every 8-output item of the pass below: ~20 of its ~50 instructions)
        if (tid < QBM) {
            const float wq = bnd6[0], wk = bnd6[1], wv = bnd6[2], bq = bnd6[3], bk = bnd6[4], bvv = bnd6[5];
            auto down = [](float bound, float &up) {                         // bound < 2^E: multiplier 2^(15 - E) (|v| 2^(15 - E) < 2^15) and, through `up`, 2^(E - 15)
                const int E = (int)((__builtin_bit_cast(uint32_t, fmaxf(bound, 1e-30f)) >> 23) & 0xff) - 126;
                up = __builtin_bit_cast(float, (uint32_t)((127 + E - 15) << 23));
                return __builtin_bit_cast(float, (uint32_t)((127 + 15 - E) << 23));
            };
            const float e2 = Sc[tid];
            float uq, uk, uv;
            const float dq = down(e2 * wq + bq, uq), dk = down(e2 * wk + bk, uk), dv = down(e2 * wv + bvv, uv);
            *reinterpret_cast<float4 *>(Sd + 8 * tid) = make_float4(dq, dk, dv, 0.f);
            *reinterpret_cast<float4 *>(Sd + 8 * tid + 4) = make_float4(uq, uk, uv, 0.f);
        }
    }
    __syncthreads();
    if constexpr (PLANES) {
#pragma unroll
        for (int it = 0; it < (QBM * (QHS / 8) + NTH - 1) / NTH; ++it) {
            const int idx = tid + it * NTH, row = idx / (QHS / 8), c8 = (idx - row * (QHS / 8)) << 3, gr = m0 + row, gc = n0 + c8;
            if (idx < QBM * (QHS / 8) && gr < M && gc < N) {
                const int t = gc >> 8, grp = gc >> 6, o = gc & 63;
                const float dn = Sd[8 * row + t];
                const float4 v0 = *reinterpret_cast<const float4 *>(Cs + row * QCS + c8), v1 = *reinterpret_cast<const float4 *>(Cs + row * QCS + c8 + 4);
                uint2 h0, l0, h1, l1;
                split4_pk(make_float4(v0.x * dn, v0.y * dn, v0.z * dn, v0.w * dn), h0, l0);
                split4_pk(make_float4(v1.x * dn, v1.y * dn, v1.z * dn, v1.w * dn), h1, l1);
                float *dst = planes_out + ((size_t)gr * 12 + grp) * 64 + (o >> 1);      // (float words: 128 halves per (row, group); hi plane first)
                idf_store16_wt(dst, __builtin_bit_cast(float4, make_uint4(h0.x, h0.y, h1.x, h1.y)));
                idf_store16_wt(dst + 32, __builtin_bit_cast(float4, make_uint4(l0.x, l0.y, l1.x, l1.y)));
                if (sl == 0 && c8 == 0) idf_store16_wt(scales_out + (size_t)gr * 4, *reinterpret_cast<const float4 *>(Sd + 8 * row + 4));
            }
        }
    } else {
#pragma unroll
        for (int it = 0; it < (QBM * (QHS / 4) + NTH - 1) / NTH; ++it) {
            const int idx = tid + it * NTH, row = idx / (QHS / 4), c4 = (idx - row * (QHS / 4)) << 2, gr = m0 + row;
            if (idx < QBM * (QHS / 4) && gr < M && n0 + c4 < N) idf_store16_wt(C + (size_t)gr * ldc + n0 + c4, *reinterpret_cast<const float4 *>(Cs + row * QCS + c4));
        }
    }
}

#ifndef IDF_QKV_LOADER_WAVES
#define IDF_QKV_LOADER_WAVES 0            // 0 (shipped): eight waves; 8: sixteen waves, eight of them loaders -- built and measured in round 6 (-DIDF_QKV_LOADER_WAVES=8): the same time in situ
#endif                                    // (8.45 / 7.47 vs 8.49 / 7.63 us and 8.53 / 7.57 vs 8.46 / 7.51 us, whole samples within 0.2 %: profiles/r06_qkv_waves_ab.txt) -- unlike the feed-forward block,
                                          // this kernel has only 160 KiB of stream and a 20-KiB epilogue per workgroup for the extra waves to shorten
template <int NP>
inline int launch_ln_linear_h2(hipStream_t s, const float *A, size_t a_pstride, const float *lnw, const float *lnb, int M, int N,
                               const float *pack, const float *bias, float *C, int ldc, float *xn_out, int64_t *step_state = nullptr,
                               int64_t *step_ts = nullptr, int step_B = 0, float *planes_out = nullptr, float *scales_out = nullptr) {
    constexpr int LW = IDF_QKV_LOADER_WAVES, NTH = NT + 64 * LW;
    static idf_excl_cache excl, excl_p;
    if (planes_out) {                                  // the plane-pair output of the QKV projection (N = 768, the bounds behind the stream): see the kernel
        const int dyn = idf_exclusive_cu(reinterpret_cast<const void *>(&ln_linear_h2_kernel<NP, true, LW>), NP == 1 ? "ln_linear_h2_kernel<1 slab, planes out>" : "ln_linear_h2_kernel<5 slabs, planes out>", NTH, excl_p);
        if (dyn != LDS_REQUEST) return IDF_NOT_EXCLUSIVE;
        if (!A) return IDF_OK;                         // (availability query: nothing to launch)
        const int nsl = (int)idf_cdiv(N, QHS);
        hipLaunchKernelGGL((ln_linear_h2_kernel<NP, true, LW>), dim3((unsigned)(idf_cdiv(M, QBM) * nsl)), dim3(NTH), LDS_REQUEST, s, A, a_pstride, M,
                           (int)(idf_cdiv(M, QBM) * nsl), pack, lnw, lnb, nsl, step_B, bias, C, ldc, N, xn_out, step_state, step_ts, planes_out, scales_out);
        return IDF_OK;
    }
    const int dyn = idf_exclusive_cu(reinterpret_cast<const void *>(&ln_linear_h2_kernel<NP, false, LW>), NP == 1 ? "ln_linear_h2_kernel<1 slab>" : "ln_linear_h2_kernel<5 slabs>", NTH, excl);
    if (dyn != LDS_REQUEST) return IDF_NOT_EXCLUSIVE;
    const int nsl = (int)idf_cdiv(N, QHS);
    hipLaunchKernelGGL((ln_linear_h2_kernel<NP, false, LW>), dim3((unsigned)(idf_cdiv(M, QBM) * nsl)), dim3(NTH), LDS_REQUEST, s, A, a_pstride, M,
                       (int)(idf_cdiv(M, QBM) * nsl), pack, lnw, lnb, nsl, step_B, bias, C, ldc, N, xn_out, step_state, step_ts, (float *)nullptr, (float *)nullptr);
    return IDF_OK;
}
}  // namespace idf_ffn_h2
