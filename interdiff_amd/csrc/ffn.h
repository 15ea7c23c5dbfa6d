// Fused feed-forward block of a decoder / encoder layer for gfx950:
//
//     u3 = x2 + gelu(x2 . W1^T + b1) . W2^T + b2          (model/diffusion_smpl.py:73-120 linear1 / gelu / linear2 + residual;
//                                                           torch.nn.TransformerDecoderLayer._ff_block, sublayers.py:331-341)
//
// ONE launch per layer instead of two GEMMs with a [N,1024] round trip through HBM in between.  The token matrix of the benchmark
// (N = 1600 rows) is far too small to fill 256 CUs with M tiles alone, so the grid is (32-row M tile) x (slice of the 1024 hidden
// units): 50 x 5 = 250 workgroups at N = 1600, one per CU, each
//     phase 1   hid[32, HS]   = gelu(x2[32,256] . W1[slice]^T + b1[slice])      HS = 208 (13 MFMA column tiles; 5 x 208 = 1040: the last
//                                                                                  16 units of the last slice are zero weights)
//     phase 2   part[32, 256] = hid[32, HS] . W2[:, slice]^T                     (K = HS)
// and writes its partial [32,256] tile to slab `slice` of `parts[5][N][256]` (slab 0 also carries x2 + b2).  The five slabs are summed
// by whoever reads the layer output next (LayerNorm-on-load of the next row block / QKV / heads GEMM: common.h ld4_sum), in a fixed
// order, so the result is deterministic -- there is no atomic and no cross-workgroup hand-off (MI355X_MICROARCH.md prices a GEMM->GEMM
// seam inside one launch at 5-13 us; a consumer-side sum costs four extra L2 reads per element).
//
// Work balance: 6400 (16-row, 16-hidden) units of 128 MFMAs each = 25.6 per workgroup; every workgroup does 26 (the padded tile costs
// nothing: the launch lasts as long as its slowest workgroup), 250 of 256 CUs are busy.  Per workgroup 3328 MFMAs = 832 per SIMD =
// 26.6 k cycles of matrix pipe.
//
// Operand movement:
//  * x2 rows (32 x 1 KiB) go global -> LDS once by LDS-DMA; the GELU output overwrites them in place (same XOR-swizzled image), so the
//    hidden activations never leave the CU;
//  * the weights of a slice are PRE-PACKED on the host (interdiff_amd/mdm.py: pack_ffn) in exactly the order and LDS image the kernel
//    consumes: a stream of k-group chunks ([HS rows][16 k] of W1, then [256 rows][16 k] of W2), each already swizzled, so a DMA
//    instruction is a linear 1-KiB copy (perfectly coalesced, no address arithmetic in the loop);
//  * ring of 3 pair slots (32 KiB each); pairs are issued two ahead, the only synchronisation in the loop is ONE counted
//    s_waitcnt vmcnt + ONE s_barrier per PAIR of k-groups, placed in the middle of the pair's MFMA stream: the fragments of the second
//    k-group are already in registers when the barrier is taken and cover the first LDS reads of the next pair.
// Arithmetic: v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulate), gelu = erf form (common.h gelu_fast).
#pragma once
#include "common.h"
#include "philox.h"

namespace idf_ffn {

constexpr int D = IDF_MDM_D, FF = IDF_MDM_FF;
constexpr int BM = 32;                 // token rows per workgroup
constexpr int NSL = IDF_FFN_SLICES;    // hidden slices = partial slabs (5)
constexpr int NW = 8, NT = NW * 64;
constexpr int HS = 208, NTILE = HS / 16;              // hidden units / MFMA column tiles per slice: 5 x 208 = 1040 >= 1024, the tail of
                                                      // the last slice is zero weights (a workgroup's time is the slowest slice's anyway)
constexpr int W1C = HS * 16, W2C = D * 16;            // floats of one k-group chunk: W1 [208 rows][16 k], W2 [256 rows][16 k]
constexpr int NP1 = 8, NP2 = (NTILE + 1) / 2, NPAIR = NP1 + NP2;     // chunk PAIRS: 8 of phase 1 (K = 256), 7 of phase 2 (K = 208; the last is single)
constexpr int SLICE_FLOATS = 16 * W1C + NTILE * W2C;  // 106496 floats = 416 KiB of packed weight stream per slice
constexpr int PSLOT = 2 * W2C;                        // floats per ring slot = one pair (32 KiB); 3 slots
constexpr int XS = BM * D;
constexpr int CSS = D + 4;                            // row stride of the output staging tile
static_assert(BM * CSS <= 3 * PSLOT && NSL * HS >= FF, "geometry");

__host__ __device__ constexpr int pair_off(int P) { return P < NP1 ? P * 2 * W1C : 16 * W1C + (P - NP1) * 2 * W2C; }     // floats into the slice stream
__host__ __device__ constexpr int pair_ins(int P) {   // 1-KiB DMA instructions of pair P
    return P < NP1 ? 2 * W1C / 256 : (P < NPAIR - 1 || NTILE % 2 == 0 ? 2 * W2C / 256 : (P == NPAIR - 1 ? W2C / 256 : 0));
}

__device__ __forceinline__ float4 ldsv4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

// Workgroup -> (M tile, slice): M-tile-major over XCD-AFFINE logical ids (round 4).  Workgroup id runs on XCD id % 8, each with its own L2;
// giving one XCD consecutive logical ids puts ALL slices of an M tile on one XCD, so the tile's input rows (x2: 32 KiB; for the QKV
// kernel 32 rows x up to five slabs = 160 KiB) cross the fabric once instead of once per slice, and the tile's output slabs are written
// from one XCD.  Measured on the split-f16 kernel in one process (tools/ffn_h2_ab.py, profiles/r04_ffn_split_f16_ab.txt): denoiser forward
// 212 vs 226-230 us, whole samples 0.2363 vs 0.2437 ms/step against plain ids (where the five slices of a tile land on five XCDs);
// slice-major affine ids (an XCD streams one or two of the five weight streams instead of all) win a burst of launches and lose in situ.
__device__ __forceinline__ void xcd_affine_tile(int nwg, int id, int nsl, int &mt, int &sl) {
    const int xq = nwg >> 3, xr = nwg & 7, xcd = id & 7;
    const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (id >> 3);
    mt = wg / nsl;
    sl = wg - mt * nsl;
}

// s_waitcnt vmcnt(n): the literal has to be an immediate; n is a constant after unrolling (the switch folds away)
__device__ __forceinline__ void wait_vmcnt_n(int n) {
    switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    }
}

#define IDF_FFN_MFMA(acc, a, b) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0)

// acc[j] += a(16 x 16k) . b[j](16k x 16) for 3 tiles (+ a 4th behind ONE wave-uniform branch): rounds over x,y,z,w so that
// consecutive MFMAs never share an accumulator
template <int MODE>
__device__ __forceinline__ void mma_group(f32x4 (&acc)[4], const float4 &a, const float4 (&b)[4], bool four) {
    if constexpr (MODE == 1) {                        // ablation: keep the operands live, issue no MFMA
        asm volatile("" ::"v"(a.x), "v"(b[0].x), "v"(b[1].y), "v"(b[2].z), "v"(b[3].w));
        return;
    }
    IDF_FFN_MFMA(acc[0], a.x, b[0].x); IDF_FFN_MFMA(acc[1], a.x, b[1].x); IDF_FFN_MFMA(acc[2], a.x, b[2].x);
    IDF_FFN_MFMA(acc[0], a.y, b[0].y); IDF_FFN_MFMA(acc[1], a.y, b[1].y); IDF_FFN_MFMA(acc[2], a.y, b[2].y);
    IDF_FFN_MFMA(acc[0], a.z, b[0].z); IDF_FFN_MFMA(acc[1], a.z, b[1].z); IDF_FFN_MFMA(acc[2], a.z, b[2].z);
    IDF_FFN_MFMA(acc[0], a.w, b[0].w); IDF_FFN_MFMA(acc[1], a.w, b[1].w); IDF_FFN_MFMA(acc[2], a.w, b[2].w);
    if (four) {
        IDF_FFN_MFMA(acc[3], a.x, b[3].x); IDF_FFN_MFMA(acc[3], a.y, b[3].y); IDF_FFN_MFMA(acc[3], a.z, b[3].z); IDF_FFN_MFMA(acc[3], a.w, b[3].w);
    }
}
// the same group in two parts, so that LDS reads can be issued BETWEEN them: right after a barrier all eight waves want to fetch
// their next fragments at once, and a wave whose MFMAs sit behind its own reads in program order leaves the matrix pipe idle until
// the LDS has taken them
template <int MODE>
__device__ __forceinline__ void mma_head(f32x4 (&acc)[4], const float4 &a, const float4 (&b)[4], bool ext, bool all4) {
    if constexpr (MODE == 1) return;
    IDF_FFN_MFMA(acc[0], a.x, b[0].x); IDF_FFN_MFMA(acc[1], a.x, b[1].x); IDF_FFN_MFMA(acc[2], a.x, b[2].x);
    if (all4) IDF_FFN_MFMA(acc[3], a.x, b[3].x);
    (void)ext;
}
template <int MODE>
__device__ __forceinline__ void mma_tail(f32x4 (&acc)[4], const float4 &a, const float4 (&b)[4], bool ext, bool all4) {
    if constexpr (MODE == 1) {
        asm volatile("" ::"v"(a.x), "v"(b[0].x), "v"(b[1].y), "v"(b[2].z), "v"(b[3].w));
        return;
    }
    if (all4) {
        IDF_FFN_MFMA(acc[0], a.y, b[0].y); IDF_FFN_MFMA(acc[1], a.y, b[1].y); IDF_FFN_MFMA(acc[2], a.y, b[2].y); IDF_FFN_MFMA(acc[3], a.y, b[3].y);
        IDF_FFN_MFMA(acc[0], a.z, b[0].z); IDF_FFN_MFMA(acc[1], a.z, b[1].z); IDF_FFN_MFMA(acc[2], a.z, b[2].z); IDF_FFN_MFMA(acc[3], a.z, b[3].z);
        IDF_FFN_MFMA(acc[0], a.w, b[0].w); IDF_FFN_MFMA(acc[1], a.w, b[1].w); IDF_FFN_MFMA(acc[2], a.w, b[2].w); IDF_FFN_MFMA(acc[3], a.w, b[3].w);
    } else {
        IDF_FFN_MFMA(acc[0], a.y, b[0].y); IDF_FFN_MFMA(acc[1], a.y, b[1].y); IDF_FFN_MFMA(acc[2], a.y, b[2].y);
        IDF_FFN_MFMA(acc[0], a.z, b[0].z); IDF_FFN_MFMA(acc[1], a.z, b[1].z); IDF_FFN_MFMA(acc[2], a.z, b[2].z);
        IDF_FFN_MFMA(acc[0], a.w, b[0].w); IDF_FFN_MFMA(acc[1], a.w, b[1].w); IDF_FFN_MFMA(acc[2], a.w, b[2].w);
        if (ext) {
            IDF_FFN_MFMA(acc[3], a.x, b[3].x); IDF_FFN_MFMA(acc[3], a.y, b[3].y); IDF_FFN_MFMA(acc[3], a.z, b[3].z); IDF_FFN_MFMA(acc[3], a.w, b[3].w);
        }
    }
}
template <int MODE>
__device__ __forceinline__ void mma_group4(f32x4 (&acc)[4], const float4 &a, const float4 (&b)[4]) {
    if constexpr (MODE == 1) {
        asm volatile("" ::"v"(a.x), "v"(b[0].x), "v"(b[1].y), "v"(b[2].z), "v"(b[3].w));
        return;
    }
    IDF_FFN_MFMA(acc[0], a.x, b[0].x); IDF_FFN_MFMA(acc[1], a.x, b[1].x); IDF_FFN_MFMA(acc[2], a.x, b[2].x); IDF_FFN_MFMA(acc[3], a.x, b[3].x);
    IDF_FFN_MFMA(acc[0], a.y, b[0].y); IDF_FFN_MFMA(acc[1], a.y, b[1].y); IDF_FFN_MFMA(acc[2], a.y, b[2].y); IDF_FFN_MFMA(acc[3], a.y, b[3].y);
    IDF_FFN_MFMA(acc[0], a.z, b[0].z); IDF_FFN_MFMA(acc[1], a.z, b[1].z); IDF_FFN_MFMA(acc[2], a.z, b[2].z); IDF_FFN_MFMA(acc[3], a.z, b[3].z);
    IDF_FFN_MFMA(acc[0], a.w, b[0].w); IDF_FFN_MFMA(acc[1], a.w, b[1].w); IDF_FFN_MFMA(acc[2], a.w, b[2].w); IDF_FFN_MFMA(acc[3], a.w, b[3].w);
}

// x2 [M][256] token rows (LayerNorm-2 output: FFN input and residual), pack = the layer's packed weight stream (NSL x 416 KiB),
// b1p [NSL*208] (linear1 bias, zero-padded), b2 [256], parts [NSL][M][256].  grid = ceil(M/32) * NSL workgroups of 512 threads.
// MODE 0 is the product kernel.  The other instantiations exist only in tools/ffn_probe.hip (ablations for the time budget: 1 = no
// MFMAs, 2 = no DMA after the prologue, 3 = s_memtime stamps of workgroup phases behind the slabs, 4 = no LDS fragment reads in the
// loops, 5 = no barrier in the loops (wrong results: what the per-pair synchronisation costs), 6 = no slab stores, 7 = plain instead of
// write-through slab stores); `if constexpr` keeps every trace of them out of MODE 0.
//
// Every loop over chunk pairs is FULLY UNROLLED and the geometry is constant, so stream offsets, ring slots, DMA counts and LDS
// fragment addresses are immediates: the scalar unit (one per CU, shared by the eight waves) has almost nothing to do.  A first
// version kept these as run-time arithmetic -- ~250 scalar instructions per wave per pair, i.e. ~2000 cycles of the CU's scalar
// issue per pair: as long as the 1800-2000 cycles of MFMAs the pair is supposed to hide it behind (tools/ffn_probe.hip).
template <int MODE = 0>
__global__ __launch_bounds__(NT) void ffn_fused_kernel(const float *__restrict__ x2, int M, const float *__restrict__ pack,
                                                        const float *__restrict__ b1p, const float *__restrict__ b2,
                                                        float *__restrict__ parts) {
    __shared__ __attribute__((aligned(1024))) float smem[XS + 3 * PSLOT + 256 + 512];
    float *Xs = smem, *ring = smem + XS, *Bs = ring + 3 * PSLOT;             // Bs: the slice's linear1 bias (208 floats)
    float *scr = Bs + 256;                                                  // 2 x 64 x 4 floats: the helpers' half of the shared tile (below)
    idf_args_now(x2, M, pack, b1p, b2, parts);                              // every kernel argument into SGPRs now (common.h)

    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int mt, sl;
    xcd_affine_tile(gridDim.x, blockIdx.x, NSL, mt, sl);
    const int m0 = mt * BM;
    const float *stream = idf_uniform_ptr(pack + (size_t)sl * SLICE_FLOATS);
    const uint32_t lane16 = lane << 4;
    const uint32_t vsrc = (uint32_t)(wave * 1024) + lane16;                 // byte offset of this lane's 16 B inside a pair: instruction wave + 8 j adds 8192 j
    const uint32_t sdst = idf_lds_addr(ring) + (uint32_t)(wave * 1024);     // LDS byte address of this wave's first instruction in slot 0
    const bool lt2 = wave < 2;                                              // 26 = 3 * 8 + 2: waves 0 and 1 issue a 4th instruction for a W1 pair
    const int key = (4 - (li >> 2)) & 3;                                    // g4 = {0,3,2,1}[(row >> 2) & 3]: 16-B position swizzle of a [rows][16] chunk (mdm.py pack_ffn)

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
    // DMA instructions of pair P (P is a constant after unrolling): instruction i = wave + 8 j copies stream bytes
    // [pair_off + 1024 i, +1024) to the same offset of ring slot P % 3 -- a pair is contiguous in the stream AND in LDS
    auto issue_pair = [&](int P) {
        if (P >= NPAIR) return;
        if constexpr (MODE == 2) { if (P >= 2) return; }
        const int nins = pair_ins(P);
        const uint32_t so = (uint32_t)(pair_off(P) * 4), dof = (uint32_t)((P % 3) * PSLOT * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (8 * j + 7 < nins) idf_dma16_s(stream, vsrc + so + 8192u * j, sdst + dof + 8192u * j);
            else if (8 * j < nins && lt2) idf_dma16_s(stream, vsrc + so + 8192u * j, sdst + dof + 8192u * j);     // nins = 26, j = 3: waves 0, 1
        }
    };
    auto wait_pair_before = [&](int P) {              // wait until everything older than this wave's DMAs of pair P has landed
        if (P >= NPAIR) { wait_vmcnt_n(0); return; }
        if constexpr (MODE == 2) { if (P >= 2) { wait_vmcnt_n(0); return; } }
        const int nins = pair_ins(P);
        if (nins % 8 == 0) wait_vmcnt_n(nins / 8);
        else if (lt2) wait_vmcnt_n(nins / 8 + 1);
        else wait_vmcnt_n(nins / 8);
    };

    // ---- phase-1 tile map: SIMD (w & 3) holds waves w and w+4; row tile (w>>1)&1; the 13 column tiles go 4|3 (waves w, w+4 of the
    // SIMDs with w even) and 3|3 (w odd): 7,6,7,6 tiles per SIMD -- so the fourth tile of waves 0 / 2 (column tile 3) is SHARED in K
    // with waves 1 / 3 (the same row tile on the neighbouring SIMD): the owner accumulates it over the even chunks and the last one,
    // the helper over the other odd ones; the helper's partial tile crosses through LDS around the last barrier of the loop.  Between
    // two barriers (one per pair, in its middle) every SIMD then issues 13 tile groups instead of 14 | 12.
    const int r1 = (wave >> 1) & 1;
    const int c0 = (wave & 1) ? (wave < 4 ? 7 : 10) : (wave < 4 ? 0 : 4);
    const bool four = wave == 0 || wave == 2;        // owner of (row tile r1, column tile 3): even chunks
    const bool help = wave == 1 || wave == 3;        // helper for that tile: odd chunks

    // ---- prologue: the bias slice (one DMA, the oldest in the queue), the x2 rows (row i = DMA instruction i, source chunk =
    // position ^ (i & 15)), the first two chunk pairs.  Every global->LDS move of this kernel is an asm DMA: no compiler-visible
    // vector load is in flight in the loops.
    if (wave == 0) idf_dma16_s(idf_uniform_ptr(b1p + sl * HS), lane16, idf_lds_addr(Bs));
    const uint32_t xs_lds = idf_lds_addr(Xs);
#pragma unroll
    for (int j = 0; j < BM / NW; ++j) {
        const int i = wave + NW * j;
        idf_dma16_s(idf_uniform_ptr(x2 + (size_t)min(m0 + i, M - 1) * D), (uint32_t)((lane ^ (i & 15)) << 4), xs_lds + (uint32_t)(i * D * 4));
    }
    issue_pair(0);
    issue_pair(1);
    wait_pair_before(1);                               // bias, x2 rows and pair 0 have landed (they are older); pair 1 may still fly
    __builtin_amdgcn_s_barrier();
    stamp();

    // ---- fragment address bases.  x2 / hid row r, logical 16-B chunk t lives at position t ^ (r & 15); with t = 4 c + kq:
    // 16 (t ^ li) = 16 ((kq ^ li) ^ 4 (c & 3)) + 256 (c >> 2): four per-lane bases (c & 3), the rest is an immediate offset.
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 a0, a1, b0[4], b1f[4];
    const float *xb1[4], *wb1[3], *wbx[3];
#pragma unroll
    for (int m = 0; m < 4; ++m) xb1[m] = Xs + (r1 * 16 + li) * D + (((kq ^ li) ^ (4 * m)) << 2);
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) {
        wb1[s3] = ring + s3 * PSLOT + ((kq ^ key) << 2) + li * 16 + c0 * 256;
        wbx[s3] = ring + s3 * PSLOT + ((kq ^ key) << 2) + li * 16 + 3 * 256;          // the shared column tile 3
    }
    auto read1 = [&](int c, float4 &a, float4 (&b)[4]) {        // fragments of W1 k-group c (chunk c & 1 of pair c >> 1)
        if constexpr (MODE == 4) { if (c > 1) return; }
        a = ldsv4(xb1[c & 3] + 64 * (c >> 2));
        const float *sb = wb1[(c >> 1) % 3] + (c & 1) * W1C;
        b[0] = ldsv4(sb);
        b[1] = ldsv4(sb + 256);
        b[2] = ldsv4(sb + 512);
        if ((c & 1) && c != 2 * NP1 - 1 ? help : four) b[3] = ldsv4(wbx[(c >> 1) % 3] + (c & 1) * W1C);
    };
    read1(0, a0, b0);
#pragma unroll
    for (int P = 0; P < NP1; ++P) {
        issue_pair(P + 2);                               // into the slot of pair P-1 (every wave is past the barrier that followed its reads)
        read1(2 * P + 1, a1, b1f);
        // the helpers' half of the shared tile is complete after chunk 13: it crosses to the owners around the last barrier
        if (P == NP1 - 1 && help) *reinterpret_cast<f32x4 *>(scr + ((wave >> 1) * 64 + lane) * 4) = acc[3];
        mma_group<MODE>(acc, a0, b0, four);              // even chunk: the owners take the shared tile
        wait_pair_before(P + 2);                         // pair P+1 has landed; pair P+2 may keep flying
        __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0), as a builtin: the compiler then KNOWS the LDS queue is empty (an asm wait is invisible to its own counting)
        if constexpr (MODE != 5) __builtin_amdgcn_s_barrier();
        const bool ext_odd = P == NP1 - 1 ? four : help;                    // odd chunk: the helpers take the shared tile, except the last one
        mma_head<MODE>(acc, a1, b1f, ext_odd, false);                       // matrix pipe first, then the LDS requests of the next chunk
        __builtin_amdgcn_sched_barrier(0);
        if (P + 1 < NP1) read1(2 * P + 2, a0, b0);
        f32x4 other = f32x4{0.f, 0.f, 0.f, 0.f};
        if (P == NP1 - 1 && four) other = *reinterpret_cast<const f32x4 *>(scr + ((wave >> 1) * 64 + lane) * 4);
        __builtin_amdgcn_sched_barrier(0);
        mma_tail<MODE>(acc, a1, b1f, ext_odd, false);
        if (P == NP1 - 1 && four) acc[3] += other;
        stamp();
    }

    // ---- hid = gelu(acc + b1) overwrites the x2 rows (every read of them is behind the last barrier), same swizzled image
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (j < 3 || four) {
            const int col = (c0 + j) * 16 + li;
            const float bv = Bs[col];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int row = r1 * 16 + kq * 4 + rr;
                Xs[row * D + ((((col >> 2) ^ (row & 15))) << 2) + (col & 3)] = gelu_fast(acc[j][rr] + bv);
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0), as a builtin: the compiler then KNOWS the LDS queue is empty (an asm wait is invisible to its own counting)
    __builtin_amdgcn_s_barrier();
    stamp();

    // ---- phase 2: part[32,256] = hid[32,HS] . W2[:, slice]^T ; wave w: row tile w & 1, output column tiles 4 (w>>1) .. +3
    const int r2 = wave & 1, nb = (wave >> 1) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float *xb2[4], *wb2[3];
#pragma unroll
    for (int m = 0; m < 4; ++m) xb2[m] = Xs + (r2 * 16 + li) * D + (((kq ^ li) ^ (4 * m)) << 2);
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) wb2[s3] = ring + s3 * PSLOT + ((kq ^ key) << 2) + (nb * 16 + li) * 16;
    auto read2 = [&](int q, float4 &a, float4 (&b)[4]) {        // fragments of W2 k-group q (hidden units 16q .. 16q+15 of the slice)
        if constexpr (MODE == 4) { if (q > 1) return; }
        a = ldsv4(xb2[q & 3] + 64 * (q >> 2));
        const float *sb = wb2[(NP1 + (q >> 1)) % 3] + (q & 1) * W2C;
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = ldsv4(sb + j * 256);
    };
    read2(0, a0, b0);
    // slab 0 carries the residual and the output bias: its workgroups fetch them two pairs before the end (plain loads, younger than
    // every DMA, consumed after the loop) instead of serialising them behind the last barrier -- they are the launch's slowest workgroups
    float4 xres[BM * (D / 4) / NT], bres[BM * (D / 4) / NT];
#pragma unroll
    for (int P = NP1; P < NPAIR; ++P) {
        const int q = 2 * (P - NP1);
        const bool two = q + 1 < NTILE;
        issue_pair(P + 2);
        if (P == NPAIR - 2 && sl == 0) {
#pragma unroll
            for (int it = 0; it < BM * (D / 4) / NT; ++it) {
                const int idx = tid + it * NT, row = idx >> 6, c4 = (idx & 63) << 2;
                xres[it] = *reinterpret_cast<const float4 *>(x2 + (size_t)min(m0 + row, M - 1) * D + c4);
                bres[it] = *reinterpret_cast<const float4 *>(b2 + c4);
            }
        }
        if (two) read2(q + 1, a1, b1f);
        mma_group4<MODE>(acc, a0, b0);
        wait_pair_before(P + 2);
        __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0), as a builtin: the compiler then KNOWS the LDS queue is empty (an asm wait is invisible to its own counting)
        if constexpr (MODE != 5) __builtin_amdgcn_s_barrier();
        if (q + 2 < NTILE) read2(q + 2, a0, b0);
        if (two) mma_group4<MODE>(acc, a1, b1f);
        stamp();
    }

    // ---- partial tile leaves through LDS as 16-byte row stores (the ring is idle: every chunk has landed and been consumed)
    float *Cs = ring;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) Cs[(r2 * 16 + kq * 4 + rr) * CSS + (nb + j) * 16 + li] = acc[j][rr];
    __syncthreads();
    float *out = parts + (size_t)sl * M * D;
#pragma unroll
    for (int it = 0; it < BM * (D / 4) / NT; ++it) {
        const int idx = tid + it * NT, row = idx >> 6, c4 = (idx & 63) << 2, gr = m0 + row;
        if (gr >= M) continue;
        float4 v = ldsv4(Cs + row * CSS + c4);
        if (sl == 0) {
            const float4 x = xres[it], bb = bres[it];
            v.x += x.x + bb.x; v.y += x.y + bb.y; v.z += x.z + bb.z; v.w += x.w + bb.w;
        }
        if constexpr (MODE == 6) { if (v.x == 12345.678f) idf_store16_wt(out + (size_t)gr * D + c4, v); }      // ablation: no slab stores (never true)
        else if constexpr (MODE == 7) *reinterpret_cast<float4 *>(out + (size_t)gr * D + c4) = v;            // ablation: plain (write-back) stores
        else idf_store16_wt(out + (size_t)gr * D + c4, v);          // the slabs are read next by other XCDs: write through (common.h)
    }
    stamp();
}

// ------------------------------------------------------------------------------------------------------------------------------
// The same block for SMALL token counts: 16-row M tiles.  With M <= 800 rows (8 clips of 100 frames: BASELINE config #4's share of a
// GPU, or any batch of that size) the 32-row grid is ceil(M/32) x 5 <= 125 workgroups -- half the chip idle while every workgroup still
// takes the full 40 k cycles.  Halving the tile doubles the workgroups (<= 250, one per CU) and halves each one's matrix work; the
// weight stream per workgroup is the same 416 KiB (twice the L2 -> LDS traffic in total, still under the shared-stream ceiling).
// Same stream, same ring, same pair schedule and barrier placement as above; only the tile maps differ:
//   phase 1   one row tile x 13 column tiles: waves 0..4 own two (2w, 2w+1), waves 5..7 one (10, 11, 12)
//   phase 2   one row tile x 16 column tiles: wave w owns 2w, 2w+1
// The sum of a tile-3 column is taken in one accumulator here and in two (owner / helper) above: the two kernels agree to rounding,
// not bit for bit, so a caller picks ONE of them for a given batch (launch_ffn: by the rows of the WHOLE batch, see there).
constexpr int BMH = 16;
template <int MODE = 0>
__global__ __launch_bounds__(NT) void ffn_fused16_kernel(const float *__restrict__ x2, int M, const float *__restrict__ pack,
                                                          const float *__restrict__ b1p, const float *__restrict__ b2,
                                                          float *__restrict__ parts) {
    __shared__ __attribute__((aligned(1024))) float smem[BMH * D + 3 * PSLOT + 256];
    float *Xs = smem, *ring = smem + BMH * D, *Bs = ring + 3 * PSLOT;
    idf_args_now(x2, M, pack, b1p, b2, parts);

    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int mt, sl;
    xcd_affine_tile(gridDim.x, blockIdx.x, NSL, mt, sl);
    const int m0 = mt * BMH;
    const float *stream = idf_uniform_ptr(pack + (size_t)sl * SLICE_FLOATS);
    const uint32_t lane16 = lane << 4;
    const uint32_t vsrc = (uint32_t)(wave * 1024) + lane16;
    const uint32_t sdst = idf_lds_addr(ring) + (uint32_t)(wave * 1024);
    const bool lt2 = wave < 2;
    const int key = (4 - (li >> 2)) & 3;
    auto issue_pair = [&](int P) {
        if (P >= NPAIR) return;
        const int nins = pair_ins(P);
        const uint32_t so = (uint32_t)(pair_off(P) * 4), dof = (uint32_t)((P % 3) * PSLOT * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (8 * j + 7 < nins) idf_dma16_s(stream, vsrc + so + 8192u * j, sdst + dof + 8192u * j);
            else if (8 * j < nins && lt2) idf_dma16_s(stream, vsrc + so + 8192u * j, sdst + dof + 8192u * j);
        }
    };
    auto wait_pair_before = [&](int P) {
        if (P >= NPAIR) { wait_vmcnt_n(0); return; }
        const int nins = pair_ins(P);
        if (nins % 8 == 0) wait_vmcnt_n(nins / 8);
        else if (lt2) wait_vmcnt_n(nins / 8 + 1);
        else wait_vmcnt_n(nins / 8);
    };
    const bool two1 = wave < 5;                            // phase 1: two column tiles (waves 0..4) or one (waves 5..7)
    const int c0 = two1 ? 2 * wave : 5 + wave;

    if (wave == 0) idf_dma16_s(idf_uniform_ptr(b1p + sl * HS), lane16, idf_lds_addr(Bs));
    const uint32_t xs_lds = idf_lds_addr(Xs);
#pragma unroll
    for (int j = 0; j < BMH / NW; ++j) {
        const int i = wave + NW * j;
        idf_dma16_s(idf_uniform_ptr(x2 + (size_t)min(m0 + i, M - 1) * D), (uint32_t)((lane ^ (i & 15)) << 4), xs_lds + (uint32_t)(i * D * 4));
    }
    issue_pair(0);
    issue_pair(1);
    wait_pair_before(1);
    __builtin_amdgcn_s_barrier();

    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    float4 a0, a1, b0[2], b1f[2];
    const float *xb[4], *wb1[3];
#pragma unroll
    for (int m = 0; m < 4; ++m) xb[m] = Xs + li * D + (((kq ^ li) ^ (4 * m)) << 2);
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) wb1[s3] = ring + s3 * PSLOT + ((kq ^ key) << 2) + li * 16 + c0 * 256;
    auto read1 = [&](int c, float4 &a, float4 (&b)[2]) {
        a = ldsv4(xb[c & 3] + 64 * (c >> 2));
        const float *sb = wb1[(c >> 1) % 3] + (c & 1) * W1C;
        b[0] = ldsv4(sb);
        if (two1) b[1] = ldsv4(sb + 256);
    };
    auto mma1 = [&](const float4 &a, const float4 (&b)[2]) {
        if (two1) {
            IDF_FFN_MFMA(acc[0], a.x, b[0].x); IDF_FFN_MFMA(acc[1], a.x, b[1].x);
            IDF_FFN_MFMA(acc[0], a.y, b[0].y); IDF_FFN_MFMA(acc[1], a.y, b[1].y);
            IDF_FFN_MFMA(acc[0], a.z, b[0].z); IDF_FFN_MFMA(acc[1], a.z, b[1].z);
            IDF_FFN_MFMA(acc[0], a.w, b[0].w); IDF_FFN_MFMA(acc[1], a.w, b[1].w);
        } else {
            IDF_FFN_MFMA(acc[0], a.x, b[0].x); IDF_FFN_MFMA(acc[0], a.y, b[0].y); IDF_FFN_MFMA(acc[0], a.z, b[0].z); IDF_FFN_MFMA(acc[0], a.w, b[0].w);
        }
    };
    read1(0, a0, b0);
#pragma unroll
    for (int P = 0; P < NP1; ++P) {
        issue_pair(P + 2);
        read1(2 * P + 1, a1, b1f);
        mma1(a0, b0);
        wait_pair_before(P + 2);
        __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0), as a builtin (see above)
        __builtin_amdgcn_s_barrier();
        if (P + 1 < NP1) read1(2 * P + 2, a0, b0);
        mma1(a1, b1f);
    }
    // hid = gelu(acc + b1) over the x2 rows, same swizzled image
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (j == 0 || two1) {
            const int col = (c0 + j) * 16 + li;
            const float bv = Bs[col];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int row = kq * 4 + rr;
                Xs[row * D + ((((col >> 2) ^ (row & 15))) << 2) + (col & 3)] = gelu_fast(acc[j][rr] + bv);
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();

    // phase 2: wave w owns output column tiles 2w, 2w+1
    const int nb = wave * 2;
    acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float *wb2[3];
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) wb2[s3] = ring + s3 * PSLOT + ((kq ^ key) << 2) + (nb * 16 + li) * 16;
    auto read2 = [&](int q, float4 &a, float4 (&b)[2]) {
        a = ldsv4(xb[q & 3] + 64 * (q >> 2));
        const float *sb = wb2[(NP1 + (q >> 1)) % 3] + (q & 1) * W2C;
        b[0] = ldsv4(sb);
        b[1] = ldsv4(sb + 256);
    };
    auto mma2 = [&](const float4 &a, const float4 (&b)[2]) {
        IDF_FFN_MFMA(acc[0], a.x, b[0].x); IDF_FFN_MFMA(acc[1], a.x, b[1].x);
        IDF_FFN_MFMA(acc[0], a.y, b[0].y); IDF_FFN_MFMA(acc[1], a.y, b[1].y);
        IDF_FFN_MFMA(acc[0], a.z, b[0].z); IDF_FFN_MFMA(acc[1], a.z, b[1].z);
        IDF_FFN_MFMA(acc[0], a.w, b[0].w); IDF_FFN_MFMA(acc[1], a.w, b[1].w);
    };
    read2(0, a0, b0);
    constexpr int NST = BMH * (D / 4) / NT;                 // float4 stores per thread (2)
    float4 xres[NST], bres[NST];
#pragma unroll
    for (int P = NP1; P < NPAIR; ++P) {
        const int q = 2 * (P - NP1);
        const bool two = q + 1 < NTILE;
        issue_pair(P + 2);
        if (P == NPAIR - 2 && sl == 0) {
#pragma unroll
            for (int it = 0; it < NST; ++it) {
                const int idx = tid + it * NT, row = idx >> 6, c4 = (idx & 63) << 2;
                xres[it] = *reinterpret_cast<const float4 *>(x2 + (size_t)min(m0 + row, M - 1) * D + c4);
                bres[it] = *reinterpret_cast<const float4 *>(b2 + c4);
            }
        }
        if (two) read2(q + 1, a1, b1f);
        mma2(a0, b0);
        wait_pair_before(P + 2);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        if (q + 2 < NTILE) read2(q + 2, a0, b0);
        if (two) mma2(a1, b1f);
    }
    float *Cs = ring;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) Cs[(kq * 4 + rr) * CSS + (nb + j) * 16 + li] = acc[j][rr];
    __syncthreads();
    float *out = parts + (size_t)sl * M * D;
#pragma unroll
    for (int it = 0; it < NST; ++it) {
        const int idx = tid + it * NT, row = idx >> 6, c4 = (idx & 63) << 2, gr = m0 + row;
        if (gr >= M) continue;
        float4 v = ldsv4(Cs + row * CSS + c4);
        if (sl == 0) {
            const float4 x = xres[it], bb = bres[it];
            v.x += x.x + bb.x; v.y += x.y + bb.y; v.z += x.z + bb.z; v.w += x.w + bb.w;
        }
        idf_store16_wt(out + (size_t)gr * D + c4, v);
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// The same block for LARGE token counts: 64-row M tiles.  Above 1632 rows (51 x 32) the 32-row grid no longer fits the chip in one round
// of workgroups, and every further round pays the kernel's fixed part again (prologue, GELU phase, store tail: ~ 40 % of a 32-row
// workgroup's time; launch_ffn below says from where on that pays).  A 64-row workgroup does two tiles' matrix work behind ONE fixed part and ONE pass over the 416-KiB weight stream:
// 3200 rows = 250 workgroups in one round instead of 500 in two; as two half-batch chains, 125 + 125 side by side.
// LDS: 64 KiB of x2 / hid rows + a TWO-slot ring (64 KiB) + the bias slice = 129 KiB.  A slot is refilled right after the barrier that
// certifies every wave has read it -- one pair-time ahead instead of two, and a pair-time is twice as long here.
//   phase 1   4 row tiles x 13 column tiles: wave w owns row tile w & 3 and column tiles 0..6 (w < 4) or 7..12 (w >= 4): the two waves
//             of a SIMD (w, w + 4) hold one row tile's 13 column tiles -- balanced without the shared tile of the 32-row kernel
//   phase 2   4 row tiles x 16 column tiles: wave w owns row tile w & 3 and output column tiles 8 (w >> 2) .. +7
// Every tile is summed in one accumulator over the chunks in order: bit-identical to the 16-row kernel, to rounding with the 32-row one.
constexpr int BMX = 64;
static_assert(BMX * CSS <= BMX * D + 2 * PSLOT, "output staging fits over the dead x2 / hid rows and the ring");
template <int MODE = 0>
__global__ __launch_bounds__(NT) void ffn_fused64_kernel(const float *__restrict__ x2, int M, const float *__restrict__ pack,
                                                          const float *__restrict__ b1p, const float *__restrict__ b2,
                                                          float *__restrict__ parts) {
    __shared__ __attribute__((aligned(1024))) float smem[BMX * D + 2 * PSLOT + 256];
    float *Xs = smem, *ring = smem + BMX * D, *Bs = ring + 2 * PSLOT;
    idf_args_now(x2, M, pack, b1p, b2, parts);

    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int mt, sl;
    xcd_affine_tile(gridDim.x, blockIdx.x, NSL, mt, sl);
    const int m0 = mt * BMX;
    const float *stream = idf_uniform_ptr(pack + (size_t)sl * SLICE_FLOATS);
    const uint32_t lane16 = lane << 4;
    const uint32_t vsrc = (uint32_t)(wave * 1024) + lane16;
    const uint32_t sdst = idf_lds_addr(ring) + (uint32_t)(wave * 1024);
    const bool lt2 = wave < 2;
    const int key = (4 - (li >> 2)) & 3;
    auto issue_pair = [&](int P) {                    // pair P -> slot P & 1
        if (P >= NPAIR) return;
        const int nins = pair_ins(P);
        const uint32_t so = (uint32_t)(pair_off(P) * 4), dof = (uint32_t)((P & 1) * PSLOT * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (8 * j + 7 < nins) idf_dma16_s(stream, vsrc + so + 8192u * j, sdst + dof + 8192u * j);
            else if (8 * j < nins && lt2) idf_dma16_s(stream, vsrc + so + 8192u * j, sdst + dof + 8192u * j);
        }
    };
    const int rt = wave & 3;                          // this wave's row tile, both phases
    const bool hi = wave >= 4;                        // phase 1: column tiles 7..12 (six) instead of 0..6 (seven)
    const int c0 = hi ? 7 : 0;

    if (wave == 0) idf_dma16_s(idf_uniform_ptr(b1p + sl * HS), lane16, idf_lds_addr(Bs));
    const uint32_t xs_lds = idf_lds_addr(Xs);
#pragma unroll
    for (int j = 0; j < BMX / NW; ++j) {
        const int i = wave + NW * j;
        idf_dma16_s(idf_uniform_ptr(x2 + (size_t)min(m0 + i, M - 1) * D), (uint32_t)((lane ^ (i & 15)) << 4), xs_lds + (uint32_t)(i * D * 4));
    }
    issue_pair(0);
    issue_pair(1);
    {   // bias, x2 rows and pair 0 have landed (they are older than this wave's share of pair 1)
        const int nins = pair_ins(1);
        if (nins % 8 == 0) wait_vmcnt_n(nins / 8);
        else if (lt2) wait_vmcnt_n(nins / 8 + 1);
        else wait_vmcnt_n(nins / 8);
    }
    __builtin_amdgcn_s_barrier();

    f32x4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 a0, a1, b0[8], b1f[8];
    const float *xb[4], *wb1[2], *wb2[2];
#pragma unroll
    for (int m = 0; m < 4; ++m) xb[m] = Xs + (rt * 16 + li) * D + (((kq ^ li) ^ (4 * m)) << 2);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        wb1[s2] = ring + s2 * PSLOT + ((kq ^ key) << 2) + li * 16 + c0 * 256;
        wb2[s2] = ring + s2 * PSLOT + ((kq ^ key) << 2) + ((wave >> 2) * 8 * 16 + li) * 16;
    }
    auto read1 = [&](int c, float4 &a, float4 (&b)[8]) {
        a = ldsv4(xb[c & 3] + 64 * (c >> 2));
        const float *sb = wb1[(c >> 1) & 1] + (c & 1) * W1C;
#pragma unroll
        for (int j = 0; j < 6; ++j) b[j] = ldsv4(sb + j * 256);
        if (!hi) b[6] = ldsv4(sb + 6 * 256);
    };
    auto mma1 = [&](const float4 &a, const float4 (&b)[8]) {
#pragma unroll
        for (int j = 0; j < 6; ++j) IDF_FFN_MFMA(acc[j], a.x, b[j].x);
        if (!hi) IDF_FFN_MFMA(acc[6], a.x, b[6].x);
#pragma unroll
        for (int j = 0; j < 6; ++j) IDF_FFN_MFMA(acc[j], a.y, b[j].y);
        if (!hi) IDF_FFN_MFMA(acc[6], a.y, b[6].y);
#pragma unroll
        for (int j = 0; j < 6; ++j) IDF_FFN_MFMA(acc[j], a.z, b[j].z);
        if (!hi) IDF_FFN_MFMA(acc[6], a.z, b[6].z);
#pragma unroll
        for (int j = 0; j < 6; ++j) IDF_FFN_MFMA(acc[j], a.w, b[j].w);
        if (!hi) IDF_FFN_MFMA(acc[6], a.w, b[6].w);
    };
    read1(0, a0, b0);
#pragma unroll
    for (int P = 0; P < NP1; ++P) {
        read1(2 * P + 1, a1, b1f);
        mma1(a0, b0);
        wait_vmcnt_n(0);                                     // pair P + 1 has landed (nothing younger is in flight)
        __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0), as a builtin (see the 32-row kernel)
        __builtin_amdgcn_s_barrier();
        issue_pair(P + 2);                                   // every wave has read pair P: its slot is free
        if (P + 1 < NP1) read1(2 * P + 2, a0, b0);
        mma1(a1, b1f);
    }
    // hid = gelu(acc + b1) over the x2 rows, same swizzled image
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        if (j < 6 || !hi) {
            const int col = (c0 + j) * 16 + li;
            const float bv = Bs[col];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int row = rt * 16 + kq * 4 + rr;
                Xs[row * D + ((((col >> 2) ^ (row & 15))) << 2) + (col & 3)] = gelu_fast(acc[j][rr] + bv);
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();

    // phase 2
    const int nb = (wave >> 2) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto read2 = [&](int q, float4 &a, float4 (&b)[8]) {
        a = ldsv4(xb[q & 3] + 64 * (q >> 2));
        const float *sb = wb2[(NP1 + (q >> 1)) & 1] + (q & 1) * W2C;
#pragma unroll
        for (int j = 0; j < 8; ++j) b[j] = ldsv4(sb + j * 256);
    };
    auto mma2 = [&](const float4 &a, const float4 (&b)[8]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) IDF_FFN_MFMA(acc[j], a.x, b[j].x);
#pragma unroll
        for (int j = 0; j < 8; ++j) IDF_FFN_MFMA(acc[j], a.y, b[j].y);
#pragma unroll
        for (int j = 0; j < 8; ++j) IDF_FFN_MFMA(acc[j], a.z, b[j].z);
#pragma unroll
        for (int j = 0; j < 8; ++j) IDF_FFN_MFMA(acc[j], a.w, b[j].w);
    };
    read2(0, a0, b0);
    constexpr int NST = BMX * (D / 4) / NT;                 // float4 stores per thread (8)
    float4 xres[NST], bres = float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int P = NP1; P < NPAIR; ++P) {
        const int q = 2 * (P - NP1);
        const bool two = q + 1 < NTILE;
        if (two) read2(q + 1, a1, b1f);
        mma2(a0, b0);
        wait_vmcnt_n(0);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        issue_pair(P + 2);
        if (P == NPAIR - 3 && sl == 0) {                     // slab 0: the residual rows and the output bias land with the last pair
            bres = *reinterpret_cast<const float4 *>(b2 + ((tid & 63) << 2));
#pragma unroll
            for (int it = 0; it < NST; ++it)
                xres[it] = *reinterpret_cast<const float4 *>(x2 + (size_t)min(m0 + (tid >> 6) + it * NW, M - 1) * D + ((tid & 63) << 2));
        }
        if (q + 2 < NTILE) read2(q + 2, a0, b0);
        if (two) mma2(a1, b1f);
    }
    float *Cs = smem;                                       // [64][CSS] over the dead hid rows (+ 1 KiB of slot 0): nobody reads LDS after the last barrier
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) Cs[(rt * 16 + kq * 4 + rr) * CSS + (nb + j) * 16 + li] = acc[j][rr];
    __syncthreads();
    float *out = parts + (size_t)sl * M * D;
#pragma unroll
    for (int it = 0; it < NST; ++it) {
        const int row = (tid >> 6) + it * NW, c4 = (tid & 63) << 2, gr = m0 + row;
        if (gr >= M) continue;
        float4 v = ldsv4(Cs + row * CSS + c4);
        if (sl == 0) {
            const float4 x = xres[it];
            v.x += x.x + bres.x; v.y += x.y + bres.y; v.z += x.z + bres.z; v.w += x.w + bres.w;
        }
        idf_store16_wt(out + (size_t)gr * D + c4, v);
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// LayerNorm + linear for the QKV projection of the two standard layers, on the same skeleton as phase 1 above:
//     C[M, N] = LN(sum of NP slabs of A)[M,256] . W[N,256]^T + bias            (nn.MultiheadAttention in_proj after the previous
//                                                                               layer's norm3; torch TransformerDecoderLayer)
// grid = ceil(M/32) x ceil(N/160) workgroups of 512 threads (50 x 5 = 250 at M = 1600, N = 768: one per CU; the last slice's
// columns >= N are zero weights and are not stored).  A workgroup normalises its 32 rows ONCE (wave w: rows w, w+8, w+16, w+24;
// the slab sum of common.h ld4_sum, all loads in flight together), parks them in the swizzled LDS image and streams its 160
// weight rows as 16 pre-packed k-group chunks ([160 rows][16 k] = 10 KiB, two per ring slot, two slots ahead).  The kernel is
// bound by its CU's matrix pipe: 2 row tiles x 10 column tiles = 5 tiles per SIMD (waves 0..3 own three column tiles, waves 4..7
// two; SIMD s holds waves s and s+4).  An earlier geometry -- 4 slices of 192 columns, 200 workgroups -- left 56 CUs idle and
// put 6 tiles on every SIMD (12.3 k instead of 10.2 k MFMA cycles per workgroup); the generic GEMM before it tiled N by 64, so
// every row block was fetched (five slabs) and normalised by TWELVE workgroups.
// The slice-0 workgroups also write the normalised rows (the residual of the attention block) to xn_out.
constexpr int LCT = 10, LHS = LCT * 16;                 // column tiles / columns per workgroup
constexpr int LW1C = LHS * 16;                          // floats per k-group chunk
constexpr int LPSLOT = 2 * LW1C;                        // floats per ring slot (20 KiB)
constexpr int LPI = 2 * LW1C / 256;                     // 1-KiB DMA instructions per pair: 20 = 8 + 8 + 4 (waves 0..3 issue a third one)
static_assert(LPI == 2 * NW + NW / 2, "pair = two instructions per wave + a third for the low half");

template <int NP>
__global__ __launch_bounds__(NT) void ln_linear_kernel(const float *__restrict__ A, size_t a_pstride, const float *__restrict__ lnw,
                                                        const float *__restrict__ lnb, int M, const float *__restrict__ pack,
                                                        const float *__restrict__ bias, float *__restrict__ C, int ldc, int N,
                                                        float *__restrict__ xn_out, int64_t *__restrict__ step_state,
                                                        int64_t *__restrict__ step_ts, int step_B, int nsl_grid) {
    __shared__ __attribute__((aligned(1024))) float smem[XS + 3 * LPSLOT];
    idf_args_now(A, a_pstride, lnw, lnb, M, pack, bias, C, ldc, N, xn_out, step_state, step_ts, step_B, nsl_grid, gridDim.x);
    // sampler bookkeeping of a fused plain step (philox.h): nobody else touches these words while this kernel runs
    if (step_state && blockIdx.x == 0 && threadIdx.x == 0) sampler_prepare_step(step_state, step_ts, step_B);
    float *Xs = smem, *ring = smem + XS;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int mt, sl;
    xcd_affine_tile(gridDim.x, blockIdx.x, nsl_grid, mt, sl);             // all column slices of an M tile on one XCD: its (up to five-slab) input rows cross the fabric once
    const int m0 = mt * BM, n0 = sl * LHS;
    const float *stream = idf_uniform_ptr(pack + (size_t)sl * (16 * LW1C));
    const uint32_t vsrc = (uint32_t)(wave * 1024) + (uint32_t)(lane << 4);
    const uint32_t sdst = idf_lds_addr(ring) + (uint32_t)(wave * 1024);
    const int key = (4 - (li >> 2)) & 3;
    const bool low = wave < NW / 2;                       // waves 0..3: a third DMA instruction per pair, a third column tile
    auto issue_pair = [&](int P) {                        // pair P = chunks 2P, 2P+1: 20 KiB contiguous in the stream and in slot P % 3
        if (P >= 8) return;
        const uint32_t so = (uint32_t)(P * 2 * LW1C * 4), dof = (uint32_t)((P % 3) * LPSLOT * 4);
        idf_dma16_s(stream, vsrc + so, sdst + dof);
        idf_dma16_s(stream, vsrc + so + 8192u, sdst + dof + 8192u);
        if (low) idf_dma16_s(stream, vsrc + so + 16384u, sdst + dof + 16384u);
    };
    auto wait_one_pair_flying = [&]() {                   // everything older than this wave's instructions of the youngest pair has landed
        if (low) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    };
    issue_pair(0);
    issue_pair(1);
    // rows: slab sum, LayerNorm (null lnw: layer 0 takes the embedding as it is), swizzled LDS image (chunk `lane` of row r at
    // position lane ^ (r & 15)), residual copy
    {
        const float4 gw = lnw ? *reinterpret_cast<const float4 *>(lnw + lane * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 gb = lnw ? *reinterpret_cast<const float4 *>(lnb + lane * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 v[BM / NW];
#pragma unroll
        for (int i = 0; i < BM / NW; ++i) v[i] = ld4_sum<NP>(A + (size_t)min(m0 + wave + NW * i, M - 1) * D + lane * 4, a_pstride);
#pragma unroll
        for (int i = 0; i < BM / NW; ++i) {
            const int row = wave + NW * i;
            float4 x = v[i];
            if (lnw) {
                float mean, rstd;
                ln_row_stats(x, mean, rstd);
                x.x = (x.x - mean) * rstd * gw.x + gb.x;
                x.y = (x.y - mean) * rstd * gw.y + gb.y;
                x.z = (x.z - mean) * rstd * gw.z + gb.z;
                x.w = (x.w - mean) * rstd * gw.w + gb.w;
            }
            *reinterpret_cast<float4 *>(Xs + row * D + ((lane ^ (row & 15)) << 2)) = x;
            if (xn_out && sl == 0 && m0 + row < M) idf_store16_wt(xn_out + (size_t)(m0 + row) * D + lane * 4, x);
        }
    }
    wait_one_pair_flying();                               // pair 0 (and everything older) has landed; pair 1 may fly
    __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0), as a builtin: the compiler then KNOWS the LDS queue is empty (an asm wait is invisible to its own counting)
    __builtin_amdgcn_s_barrier();

    // wave w: row tile w & 1; column tiles 3 (w >> 1) .. +2 (w < 4) or 6 + 2 ((w - 4) >> 1) .. +1 (w >= 4)
    const int r1 = wave & 1, c0 = low ? (wave >> 1) * 3 : 6 + ((wave - 4) >> 1) * 2;
    f32x4 acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 a0, a1, b0[3], b1f[3];
    const float *xb[4], *wb[3];
#pragma unroll
    for (int m = 0; m < 4; ++m) xb[m] = Xs + (r1 * 16 + li) * D + (((kq ^ li) ^ (4 * m)) << 2);
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) wb[s3] = ring + s3 * LPSLOT + ((kq ^ key) << 2) + li * 16 + c0 * 256;
    auto rd = [&](int c, float4 &a, float4 (&b)[3]) {
        a = ldsv4(xb[c & 3] + 64 * (c >> 2));
        const float *sb = wb[(c >> 1) % 3] + (c & 1) * LW1C;
        b[0] = ldsv4(sb);
        b[1] = ldsv4(sb + 256);
        if (low) b[2] = ldsv4(sb + 512);
    };
    auto mma3 = [&](const float4 &a, const float4 (&b)[3]) {
        IDF_FFN_MFMA(acc[0], a.x, b[0].x); IDF_FFN_MFMA(acc[1], a.x, b[1].x);
        IDF_FFN_MFMA(acc[0], a.y, b[0].y); IDF_FFN_MFMA(acc[1], a.y, b[1].y);
        IDF_FFN_MFMA(acc[0], a.z, b[0].z); IDF_FFN_MFMA(acc[1], a.z, b[1].z);
        IDF_FFN_MFMA(acc[0], a.w, b[0].w); IDF_FFN_MFMA(acc[1], a.w, b[1].w);
        if (low) {
            IDF_FFN_MFMA(acc[2], a.x, b[2].x); IDF_FFN_MFMA(acc[2], a.y, b[2].y); IDF_FFN_MFMA(acc[2], a.z, b[2].z); IDF_FFN_MFMA(acc[2], a.w, b[2].w);
        }
    };
    rd(0, a0, b0);
#pragma unroll
    for (int P = 0; P < 8; ++P) {
        issue_pair(P + 2);
        rd(2 * P + 1, a1, b1f);
        mma3(a0, b0);
        if (P + 2 < 8) wait_one_pair_flying();            // pair P+1 has landed; pair P+2 may keep flying
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0), as a builtin: the compiler then KNOWS the LDS queue is empty (an asm wait is invisible to its own counting)
        __builtin_amdgcn_s_barrier();
        if (P + 1 < 8) rd(2 * P + 2, a0, b0);
        mma3(a1, b1f);
    }
    // epilogue: + bias, through LDS, 16-byte row stores (write-through: the attention kernel reads this from other XCDs)
    constexpr int LCS = LHS + 4;
    float *Cs = ring;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        if (j < 2 || low) {
            const float bv = bias[min(n0 + (c0 + j) * 16 + li, N - 1)];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) Cs[(r1 * 16 + kq * 4 + rr) * LCS + (c0 + j) * 16 + li] = acc[j][rr] + bv;
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < (BM * (LHS / 4) + NT - 1) / NT; ++it) {
        const int idx = tid + it * NT, row = idx / (LHS / 4), c4 = (idx - row * (LHS / 4)) << 2, gr = m0 + row;
        if (idx < BM * (LHS / 4) && gr < M && n0 + c4 < N) idf_store16_wt(C + (size_t)gr * ldc + n0 + c4, ldsv4(Cs + row * LCS + c4));
    }
}

template <int NP>
inline void launch_ln_linear(hipStream_t s, const float *A, size_t a_pstride, const float *lnw, const float *lnb, int M, int N,
                             const float *pack, const float *bias, float *C, int ldc, float *xn_out, int64_t *step_state = nullptr,
                             int64_t *step_ts = nullptr, int step_B = 0) {
    const int nsl = (int)idf_cdiv(N, LHS);
    hipLaunchKernelGGL(ln_linear_kernel<NP>, dim3((unsigned)(idf_cdiv(M, BM) * nsl)), dim3(NT), 0, s, A, a_pstride, lnw, lnb,
                       M, pack, bias, C, ldc, N, xn_out, step_state, step_ts, step_B, nsl);
}

// rows: 16 / 32 / 64 = the M tile to use; 0 = choose by THIS launch's rows (ffn_tile_for_rows).  The kernels differ in the rounding of one
// column tile (32 vs 16 / 64), so a caller that splits a batch into chains must pass the choice made for the WHOLE batch
// (idf_mdm_weights.tune[IDF_TUNE_FFN]: 0 auto, 1 = 32, 2 = 16, 3 = 64; interdiff_amd/mdm.py sets it from the batch it is handed).
//   <= 800 rows: 16-row tiles (their grid still fits the chip in one round).
//   >= 2800 rows: 64-row tiles.  3200 rows = 250 x 64-row workgroups in one round (34.4 us) instead of 500 x 32-row in two (37.5); inside
//   the sampler, where such a batch steps as two chains whose launches interleave, the 64-row tile is ahead at every size measured
//   (30 / 32 / 40 / 48 / 64 / 96 clips of 100 frames: 2 / 2.6 / 3.6 / 4 / 8 / 1.5 % of a step, equal at 28, profiles/r03_batch_scaling.txt).  A lone launch just
//   past a multiple of 3264 rows is the exception (3300 rows: 260 workgroups in two rounds, 62 us, against 520 in three, 51): accepted.
//   in between: 32-row tiles (a batch of 24 clips steps as two chains of 1200 rows whose 32-row launches overlap: 0.356 vs 0.407 ms/step).
constexpr int FFN16_MAX_ROWS = 800, FFN64_MIN_ROWS = 2800;
inline int ffn_tile_for_rows(int rows) { return rows <= FFN16_MAX_ROWS ? 16 : (rows < FFN64_MIN_ROWS ? 32 : 64); }
inline int ffn_rows_of_tune(int t) { return t == 1 ? 32 : (t == 2 ? 16 : (t == 3 ? 64 : 0)); }
inline void launch_ffn(hipStream_t s, const float *x2, int M, const float *pack, const float *b1p, const float *b2, float *parts, int rows = 0) {
    if (rows == 0) rows = ffn_tile_for_rows(M);
    if (rows == 16) hipLaunchKernelGGL(ffn_fused16_kernel<0>, dim3((unsigned)(idf_cdiv(M, BMH) * NSL)), dim3(NT), 0, s, x2, M, pack, b1p, b2, parts);
    else if (rows == 64) hipLaunchKernelGGL(ffn_fused64_kernel<0>, dim3((unsigned)(idf_cdiv(M, BMX) * NSL)), dim3(NT), 0, s, x2, M, pack, b1p, b2, parts);
    else hipLaunchKernelGGL(ffn_fused_kernel<0>, dim3((unsigned)(idf_cdiv(M, BM) * NSL)), dim3(NT), 0, s, x2, M, pack, b1p, b2, parts);
}
}  // namespace idf_ffn
