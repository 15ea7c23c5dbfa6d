// Phase stamps of the row-block kernel inside a real denoiser forward (not product code):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-inline-asm -I interdiff_amd/csrc tools/rowblock_probe.hip -o build_tools/rowblock_probe
// Builds csrc/denoiser.hip as ONE translation unit with IDF_RB_STAMP defined: thread 0 of every workgroup of the QaN row block
// writes the shader clock at each phase boundary.  Weights / activations are random (timing only).  Prints the mean cycles per
// phase over the workgroups of the LAST QaN launch of a forward at B x T.
#include <hip/hip_runtime.h>
__device__ long long g_rb_stamps[8192 * 16];
#define IDF_RB_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0) g_rb_stamps[(blockIdx.x + gridDim.x * blockIdx.y) * 16 + (i)] = clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
__device__ long long g_at_stamps[8192 * 8];
#define IDF_AT_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0) g_at_stamps[(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8 + (i)] = clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
__device__ long long g_ah2_stamps[8192 * 8];
#define IDF_AH2_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0) g_ah2_stamps[blockIdx.x * 8 + (i)] = clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#include "denoiser.hip"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
bool g_idf_prof_on = false;
void idf_prof_mark_slow(int, hipStream_t) {}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 16, T = argc > 2 ? atoi(argv[2]) : 100, h2 = argc > 3 ? atoi(argv[3]) : 0;     // h2 = 1: the split-f16 row block (round 5: the eight-wave rowblock8_kernel)
    const int tokens = argc > 5 ? atoi(argv[5]) : 0;                                                                       // tokens per workgroup of the eight-wave kernel: 8 / 16 (0: the launcher's choice)
    const int waves4 = argc > 4 ? atoi(argv[4]) : 0;                                                                       // 1 (with h2 = 1): round 4's four-wave split-f16 kernel instead
    // a weights struct whose every offset points into one big random arena (layout irrelevant for timing)
    const size_t arena_floats = (size_t)40 << 20;
    std::vector<float> h(arena_floats);
    srand(1);
    for (auto &v : h) v = (rand() / (float)RAND_MAX - 0.5f) * 0.05f;
    float *arena;
    CK(hipMalloc(&arena, arena_floats * 4));
    CK(hipMemcpy(arena, h.data(), arena_floats * 4, hipMemcpyHostToDevice));
    idf_mdm_weights w{};
    w.C = 144; w.n_steps = 1000; w.arena = arena; w.max_T = 512;
    size_t off = 0;
    auto take = [&](size_t n) { const size_t o = off; off += (n + 63) / 64 * 64; return (int64_t)o; };
    w.in_w = take(256 * 144); w.in_b = take(256); w.out_w = take(144 * 256); w.out_b = take(144);
    w.temb_table = take(1000 * 256); w.pe = take(512 * 256);
    for (int l = 0; l < 8; ++l) {
        idf_mdm_layer &ly = w.layer[l];
        ly.is_qan = (l >= 1);                 // the last launch is a QaN row block: its stamps are the ones read back
        ly.sa_in_w = take(768 * 256); ly.sa_in_b = take(768); ly.sa_out_w = take(256 * 256); ly.sa_out_b = take(256); ly.sa_out_frag = take(256 * 256);
        ly.qc = take(16 * 3 * 40 * 4); ly.wk = take(64);
        if (h2) { ly.qc_h2 = take(16 * 3 * 40 * 4); ly.rb_h2_ok = 1; }
        if (h2 == 2) ly.sa_out_frag_h2 = take(256 * 256);
        ly.ca_out_b = take(256);
        ly.ff1_b = take(1024); ly.ff2_b = take(256); ly.ffn_pack = take(5 * 106496); ly.ffn_b1p = take(5 * 208 + 256);
        for (int k = 0; k < 3; ++k) { ly.ln_w[k] = take(256); ly.ln_b[k] = take(256); }
    }
    if (h2) w.tune[IDF_TUNE_FFN_MATH] = 1;
    if (h2 == 1 && waves4) w.tune[IDF_TUNE_MISC] = 8;
    w.rb_tokens = tokens;
    const int tv = (h2 == 1 && !waves4) ? (tokens ? tokens : (B * ((T + 7) / 8) <= 256 ? 8 : 16)) : 16;
    if (h2 == 2) w.tune[IDF_TUNE_MISC] = 5;       // the split-f16 attention kernel (csrc/attn_h2.h)        // (no split-f16 FFN / QKV streams are set: those stay exact)
    if (off > arena_floats) { printf("arena too small\n"); return 1; }
    float *memctx, *x, *x0;
    int64_t *ts;
    void *ws;
    const size_t wsb = interdiff_mdm_workspace_bytes(B, T);
    CK(hipMalloc(&memctx, interdiff_mdm_memctx_floats(B) * 4));
    CK(hipMemcpy(memctx, h.data(), interdiff_mdm_memctx_floats(B) * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&x, (size_t)B * 144 * T * 4)); CK(hipMalloc(&x0, (size_t)B * 144 * T * 4)); CK(hipMalloc(&ts, B * 8)); CK(hipMalloc(&ws, wsb));
    CK(hipMemcpy(x, h.data(), (size_t)B * 144 * T * 4, hipMemcpyHostToDevice));
    CK(hipMemset(ts, 0, B * 8));
    for (int i = 0; i < 5; ++i)
        if (interdiff_mdm_forward(&w, memctx, x, ts, B, T, x0, ws, wsb, nullptr) != 0) { printf("forward failed\n"); return 1; }
    CK(hipDeviceSynchronize());
    const bool replay = argc > 6 && atoi(argv[6]) != 0;       // 7th argument 1: the stamps of a captured forward REPLAYED as a hipGraph (how the product runs it) instead of plain launches
    if (replay) {
        hipStream_t cs;
        hipGraph_t gr;
        hipGraphExec_t ge;
        CK(hipStreamCreate(&cs));
        CK(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 2; ++i)
            if (interdiff_mdm_forward(&w, memctx, x, ts, B, T, x0, ws, wsb, cs) != 0) { printf("forward failed\n"); return 1; }
        CK(hipStreamEndCapture(cs, &gr));
        CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
        for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, cs));
        CK(hipStreamSynchronize(cs));
        printf("(stamps of a hipGraph replay: two forwards per graph, five replays)\n");
    }
    const int nwg = ((T + tv - 1) / tv) * B;
    std::vector<long long> st((size_t)nwg * 16);
    CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_rb_stamps), st.size() * 8));
    const char *names[9] = {"", "rows + slab sum + LN_prev (+ Qc requested)", "logits MFMA (waits Qc)", "tap softmax + coefficients", "stencil + LN1",
                            "folded scores MFMA (waits G)", "head softmax", "P.VW MFMA (waits VW)", "LN2 + store"};
    double acc[9] = {0}, tot = 0;
    for (int wgi = 0; wgi < nwg; ++wgi)
        for (int i = 1; i < 9; ++i) acc[i] += (double)(st[(size_t)wgi * 16 + i] - st[(size_t)wgi * 16 + i - 1]);
    printf("QaN row block%s (last launch of the forward), B=%d T=%d, %d workgroups; mean cycles per phase:\n", h2 ? (waves4 || h2 != 1 ? ", split-f16 contractions, four waves" : (tv == 8 ? ", split-f16 contractions, EIGHT waves, 8 tokens per workgroup (rowblock8_kernel)" : ", split-f16 contractions, EIGHT waves, 16 tokens per workgroup (rowblock8_kernel)")) : "", B, T, nwg);
    for (int i = 1; i < 9; ++i) { printf("  %-46s %8.0f\n", names[i], acc[i] / nwg); tot += acc[i] / nwg; }
    printf("  total %.0f\n", tot);
    if (h2 == 1 && !waves4) {     // rowblock8_kernel: finer stamps of wave 0, all counted from the kernel's first instruction (stamp 0)
        double a[16] = {0};
        for (int wgi = 0; wgi < nwg; ++wgi)
            for (int i = 9; i < 16; ++i) a[i] += (double)(st[(size_t)wgi * 16 + i] - st[(size_t)wgi * 16 + (i == 13 || i == 14 ? 1 : 0)]);
        printf("  inside the first phase, from the kernel's first instruction (wave 0): token rows requested %.0f, rest of the argument segment read + every request issued %.0f, its rows landed %.0f,\n"
               "    past the barrier behind the landing (every wave's rows are there) %.0f, wave 0 done with its rows (two LayerNorms side by side: its own + a halo row) %.0f\n",
               a[9] / nwg, a[15] / nwg, a[10] / nwg, a[11] / nwg, a[12] / nwg);
        printf("  inside the logits phase (wave 0): operands read + MFMAs issued %.0f, partial tiles stored + VW requested %.0f (then the barrier)\n", a[13] / nwg, a[14] / nwg);
    } else {
        double a9 = 0, a10 = 0;
        for (int wgi = 0; wgi < nwg; ++wgi) {
            a9 += (double)(st[(size_t)wgi * 16 + 9] - st[(size_t)wgi * 16]);
            a10 += (double)(st[(size_t)wgi * 16 + 10] - st[(size_t)wgi * 16]);
        }
        printf("  inside the first phase (four-wave kernel: counted from behind the argument wait): all requests issued %.0f, first batch landed (wave 0) %.0f\n", a9 / nwg, a10 / nwg);
    }
    {
        const int nat = ((T + 16 * ATTN_RT - 1) / (16 * ATTN_RT)) * 4 * B;
        std::vector<long long> sa((size_t)nat * 8);
        CK(hipMemcpyFromSymbol(sa.data(), HIP_SYMBOL(g_at_stamps), sa.size() * 8));
        const char *an[6] = {"", "K, V, Q -> LDS (+ W_o requested)", "S = Q K^T", "row softmax", "P V", "out-projection partial + store"};
        double aa[6] = {0}, at = 0;
        for (int w = 0; w < nat; ++w)
            for (int i = 1; i < 6; ++i) aa[i] += (double)(sa[(size_t)w * 8 + i] - sa[(size_t)w * 8 + i - 1]);
        printf("self-attention (last launch), %d workgroups; mean cycles per phase:\n", nat);
        for (int i = 1; i < 6; ++i) { printf("  %-46s %8.0f\n", an[i], aa[i] / nat); at += aa[i] / nat; }
        printf("  total %.0f\n", at);
        double b7 = 0, b6 = 0;
        for (int w = 0; w < nat; ++w) { b7 += (double)(sa[(size_t)w * 8 + 7] - sa[(size_t)w * 8 + 4]); b6 += (double)(sa[(size_t)w * 8 + 6] - sa[(size_t)w * 8 + 4]); }
        printf("  inside the last phase: context tile in LDS + barrier %.0f, out-projection MFMAs issued %.0f, then the 16 partial-slab stores per lane\n", b7 / nat, b6 / nat);
    }
    if (h2 == 2) {
        const int nw = ((T + 31) / 32) * 4 * B;
        std::vector<long long> sa((size_t)nw * 8);
        CK(hipMemcpyFromSymbol(sa.data(), HIP_SYMBOL(g_ah2_stamps), sa.size() * 8));
        const char *an[7] = {"", "operand fetch", "tile scales + Q / K / V^T planes", "S = Q K^T", "softmax + probability planes", "P V + context planes", "out-projection + stores issued"};
        double aa[7] = {0}, at = 0;
        for (int w = 0; w < nw; ++w)
            for (int i = 1; i < 7; ++i) aa[i] += (double)(sa[(size_t)w * 8 + i] - sa[(size_t)w * 8 + i - 1]);
        printf("split-f16 self-attention (last launch), %d workgroups; mean cycles per phase:\n", nw);
        for (int i = 1; i < 7; ++i) { printf("  %-46s %8.0f\n", an[i], aa[i] / nw); at += aa[i] / nw; }
        printf("  total %.0f\n", at);
    }
    return 0;
}
