"""Kernel-level A/B bench on one MI355X (not part of the product path).

    python tools/kbench.py [--B 16] [--T 100] [--reps 20] [--sweep]

Times every kernel kind of one denoiser forward (HIP events on the launch stream, via the library's profile
hooks) and, with --sweep, re-times the forward under every tile configuration of each GEMM call site
(idf_mdm_weights.tune, a field of the model handle), and times the fused FFN kernel against the two-GEMM form it replaced.  Output: one table per sweep + a JSON line, so the numbers can be pasted into profiles/.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from interdiff_amd import synthetic as syn, _lib          # noqa: E402
from interdiff_amd.mdm import MDM, ffn_parts, linear        # noqa: E402

TUNE = _lib.TUNE
KIND_OF = dict(embed='embed', qkv='gemm_qkv', outproj='gemm_outproj', heads='gemm_heads')
CFGS = {0: 'default', 1: '32x64 ks1 kc32', 2: '32x64 ks1 kc64', 3: '32x64 ks2 kc64', 4: '64x64 ks1 kc32', 5: '32x32 ks1 kc64',
        6: '32x32 ks2 kc64', 7: '64x32 ks1 kc32', 8: '64x32 ks2 kc64', 9: 'reg-staged 32x64 kc32'}


def profile_forward(lib, model, x, ts, y, reps):
    for _ in range(3):
        model(x, ts, y=y)
    torch.cuda.synchronize()
    _lib.check(lib.interdiff_profile_begin(100000))
    for _ in range(reps):
        model(x, ts, y=y)
    ms = (C.c_double * len(_lib.KERNEL_KINDS))()
    cnt = (C.c_int64 * len(_lib.KERNEL_KINDS))()
    _lib.check(lib.interdiff_profile_end(ms, cnt))
    out = {k: 1e3 * ms[i] / cnt[i] for i, k in enumerate(_lib.KERNEL_KINDS) if cnt[i]}
    out['_launches_per_forward'] = sum(cnt[i] for i in range(len(_lib.KERNEL_KINDS))) / reps
    out['_sum_us_per_forward'] = 1e3 * sum(ms[i] for i in range(len(_lib.KERNEL_KINDS))) / reps
    return out


def wall_forward(model, x, ts, y, reps):
    for _ in range(3):
        model(x, ts, y=y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        model(x, ts, y=y)
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / reps


def burst_us(fn, reps=200, per_graph=50):
    """us per call of fn, replayed back to back from a hipGraph on a side stream (the GPU's time, whatever the host does)."""
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    side, graph = torch.cuda.Stream(), torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            for _ in range(per_graph):
                fn()
        graph.replay()
        side.synchronize()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side)
            for _ in range(reps // per_graph):
                graph.replay()
            e1.record(side)
            e1.synchronize()
            ts.append(1e3 * e0.elapsed_time(e1) / (reps // per_graph * per_graph))
    return dict(mean=sum(ts) / len(ts), best=min(ts))


def ffn_ab(model, M):
    """The fused FFN kernel (csrc/ffn.h) vs the two LDS-DMA GEMMs it replaced (linear1+gelu -> HBM -> linear2+residual), same
    weights (decoder layer 1), same M; also the parity of the two forms."""
    g = torch.Generator().manual_seed(5)
    x2 = torch.randn(M, 256, generator=g).cuda()
    ly = model.w.layer[1]
    A = model.arena
    w1, b1 = A[ly.ff1_w:ly.ff1_w + 1024 * 256].view(1024, 256), A[ly.ff1_b:ly.ff1_b + 1024]
    w2, b2 = A[ly.ff2_w:ly.ff2_w + 256 * 1024].view(256, 1024), A[ly.ff2_b:ly.ff2_b + 256]
    parts = torch.empty(_lib.FFN_SLICES, M, 256, device='cuda')
    hid, out = torch.empty(M, 1024, device='cuda'), torch.empty(M, 256, device='cuda')
    fused = burst_us(lambda: ffn_parts(model, x2, 1, out=parts))

    def two():
        linear(x2, w1, b1, gelu=True, out=hid, cfg=7)
        linear(hid, w2, b2, residual=x2, out=out, cfg=6)
    pair = burst_us(two)
    two()
    ffn_parts(model, x2, 1, out=parts)
    err = float((parts.sum(0) - out).abs().max() / out.abs().max())
    fl = 2 * 2.0 * M * 256 * 1024
    return dict(fused_us=fused, two_gemms_us=pair, fused_tflops=fl / fused['mean'] / 1e6, two_gemms_tflops=fl / pair['mean'] / 1e6,
                fused_frac_of_157p3=fl / fused['mean'] / 1e6 / 157.3, rel_diff_fused_vs_two_gemms=err)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--B', type=int, default=16)
    ap.add_argument('--T', type=int, default=100)
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--sweep', action='store_true')
    ap.add_argument('--sites', default='outproj,qkv,heads,embed')
    args = ap.parse_args()
    torch.set_grad_enabled(False)
    lib = _lib.load()
    dev = 'cuda'
    model = MDM(syn.mdm_state_dict(233), device=dev)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(args.B, 1, 144, args.T, generator=g).to(dev)
    cond = torch.randn(10, args.B, 256, generator=g).to(dev)
    ts = torch.full((args.B,), 500, dtype=torch.int64, device=dev)
    y = {'cond': cond}
    base = profile_forward(lib, model, x, ts, y, args.reps)
    wall = wall_forward(model, x, ts, y, 200)
    print('== default configuration, B=%d T=%d: forward wall %.1f us, sum of kernels %.1f us, %d launches'
          % (args.B, args.T, wall, base['_sum_us_per_forward'], base['_launches_per_forward']))
    for k, v in base.items():
        if not k.startswith('_'):
            print('   %-14s %8.2f us' % (k, v))
    result = dict(B=args.B, T=args.T, default=base, wall_us=wall, sweeps={})
    result['ffn'] = ffn_ab(model, args.B * args.T)
    print('== feed-forward block at M=%d (graph-replayed bursts, us per layer): %s' % (args.B * args.T, json.dumps(result['ffn'])))
    if args.sweep:
        ref = model(x, ts, y=y).clone()
        for site in args.sites.split(','):
            cfgs = range(5) if site == 'embed' else range(10)
            row = {}
            for c in cfgs:
                model.w.tune[TUNE[site]] = c
                p = profile_forward(lib, model, x, ts, y, args.reps)
                # out-projection: cfg 0 = folded into the attention kernel (no GEMM launch), so compare the whole attention block
                row[c] = (sum(p.get(k, 0.0) for k in ('self_attn', 'gemm_outproj', 'rowblock_std')) if site == 'outproj' else p[KIND_OF[site]])
                err = ((model(x, ts, y=y) - ref).abs().max() / ref.abs().max()).item()
                assert err < (2e-5 if site == 'outproj' else 1e-5), (site, c, err)      # (per-head partial sums round differently)
            model.w.tune[TUNE[site]] = 0
            result['sweeps'][site] = row
            print('== %s' % site)
            for c, v in row.items():
                print('   cfg %d %-18s %8.2f us%s' % (c, CFGS[c] if site != 'embed' else 'embed variant', v,
                                                     ' (self_attn + out-projection + row block)' if site == 'outproj' else ''))
    print(json.dumps(result))


if __name__ == '__main__':
    main()
