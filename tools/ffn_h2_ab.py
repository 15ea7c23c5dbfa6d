"""Exact-fp32 vs split-f16 fused feed-forward kernel (csrc/ffn.h vs csrc/ffn_h2.h), same process, same box (not product code):
per-launch time from a layer-cycling hipGraph by token count and row tile, error of both against torch fp64 on the model's own
weights, denoiser forward and whole-sample time under each.  Output -> profiles/r04_ffn_split_f16_ab.txt."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                                      # noqa: E402
from interdiff_amd.mdm import ffn_parts                                           # noqa: E402
from interdiff_amd.diffusion import create_gaussian_diffusion                     # noqa: E402
from tools.ffn16_ab import time_ffn                                               # noqa: E402


def err_fp64(model, sd, N, layer=3):
    g = torch.Generator().manual_seed(11)
    x2 = torch.randn(N, 256, generator=g)
    got = ffn_parts(model, x2.cuda(), layer).sum(0).cpu().double()
    pre = 'decoder.layers.%d.' % layer
    w1, b1, w2, b2 = (torch.as_tensor(sd[pre + k]).double() for k in ('linear1.weight', 'linear1.bias', 'linear2.weight', 'linear2.bias'))
    xd = x2.double()
    ref = xd + torch.nn.functional.gelu(xd @ w1.T + b1) @ w2.T + b2
    return float((got - ref).abs().max() / ref.abs().max())


def main():
    torch.set_grad_enabled(False)
    dev = torch.device('cuda:0')
    model, corr, bt, y, (sd, _, _) = bench.build_world(dev, 0)
    diff = create_gaussian_diffusion('cosine', bench.STEPS)
    for N in (800, 1600, 3200):
        row = {}
        for math in ('exact', 'split'):
            model.ffn_math = math
            for rows in (16, 32, 64):
                model.ffn_rows = rows
                row['%s_rows%d_us' % (math, rows)] = time_ffn(model, dev, N)
            model.ffn_rows = 0
            row[math + '_err_vs_fp64'] = err_fp64(model, sd, N)
        print('ffn', N, json.dumps(row), flush=True)
    from interdiff_amd import _lib
    for rep in range(2):
        for math in ('exact', 'split', 'split_slice_major', 'split_plain_ids'):
            model.ffn_math = 'exact' if math == 'exact' else 'split'
            model.w.tune[_lib.TUNE['misc']] = {'split_slice_major': 2, 'split_plain_ids': 3}.get(math, 0)
            if rep == 0:
                print('burst', math, time_ffn(model, dev, 1600), time_ffn(model, dev, 800), flush=True)
            model.__dict__.pop('_graph_cache', None)
            out = dict(math=math)
            out['forward_us'] = round(bench.time_forward_graph(model, bt, y, dev), 2)
            bench.run_steps(diff, model, None, bt, y, 57, seed=7)
            ts = []
            for _ in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                bench.run_steps(diff, model, None, bt, y, 1000, seed=3)
                torch.cuda.synchronize()
                ts.append(round(time.perf_counter() - t0, 5))
            out['no_correction_ms_per_step'] = ts
            ts = []
            bench.run_steps(diff, model, corr, bt, y, 1000, seed=3)
            for _ in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                bench.run_steps(diff, model, corr, bt, y, 1000, seed=3)
                torch.cuda.synchronize()
                ts.append(round(time.perf_counter() - t0, 5))
            out['correction_ms_per_step'] = ts
            print('sample', json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
