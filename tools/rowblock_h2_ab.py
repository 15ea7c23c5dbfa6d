"""Exact-fp32 vs split-f16 row block (csrc/denoiser.hip rowblock_kernel<.., H2>) under the split-f16 feed-forward / QKV arithmetic: denoiser forward and
whole samples, alternating in one process on one box (not product code).  Output -> profiles/r04_rowblock_split_f16_ab.txt."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                                      # noqa: E402
from interdiff_amd.diffusion import create_gaussian_diffusion                     # noqa: E402


def main():
    torch.set_grad_enabled(False)
    dev = torch.device('cuda:0')
    model, corr, bt, y, _ = bench.build_world(dev, 0)
    diff = create_gaussian_diffusion('cosine', bench.STEPS)
    model.ffn_math = 'split'
    for rep in range(3):
        for rb in ('exact', 'split'):
            model.rowblock_math = rb
            out = dict(rowblock_math=rb, forward_us=round(bench.time_forward_graph(model, bt, y, dev), 2))
            bench.run_steps(diff, model, None, bt, y, 57, seed=7)
            for name, hook in (('no_correction_ms_per_step', None), ('correction_ms_per_step', corr)):
                bench.run_steps(diff, model, hook, bt, y, 1000, seed=3)
                ts = []
                for _ in range(2):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    bench.run_steps(diff, model, hook, bt, y, 1000, seed=3)
                    torch.cuda.synchronize()
                    ts.append(round(time.perf_counter() - t0, 5))
                out[name] = ts
            print('sample', json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
