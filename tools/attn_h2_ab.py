"""fp32 vs split-f16 self-attention kernel (csrc/denoiser.hip self_attn_kernel vs csrc/attn_h2.h) under the split arithmetic: denoiser forward and whole
samples, alternating in one process on one box (not product code).  Output -> profiles/r04_attn_split_f16_ab.txt (round 4: V transposed through LDS), profiles/r05_attn_split_f16_ab.txt (round 5: transposing reads)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                                      # noqa: E402
from interdiff_amd import _lib                                                    # noqa: E402
from interdiff_amd.diffusion import create_gaussian_diffusion                     # noqa: E402


def main():
    torch.set_grad_enabled(False)
    dev = torch.device('cuda:0')
    model, corr, bt, y, _ = bench.build_world(dev, 0)
    diff = create_gaussian_diffusion('cosine', bench.STEPS)
    for rep in range(3):
        settings = (('fp32 attention', 6), ('split-f16 attention', 0))       # (round 5: the split-f16 kernel is the default; 6 selects the fp32 kernel)
        if os.environ.get('ATTN_AB') == 'planes':                           # the QKV kernel's output form in front of the split-f16 attention: fp32 rows (9) vs plane pairs (default)
            settings = (('split-f16 attention, fp32 rows from the QKV kernel', 9), ('split-f16 attention, plane pairs from the QKV kernel', 0))
        for name, misc in settings:
            model.w.tune[_lib.TUNE['misc']] = misc
            model.__dict__.pop('_graph_cache', None)
            out = dict(attention=name, forward_us=round(bench.time_forward_graph(model, bt, y, dev), 2))
            bench.run_steps(diff, model, None, bt, y, 57, seed=7)
            for key, hook in (('no_correction_ms_per_step', None), ('correction_ms_per_step', corr)):
                bench.run_steps(diff, model, hook, bt, y, 1000, seed=3)
                ts = []
                for _ in range(2):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    bench.run_steps(diff, model, hook, bt, y, 1000, seed=3)
                    torch.cuda.synchronize()
                    ts.append(round(time.perf_counter() - t0, 5))
                out[key] = ts
            print('sample', json.dumps(out), flush=True)
    model.w.tune[_lib.TUNE['misc']] = 0


if __name__ == '__main__':
    main()
