"""Sizes of the captured plain-step blocks (diffusion.GRAPH_BLOCKS), same process (not product code).  rocprofv3 shows an idle gap
between the last launch of a two-chain block and the first launch of the next replay (~0.3 ms for a 49-step block; none in the one-chain
form); it scales with the block's node count, so smaller blocks do not lose and larger ones do.  Whole samples, ms per step."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                                      # noqa: E402
from interdiff_amd import diffusion as dfn                                        # noqa: E402


def main():
    torch.set_grad_enabled(False)
    dev = torch.device('cuda:0')
    bench.B_PER_GPU = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    model, corr, bt, y, _ = bench.build_world(dev, 0)
    for blocks in ((12, 7, 1), (24, 7, 1), (49, 7, 1), (70, 49, 7, 1), (98, 49, 7, 1), (140, 49, 7, 1), (245, 49, 7, 1)):
        dfn.GRAPH_BLOCKS = blocks
        model.__dict__.pop('_graph_cache', None)
        diff = dfn.create_gaussian_diffusion('cosine', bench.STEPS)
        row = dict(B=bench.B_PER_GPU, blocks=blocks)
        for name, c in (('no_correction', None), ('correction', corr)):
            t0 = time.perf_counter()
            bench.run_steps(diff, model, c, bt, y, 1000, seed=7)
            torch.cuda.synchronize()
            row[name + '_first_sample_s'] = round(time.perf_counter() - t0, 3)          # includes the captures
            ts = []
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                bench.run_steps(diff, model, c, bt, y, 1000, seed=3)
                torch.cuda.synchronize()
                ts.append(round(time.perf_counter() - t0, 5))
            row[name + '_ms_per_step'] = ts
        print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
