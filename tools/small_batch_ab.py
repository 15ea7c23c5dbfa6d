"""Small batches (BASELINE config #4's 8 clips per GPU and below): whole 1000-step samples with the 32-row / 16-row feed-forward tile
and with / without the two-chain split, one box, one process (not product code)."""
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
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument('--batches', type=int, nargs='*', default=[8, 4, 16])
    ap.add_argument('--rows', type=int, nargs='*', default=[32, 16])
    ap.add_argument('--splits', type=int, nargs='*', default=[1, 0])
    a = ap.parse_args()
    torch.set_grad_enabled(False)
    dev = torch.device('cuda:0')
    for B in a.batches:
        bench.B_PER_GPU = B
        model, corr, bt, y, _ = bench.build_world(dev, 0)
        for rows in a.rows:
            for split in map(bool, a.splits):
                model.ffn_rows = rows
                model.__dict__.pop('_graph_cache', None)
                diff = create_gaussian_diffusion('cosine', bench.STEPS)
                diff.split_chains = split
                row = dict(B=B, ffn_rows=rows, split=split)
                for name, c in (('no_correction', None), ('correction', corr)):
                    bench.run_steps(diff, model, c, bt, y, 57, seed=7)
                    ts = []
                    for _ in range(2):
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        o = bench.run_steps(diff, model, c, bt, y, 1000, seed=3)
                        torch.cuda.synchronize()
                        ts.append(round(time.perf_counter() - t0, 5))
                    assert torch.isfinite(o).all()
                    row[name + '_ms_per_step'] = ts
                print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
