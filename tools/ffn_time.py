"""FFN / forward timing of the tree this file sits in (not product code): used to A/B two builds on ONE box by running this script
from two checkouts in alternation (box-to-box spread is several per cent).  Prints one JSON line."""
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
    bench.run_steps(diff, model, None, bt, y, 57, seed=7)
    out = dict(tree=ROOT)
    out['ffn_cycling_us'] = [round(bench.time_dominant_kernel(model, dev)[0], 3) for _ in range(2)]
    out['ffn_one_layer_us'] = round(bench.time_dominant_kernel(model, dev, cycle_layers=False)[0], 3)
    out['ffn_pair_us'] = round(bench.time_dominant_kernel(model, dev, chains=2)[0], 3)
    out['forward_us'] = [round(bench.time_forward_graph(model, bt, y, dev), 2) for _ in range(2)]
    ts = []
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bench.run_steps(diff, model, None, bt, y, 1000, seed=3)
        torch.cuda.synchronize()
        ts.append(round(time.perf_counter() - t0, 5))
    out['no_correction_ms_per_step'] = ts
    print(json.dumps(out))


if __name__ == '__main__':
    main()
