"""Three vs four ring slots in the split-f16 feed-forward kernel (csrc/ffn_h2.h; tune[IDF_TUNE_MISC] = 6 selects three): per-launch time from a layer-cycling
graph, denoiser forward and whole samples, alternating in one process (not product code).  Output -> profiles/r04_ffn_ring_slots_ab.txt."""
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
from tools.ffn16_ab import time_ffn                                               # noqa: E402


def main():
    torch.set_grad_enabled(False)
    dev = torch.device('cuda:0')
    model, corr, bt, y, _ = bench.build_world(dev, 0)
    diff = create_gaussian_diffusion('cosine', bench.STEPS)
    for rep in range(3):
        for name, misc in (('three slots', 6), ('four slots', 0)):
            model.w.tune[_lib.TUNE['misc']] = misc
            model.__dict__.pop('_graph_cache', None)
            out = dict(ring=name, ffn_us_1600=round(time_ffn(model, dev, 1600), 3), ffn_us_800=round(time_ffn(model, dev, 800), 3),
                       forward_us=round(bench.time_forward_graph(model, bt, y, dev), 2))
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
