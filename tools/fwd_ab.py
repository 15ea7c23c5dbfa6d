"""Same-process A/B of denoiser-forward variants on one MI355X (not product code): idf_mdm_weights.tune[<slot>] switches of whatever
experiment is compiled in (round 3 used the reserved IDF_TUNE_MISC slot for the L2 warm-ups, the static wave priority and the side-stream
prefetch -- all measured, none kept: the shipped library reads no bit of it), each setting timed as
(a) one forward replayed from a hipGraph and (b) whole 1000-step samples without correction (the sampler's two-chain graphs), in
alternation.  Box-to-box spread is several per cent: only same-call ratios mean anything.
    python tools/fwd_ab.py [--settings 3,0,2,1] [--reps 3] [--B 16]"""
import argparse
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
    ap = argparse.ArgumentParser()
    ap.add_argument('--settings', default='3,0,2,1')
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--B', type=int, default=16)
    ap.add_argument('--slot', default='misc')
    args = ap.parse_args()
    torch.set_grad_enabled(False)
    dev = torch.device('cuda:0')
    bench.B_PER_GPU = args.B
    model, corr, bt, y, _ = bench.build_world(dev, 0)
    settings = [int(s) for s in args.settings.split(',')]
    diffs, ref, out = {}, None, {}
    for m in settings:
        model.w.tune[_lib.TUNE[args.slot]] = m
        diffs[m] = create_gaussian_diffusion('cosine', bench.STEPS)              # its own graph cache entry: the captures bake the switches in
        x = bench.run_steps(diffs[m], model, None, bt, y, 1000, seed=3)
        ref = x if ref is None else ref
        out['setting%d_identical' % m] = bool(torch.equal(x, ref))
    for rep in range(args.reps):
        for m in settings:
            model.w.tune[_lib.TUNE[args.slot]] = m
            out.setdefault('setting%d_forward_us' % m, []).append(round(bench.time_forward_graph(model, bt, y, dev), 2))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            bench.run_steps(diffs[m], model, None, bt, y, 1000, seed=3)
            torch.cuda.synchronize()
            out.setdefault('setting%d_sample_ms_per_step' % m, []).append(round(time.perf_counter() - t0, 5))
    model.w.tune[_lib.TUNE[args.slot]] = 0
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
