"""Same-process A/B of the sampler's chain forms on one MI355X (not product code): whole 1000-step samples with correction at
B = 16 and B = 32, alternating  joined chains + whole-batch hook steps (stagger 0)  and  chains on their own streams, hook per
half batch, chain 1 `s` plain steps behind (stagger s).  Box-to-box spread is ~8 %, so only same-call ratios mean anything.
    python tools/stagger_ab.py [--staggers 0,1,7,25] [--reps 3]"""
import argparse
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
    ap = argparse.ArgumentParser()
    ap.add_argument('--staggers', default='0,1,7,25')
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--batches', default='16,32')
    args = ap.parse_args()
    torch.set_grad_enabled(False)
    dev = torch.device('cuda:0')
    out = {}
    for B in (int(b) for b in args.batches.split(',')):
        bench.B_PER_GPU = B
        model, corr, bt, y, _ = bench.build_world(dev, 0)
        diff = create_gaussian_diffusion('cosine', bench.STEPS)
        res, ref = {}, None
        staggers = [int(s) for s in args.staggers.split(',')]
        for s in staggers:                                        # warm-up + captures
            diff.stagger_steps = s
            x = bench.run_steps(diff, model, corr, bt, y, 1000, seed=3)
            if ref is None:
                ref = x
            res['stagger%d_identical' % s] = bool(torch.equal(x, ref))
        for rep in range(args.reps):
            for s in staggers:
                diff.stagger_steps = s
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                bench.run_steps(diff, model, corr, bt, y, 1000, seed=3)
                torch.cuda.synchronize()
                res.setdefault('stagger%d_ms_per_step' % s, []).append(round(time.perf_counter() - t0, 5))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bench.run_steps(diff, model, None, bt, y, 1000, seed=3)
        torch.cuda.synchronize()
        res['no_correction_ms_per_step'] = round(time.perf_counter() - t0, 5)
        out['B%d' % B] = res
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
