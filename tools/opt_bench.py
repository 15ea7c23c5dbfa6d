"""Timing of the physics post-optimisation (row N4) at BASELINE.json configs[4] per-GPU scale: B=16 clips of T=20 frames
(optimization.py:216: past 10 + future 10), 2048 object points, the full 200-iteration schedule.

    python tools/opt_bench.py [--clips 16] [--frames 20] [--points 2048] [--reps 3]
Prints one JSON line: iterations/s, clips/s, ms per Adam iteration."""
import argparse
import json
import os
import sys
import time
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from interdiff_amd import synthetic as syn            # noqa: E402
from interdiff_amd.smpl import SMPL_Layer             # noqa: E402
from interdiff_amd.optimize import PhysicsOptimizer   # noqa: E402


OPT_KEYS = ('pose', 'trans', 'obj_angles', 'obj_trans', 'betas', 'obj_points')


def clip_batch(B, T, P, seed=1):
    bt = syn.make_optim_batch(seed=seed, B=B, T=T, n_points=P)
    return [torch.from_numpy(bt[k]).cuda() for k in OPT_KEYS]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--clips', type=int, default=16)
    ap.add_argument('--frames', type=int, default=20)
    ap.add_argument('--points', type=int, default=2048)
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--brute-force', action='store_true', help='round-2 nearest-neighbour kernel (no scan order)')
    a = ap.parse_args()
    opt = PhysicsOptimizer(SMPL_Layer({k: torch.from_numpy(v) for k, v in syn.smplh_model().items()}, device='cuda'), scan_order=not a.brute_force)
    batch = clip_batch(a.clips, a.frames, a.points)
    opt.optimize(*batch, iters=range(0, 5))              # warm-up
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(a.reps):
        t0 = time.perf_counter()
        res = opt.optimize(*batch)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    ls = res['losses'].cpu().numpy()
    print(json.dumps(dict(metric='adam-iterations/sec (B clips side by side)', clips=a.clips, frames=a.frames, points=a.points,
                          seconds_per_200_iterations=best, ms_per_iteration=best / 200 * 1e3, clips_per_sec=a.clips / best,
                          frame_iterations_per_sec=a.clips * a.frames * 200 / best, saved=bool(res['saved'].all()),
                          loss_first=float(ls[0, :, 0].mean()), loss_last=float(ls[-1, :, 0].mean()))))


if __name__ == '__main__':
    main()
