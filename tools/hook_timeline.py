"""The correction hook replayed from a captured graph, product form (tune 0: the predictor's stacks inside the scan's launch; in the round-6 experiment this was the side-stream
form, profiles/r06_hook_overlap.txt) vs the predictor launched after the scan (tune 2) (not product code):
    python tools/hook_timeline.py            -> us per replay of a graph holding ONE hook call, both routes (torch events)
    python tools/hook_timeline.py --db X.db  -> from a rocprofv3 kernel trace of the run above: the launches of the last replays with start / end relative to the first"""
import argparse
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timeline(db):
    con = sqlite3.connect(db)
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
    idx = [i for i, r in enumerate(rows) if 'corr_prepare' in r[0]]
    for which in (idx[len(idx) // 2 - 1], idx[-1]):          # the last replay of each route
        t0 = rows[which][1]
        print('---')
        for name, s, e in rows[which:which + 12]:
            short = name.split('(')[0].replace('(anonymous namespace)::', '')[-60:]
            print('%-60s start %9.1f us  end %9.1f us' % (short, (s - t0) / 1e3, (e - t0) / 1e3))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--db', default='')
    args = ap.parse_args()
    if args.db:
        return timeline(args.db)
    import numpy as np
    import torch
    from interdiff_amd import synthetic as syn
    from interdiff_amd.smpl import SMPL_Layer
    from interdiff_amd.objprojector import ObjProjector
    from interdiff_amd.correction import HipCorrection
    torch.set_grad_enabled(False)
    dev, B, T, P, past = 'cuda', 16, 100, 2048, 10
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'correction_ckpt.npz'))
    smpl = SMPL_Layer(syn.smplh_model(7), device=dev)
    bt = {k: (torch.from_numpy(v).to(dev) if isinstance(v, np.ndarray) else v) for k, v in syn.make_clip_batch(seed=233, B=B, T=T, past_len=past, n_points=P).items()}
    pad = list(range(past)) + [past - 1] * (T - past)
    y = dict(inpainted_motion=bt['gt'], hand_pose=bt['hand_pose'][pad].contiguous(), beta=bt['beta'], obj_points=bt['obj_points'])
    x0 = bt['gt'] + 0.05 * torch.randn_like(bt['gt'])
    table = torch.zeros(1000, 4, device=dev)
    table[:, 3] = 0.25
    state = torch.zeros(8, dtype=torch.int64, device=dev)
    for tune in (0, 2):
        corr = HipCorrection(smpl, ObjProjector({k: z[k] for k in z.files}, T=T, past_len=past, device=dev), n_points=P, past_len=past, device=dev)
        corr.ctx.tune = tune
        ws = corr.workspace_for(B, T)
        x = x0.clone()
        corr.apply_dev(x, table, state, y, ws)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            corr.apply_dev(x, table, state, y, ws)
        for _ in range(3):
            g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        print('tune %d (%s): %.1f us per replayed hook call' % (tune, 'predictor after the scan' if tune else 'product form', 1e3 * e0.elapsed_time(e1) / 10))


if __name__ == '__main__':
    main()
