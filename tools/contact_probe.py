"""Contact-scan probe on one MI355X (not product code): the hook's nearest-vertex scan (interdiff_contact_nn) on the clips
tools/corr_bench.py times, with the kernel's own statistics -- blocks scored / box tests per wave, thread-0 clock cycles per phase.
    python tools/contact_probe.py [--B 16] [--T 100] [--coherent]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from interdiff_amd import synthetic as syn, transforms as tr                     # noqa: E402
from interdiff_amd.smpl import SMPL_Layer                             # noqa: E402
from interdiff_amd.objprojector import ObjProjector                   # noqa: E402
from interdiff_amd.correction import HipCorrection                    # noqa: E402

PHASES = ('records->LDS', 'boxes+markers', 'own tasks (wave 0)', 'waiting for other waves', 'reductions')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--B', type=int, default=16)
    ap.add_argument('--T', type=int, default=100)
    ap.add_argument('--coherent', action='store_true')
    ap.add_argument('--noise', type=float, default=0.05)
    args = ap.parse_args()
    torch.set_grad_enabled(False)
    dev, B, T, P, past = 'cuda', args.B, args.T, 2048, 10
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'correction_ckpt.npz'))
    smpl = SMPL_Layer(syn.smplh_model(7, **({'coherent': True} if args.coherent else {})), device=dev)
    bt = {k: (torch.from_numpy(v).to(dev) if isinstance(v, np.ndarray) else v) for k, v in syn.make_clip_batch(seed=233, B=B, T=T, past_len=past, n_points=P).items()}
    x = (bt['gt'] + args.noise * torch.randn_like(bt['gt']))[:, 0].permute(2, 0, 1).contiguous()       # [T,B,144]
    pad = list(range(past)) + [past - 1] * (T - past)
    body = tr.rotation_6d_to_axis_angle(x[..., :132].reshape(T, B, 22, 6)).reshape(T, B, 66)
    pose = torch.cat([body, bt['hand_pose'][pad]], dim=2).reshape(T * B, 156)
    verts = smpl(pose, th_betas=bt['beta'].reshape(T * B, 10), th_trans=x[..., 132:135].reshape(T * B, 3))[0].reshape(T, B, -1, 3)
    objR, objT = tr.rotation_6d_to_matrix(x[..., 135:141].contiguous()), x[..., 141:144].contiguous()
    out = {}
    for name, so in (('scan_order', True), ('identity_order', False)):
        corr = HipCorrection(smpl, ObjProjector({k: z[k] for k in z.files}, T=T, past_len=past, device=dev), n_points=P, past_len=past, device=dev,
                             scan_order=so)
        for _ in range(2):
            o2h, idx, st = corr.contact_nn(verts, bt['obj_points'], objR, objT, want_stats=True)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(5):
            corr.contact_nn(verts, bt['obj_points'], objR, objT)
        ev1.record()
        torch.cuda.synchronize()
        scored, blocks, tests, nwg = st[:4]
        out[name] = dict(ms_per_call_all_frames=round(ev0.elapsed_time(ev1) / 5, 3), blocks_scored_frac=round(scored / max(blocks, 1), 4),
                         box_tests_per_block=round(tests / max(blocks, 1), 4), workgroups=nwg,
                         kcycles_per_workgroup={p: round(c / max(nwg, 1) / 1e3, 1) for p, c in zip(PHASES, st[4:])})
        if name == 'scan_order':
            keep = idx
        else:
            out['indices_identical'] = bool(torch.equal(idx, keep))
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
