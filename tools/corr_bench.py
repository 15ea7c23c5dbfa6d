"""Correction-hook timing on one MI355X (not product code): the fused denoised_fn call at B clips x T frames, per-kernel event
times for the scan-order and the brute-force (identity-order) contact scan (the scan's figure includes the predictor's stacks, which ride in its launch; `objproj` is the pick
kernel), then the whole call in the product form and with the predictor launched after the scan (idf_correction_ctx.tune = 0 / 2).
    python tools/corr_bench.py [--B 16] [--T 100]"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from interdiff_amd import synthetic as syn, _lib                     # noqa: E402
from interdiff_amd.smpl import SMPL_Layer                             # noqa: E402
from interdiff_amd.objprojector import ObjProjector                   # noqa: E402
from interdiff_amd.correction import HipCorrection                    # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--B', type=int, default=16)
    ap.add_argument('--T', type=int, default=100)
    ap.add_argument('--only', default='', help='scan_order | identity_order: run one variant (counter passes)')
    args = ap.parse_args()
    torch.set_grad_enabled(False)
    lib = _lib.load()
    dev, B, T, P, past = 'cuda', args.B, args.T, 2048, 10
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'correction_ckpt.npz'))
    smpl = SMPL_Layer(syn.smplh_model(7), device=dev)
    mk = lambda so: HipCorrection(smpl, ObjProjector({k: z[k] for k in z.files}, T=T, past_len=past, device=dev), n_points=P, past_len=past, device=dev,
                                  scan_order=so)
    bt = {k: (torch.from_numpy(v).to(dev) if isinstance(v, np.ndarray) else v) for k, v in syn.make_clip_batch(seed=233, B=B, T=T, past_len=past, n_points=P).items()}
    pad = list(range(past)) + [past - 1] * (T - past)
    y = dict(inpainted_motion=bt['gt'], hand_pose=bt['hand_pose'][pad].contiguous(), beta=bt['beta'], obj_points=bt['obj_points'])
    x = bt['gt'] + 0.05 * torch.randn_like(bt['gt'])
    out = {}
    ref = None
    for name, so, tune in (('scan_order', True, 0), ('identity_order', False, 0)):
        if args.only and name != args.only:
            continue
        corr = mk(so)
        corr.ctx.tune = tune
        corr.debug = {}
        got = corr.apply(x.clone(), 250, y)
        dec = (corr.debug['condition'].clone(), corr.debug['contact'].clone())
        if ref is None:
            ref = (got, dec)
        same = bool(torch.equal(got, ref[0]) and torch.equal(dec[0], ref[1][0]) and torch.equal(dec[1], ref[1][1]))
        corr.debug = None
        for _ in range(3):
            corr.apply(x.clone(), 250, y)
        torch.cuda.synchronize()
        _lib.check(lib.interdiff_profile_begin(10000))
        for _ in range(10):
            corr.apply(x.clone(), 250, y)
        ms = (C.c_double * len(_lib.KERNEL_KINDS))()
        cnt = (C.c_int64 * len(_lib.KERNEL_KINDS))()
        _lib.check(lib.interdiff_profile_end(ms, cnt))
        out[name] = dict(identical_to_first=same, **{k: round(1e3 * ms[i] / cnt[i], 1) for i, k in enumerate(_lib.KERNEL_KINDS) if cnt[i]})
        # the whole call, as the sampler issues it (per-kernel profile off): the predictor's stacks inside the scan's launch (product) against scan, then one-launch predictor
        for label, tn in (('hook_us', 0), ('hook_us_predictor_after_the_scan', 2)):
            corr.ctx.tune = tn
            for _ in range(3):
                corr.apply(x.clone(), 250, y)
            xs = [x.clone() for _ in range(10)]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for xx in xs:
                corr.apply(xx, 250, y)
            e1.record()
            torch.cuda.synchronize()
            out[name][label] = round(1e3 * e0.elapsed_time(e1) / len(xs), 1)
        corr.ctx.tune = tune
    print(json.dumps(out))


if __name__ == '__main__':
    main()
