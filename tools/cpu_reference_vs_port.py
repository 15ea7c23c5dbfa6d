"""The reference's own CPU time next to the oracle's ("port"), build container only.

    python tools/cpu_reference_vs_port.py [threads]      ->  profiles/r06_cpu_reference_vs_port.json

`bench.py`'s cpu_baseline times oracle/ on the GPU box's host cores (kind "port"): /root/reference cannot travel there.  This script answers the
question the port leaves open -- is the port slower or faster than the source it stands in for? -- by timing BOTH on the same cores, same inputs, same
thread count, interleaved in one process:
  * plain denoising steps at BASELINE config #2's shape (B=16, T=100): the reference's MDM.forward + GaussianDiffusion.p_sample (imported read-only from
    /root/reference through tests/golden/refshim.py) against oracle/denoiser.mdm_forward + the posterior update bench.py times;
  * one correction call, eval_smpl_short.denoised_fn at t = 250 on ONE clip (T=100, P=2048), against oracle/correction.denoised_fn on the same clip.
The ratios (port seconds / reference seconds) are what bench.py reports as cpu_baseline.port_vs_reference (recorded, not measured on the GPU box).
"""
import json
import os
import sys
import time
from argparse import Namespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import refshim                                            # noqa: E402
import make_golden as mg                                  # noqa: E402
from tests import fixtures as fx                          # noqa: E402
from oracle import diffusion as odf, denoiser as oden, correction as ocor       # noqa: E402


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 8)
    torch.set_num_threads(threads)
    torch.set_grad_enabled(False)
    refshim.install()
    T, B, P = fx.FULL_SHAPE
    past = fx.PAST
    batch, noise, _ = fx.full_inputs()
    sd = fx.mdm_weights()
    net = mg.ref_mdm()
    diff = mg.ref_diffusion(1000)
    sched = odf.make_schedule(1000)
    cond = batch['cond']
    mask = torch.ones_like(batch['gt'], dtype=torch.bool)
    mask[..., past:] = False
    y = dict(cond=cond, inpainted_motion=batch['gt'], inpainting_mask=mask)
    ts = torch.full((B,), 999, dtype=torch.int64)
    x = noise.clone()

    def ref_plain():                                       # gaussian_diffusion.py p_sample: model forward, inpainting, posterior mean + noise
        return diff.p_sample(net, x, ts, clip_denoised=False, model_kwargs={'y': y})['sample']

    def port_plain():                                      # what bench.py's cpu_baseline times
        x0 = oden.mdm_forward(sd, x, ts, cond)
        x0 = x0 * (~mask) + batch['gt'] * mask
        return float(sched['posterior_mean_coef1'][999]) * x0 + float(sched['posterior_mean_coef2'][999]) * x + 0.1 * torch.randn_like(x)

    def timeit(f, n):
        t0 = time.perf_counter()
        for _ in range(n):
            f()
        return (time.perf_counter() - t0) / n
    for f in (ref_plain, port_plain):
        for _ in range(2):
            f()
    t_ref, t_port = [], []
    for _ in range(3):                                     # interleaved: drift of the box hits both alike
        t_ref.append(timeit(ref_plain, 3))
        t_port.append(timeit(port_plain, 3))
    err = float((ref_plain() - 0).abs().max())            # (finite check only; parity is tests/test_oracle_golden.py's business)
    assert np.isfinite(err)

    # one correction call on one clip
    ev = refshim.load('eval_smpl_short')
    ev.args = Namespace(smpl_dim=132, past_len=past)
    L = mg.ref_smpl(fx.smpl_model())

    class Holder:
        pass
    om = Holder()
    om.model = mg.ref_objproj(T)
    yk = fx.model_kwargs_y(dict(batch, noise=noise), T) if hasattr(fx, 'model_kwargs_y') else None
    sl = slice(0, 1)
    ysub = {k: (v[:, sl] if k in ('cond', 'hand_pose', 'beta') else v[sl]) if isinstance(v, torch.Tensor) else v for k, v in yk.items()}
    xin = batch['gt'][sl] + 0.05 * torch.randn(batch['gt'][sl].shape, generator=torch.Generator().manual_seed(5))
    tt = torch.full((1,), 250, dtype=torch.int64)
    yref = dict(ysub, smpl=L, obj_model=om)
    yport = dict(ysub, smpl={k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in fx.smpl_model().items()},
                 obj_model=fx.objproj_weights())
    c_ref, c_port = [], []
    for _ in range(2):
        t0 = time.perf_counter()
        ev.denoised_fn(xin.clone(), tt, {'y': yref})
        c_ref.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        ocor.denoised_fn(xin.clone(), tt, {'y': yport}, past_len=past)
        c_port.append(time.perf_counter() - t0)
    out = dict(threads=threads, shape=dict(B=B, T=T, P=P),
               plain_step_s=dict(reference=min(t_ref), port=min(t_port), all_reference=t_ref, all_port=t_port),
               correction_call_one_clip_s=dict(reference=min(c_ref), port=min(c_port), all_reference=c_ref, all_port=c_port),
               port_vs_reference=dict(plain_step=min(t_port) / min(t_ref), correction_call=min(c_port) / min(c_ref)),
               note='seconds; port_vs_reference > 1 means the oracle (what bench.py times as cpu_baseline, kind "port") is SLOWER than the reference source it restates, '
                    'i.e. a GPU/CPU ratio quoted against the port overstates the ratio against the reference by that factor')
    mix = lambda tp, tc: 1000.0 / (989 * tp + 11 * (tp + 16 * tc))
    out['blended_steps_per_s_16_clips'] = dict(reference=mix(min(t_ref), min(c_ref)), port=mix(min(t_port), min(c_port)))
    path = os.path.join(ROOT, 'profiles', 'r06_cpu_reference_vs_port.json')
    json.dump(out, open(path, 'w'), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
