"""Where do the staggered chains differ from the joined form?  (not product code)"""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import fixtures as fx                                                   # noqa: E402
from tests.test_hip_parity import make_correction, dev                             # noqa: E402
from interdiff_amd.mdm import MDM                                                  # noqa: E402
from interdiff_amd.smpl import SMPL_Layer                                          # noqa: E402
from interdiff_amd.diffusion import create_gaussian_diffusion                      # noqa: E402

torch.set_grad_enabled(False)
mdm = MDM(fx.mdm_weights(), device='cuda')
smpl = SMPL_Layer(fx.smpl_model(), device='cuda')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T, P = fx.TIMED_T, fx.TIMED_P
bt, y = fx.timed_inputs(B)
y, x_t = dev(y), bt['noise'].to('cuda')
corr = make_correction(smpl, T, P)
for math in ('exact', 'split'):
    mdm.ffn_math = math
    diff = create_gaussian_diffusion('cosine', 1000)
    dumps = list(range(0, 120, 10)) + [59, 60, 61, 109, 110, 111, 119]
    dumps = sorted(set(dumps))
    run = lambda **kw: diff.p_sample_loop(mdm, tuple(x_t.shape), noise=x_t, clip_denoised=False, model_kwargs={'y': y}, denoised_fn=corr,
                                          n_steps=120, first_t=560, seed=99, dump_steps=dumps, **kw)
    ref = run()
    for rep in range(4):
        diff.stagger_steps = 7 if rep < 3 else 0
        got = run()
        diff.stagger_steps = 0
        msg = []
        for d, a, b in zip(dumps, ref, got):
            nb = (a != b)
            if nb.any():
                idx = nb.nonzero()
                msg.append('it%d: n=%d clips=%s chans=[%d..%d] maxdiff=%.3g' % (d, int(nb.sum()), sorted(set(idx[:, 0].tolist()))[:8], int(idx[:, 2].min()), int(idx[:, 2].max()),
                                                                                 float((a - b).abs().max())))
        print(math, 'stagger' if rep < 3 else 'joined-again', rep, msg[:3] if msg else 'IDENTICAL', flush=True)
