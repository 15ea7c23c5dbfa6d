"""Is the correction hook's output independent of what else runs on the GPU?  (not product code)
The hook on stream A, a loop of feed-forward launches (or nothing) on stream B, repeated; every repetition must give the same bits."""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import fixtures as fx                                                   # noqa: E402
from tests.test_hip_parity import make_correction, dev                             # noqa: E402
from interdiff_amd.mdm import MDM, ffn_parts                                       # noqa: E402
from interdiff_amd.smpl import SMPL_Layer                                          # noqa: E402

torch.set_grad_enabled(False)
mdm = MDM(fx.mdm_weights(), device='cuda')
smpl = SMPL_Layer(fx.smpl_model(), device='cuda')
B = 8
T, P = fx.TIMED_T, fx.TIMED_P
bt, y = fx.timed_inputs(B)
y = dev(y)
corr = make_correction(smpl, T, P)
x0 = (bt['gt'] + 0.05 * bt['noise']).to('cuda')
A, Bs = torch.cuda.Stream(), torch.cuda.Stream()
x2 = torch.randn(800, 256, device='cuda')
parts = torch.empty(5, 800, 256, device='cuda')


def hook_once(load):
    corr.debug = {}
    x = x0.clone()
    torch.cuda.synchronize()
    with torch.cuda.stream(Bs):
        for i in range(400 if load else 0):
            mdm.ffn_math = load
            ffn_parts(mdm, x2, i % 8, out=parts)
    with torch.cuda.stream(A):
        corr.apply(x, 450, y)
    torch.cuda.synchronize()
    d = corr.debug
    return x, d['condition'].clone(), d['contact'].clone(), d['distance'].clone(), d['loss'].clone()


ref = hook_once(None)
for load in (None, 'exact', 'split', 'split', 'exact', 'split'):
    for rep in range(4):
        got = hook_once(load)
        names = ('x', 'condition', 'contact', 'distance', 'loss')
        bad = [n for n, a, b in zip(names, ref, got) if not torch.equal(a, b)]
        detail = ''
        if 'x' in bad:
            nb = (ref[0] != got[0]).nonzero()
            detail = ' clips=%s chans=[%d..%d] n=%d maxdiff=%.3g' % (sorted(set(nb[:, 0].tolist())), int(nb[:, 2].min()), int(nb[:, 2].max()), nb.shape[0],
                                                                      float((ref[0] - got[0]).abs().max()))
        print('load', load, 'rep', rep, 'differs:' if bad else 'identical', bad, detail, flush=True)
