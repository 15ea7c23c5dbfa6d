"""Where do the 8-token and the 16-token form of the eight-wave row block differ?  (not product code)  One denoiser forward at B x T under rb_tokens = 8 and = 16 on the same
input, exact arithmetic elsewhere identical; prints which token positions / channels differ and by how much, and the same with only layer subsets active is left to the reader."""
import sys
import os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import fixtures as fx                                                   # noqa: E402
from interdiff_amd.mdm import MDM                                                  # noqa: E402

torch.set_grad_enabled(False)
B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (16, 100)
x, ts, cond = fx.mdm_inputs(B, T)
m = MDM(fx.mdm_weights(), device='cuda')
out = {}
for tok in (8, 16, 8):
    m.w.rb_tokens = tok
    o = m(x.cuda(), ts.cuda(), y={'cond': cond.cuda()}).clone()
    if tok in out:
        print('rb_tokens %d twice: identical %s' % (tok, torch.equal(out[tok], o)))
    out[tok] = o
d = (out[8] - out[16]).abs()
print('B=%d T=%d: max |delta| %.3e of max |x0| %.3e; elements differing %d of %d' % (B, T, d.max().item(), out[16].abs().max().item(), int((d > 0).sum()), d.numel()))
per_t = (d.squeeze(1) > 0).any(dim=1)                       # [B, T]
print('token positions that differ, per clip (first 4 clips):')
for b in range(min(B, 4)):
    print('  clip %d:' % b, [t for t in range(T) if per_t[b, t]])
print('positions t mod 16 histogram:', torch.bincount(torch.nonzero(per_t)[:, 1] % 16, minlength=16).tolist())
print('positions t mod 8 histogram: ', torch.bincount(torch.nonzero(per_t)[:, 1] % 8, minlength=8).tolist())
