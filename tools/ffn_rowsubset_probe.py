"""Is the feed-forward block row-local?  parts(x2[:m]) vs parts(x2)[:, :m] for both arithmetics and every row tile (not product code)."""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import fixtures as fx                                                   # noqa: E402
from interdiff_amd.mdm import MDM, ffn_parts                                       # noqa: E402

torch.set_grad_enabled(False)
mdm = MDM(fx.mdm_weights(), device='cuda')
g = torch.Generator().manual_seed(1)
x2 = torch.randn(48, 256, generator=g).cuda()
for math in ('exact', 'split'):
    mdm.ffn_math = math
    for rows in (16, 32, 64):
        mdm.ffn_rows = rows
        full = ffn_parts(mdm, x2, 1)
        for m in (1, 8, 16, 24, 40):
            sub = ffn_parts(mdm, x2[:m].contiguous(), 1)
            d = (sub - full[:, :m]).abs()
            bad = (d > 0).nonzero()
            print(math, rows, m, 'max diff %.3g' % d.max().item(), 'n_bad', bad.shape[0], bad[:4].tolist(), flush=True)
        again = ffn_parts(mdm, x2, 1)
        print(math, rows, 'repeat equal', torch.equal(again, full), flush=True)
