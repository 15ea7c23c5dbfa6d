"""Does a co-resident workgroup's LDS stay intact while feed-forward kernels run on another stream?  (not product code)"""
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import fixtures as fx                                                   # noqa: E402
from interdiff_amd import _lib                                                     # noqa: E402
from interdiff_amd.mdm import MDM, ffn_parts                                       # noqa: E402

torch.set_grad_enabled(False)
lib = _lib.load()
mdm = MDM(fx.mdm_weights(), device='cuda')
A, Bs = torch.cuda.Stream(), torch.cuda.Stream()
x2 = torch.randn(800, 256, device='cuda')
parts = torch.empty(5, 800, 256, device='cuda')
for kind in (None, ('exact', 16), ('split', 16), ('split', 32), ('exact', 32), ('split', 16)):
    out = torch.zeros(4 + 4000, dtype=torch.int32, device='cuda')
    torch.cuda.synchronize()
    with torch.cuda.stream(Bs):
        for i in range(400 if kind else 0):
            mdm.ffn_math, mdm.ffn_rows = kind
            ffn_parts(mdm, x2, i % 8, out=parts)
    with torch.cuda.stream(A):
        for rep in range(6):
            _lib.check(lib.interdiff_debug_lds_sentinel(_lib.dptr(out), 1600, 60, _lib.stream()), 'sentinel')
    torch.cuda.synchronize()
    mdm.ffn_rows = 0
    o = out.cpu().numpy().view(np.uint32)
    n = int(o[0])
    print('load', kind, 'foreign writes seen:', n, flush=True)
    rec = o[4:4 + 4 * min(n, 1000)].reshape(-1, 4)
    for r in rec[:24]:
        print('   wg %d word %d (byte 0x%x) value 0x%08x pass %d' % (r[0], r[1], 4 * r[1], r[2], r[3]))
    if n:
        words = rec[:, 1]
        print('   words min %d max %d ; distinct values %d ; distinct wgs %d' % (words.min(), words.max(), len(set(rec[:, 2].tolist())), len(set(rec[:, 0].tolist()))))
