"""Does splitting the clip batch over S streams (independent clips -> independent kernel chains) raise throughput?
   python tools/stream_probe.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from interdiff_amd import synthetic as syn, _lib
from interdiff_amd.mdm import MDM

torch.set_grad_enabled(False)
dev = 'cuda'
B, T = 16, 100
sd = syn.mdm_state_dict(233)
g = torch.Generator().manual_seed(1)
X = torch.randn(B, 1, 144, T, generator=g).to(dev)
COND = torch.randn(10, B, 256, generator=g).to(dev)
for S in (1, 2, 4, 8):
    bs = B // S
    models = [MDM(sd, device=dev) for _ in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    xs = [X[i * bs:(i + 1) * bs].contiguous() for i in range(S)]
    conds = [COND[:, i * bs:(i + 1) * bs].contiguous() for i in range(S)]
    ts = torch.full((bs,), 500, dtype=torch.int64, device=dev)
    outs = [torch.empty_like(x) for x in xs]
    def run(n):
        for _ in range(n):
            for m, st, x, c, o in zip(models, streams, xs, conds, outs):
                with torch.cuda.stream(st):
                    m(x, ts, y={'cond': c}, out=o)
    run(5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 100
    run(n)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    print('S=%d streams x B=%d: %.1f us per full-batch forward (host enqueue %.1f us)' % (S, bs, 1e6 * t / n, 1e6 * t_host / n))
