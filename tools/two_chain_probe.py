"""Do two dependent kernel chains on two streams overlap on one MI355X?  (not product code)
Chain = N fused-FFN launches (each depends on the previous through its buffer).  One chain at M=1600 vs two concurrent chains at
M=800 (half the clips each), eager on two streams and as a two-branch hipGraph.
    python tools/two_chain_probe.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interdiff_amd import synthetic as syn, _lib
from interdiff_amd.mdm import MDM, ffn_parts
torch.set_grad_enabled(False)
dev = 'cuda'
model = MDM(syn.mdm_state_dict(233), device=dev)
N = 48


def chain(M, stream):
    x2 = torch.randn(M, 256, device=dev)
    parts = torch.empty(_lib.FFN_SLICES, M, 256, device=dev)
    def run():
        with torch.cuda.stream(stream):
            for i in range(N):
                ffn_parts(model, x2, i % 8, out=parts)
    return run


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / reps


s0, s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
one = chain(1600, s0)
a, b = chain(800, s1), chain(800, s2)
print('eager: one chain M=1600: %.1f us per launch; two chains M=800 on two streams: %.1f us per launch pair; one chain M=800 alone: %.1f'
      % (timed(one) / N, timed(lambda: (a(), b())) / N, timed(a) / N))
# graphs
def graph_of(fns):
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            cur = torch.cuda.current_stream()
            evs = []
            for f, st in fns:
                st.wait_stream(cur)
                f()
                evs.append(st)
            for st in evs:
                cur.wait_stream(st)
    return g
g1 = graph_of([(one, s0)])
g2 = graph_of([(a, s1), (b, s2)])
g3 = graph_of([(a, s1)])
for name, g in (('one chain M=1600', g1), ('two chains M=800', g2), ('one chain M=800', g3)):
    print('graph: %-18s %.1f us per launch (pair)' % (name, timed(g.replay) / N))
