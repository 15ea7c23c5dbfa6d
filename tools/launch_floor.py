"""Cost of a kernel boundary inside a replayed hipGraph on one MI355X (not product code): N tiny dependent kernels per graph.
    python tools/launch_floor.py"""
import torch
x = torch.zeros(64, device='cuda')
big = torch.zeros(1600 * 256, device='cuda')
for name, fn in (('tiny (64 floats)', lambda: x.add_(1.0)), ('1.6 MB in place', lambda: big.add_(1.0))):
    for n in (1, 8, 24, 96):
        side, g = torch.cuda.Stream(), torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            fn()
            side.synchronize()
            with torch.cuda.graph(g, stream=side):
                for _ in range(n):
                    fn()
            g.replay()
            side.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side)
            for _ in range(50):
                g.replay()
            e1.record(side)
            e1.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / 50
        print('%-18s %3d kernels per graph: %8.1f us per replay, %6.2f us per kernel' % (name, n, us, us / n))
