"""How the FFN GEMMs scale with M around the benchmark's M = 1600 (not product code): if the time is flat between tile-count
multiples of the 256 CUs, the quantisation of tiles over CUs is what a better schedule could win back.
    python tools/msweep.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from interdiff_amd.mdm import linear          # noqa: E402


def t_us(fn, reps=200):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    g = torch.Generator().manual_seed(0)
    w1, b1 = torch.randn(1024, 256, generator=g).cuda() * 0.05, torch.randn(1024, generator=g).cuda()
    w2, b2 = torch.randn(256, 1024, generator=g).cuda() * 0.05, torch.randn(256, generator=g).cuda()
    print('%6s %10s %10s %12s %12s' % ('M', 'ffn1 us', 'ffn2 us', 'ffn1 TF/s', 'ffn2 TF/s'))
    for M in (768, 1024, 1280, 1536, 1600, 1792, 2048, 2560, 3072, 3200, 4096, 6400):
        x = torch.randn(M, 256, generator=g).cuda()
        h = torch.randn(M, 1024, generator=g).cuda()
        o1, o2 = torch.empty(M, 1024, device='cuda'), torch.empty(M, 256, device='cuda')
        a = t_us(lambda: linear(x, w1, b1, gelu=True, out=o1, cfg=7))
        b = t_us(lambda: linear(h, w2, b2, residual=x, out=o2, cfg=6))
        fl = 2.0 * M * 256 * 1024
        print('%6d %10.2f %10.2f %12.1f %12.1f' % (M, a, b, fl / a / 1e6, fl / b / 1e6))


if __name__ == '__main__':
    main()
