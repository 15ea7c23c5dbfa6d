"""Experiment (not product code): one 16-clip sampling loop vs two concurrent 8-clip loops on two streams of ONE MI355X, each loop
driven by its own Python thread with its own denoiser handle, correction hook and captured graphs.
    python tools/two_sampler_probe.py"""
import os, sys, time, threading
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from interdiff_amd.diffusion import create_gaussian_diffusion
torch.set_grad_enabled(False)
dev = torch.device('cuda:0')


def world(B, rank):
    bench.B_PER_GPU = B
    model, corr, bt, y, _ = bench.build_world(dev, rank)
    diff = create_gaussian_diffusion('cosine', bench.STEPS)
    return diff, model, corr, bt, y


def sample(w, stream, n=1, corr=True):
    diff, model, c, bt, y = w
    with torch.cuda.stream(stream):
        for i in range(n):
            out = bench.run_steps(diff, model, c if corr else None, bt, y, bench.STEPS, seed=5 + i)
    return out


for corr in (True, False):
    w16 = world(16, 0)
    s0 = torch.cuda.Stream()
    sample(w16, s0, corr=corr); torch.cuda.synchronize()
    t0 = time.perf_counter(); sample(w16, s0, corr=corr); torch.cuda.synchronize()
    t16 = time.perf_counter() - t0
    del w16
    wa, wb = world(8, 1), world(8, 2)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for w, s in ((wa, sa), (wb, sb)):
        sample(w, s, corr=corr)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); sample(wa, sa, corr=corr); torch.cuda.synchronize()
    t8 = time.perf_counter() - t0
    th = [threading.Thread(target=sample, args=(w, s, 1, corr)) for w, s in ((wa, sa), (wb, sb))]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    t2 = time.perf_counter() - t0
    print('correction=%s: one 16-clip loop %.1f ms; one 8-clip loop alone %.1f ms; two 8-clip loops concurrently %.1f ms (%.3f ms/step for 16 clips)'
          % (corr, 1e3 * t16, 1e3 * t8, 1e3 * t2, 1e3 * t2 / bench.STEPS))
    del wa, wb
