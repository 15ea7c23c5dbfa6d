"""Round-6 experiment, kept for the record (profiles/r06_hook_overlap.txt): written when the hook ran the predictor on a library-owned SIDE STREAM (ctx.tune 0 then; the shipped
library has no side stream any more -- tune 0 is now the one-launch form -- so re-running this compares that form).
Does using the hook's side stream change how the PLAIN steps run afterwards?  (not product code)
    python tools/side_stream_probe.py
Times 490 plain steps (no hook; graphs of 49) before and after one overlapped hook call, one one-stream hook call, and a pure torch fork / join on a fresh stream."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                            # noqa: E402
from interdiff_amd.diffusion import create_gaussian_diffusion   # noqa: E402


def main():
    torch.set_grad_enabled(False)
    dev = torch.device('cuda', 0)
    model, corr, bt, y, _ = bench.build_world(dev, 0)
    diff = create_gaussian_diffusion('cosine', bench.STEPS)
    bench.run_steps(diff, model, None, bt, y, 57, seed=7)
    corr.ctx.tune = 2
    corr.apply(bt['noise'].clone(), 500, y)
    torch.cuda.synchronize()

    def plain(label):
        ts = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            bench.run_steps(diff, model, None, bt, y, 490, seed=7)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 490 * 1e3)
        print('%-70s plain step %s ms' % (label, ' '.join('%.4f' % t for t in ts)), flush=True)

    plain('start (hook called once on one stream)')
    corr.ctx.tune = 2
    corr.apply(bt['noise'].clone(), 500, y)
    plain('after another one-stream hook call')
    st = torch.cuda.Stream()
    cur = torch.cuda.current_stream()
    st.wait_stream(cur)
    with torch.cuda.stream(st):
        z = torch.zeros(1024, device=dev) + 1
    cur.wait_stream(st)
    torch.cuda.synchronize()
    plain('after a torch fork / join on a fresh stream')
    corr.ctx.tune = 0
    corr.apply(bt['noise'].clone(), 500, y)
    torch.cuda.synchronize()
    plain('after ONE overlapped hook call (library side stream)')
    corr.ctx.tune = 2
    corr.apply(bt['noise'].clone(), 500, y)
    torch.cuda.synchronize()
    plain('after a one-stream hook call again')


if __name__ == '__main__' and 'host' not in sys.argv[1:]:
    main()


def host_block_probe():
    """Host time of a hook call issued while the stream still has ~85 ms of queued plain steps: a call that blocks the host shows the queue's length."""
    torch.set_grad_enabled(False)
    dev = torch.device('cuda', 0)
    model, corr, bt, y, _ = bench.build_world(dev, 0)
    diff = create_gaussian_diffusion('cosine', bench.STEPS)
    bench.run_steps(diff, model, None, bt, y, 57, seed=7)
    for tune in (2, 0):
        corr.ctx.tune = tune
        corr.apply(bt['noise'].clone(), 500, y)
    torch.cuda.synchronize()
    x = bt['noise'].clone()
    for tune in (2, 0, 2, 0):
        corr.ctx.tune = tune
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bench.run_steps(diff, model, None, bt, y, 490, seed=7)        # ~85 ms of GPU work, returns when enqueued (the final clone is asynchronous too)
        t1 = time.perf_counter()
        corr.apply(x, 500, y)
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        print('tune %d: enqueue 490 plain steps %.1f ms host | hook call %.2f ms host | drain %.1f ms' % (tune, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2)), flush=True)


if __name__ == '__main__' and 'host' in sys.argv[1:]:
    host_block_probe()
