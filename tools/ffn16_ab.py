"""32-row vs 16-row vs 64-row fused feed-forward kernel (csrc/ffn.h) by token count, on one box in one process (not product code):
per-launch time from a layer-cycling hipGraph (bench.time_dominant_kernel's recipe) and the error of both against torch fp64."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                                      # noqa: E402
from interdiff_amd import _lib                                                    # noqa: E402
from interdiff_amd.mdm import ffn_parts                                           # noqa: E402


def time_ffn(model, dev, N, per_graph=48, reps=5):
    g = torch.Generator().manual_seed(5)
    x2 = [torch.randn(N, 256, generator=g).to(dev) for _ in range(2)]
    parts = [torch.empty(_lib.FFN_SLICES, N, 256, device=dev) for _ in range(2)]
    for i in range(16):
        ffn_parts(model, x2[i & 1], i % 8, out=parts[i & 1])
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=dev)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            for i in range(per_graph):
                ffn_parts(model, x2[i & 1], i % 8, out=parts[i & 1])
        graph.replay()
        side.synchronize()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side)
            for _ in range(reps):
                graph.replay()
            e1.record(side)
            e1.synchronize()
            ts.append(1e3 * e0.elapsed_time(e1) / (reps * per_graph))
    return round(sum(ts) / 3, 3)


def main():
    torch.set_grad_enabled(False)
    dev = torch.device('cuda:0')
    model, corr, bt, y, _ = bench.build_world(dev, 0)
    out = {}
    for N in (100, 800, 1200, 1600, 1700, 2400, 3200, 3300, 6400):
        row = {}
        res = {}
        for name, v in (('rows32', 32), ('rows16', 16), ('rows64', 64)):
            model.ffn_rows = v
            row[name + '_us'] = time_ffn(model, dev, N)
            g = torch.Generator().manual_seed(11)
            x2 = torch.randn(N, 256, generator=g).to(dev)
            res[name] = ffn_parts(model, x2, 3).sum(0).cpu()
        row['max_abs_diff_16_vs_32'] = float((res['rows16'] - res['rows32']).abs().max())
        row['rows64_equals_rows16'] = bool(torch.equal(res['rows16'], res['rows64']))
        row['scale'] = float(res['rows32'].abs().max())
        out[N] = row
        print(N, json.dumps(row), flush=True)
    model.ffn_rows = 0


if __name__ == '__main__':
    main()
