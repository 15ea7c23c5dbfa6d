"""Is the two-chain form's concurrency stable?  (not product code)  Whole B = 32 samples with correction, the sampler's graphs re-captured
several times in one process; run the script several times for process-to-process spread."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                                      # noqa: E402
from interdiff_amd.diffusion import create_gaussian_diffusion                     # noqa: E402


def main():
    torch.set_grad_enabled(False)
    dev = torch.device('cuda:0')
    bench.B_PER_GPU = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    model, corr, bt, y, _ = bench.build_world(dev, 0)
    out = []
    for rep in range(5):
        model.__dict__.pop('_graph_cache', None)
        diff = create_gaussian_diffusion('cosine', bench.STEPS)
        bench.run_steps(diff, model, None, bt, y, 120, seed=7)
        ts = []
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            bench.run_steps(diff, model, None, bt, y, 1000, seed=3)
            torch.cuda.synchronize()
            ts.append(round(time.perf_counter() - t0, 4))
        out.append(ts)
        extra = [torch.cuda.Stream(dev) for _ in range(rep + 1)]          # perturb the stream -> hardware-queue round robin
        for s_ in extra:
            with torch.cuda.stream(s_):
                torch.zeros(1, device=dev)
    print(json.dumps(dict(B=bench.B_PER_GPU, no_correction_s=out)), flush=True)


if __name__ == '__main__':
    main()
