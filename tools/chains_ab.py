"""One chain vs two chains per sample under the current kernels, same process (not product code)."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                                      # noqa: E402
from interdiff_amd.diffusion import create_gaussian_diffusion                     # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device('cuda:0')
for B in [int(a) for a in sys.argv[1:]] or [16]:
    bench.B_PER_GPU = B
    model, corr, bt, y, _ = bench.build_world(dev, 0)
    for rep in range(2):
        for split in (True, False):
            diff = create_gaussian_diffusion('cosine', bench.STEPS)
            diff.split_chains = split
            model.__dict__.pop('_graph_cache', None)
            out = dict(B=B, two_chains=split)
            for name, c in (('no_correction', None), ('correction', corr)):
                bench.run_steps(diff, model, c, bt, y, 1000, seed=3)
                ts = []
                for _ in range(3):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    bench.run_steps(diff, model, c, bt, y, 1000, seed=3)
                    torch.cuda.synchronize()
                    ts.append(round(time.perf_counter() - t0, 5))
                out[name] = ts
            print(json.dumps(out), flush=True)
