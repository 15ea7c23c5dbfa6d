"""A short driver for counter passes on the 16- / 64-row feed-forward kernels (not product code): 60 launches of interdiff_mdm_ffn at 800
(or argv[1]: 3200 -> the 64-row kernel) rows, walking through the eight layers like a denoiser step.
    tools/gpu_pmc.sh <tag> "<counters>" python tools/ffn16_pmc.py [rows]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                                      # noqa: E402
from interdiff_amd import _lib                                                    # noqa: E402
from interdiff_amd.mdm import ffn_parts                                           # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device('cuda:0')
model, corr, bt, y, _ = bench.build_world(dev, 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 800
x2 = [torch.randn(N, 256, generator=torch.Generator().manual_seed(5 + i)).to(dev) for i in range(2)]
parts = [torch.empty(_lib.FFN_SLICES, N, 256, device=dev) for _ in range(2)]
for i in range(60):
    ffn_parts(model, x2[i & 1], i % 8, out=parts[i & 1])
torch.cuda.synchronize()
