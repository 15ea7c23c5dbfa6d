"""Time of the K/V-tiled self-attention (csrc/denoiser.hip self_attn_tiled_kernel: clips longer than 208 frames) next to the one-shot kernels, once (not product code):
    python tools/tiled_attn_time.py   -> per shape: us per launch of the self-attention stage inside a denoiser forward (library profile hooks, HIP events), forward us."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from interdiff_amd import synthetic as syn, _lib        # noqa: E402
from interdiff_amd.mdm import MDM                        # noqa: E402
import kbench                                            # noqa: E402


def main():
    torch.set_grad_enabled(False)
    dev = 'cuda'
    sd = {k: torch.from_numpy(v) for k, v in syn.mdm_state_dict(233).items()}
    model = MDM(sd, device=dev, n_steps=1000)
    lib = _lib.load()
    out = {}
    for B, T in ((16, 100), (8, 192), (8, 208), (7, 240), (5, 300), (3, 512)):
        g = torch.Generator().manual_seed(T)
        x = torch.randn(B, 1, 144, T, generator=g).to(dev)
        ts = torch.randint(0, 1000, (B,), generator=g).to(dev)
        y = {'cond': torch.randn(10, B, 256, generator=g).to(dev)}
        if T > model.w.max_T:
            continue
        p = kbench.profile_forward(lib, model, x, ts, y, 30)
        out['B=%d T=%d (%d rows)' % (B, T, B * T)] = dict(self_attention_us_per_launch=round(p.get('self_attn', float('nan')), 2), forward_sum_us=round(p['_sum_us_per_forward'], 1),
                                                           kernel='split-f16 one-shot (planes)' if T <= 192 else ('fp32 one-shot' if T <= 208 else 'self_attn_tiled_kernel (64-key tiles, running max / sum)'))
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
