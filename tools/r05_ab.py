"""Round-5 same-process A/Bs on one box (not product code): settings alternate, every setting measured `reps` times.

    python tools/r05_ab.py rb_tokens [reps]       eight-wave row block with 8 vs 16 tokens per workgroup (R05_CLIPS=32: at BASELINE config #3's batch, two chains)
    python tools/r05_ab.py rowblock_waves [reps]  the split-f16 row block: eight-wave kernel (shipped) vs round 4's four-wave kernel (tune[IDF_TUNE_MISC] = 8)
    python tools/r05_ab.py tail_order [reps]      step-tail workgroup order: XCD-affine row tiles (shipped, tune[IDF_TUNE_MISC] = 0) vs plain ids (= 7, round 4)
    INTERDIFF_HIP_LIB=<variant .so> python tools/r05_ab.py once     one measurement of whatever library is loaded (forward + whole samples): for library-level A/Bs
                                                                   (LDS strides, store modes: build variants under build_ab/ with IDF_BUILD_DIR / IDF_EXTRA_HIPCC_FLAGS)
Each measurement: denoiser forward as one graph replay (us), whole 1000-step samples without / with correction (ms per step, two samples each)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                                      # noqa: E402
from interdiff_amd import _lib                                                    # noqa: E402
from interdiff_amd.diffusion import create_gaussian_diffusion                     # noqa: E402


def measure(model, corr, bt, y, diff, dev, samples=2):
    out = dict(forward_us=round(bench.time_forward_graph(model, bt, y, dev), 2))
    bench.run_steps(diff, model, None, bt, y, 57, seed=7)
    for name, hook in (('no_correction_ms_per_step', None), ('correction_ms_per_step', corr)):
        bench.run_steps(diff, model, hook, bt, y, 1000, seed=3)
        ts = []
        for _ in range(samples):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            bench.run_steps(diff, model, hook, bt, y, 1000, seed=3)
            torch.cuda.synchronize()
            ts.append(round(time.perf_counter() - t0, 5))
        out[name] = ts
    return out


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else 'tail_order'
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    torch.set_grad_enabled(False)
    dev = torch.device('cuda:0')
    bench.B_PER_GPU = int(os.environ.get('R05_CLIPS', bench.B_PER_GPU))       # (32: BASELINE config #3, stepped as two chains)
    model, corr, bt, y, _ = bench.build_world(dev, 0)
    diff = create_gaussian_diffusion('cosine', bench.STEPS)
    print('library', _lib.LIB_PATH, flush=True)
    if what == 'once':
        print('sample', json.dumps(dict(setting=os.environ.get('R05_LABEL', 'loaded library'), **measure(model, corr, bt, y, diff, dev))), flush=True)
        return
    if what == 'rb_tokens':                            # tokens per workgroup of the eight-wave row block: the launcher's choice (8 while the launch fits one round) vs 16 forced
        for rep in range(reps):
            for label, tok in (('rb_tokens auto (8 at this shape)', 0), ('rb_tokens 16 (forced)', 16), ('rb_tokens 8 (forced)', 8)):
                model.w.rb_tokens = tok
                model.__dict__.pop('_graph_cache', None)
                print('sample', json.dumps(dict(setting=label, clips=bench.B_PER_GPU, **measure(model, corr, bt, y, diff, dev))), flush=True)
        model.w.rb_tokens = 0
        return
    settings = {'tail_order': (('xcd_affine_tail (shipped)', 0), ('plain_id_tail (round 4)', 7)),
                'rowblock_waves': (('rowblock8_kernel (eight waves, shipped)', 0), ('rowblock_kernel (four waves, round 4)', 8))}[what]
    for rep in range(reps):
        for label, misc in settings:
            model.rowblock_waves = 4 if misc == 8 else 8
            model.w.tune[_lib.TUNE['misc']] = misc
            model.__dict__.pop('_graph_cache', None)          # captured launches bake the kernel arguments in
            print('sample', json.dumps(dict(setting=label, **measure(model, corr, bt, y, diff, dev))), flush=True)
    model.w.tune[_lib.TUNE['misc']] = 0


if __name__ == '__main__':
    main()
