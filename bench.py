"""bench.py -- denoising-sampler throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one DDPM reverse step (p_sample: denoiser forward + x0-inpainting + [gated correction] + posterior update with
in-kernel noise) over one batch of B=16 BEHAVE-shaped synthetic clips of T=100 frames per GPU (BASELINE config #2:
eval_smpl_short.py, B=16, T=100, 1000-step DDPM, correction mode).

The timed region is ALWAYS made of whole samples: max(3, ceil(K / 1000)) complete 1000-step samples, each with its 989 plain steps, its
11 gated correction steps (t in {500,450,..,0}: SMPL-H FK + LBS, normals, signed nearest neighbours, contact-frame predictor) and
the once-per-sample memory folding -- whatever --steps says, so that the headline is the rate of the workload eval_smpl_short.py runs,
never a window of plain steps.  Every sample is clocked on its own (barrier + synchronize on both sides, max over ranks); `value` is
computed from the MEDIAN sample, min / max / all travel in `ms_per_step_samples`.  `steps` in the JSON line is the number of steps
really timed, `steps_requested` echoes --steps.  Inputs are resident in HBM before the clock starts.

value = frame-steps/s = 1000 * B_total * T / (median seconds per sample)  (whole job, all ranks; weak scaling: B=16 per GPU).

`--gpus N` with N > 1 and no WORLD_SIZE in the environment re-launches itself as N ranks (torch.distributed.run, one per GPU).
After the timed region every rank scores its clips and the six per-clip metric vectors are collated with the path's one collective
(RCCL all-gather); rank 0 then measures, outside the clock: the same sample in no_correction mode, BASELINE config #3 (B=32),
a K-step window of plain steps, the per-kernel profile + roofline kernel, the conditioning path, the post-optimisation row and the
CPU baseline.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
_T0 = time.perf_counter()
from interdiff_amd import synthetic as syn, _lib, dist as idist   # noqa: E402
from interdiff_amd.mdm import MDM                                  # noqa: E402
from interdiff_amd.smpl import SMPL_Layer                          # noqa: E402
from interdiff_amd.objprojector import ObjProjector                # noqa: E402
from interdiff_amd.correction import HipCorrection                 # noqa: E402
from interdiff_amd.diffusion import create_gaussian_diffusion      # noqa: E402

B_PER_GPU, T, PAST, P, STEPS = 16, 100, 10, 2048, 1000
# algorithmic FLOP per token of one denoiser step and of the kernels bench reports on
FLOP_PER_TOKEN = 11978752 + 2048 * T
FFN_FLOP_PER_TOKEN = 2 * 2 * 256 * 1024                     # linear1 + linear2 of one layer: what ONE launch of the fused kernel computes
PEAK_F32_MFMA_TFLOPS = 157.3                                # MI355X_MICROARCH.md: fp32-input MFMA, dense
PEAK_F16_MFMA_TFLOPS = 2500.0                               # MI355X_MICROARCH.md: f16 / bf16 MFMA, dense
MALL_STREAM_TBPS = 12 * 256 * 2.4e9 / 1e12                  # Infinity-Cache-resident private streams: 12 B/clk/CU measured (tools/dma_ceiling.hip, profiles/r02_dma_ceiling.txt) = 7.4 TB/s
DOMINANT_KERNEL_ID = 'idf_ffn_h2::ffn_h2_kernel r06 (8 computing + 8 loader waves)'        # the build the roofline block (and profiles/traffic.json) speaks about
ROCPROF_STATS = 'profiles/r06_kernel_stats_bench.txt'       # rocprofv3 --kernel-trace --stats of `python bench.py` on the same build (tools/profile_round_r06.sh)


def ffn_issued_f16_flop(rows, tile):
    """f16 FLOP the split-f16 feed-forward kernel ISSUES for one launch over `rows` token rows with `tile`-row workgroups (csrc/ffn_h2.h): per workgroup
    (13 hidden tiles x 8 K steps + 16 output tiles x 7 K steps) x tile/16 token tiles x 3 products, each one v_mfma_f32_16x16x32_f16 = 16 384 FLOP;
    ceil(rows / tile) x 5 slices workgroups.  324 000 MFMAs at 1600 rows = SQ_VALU_MFMA_BUSY_CYCLES / 16 of the PMC pass (profiles/)."""
    wgs = -(-rows // tile) * 5
    return wgs * (13 * 8 + 16 * 7) * (tile // 16) * 3 * 16384


def EXECUTED_FLOP_PER_TOKEN(T):
    """What the kernels really contract per token and step: the cross-attention to the constant memory is folded per sample (x.G^T, P.VW instead of the q / out
    projections), so ~16 % less than SURVEY.md's count of the reference's own GEMMs."""
    ffn = 8 * 2 * 2 * 256 * 1024
    qkv, attn, outp = 2 * 2 * 256 * 768, 2 * 2 * 2 * T * 256, 2 * 2 * 256 * 256
    qan, cross = 6 * 2 * 256 * 30, 8 * (2 * 256 * 40 + 2 * 40 * 256)
    ends = 2 * 2 * 144 * 256
    return ffn + qkv + attn + outp + qan + cross + ends


def STEP_WEIGHT_BYTES(B):
    """Bytes of weights and per-sample constants one step reads (each at least once): feed-forward plane streams, QKV planes, out-projection fragments,
    learned queries, step-tail fragments, and the folded memory G / VW of every (layer, clip) as plane fragments + g0."""
    ffn, qkv, outp = 8 * 5 * 442368, 2 * 5 * 163840, 2 * 262144
    qc, tail = 6 * 30720, (36864 + 40960) * 4
    mem = 8 * B * ((12288 + 16384) * 4 + 160)
    return ffn + qkv + outp + qc + tail + mem


def STEP_ACTIVATION_BYTES(rows):
    """Kernel-to-kernel hand-overs of one chained plain step, each written once and read once: 19 x [rows,256] fp32 + the two QKV matrices [rows,768]."""
    return 2 * (19 * rows * 256 * 4 + 2 * rows * 768 * 4)


def tt(d, dev=None):
    out = {}
    for k, v in d.items():
        if isinstance(v, np.ndarray):
            v = torch.from_numpy(v)
            if dev is not None:
                v = v.to(dev)
        out[k] = v
    return out


def build_world(dev, rank):
    sd = syn.mdm_state_dict(233)
    smpl_np = syn.smplh_model(7)
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'correction_ckpt.npz'))      # real ObjProjector weights
    osd = {k: z[k] for k in z.files}
    model = MDM(sd, device=dev, n_steps=STEPS)
    smpl = SMPL_Layer(smpl_np, device=dev)
    corr = HipCorrection(smpl, ObjProjector(osd, T=T, past_len=PAST, device=dev), n_points=P, past_len=PAST, device=dev)
    bt = tt(syn.make_clip_batch(seed=233 + rank, B=B_PER_GPU, T=T, past_len=PAST, n_points=P), dev)
    pad = list(range(PAST)) + [PAST - 1] * (T - PAST)
    mask = torch.ones_like(bt['gt'], dtype=torch.bool)
    mask[..., PAST:] = False
    y = dict(cond=bt['cond'], inpainted_motion=bt['gt'], inpainting_mask=mask, hand_pose=bt['hand_pose'][pad].contiguous(),
             beta=bt['beta'], obj_points=bt['obj_points'])
    return model, corr, bt, y, (sd, smpl_np, osd)


def run_steps(diff, model, corr, bt, y, n_steps, seed, use_graph=True, first_t=None):
    """n_steps iterations of the 1000-step loop from t = first_t (default 999); `corr` None = no_correction mode."""
    return diff.p_sample_loop(model, tuple(bt['noise'].shape), noise=bt['noise'], clip_denoised=False, first_t=first_t,
                              model_kwargs={'y': y}, denoised_fn=corr, seed=seed, n_steps=n_steps, use_graph=use_graph)


def kernel_profile(diff, model, corr, bt, y, n_steps=30):
    """Second, instrumented pass: HIP events around every launch (on the launch stream) -> ms per kernel kind."""
    lib = _lib.load()
    _lib.check(lib.interdiff_profile_begin(200000))
    run_steps(diff, model, corr, bt, y, n_steps, seed=1, use_graph=False)       # eager route: events between launches
    x = bt['noise'].clone()
    corr.apply(x, 500, y)                                   # one correction call so its kernels are sampled too
    ms = (C.c_double * len(_lib.KERNEL_KINDS))()
    cnt = (C.c_int64 * len(_lib.KERNEL_KINDS))()
    _lib.check(lib.interdiff_profile_end(ms, cnt))
    return {k: dict(ms_total=ms[i], launches=int(cnt[i]), us_avg=(1e3 * ms[i] / cnt[i]) if cnt[i] else None)
            for i, k in enumerate(_lib.KERNEL_KINDS) if cnt[i]}


def time_dominant_kernel(model, dev, reps=200, chains=1, cycle_layers=True, n_rows=None):
    """The roofline kernel: the fused feed-forward block of one layer (csrc/ffn.h: [1600,256] -> linear1 -> gelu -> linear2 as five
    partial slabs; 8 of the 22 launches of a denoiser forward and the bulk of its FLOP), timed live with HIP events on the launch
    stream around `reps` launches replayed from a hipGraph, so that the figure is the GPU's whatever the host is doing.
    cycle_layers (the figure `roofline` reports): consecutive launches walk through the EIGHT layers' weight streams and alternate
    two activation buffers, as a denoiser step does -- every launch finds its 2.1-MB weight stream in the Infinity Cache, not in the
    L2s where a burst on ONE layer leaves it; rocprofv3's in-situ average (profiles/) is the cross-check.  cycle_layers=False: the
    round-2 burst on layer 1 (secondary key).  Returns the MEAN of three bursts (and the best, for reference).
    chains = 2: the form the sampler's plain steps launch it in -- the batch's rows as two halves, each half a chain of
    launches on its own branch of the graph; the figure is then per PAIR of concurrent half-size launches (the same FLOP).
    n_rows: another batch's token rows (<= 800: the 16-row tile kernel, csrc/ffn.h ffn_fused16_kernel; 3200: the 64-row one) instead of the
    workload's."""
    from interdiff_amd.mdm import ffn_parts
    total = n_rows or B_PER_GPU * T
    N = total // chains
    g = torch.Generator().manual_seed(5)
    x2 = [[torch.randn(N, 256, generator=g).to(dev) for _ in range(2)] for _ in range(chains)]
    parts = [[torch.empty(_lib.FFN_SLICES, N, 256, device=dev) for _ in range(2)] for _ in range(chains)]
    layer_of = (lambda i: i % 8) if cycle_layers else (lambda i: 1)
    for i in range(20):
        for c in range(chains):
            ffn_parts(model, x2[c][i & 1], layer_of(i), out=parts[c][i & 1], batch_rows=total)
    torch.cuda.synchronize()
    per_graph = 48
    side = torch.cuda.Stream(device=dev)
    branch = [torch.cuda.Stream(device=dev) for _ in range(chains)] if chains > 1 else None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            if chains == 1:
                for i in range(per_graph):
                    ffn_parts(model, x2[0][i & 1], layer_of(i), out=parts[0][i & 1], batch_rows=total)
            else:
                cur = torch.cuda.current_stream()
                for c in range(chains):
                    branch[c].wait_stream(cur)
                    with torch.cuda.stream(branch[c]):
                        for i in range(per_graph):
                            ffn_parts(model, x2[c][i & 1], layer_of(i), out=parts[c][i & 1], batch_rows=total)
                for c in range(chains):
                    cur.wait_stream(branch[c])
        graph.replay()
        side.synchronize()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side)
            for _ in range(max(1, reps // per_graph)):
                graph.replay()
            e1.record(side)
            e1.synchronize()
            ts.append(1e3 * e0.elapsed_time(e1) / (max(1, reps // per_graph) * per_graph))
    return sum(ts) / len(ts), min(ts)


def time_forward_graph(model, bt, y, dev, per_graph=10, reps=5):
    """One denoiser forward (22 launches) replayed from a hipGraph on a side stream, HIP events around `reps` replays of
    `per_graph` forwards: the GPU's time for MDM.forward, launch gaps as they are inside the sampler's captured steps."""
    x = bt['noise'].clone()
    ts = torch.full((x.shape[0],), 500, dtype=torch.int64, device=dev)
    out = torch.empty_like(x)
    for _ in range(3):
        model(x, ts, y=y, out=out)
    torch.cuda.synchronize()
    side, graph = torch.cuda.Stream(device=dev), torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            for _ in range(per_graph):
                model(x, ts, y=y, out=out)
        graph.replay()
        side.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(side)
        for _ in range(reps):
            graph.replay()
        e1.record(side)
        e1.synchronize()
    return 1e3 * e0.elapsed_time(e1) / (reps * per_graph)


def log(msg):
    print('[bench %7.1fs] %s' % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


def usable_cores():
    """Host cores this process may really use: affinity mask, capped by the cgroup CPU quota (a container that
    reports 100+ cores but is throttled to a few would otherwise thrash) and by 32."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 32))


def cpu_baseline(assets, bt_cpu, y_cpu):
    """The oracle (CPU restatement of the reference path), timed on this box's host cores on a bounded sample of the SAME mix the
    GPU headline times (989 plain + 11 corrected steps per sample): 3 warm-up + 5 timed plain steps at the full B=16,T=100 batch,
    and 3 timed correction calls (denoised_fn at t = 500, 250, 0), each on one clip of the batch and scaled to 16 clips (the
    hook is independent per clip; a 16-clip call takes ~80 s)."""
    from oracle import diffusion as odf, denoiser as oden, correction as ocor
    sd, smpl_np, osd = assets
    sd_t = {k: torch.from_numpy(v) for k, v in sd.items()}
    cores = usable_cores()
    torch.set_num_threads(cores)
    sched = odf.make_schedule(STEPS)
    x = bt_cpu['noise'].clone()
    ts = torch.full((B_PER_GPU,), 999, dtype=torch.int64)

    def plain():
        x0 = oden.mdm_forward(sd_t, x, ts, y_cpu['cond'])
        x0 = x0 * (~y_cpu['inpainting_mask']) + y_cpu['inpainted_motion'] * y_cpu['inpainting_mask']
        return float(sched['posterior_mean_coef1'][999]) * x0 + float(sched['posterior_mean_coef2'][999]) * x + 0.1 * torch.randn_like(x)
    for _ in range(3):
        plain()
    n_plain = 5
    t0 = time.perf_counter()
    for _ in range(n_plain):
        plain()
    t_plain = (time.perf_counter() - t0) / n_plain
    smpl_t = {k: torch.from_numpy(v) for k, v in smpl_np.items()}
    osd_t = {k: torch.from_numpy(v) for k, v in osd.items()}
    t_calls = []
    for clip, tval in ((0, 500), (1, 250), (2, 0)):
        sl = slice(clip, clip + 1)
        ysub = {k: (v[:, sl] if k in ('cond', 'hand_pose', 'beta') else v[sl]) if isinstance(v, torch.Tensor) else v
                for k, v in y_cpu.items()}
        ysub.update(smpl=smpl_t, obj_model=osd_t)
        t0 = time.perf_counter()
        ocor.denoised_fn(bt_cpu['gt'][sl].clone(), torch.full((1,), tval, dtype=torch.int64), {'y': ysub}, past_len=PAST)
        t_calls.append(time.perf_counter() - t0)
    t_corr = sum(t_calls) / len(t_calls) * B_PER_GPU
    steps_per_s = STEPS / ((STEPS - 11) * t_plain + 11 * (t_plain + t_corr))
    return dict(value=steps_per_s * B_PER_GPU * T, unit='frame-steps/s', cores=cores, kind='port',
                sample='3 warm-up + %d timed plain steps at B=16,T=100 (%.3f s/step) + 3 correction calls (t=500,250,0) on one clip '
                       'each, scaled x16 (%.1f s per 16-clip call); blended over the 989 plain + 11 corrected steps of one sample = '
                       'the mix the GPU headline times; torch CPU fp32, %d threads' % (n_plain, t_plain, t_corr, cores),
                steps_per_sec=steps_per_s, plain_step_s=t_plain, correction_call_s=t_corr,
                plain_only=dict(value=B_PER_GPU * T / t_plain, unit='frame-steps/s',
                                note='plain steps alone (the denoiser): the blended figure is dominated by the oracle\'s brute-force nearest-neighbour search in the 11 correction calls'))


def postopt_bench(smpl, smpl_np, dev, with_cpu, B=16, T=20, n_points=2048):
    """Physics post-optimisation ("next" row N4, optimization.py:19-173) at BASELINE.json configs[4]'s per-GPU share:
    16 clips x (10 past + 10 future) frames, 2048 object points, the full 200 Adam iterations, clips side by side."""
    from interdiff_amd.optimize import PhysicsOptimizer
    keys = ('pose', 'trans', 'obj_angles', 'obj_trans', 'betas', 'obj_points')
    bt = syn.make_optim_batch(seed=1, B=B, T=T, n_points=n_points)
    opt = PhysicsOptimizer(smpl, device=dev)
    batch = [torch.from_numpy(bt[k]).to(dev) for k in keys]
    opt.optimize(*batch, iters=range(0, 5))
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        res = opt.optimize(*batch)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    pairs = B * T * n_points * smpl.cmodel.V                       # distances per iteration, each serves both NN questions
    out = dict(workload='optimization.py: %d clips x %d frames, %d object points, 200 Adam iterations' % (B, T, n_points),
               ms_per_iteration=best / 200 * 1e3, clips_per_sec=B / best, saved=bool(res['saved'].all()),
               nn_scan=dict(kernels='corr_contact_kernel<true> + opt_patch_kernel + opt_near_kernel', bound='valu', pairs_a_brute_force_would_score=pairs,
                            note='round 3: the two nearest-neighbour questions are asked separately, each with an exact cull (nearest vertex per point: the hook\'s block-culled scan; '
                                 'any point within 0.5 m per vertex: 64-point patches skipped by their boxes); ~0.7 ms of a 1.33-ms iteration (round 2: one brute-force scan, 1.0 of 1.65 ms), profiles/r03_postopt_kernel_stats.txt'))
    if with_cpu:
        from oracle import optimization as oo
        model = {k: torch.from_numpy(v) for k, v in smpl_np.items()}
        one = [torch.from_numpy(bt[k][0]) for k in keys]
        t0 = time.perf_counter()
        oo.optimize(model, *one, iters=[151])
        out['cpu_port_s_per_iteration_per_clip'] = time.perf_counter() - t0
        out['cpu_sample'] = '1 Adam iteration (torch autograd oracle) of 1 clip, %d threads' % torch.get_num_threads()
    return out


def self_spawn(args):
    """`python bench.py --gpus N` outside a launcher: start the N ranks ourselves (one process per GPU, RCCL over xGMI)."""
    import socket
    import subprocess
    if torch.cuda.device_count() < args.gpus:
        print('bench.py: --gpus %d but only %d GPU(s) visible' % (args.gpus, torch.cuda.device_count()), file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log('no WORLD_SIZE in the environment: launching %d ranks: %s' % (args.gpus, ' '.join(cmd)))
    return subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0')))


def timed_samples(diff, model, corr, bt, y, n, seed0=233):
    """n complete 1000-step samples back to back on the current stream; returns (wall seconds, last sample)."""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s_i in range(n):
        out = run_steps(diff, model, corr, bt, y, STEPS, seed=seed0 + s_i)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, out


def philox_stream(seed):
    """step_noise callable that materialises the in-kernel generator's stream (what the eager route is fed to reproduce the graph route)."""
    lib = _lib.load()

    def draw(it, x):
        out = torch.empty_like(x)
        _lib.check(lib.interdiff_randn(_lib.dptr(out), out.numel(), seed, it, _lib.stream()), 'randn')
        return out
    return draw


def route_check(diff, model, corr, bt, y, seed=4242, first_t=520, n=60):
    """The route a leg times (hipGraph blocks of fused steps, chains, captured hook steps, in-kernel Philox) against the EAGER route fed
    the same noise stream, on a 60-step window that crosses the corrected step t = 500: bit for bit, at the leg's own shape."""
    kw = dict(noise=bt['noise'], clip_denoised=False, model_kwargs={'y': y}, denoised_fn=corr, n_steps=n, first_t=first_t)
    a = diff.p_sample_loop(model, tuple(bt['noise'].shape), seed=seed, **kw)
    b = diff.p_sample_loop(model, tuple(bt['noise'].shape), step_noise=philox_stream(seed), use_graph=False, **kw)
    if not torch.equal(a, b):
        raise AssertionError('timed route differs from the eager route: max |delta| = %g' % float((a - b).abs().max()))
    return True


def main():
    global B_PER_GPU, T
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=STEPS)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-profile', action='store_true')
    ap.add_argument('--no-postopt', action='store_true')
    ap.add_argument('--no-extra-configs', action='store_true', help='skip the no_correction / B=32 / plain-window legs')
    ap.add_argument('--clips-per-gpu', type=int, default=B_PER_GPU,
                    help='NOT the BASELINE configuration unless 16: batch-scaling experiments only (DESIGN.md §4.1)')
    args = ap.parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_spawn(args))
    B_PER_GPU = args.clips_per_gpu
    # stdout carries exactly ONE line, the JSON: libraries that chat on file descriptor 1 (RCCL prints a version banner when its first
    # communicator comes up) are sent to stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    torch.set_grad_enabled(False)
    torch.set_num_threads(usable_cores())             # ONE thread count for every CPU leg below (cpu_baseline, the post-optimisation CPU sample)
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
    rccl_note = None
    try:                                              # a process group even for ONE rank: the path's collective then goes through RCCL for real at N = 1 too
        rank, world, local = idist.init_from_env('nccl', single_rank_group=True)
    except Exception as e:                            # (never let the collective's plumbing take a 1-GPU measurement down)
        if int(os.environ.get('WORLD_SIZE', '1')) > 1:
            raise
        rank, world, local, rccl_note = 0, 1, 0, 'single-rank nccl group failed to start: %r' % (e,)
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    K = max(1, args.steps)
    n_samples = max(3, (K + STEPS - 1) // STEPS)      # whole samples only (the headline always contains the correction path), at least three for a spread
    log('building world (weights, SMPL-H stand-in, clips) on %s' % dev)
    model, corr, bt, y, assets = build_world(dev, rank)
    log('world ready')
    diff = create_gaussian_diffusion('cosine', STEPS)

    # untimed warm-up: W plain steps + one correction call (first-use allocations, code objects)
    run_steps(diff, model, corr, bt, y, max(1, args.warmup), seed=7)
    run_steps(diff, model, corr, bt, y, 57, seed=7)        # setup, not a step count: captures every hipGraph block size (49+7+1)
    corr.apply(bt['noise'].clone(), 500, y)
    route_ok = route_check(diff, model, corr, bt, y)          # untimed: the route about to be timed == the eager route, 60 steps across t = 500, bit for bit
    torch.cuda.synchronize()
    log('warm-up done')

    # the once-per-sample memory folding runs inside the clock (p_sample_loop folds `cond` at the start of every sample).  Every sample is its
    # own timed region: barrier + synchronize on both sides, MAX over ranks; the headline is the MEDIAN sample
    sample_s, sample_ranks = [], []
    for s_i in range(n_samples):
        idist.barrier()
        wall_local, out = timed_samples(diff, model, corr, bt, y, 1, seed0=233 + s_i)
        idist.barrier()
        sample_s.append(idist.max_over_ranks(wall_local, dev))
        sample_ranks.append(idist.gather_scalar(wall_local, dev))       # every rank's own clock around the same region: a straggler shows
    assert torch.isfinite(out).all()
    wall = sorted(sample_s)[len(sample_s) // 2]                        # median sample (n odd by default: 3)
    wall_ranks = sample_ranks[sample_s.index(wall)]
    n_timed = n_samples * STEPS
    log('timed region: %d whole samples of %d steps (incl. %d correction steps each): %s s; median %.4f' % (n_samples, STEPS, 11, ['%.4f' % w for w in sample_s], wall))

    # ---- eval leg, all ranks: every rank scores its clips, ONE all-gather (RCCL over xGMI) collates the six per-clip metric vectors
    from interdiff_amd import eval as ev
    clip_batch = dict(gt=bt['gt'], cond=bt['cond'], hand_pose=bt['hand_pose'], beta=bt['beta'], obj_points=bt['obj_points'])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    per_clip, means = ev.evaluate_sharded(model, diff, corr, clip_batch, PAST, mode='correction', diverse_samples=1, seed=1000,
                                          presharded=True)
    torch.cuda.synchronize()
    eval_s = time.perf_counter() - t0
    assert all(v.numel() == B_PER_GPU * world for v in per_clip.values())
    coll = idist.collective_backend_info()
    if world == 1 and coll['backend'] == 'nccl':      # one rank: evaluate_sharded took the world == 1 shortcut -- push the same vectors through RCCL explicitly
        mine = {k: v.clone() for k, v in per_clip.items()}
        back, header = idist.gather_metrics(mine, 1, counts=[B_PER_GPU], return_header=True, force_collective=True, check_header=True)
        assert all(torch.equal(back[k], mine[k]) for k in mine)
        coll['single_rank_all_gather'] = 'issued (1-rank nccl group, header verified)'
    if rccl_note:
        coll['note'] = rccl_note
    idist.shutdown()              # no collective after this point: rank 0 alone takes the extra legs below
    if rank != 0:
        return

    extra = {}
    if not args.no_extra_configs:
        w_nc, o_nc = timed_samples(diff, model, None, bt, y, 1)
        extra['no_correction'] = dict(workload='the same clips, eval_smpl_short.py --mode no_correction (1000 plain steps)', steps=STEPS,
                                      ms_per_step=1e3 * w_nc / STEPS, value=STEPS * B_PER_GPU * T / w_nc, unit='frame-steps/s')
        # K-step window of plain steps from t = 999 (what --steps K alone would have timed): secondary, never the headline
        kw = min(K, 450)
        run_steps(diff, model, corr, bt, y, kw, seed=3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(diff, model, corr, bt, y, kw, seed=3)
        torch.cuda.synchronize()
        w_win = time.perf_counter() - t0
        extra['plain_step_window'] = dict(timesteps='999..%d' % (STEPS - kw), steps=kw, correction_steps_in_window=0,
                                          ms_per_step=1e3 * w_win / kw, value=kw * B_PER_GPU * T / w_win, unit='frame-steps/s')
        if B_PER_GPU == 16:
            # BASELINE config #3: eval_smpl_short.py + correction predictor, B=32 (the reference's own default batch), same T
            B_PER_GPU = 32
            m3, c3, bt3, y3, _ = build_world(dev, rank)
            run_steps(diff, m3, c3, bt3, y3, 57, seed=7)
            c3.apply(bt3['noise'].clone(), 500, y3)
            ok3 = route_check(diff, m3, c3, bt3, y3)
            w3s = [timed_samples(diff, m3, c3, bt3, y3, 1, seed0=233 + i) for i in range(3)]
            w3 = sorted(w[0] for w in w3s)[1]
            extra['config3_B32_correction'] = dict(workload='eval_smpl_short.py correction mode, B=32, T=%d, whole samples (median of three; all in seconds_each)' % T, steps=STEPS, seconds_each=[w[0] for w in w3s],
                                                   ms_per_step=1e3 * w3 / STEPS, value=STEPS * 32 * T / w3, unit='frame-steps/s', equals_eager_route_on_60_steps_across_t500=ok3)
            del m3, c3, bt3, y3
            B_PER_GPU = 16
        if B_PER_GPU == 16:
            # the reference's OWN default eval shape (eval_smpl_short.py:376-380,401,405): B = 32 clips of T = 35 frames (10 past + 25 future).
            # T % 4 != 0: the fused step's per-row update form (csrc/gemm.h post_prefetch), two chains of 560 token rows
            B_PER_GPU, T = 32, 35
            m5, c5, bt5, y5, _ = build_world(dev, rank)
            run_steps(diff, m5, c5, bt5, y5, 57, seed=7)
            c5.apply(bt5['noise'].clone(), 500, y5)
            ok5 = route_check(diff, m5, c5, bt5, y5)
            w5s = [timed_samples(diff, m5, c5, bt5, y5, 1, seed0=233 + i) for i in range(3)]
            w5 = sorted(w[0] for w in w5s)[1]
            extra['reference_default_B32_T35'] = dict(workload='eval_smpl_short.py with its own defaults: B=32, T=35 (10 past + 25 future), correction mode, whole 1000-step samples (median of three; all in seconds_each)',
                                                      steps=STEPS, seconds_each=[w[0] for w in w5s], ms_per_step=1e3 * w5 / STEPS, value=STEPS * 32 * 35 / w5, unit='frame-steps/s',
                                                      equals_eager_route_on_60_steps_across_t500=ok5)
            del m5, c5, bt5, y5
            B_PER_GPU, T = 16, 100
        if B_PER_GPU == 16:
            # BASELINE config #4's per-GPU share: eval_smpl_long.py, B = 64 over 8 GPUs = 8 clips per GPU, autoregressive rollout of
            # K = 4 further windows; every window = conditioning (PointNet++ + 8-layer encoder) + one whole 1000-step sample with correction
            from interdiff_amd import eval as ev4
            K4, B4 = 4, 8
            ei4 = tt(syn.make_embedding_inputs(seed=78, B=B4, T=T, n_points=P), dev)
            g4 = torch.Generator().manual_seed(4)
            raw4 = dict(ei4, hand_pose=(0.1 * torch.randn(T, B4, 90, generator=g4)).to(dev), beta=torch.randn(1, B4, 10, generator=g4).expand(T, B4, 10).contiguous().to(dev))
            ev4.sample_long(model, diff, corr, raw4, 1, PAST, seed=1)                 # warm-up: captures for the 8-clip shape
            # the rollout's timed route == its eager route: two windows (conditioning pass each), 60 steps across t = 500, same Philox streams
            a4 = ev4.sample_long(model, diff, corr, raw4, 1, PAST, seed=2, n_steps=60, first_t=520)
            b4 = ev4.sample_long(model, diff, corr, raw4, 1, PAST, seed=2, n_steps=60, first_t=520, use_graph=False, step_noise=lambda k: philox_stream(2 + k))
            assert all(torch.equal(p_, q_) for p_, q_ in zip(a4, b4)), 'config #4: timed rollout route differs from the eager route'
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            o4 = ev4.sample_long(model, diff, corr, raw4, K4, PAST, seed=2)
            torch.cuda.synchronize()
            w4 = time.perf_counter() - t0
            assert o4[0].shape[0] == T + K4 * (T - PAST) and all(torch.isfinite(a).all() for a in o4)       # (route equality asserted above)
            extra['config4_long_horizon'] = dict(workload='eval_smpl_long.py autoregressive rollout: %d clips per GPU (B=64 over 8 GPUs), T=%d, %d windows '
                                                          '(first + %d re-conditioned), each = conditioning pass + whole 1000-step sample with correction' % (B4, T, K4 + 1, K4),
                                                 windows=K4 + 1, steps=(K4 + 1) * STEPS, seconds=w4, ms_per_step=1e3 * w4 / ((K4 + 1) * STEPS),
                                                 value=(K4 + 1) * STEPS * B4 * T / w4, unit='frame-steps/s', frames_generated_per_clip=T + K4 * (T - PAST),
                                                 equals_eager_route_on_2_windows_60_steps_across_t500=True)
        log('extra configurations done')
    prof = None
    if not args.no_kernel_profile:
        prof = kernel_profile(diff, model, corr, bt, y)
        dom_us, dom_best = time_dominant_kernel(model, dev)
        burst_us, burst_best = time_dominant_kernel(model, dev, cycle_layers=False)
        pair_us, pair_best = time_dominant_kernel(model, dev, chains=2)
        small_us, small_best = time_dominant_kernel(model, dev, n_rows=800)
        big_us, big_best = time_dominant_kernel(model, dev, n_rows=3200)
        fwd_us = time_forward_graph(model, bt, y, dev)
        exact_us = exact_fwd_us = None
        if getattr(model, 'ffn_math', 'exact') == 'split':       # the exact-fp32 kernel of csrc/ffn.h stays selectable: its figures next to the shipped kernel's
            model.ffn_math = 'exact'
            exact_us, _ = time_dominant_kernel(model, dev)
            exact_fwd_us = time_forward_graph(model, bt, y, dev)
            model.ffn_math = 'split' 
        log('kernel profile done')
    # once-per-sample conditioning path ("next" row): PointNet++ object encoder + embeddings + 8-layer encoder
    ei = tt(syn.make_embedding_inputs(seed=77, B=B_PER_GPU, T=T, n_points=P), dev)
    args_e = (ei['body_pose'], ei['body_trans'], ei['obj_angles'], ei['obj_trans'], ei['obj_points'], PAST)
    model._get_embeddings(*args_e)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        model._get_embeddings(*args_e)
    e1.record()
    e1.synchronize()
    enc_ms = e0.elapsed_time(e1) / 5
    post = None
    if not args.no_postopt:
        post = postopt_bench(corr.smpl, assets[1], dev, with_cpu=(world == 1 and not args.no_cpu_baseline))
        log('post-optimisation bench done')
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(assets, tt({k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in bt.items()}),
                           {k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in y.items()})
    log('cpu baseline done' if cpu else 'cpu baseline skipped')
    Btot = B_PER_GPU * world
    split = getattr(model, 'ffn_math', 'exact') == 'split'
    line = dict(metric='denoising frame-steps/sec (denoising-steps/sec x B x T frames)', value=STEPS * Btot * T / wall,
                unit='frame-steps/s', n_gpus=world, steps=n_timed, steps_requested=K, warmup=args.warmup, ms_per_step=1e3 * wall / STEPS,
                higher_is_better=True, scaling='weak', vs_baseline=None,
                dtype='f32 (every contraction of the denoiser -- feed-forward block, QKV projection, self-attention, row-block contractions and the two token GEMMs: every fp32 operand as two f16 planes, 3 f16 MFMAs per product, fp32 accumulate; SMPL and everything else fp32 MFMA / VALU)' if split else 'f32',
                data='synthetic', steps_per_sec=STEPS / wall,
                ms_per_step_samples=dict(median=1e3 * wall / STEPS, min=1e3 * min(sample_s) / STEPS, max=1e3 * max(sample_s) / STEPS,
                                         all=[round(1e3 * w_ / STEPS, 5) for w_ in sample_s], n=n_samples,
                                         note='every whole 1000-step sample clocked on its own (barrier + synchronize both sides, max over ranks); value = from the median'),
                steps_note='--steps is advisory: the timed region is always max(3, ceil(steps / 1000)) WHOLE 1000-step samples (989 plain + 11 corrected steps each)',
                ms_per_step_by_rank=dict(min=1e3 * min(wall_ranks) / STEPS, max=1e3 * max(wall_ranks) / STEPS,
                                         all=[round(1e3 * w_ / STEPS, 5) for w_ in wall_ranks], of='the median sample'),
                timed_route_equals_eager_route=dict(ok=route_ok, how='60 steps from t = 520 (corrected step t = 500 inside), graph route vs eager route fed the same Philox stream, torch.equal'),
                config=dict(workload='eval_smpl_short.py correction mode: BEHAVE-shaped SMPL-H clips, B=%d per GPU, T=%d '
                                     '(10 past + 90 future), C=144, 1000-step cosine DDPM, 2048 object points, real '
                                     'ObjProjector checkpoint, synthetic denoiser/SMPL-H weights' % (B_PER_GPU, T),
                            global_batch=Btot, seq_len=T, samples_in_region=n_samples, correction_steps_in_region=11 * n_samples,
                            plain_steps_in_region=989 * n_samples, timesteps='%d whole sample(s): t = 999..0' % n_samples,
                            parallelism='clips sharded x%d' % world, ffn_math=getattr(model, 'ffn_math', 'exact')))
    line['eval_collation'] = dict(collective='ONE all_gather of [7, B_local] fp32 (count header + six metric rows), %d rank(s)' % world,
                                  backend=coll.get('backend'), rccl_version=coll.get('rccl_version'), single_rank_all_gather=coll.get('single_rank_all_gather'), note_rccl=coll.get('note'),
                                  seconds_sample_plus_metrics=eval_s, clips=Btot, means=means,
                                  sharding='every rank draws the whole batch\'s noise at its clips\' global position (shard=): a sharded run equals the unsharded one bit for bit (tests)',
                                  note='random-init denoiser: the metric values only serve as parity evidence against the oracle')
    line.update(extra)
    if prof:
        split = getattr(model, 'ffn_math', 'exact') == 'split'
        dom = 'ffn_fused'
        us = dom_us
        rows = B_PER_GPU * T
        flops = FFN_FLOP_PER_TOKEN * rows                     # algorithmic fp32 FLOP of one launch (1 048 576 per token)
        traffic, traffic_src = None, None
        recorded = None                                      # RECORDED, not measured by this run: the committed rocprofv3 summary of this same command
        tj = {}
        try:
            tf = os.path.join(ROOT, 'profiles', 'traffic.json')
            if os.path.exists(tf):
                tj = json.load(open(tf))
                if tj.get('kernel_id') == DOMINANT_KERNEL_ID:      # a PMC figure is only valid for the kernel build it was taken on
                    traffic, traffic_src = tj.get(dom), tj.get('_how')
                else:
                    tj = {}
        except Exception as e:                               # a stale or reformatted profile must never take the measurement down
            traffic_src, tj = 'profiles/traffic.json unreadable: %r' % (e,), {}
        if split:
            # The shipped kernel issues v_mfma_f32_16x16x32_f16 (three per product): its matrix roof is the f16 dense peak, and what it is priced with is the
            # f16 FLOP it ISSUES (zero-padded hidden units / K steps included: they occupy the pipe like the rest) -- not the fp32-equivalent work, which
            # would be priced against a pipe this kernel no longer uses.
            issued = ffn_issued_f16_flop(rows, 32)
            eq = flops / (us * 1e-6) / 1e12                  # ALGORITHMIC fp32-grade FLOP per second (SURVEY.md 8(d) x rows / measured duration)
            peak = PEAK_F16_MFMA_TFLOPS / 3                  # the roof an fp32-grade product has on the pipe the kernel issues on: f16 dense peak / 3 products
            ach = eq
            iss = issued / (us * 1e-6) / 1e12                # secondary: f16 FLOP the kernel ISSUES (3 products + zero padding of hidden slices / K steps) vs the f16 dense peak
            frac_of = lambda rows_, tile_, us_: FFN_FLOP_PER_TOKEN * rows_ / (us_ * 1e-6) / 1e12 / peak
            r_ = tj.get('rocprofv3_in_situ_us')
            if r_:
                recorded = dict(us_per_launch=r_, achieved=flops / (r_ * 1e-6) / 1e12, frac=flops / (r_ * 1e-6) / 1e12 / peak, recorded_not_measured=True, source=ROCPROF_STATS)
            stream_bytes = 5 * 442368                        # packed weight stream of one layer (what every XCD pulls through its L2 once per launch)
            line['roofline'] = dict(
                bound='mfma', kernel=dom, kernel_id=DOMINANT_KERNEL_ID, achieved=ach, peak=peak, unit='TFLOP/s', frac=ach / peak,
                traffic=traffic, us_per_launch=us, us_per_launch_best=dom_best,
                algorithmic_flop_per_launch=flops, rocprofv3_in_situ=recorded,
                peak_note='achieved = the block\'s ALGORITHMIC fp32 FLOP per launch (2 x 2*M*256*1024 = 1 048 576 per token x M rows, SURVEY.md 8(d)) / measured duration; peak = the roof an fp32-grade '
                          'product has on the pipe this kernel issues on: it multiplies as three v_mfma_f32_16x16x32_f16 per product (csrc/ffn_h2.h), so f16 dense peak 2500 / 3 = 833 TFLOP/s '
                          '(MI355X_MICROARCH.md).  Same number as 3 x algorithmic FLOP / f16 peak.  Rounds 1-4 divided the same algorithmic FLOP by the fp32-input MFMA peak (157 TFLOP/s), a pipe the '
                          'kernel no longer uses; round 5\'s top-level figure counted the ISSUED f16 FLOP (zero padding included), which is now `issued_f16` below -- redundant MFMAs must not raise frac',
                issued_f16=dict(flop_per_launch=issued, mfma_per_launch=issued // 16384, achieved_tflops=iss, frac_of_f16_dense_peak=iss / PEAK_F16_MFMA_TFLOPS,
                                note='f16 FLOP the kernel issues per launch: (13 hidden tiles x 8 K steps + 16 output tiles x 7 K steps) x 2 token tiles x 3 products x 16 384 x 250 workgroups, zero padding '
                                     '(1024 -> 5 x 208 hidden units, K 208 -> 224) included; = the share of the launch during which the matrix pipes issue'),
                mfma_busy_share=dict(analytic=iss / PEAK_F16_MFMA_TFLOPS, recorded_pmc=tj.get('mfma_busy_share'),
                                     note='SQ_VALU_MFMA_BUSY_CYCLES per launch / (1024 SIMDs x launch cycles); analytic = issued MFMAs x 16 cycles over the same denominator'),
                binding_resource=dict(what='the LDS-DMA weight stream (every workgroup pulls its slice\'s 432 KiB from its XCD\'s L2 at the ~50 B/clk a CU reaches) plus fixed phases outside the K loops '
                                           '(row fetch + split, GELU + split, staging, the 8-MB slab store burst): DESIGN.md 4.2 phase stamps',
                                      weight_stream_bytes_per_workgroup=stream_bytes // 5, weight_stream_bytes_per_layer=stream_bytes,
                                      weight_stream_gb_per_s_per_cu=(stream_bytes // 5) / (us * 1e-6) / 1e9,
                                      l2_fed_dma_ceiling_gb_per_s_per_cu=50 * 2.4, note='ceiling: tools/dma_ceiling.hip (profiles/r02_dma_ceiling.txt), ~50 B/clk/CU at 2.4 GHz'),
                fp32_input_mfma_peak_tflops=PEAK_F32_MFMA_TFLOPS,      # (orientation only: the arithmetic left that pipe, a ratio to it is not a roofline fraction)
                exact_fp32_kernel=dict(us_per_launch=exact_us, bound='mfma', achieved=flops / (exact_us * 1e-6) / 1e12, peak=PEAK_F32_MFMA_TFLOPS,
                                       frac=flops / (exact_us * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS, denoiser_forward_us=exact_fwd_us,
                                       kernel='idf_ffn::ffn_fused_kernel (v_mfma_f32_16x16x4_f32; csrc/ffn.h), selectable with MDM.ffn_math = "exact" / INTERDIFF_FFN_MATH=exact',
                                       note='same measurement recipe, same process; this kernel DOES issue the fp32-input MFMA, so its roof is that peak') if exact_us else None,
                one_layer_burst=dict(us_per_launch=burst_us, us_per_launch_best=burst_best, frac=frac_of(rows, 32, burst_us),
                                     note='back-to-back launches on ONE layer (weights stay in the L2s): optimistic; secondary'),
                two_chain_form=dict(us_per_pair=pair_us, us_per_pair_best_burst=pair_best, frac=flops / (pair_us * 1e-6) / 1e12 / peak,
                                    note='NOT the route of this shape (one chain up to one round of workgroups, MDM.one_chain_max_rows); kept for comparison: '
                                         'the same layer as two concurrent launches at M=%d on two graph branches' % (rows // 2)),
                small_batch_16_row_tile=dict(rows=800, us_per_launch=small_us, us_per_launch_best=small_best, frac=frac_of(800, 16, small_us),
                                             note='the 16-row tile batches of <= 800 token rows take (8 clips of 100 frames = BASELINE config #4\'s share of a GPU)'),
                large_batch_64_row_tile=dict(rows=3200, us_per_launch=big_us, us_per_launch_best=big_best, frac=frac_of(3200, 64, big_us),
                                             note='the 64-row tile BASELINE config #3 (32 clips of 100 frames) takes'),
                traffic_source=traffic_src or 'null: profiles/traffic.json holds no rocprofv3 FETCH_SIZE/WRITE_SIZE passes for this kernel build',
                traffic_vs_algorithmic=dict(algorithmic_bytes=5493824, note='x2 1.64 MB + packed weight planes 2.21 MB + output 1.64 MB; the design writes the output as five partial slabs (8.19 MB) '
                                                                          'and every XCD pulls the weight stream into its own L2 once (8 x 2.2 MB)') if traffic else None,
                note='one launch = linear1 + gelu + linear2 of a layer at M=%d; 8 of the 21 launches of a chained plain step; duration = mean of three bursts of '
                     '192 launches replayed from a hipGraph that walk through the eight layers\' weight streams and alternate activation buffers like a '
                     'denoiser step (every weight stream from the Infinity Cache), HIP events on the launch stream; the rocprofv3 in-situ average of the same '
                     'command is committed under profiles/ (rocprofv3_in_situ, recorded)' % rows)
        else:
            ach = flops / (us * 1e-6) / 1e12
            line['roofline'] = dict(bound='mfma', kernel=dom, kernel_id='idf_ffn::ffn_fused_kernel (exact fp32 MFMA selected)', achieved=ach, peak=PEAK_F32_MFMA_TFLOPS, unit='TFLOP/s',
                                    frac=ach / PEAK_F32_MFMA_TFLOPS, traffic=None, us_per_launch=us, us_per_launch_best=dom_best, algorithmic_flop_per_launch=flops,
                                    note='exact-fp32 arithmetic selected (INTERDIFF_FFN_MATH=exact): v_mfma_f32_16x16x4_f32, priced against the fp32-input MFMA peak')
        # the whole step against the chip: what the matrix pipes and the memory system would need for one step's algorithmic work, next to what a step takes
        step_us = 1e6 * wall / STEPS
        alg_flop = FLOP_PER_TOKEN * rows
        exe_flop = EXECUTED_FLOP_PER_TOKEN(T) * rows
        matrix_roof = PEAK_F16_MFMA_TFLOPS / 3 if split else PEAK_F32_MFMA_TFLOPS
        w_bytes = STEP_WEIGHT_BYTES(B_PER_GPU)
        a_bytes = STEP_ACTIVATION_BYTES(rows)
        mem_us = (w_bytes + a_bytes) / MALL_STREAM_TBPS / 1e6
        mat_us = alg_flop / matrix_roof / 1e6
        line['step_roofline'] = dict(
            us_per_step=step_us, algorithmic_flop_per_step=alg_flop, executed_flop_per_step=exe_flop,
            matrix_roof=dict(tflops=matrix_roof, us=mat_us, frac_of_step=mat_us / step_us,
                             note='algorithmic FLOP of one step (SURVEY.md 8(d): 11 978 752 + 2048 T per token) / the fp32-grade roof of the f16 pipe (f16 dense peak / 3 products)' if split else
                                  'algorithmic FLOP of one step / the fp32-input MFMA peak'),
            memory_roof=dict(weight_and_constant_bytes=w_bytes, activation_handover_bytes=a_bytes, stream_tb_per_s=MALL_STREAM_TBPS, us=mem_us, frac_of_step=mem_us / step_us,
                             note='bytes every step must move at least once: the packed weights + per-sample folded memory (read once per step, Infinity-Cache resident) and every '
                                  'kernel-to-kernel hand-over written once and read once ([M,256] fp32 rows; QKV 3x); rate = the Infinity-Cache stream rate measured for private streams '
                                  '(tools/dma_ceiling.hip: 12 B/clk/CU x 256 CUs x 2.4 GHz)'),
            frac=max(mat_us, mem_us) / step_us,
            gap_note='a step takes %.1fx its binding roof: 21 strictly dependent launches (~1.7 us boundary each), each a fetch -> compute -> store latency chain on <= 250 workgroups, plus the '
                     'amortised hook (11 corrected steps per 1000); DESIGN.md 4' % (step_us / max(mat_us, mem_us)))
        line['arithmetic_by_layer'] = model.arithmetic_report() if hasattr(model, 'arithmetic_report') else None
        try:
            txt, bad = _lib.exclusive_cu_report()
            line['exclusive_cu'] = dict(kernels_not_exclusive=bad, table=txt.strip().split('\n'),
                                        note='every kernel that issues the f16 MFMA must own its CU (verified at launch: occupancy query == 1, LDS == 160 KiB, >= 256 registers); a kernel that fails runs as its fp32 counterpart')
        except Exception as e:
            line['exclusive_cu'] = dict(error=repr(e))
        fl = FLOP_PER_TOKEN * B_PER_GPU * T
        line['denoiser_forward'] = dict(us=fwd_us, achieved_tflops=fl / (fwd_us * 1e-6) / 1e12,
                                        frac_of_matrix_roof=fl / (fwd_us * 1e-6) / 1e12 / (PEAK_F16_MFMA_TFLOPS / 3 if split else PEAK_F32_MFMA_TFLOPS),
                                        matrix_roof_tflops=(PEAK_F16_MFMA_TFLOPS / 3 if split else PEAK_F32_MFMA_TFLOPS), launches=22,
                                        how='MDM.forward (22 launches, B=%d T=%d) replayed from a hipGraph, HIP events on its stream' % (B_PER_GPU, T))
        line['kernels_us_event_to_event'] = {k: round(v['us_avg'], 2) for k, v in prof.items()}    # includes the launch gap + event records
    line['conditioning_ms_per_sample'] = enc_ms        # MDM._get_embeddings, outside the timed region (once per 1000 steps)
    if post:
        line['post_optimisation'] = post                    # "next" row N4 (optimization.py), outside the timed region
    if cpu:
        if 'no_correction' in extra:                         # GPU / CPU on the denoiser alone, next to the blended ratio
            cpu['plain_only']['gpu_over_cpu'] = extra['no_correction']['value'] / cpu['plain_only']['value']
            cpu['headline_gpu_over_cpu'] = dict(value=cpu['plain_only']['gpu_over_cpu'], of='plain denoising steps (the denoiser alone): the quotable figure -- 71 % of the blended CPU time is the '
                                                'oracle\'s own brute-force nearest-neighbour search in the 11 correction calls')
        cpu['gpu_over_cpu_blended'] = line['value'] / cpu['value']
        try:                                                 # RECORDED in the build container (tools/cpu_reference_vs_port.py; /root/reference cannot travel to this box): is the port slower than the source it restates?
            pr = json.load(open(os.path.join(ROOT, 'profiles', 'r06_cpu_reference_vs_port.json')))
            cpu['port_vs_reference'] = dict(plain_step=pr['port_vs_reference']['plain_step'], correction_call=pr['port_vs_reference']['correction_call'], threads=pr['threads'],
                                            reference_plain_step_s=pr['plain_step_s']['reference'], port_plain_step_s=pr['plain_step_s']['port'], recorded_not_measured=True,
                                            source='profiles/r06_cpu_reference_vs_port.json',
                                            note='seconds of the oracle / seconds of the reference\'s own source on the same cores, same inputs (8 threads, build container): ~1.0 -- the port neither '
                                                 'flatters nor penalises the CPU side; a GPU/CPU ratio against the reference itself = the ratio above / this number')
        except Exception as e:
            cpu['port_vs_reference'] = dict(error=repr(e))
        line['cpu_baseline'] = cpu
    sys.stdout.flush()
    C.CDLL(None).fflush(None)                          # whatever C stdio still holds goes to stderr, not behind the JSON
    os.write(json_fd, (json.dumps(line) + '\n').encode())


if __name__ == '__main__':
    main()
