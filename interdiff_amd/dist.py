"""Multi-GPU layer: clips are independent through the whole path, so the batch dimension is
sharded over ranks (one process per GPU), weights / body model are replicated, and the ONLY collective is one
all-gather of the six per-clip metric vectors per eval batch (RCCL over xGMI: backend "nccl" on ROCm; payload
<= 6*B_local*4 bytes, latency-bound).  ``gloo`` exercises the same code on CPU in the tests."""
import os
import torch
import torch.distributed as dist

METRIC_KEYS = ('global_mpjpe', 'local_mpjpe', 'body_translation', 'obj_translation', 'obj_rot_error', 'penetrate')


def init_from_env(backend=None):
    """torchrun-style env (RANK, WORLD_SIZE, LOCAL_RANK, MASTER_ADDR, MASTER_PORT). Returns (rank, world, local_rank)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def get_rank_world():
    """(rank, world) of the initialised process group, (0, 1) without one."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_slice(n_items, rank, world):
    """Contiguous, balanced shard of n_items for this rank."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return slice(start, start + base + (1 if rank < rem else 0))


def shard_batch(batch, rank, world, batch_dims):
    """Slice every tensor of `batch` along its clip dimension (batch_dims: name -> dim)."""
    out = {}
    for k, v in batch.items():
        if isinstance(v, torch.Tensor) and k in batch_dims:
            sl = shard_slice(v.shape[batch_dims[k]], rank, world)
            out[k] = v.narrow(batch_dims[k], sl.start, sl.stop - sl.start).contiguous()
        else:
            out[k] = v
    return out


def gather_metrics(local, world, counts=None, return_header=False):
    """local: dict name -> [B_local] tensor.  Returns dict name -> [B_total] (rank order) on every rank.

    ONE fixed-size all-gather and no host synchronisation: every rank's shard size is known by construction (``counts``: the
    ``shard_slice`` sizes of the global batch; None = equal shards of this rank's size), so the buffer is [7, max(counts)] floats --
    row 0 carries the sender's own count in slot 0 (a header the receiver can verify, ``return_header``: device tensor [world], never
    read on the hot path), rows 1..6 the six metric vectors, zero padded."""
    stacked = torch.stack([local[k].float() for k in METRIC_KEYS])                 # [6, B_local]
    n = stacked.shape[1]
    if world == 1 or not dist.is_initialized():
        out = {k: stacked[i] for i, k in enumerate(METRIC_KEYS)}
        return (out, torch.tensor([float(n)], device=stacked.device)) if return_header else out
    counts = [n] * world if counts is None else [int(c) for c in counts]
    cap = max(max(counts), 1)
    if len(counts) != world or n > cap:
        raise ValueError('counts must list every rank\'s shard size')
    buf = torch.zeros(7, cap, device=stacked.device)
    buf[0, 0] = float(n)
    buf[1:, :n] = stacked
    bufs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf)                                                      # the one collective of the path
    full = torch.cat([b[1:, :c] for b, c in zip(bufs, counts)], dim=1)
    out = {k: full[i] for i, k in enumerate(METRIC_KEYS)}
    return (out, torch.stack([b[0, 0] for b in bufs])) if return_header else out


def gather_scalar(value, device):
    """Every rank's float, in rank order, on every rank (bench.py: per-rank wall times, so that a straggler shows)."""
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if not dist.is_initialized():
        return [float(value)]
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def max_over_ranks(value, device):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized():
        dist.barrier()


def shutdown():
    """Tear the process group down (after the last collective): ranks may then finish at different times."""
    if dist.is_initialized():
        try:
            dist.destroy_process_group()
        except Exception as e:                       # teardown only: never let it take a finished measurement down
            print('[interdiff_amd.dist] destroy_process_group: %r' % (e,), flush=True)


def _selftest_worker(rank, world, port):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w, _ = init_from_env('gloo')
    B = 5                                                                           # uneven on purpose
    full = {k: torch.arange(B, dtype=torch.float32) + 10 * i for i, k in enumerate(METRIC_KEYS)}
    sl = shard_slice(B, r, w)
    counts = [shard_slice(B, q, w).stop - shard_slice(B, q, w).start for q in range(w)]
    got, header = gather_metrics({k: v[sl] for k, v in full.items()}, w, counts=counts, return_header=True)
    for k in METRIC_KEYS:
        assert torch.equal(got[k], full[k]), (k, got[k], full[k])
    assert header.tolist() == [float(c) for c in counts]                            # every sender's own count travelled in its header slot
    assert gather_scalar(r + 0.25, 'cpu') == [q + 0.25 for q in range(w)]
    batch = shard_batch({'gt': torch.arange(B * 3).reshape(B, 3), 'cond': torch.arange(2 * B).reshape(2, B), 'past_len': 10},
                        r, w, {'gt': 0, 'cond': 1})
    assert batch['gt'].shape[0] == sl.stop - sl.start and batch['cond'].shape[1] == sl.stop - sl.start and batch['past_len'] == 10
    assert max_over_ranks(r + 1.5, 'cpu') == w + 0.5
    barrier()
    shutdown()
    assert not dist.is_initialized()


def _selftest_eval_worker(rank, world, port):
    """evaluate_sharded on CPU ranks (gloo): the sampler + metrics of ``evaluate_batch`` need the GPU, so a stand-in scores every
    clip with numbers that identify the clip, the seed it was handed and the rank -- what is under test is the shard, the seed
    offsets, the ONE all-gather and the per-clip order of the collated vectors."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from . import eval as ev
    r, w, _ = init_from_env('gloo')
    B, T, P, div = (5 if world <= 4 else 64), 12, 8, 3               # 2 ranks: uneven shards on purpose; 8 ranks: BASELINE config #4's 64 clips = 8 per rank
    batch = dict(gt=torch.arange(B, dtype=torch.float32)[:, None, None, None].expand(B, 1, 144, T).contiguous(),
                 cond=torch.zeros(10, B, 256), hand_pose=torch.zeros(T, B, 90), beta=torch.zeros(T, B, 10), obj_points=torch.zeros(B, P, 3))
    calls = []

    def fake_evaluate_batch(model, diffusion, correction, local, past_len, mode, diverse_samples, seed=None, **kw):
        calls.append((local['gt'].shape[0], seed, diverse_samples))
        assert local['cond'].shape[1] == local['hand_pose'].shape[1] == local['beta'].shape[1] == local['obj_points'].shape[0] == local['gt'].shape[0]
        clip = local['gt'][:, 0, 0, 0]
        return {k: clip * 10 + i + (0.001 * seed if k == 'penetrate' else 0.0) for i, k in enumerate(METRIC_KEYS)}
    real, ev.evaluate_batch = ev.evaluate_batch, fake_evaluate_batch
    try:
        full, means = ev.evaluate_sharded(None, None, None, batch, 10, 'correction', div, seed=100)
    finally:
        ev.evaluate_batch = real
    sl = shard_slice(B, r, w)
    assert calls == [(sl.stop - sl.start, 100 + sl.start * div, div)], calls
    clip = torch.arange(B, dtype=torch.float32)
    seeds = torch.tensor([100 + shard_slice(B, q, w).start * div for q in range(w) for _ in range(shard_slice(B, q, w).stop - shard_slice(B, q, w).start)],
                         dtype=torch.float32)
    for i, k in enumerate(METRIC_KEYS):
        want = clip * 10 + i + (0.001 * seeds if k == 'penetrate' else 0.0)
        assert torch.allclose(full[k], want), (k, full[k], want)
        assert abs(means[k] - float(want.mean())) < 1e-5
    # a pre-sharded batch (bench.py builds its clips per rank) goes through the same collective
    mine = shard_batch(batch, r, 1, {})                             # whole batch as "this rank's clips"
    ev.evaluate_batch = fake_evaluate_batch
    try:
        full2, _ = ev.evaluate_sharded(None, None, None, mine, 10, 'correction', 1, seed=7, presharded=True)
    finally:
        ev.evaluate_batch = real
    assert full2['global_mpjpe'].numel() == B * w
    barrier()
    shutdown()


def _selftest_long_worker(rank, world, port):
    """eval.sample_long_sharded on CPU ranks (gloo): BASELINE config #4's partitioning -- every rank rolls its own clips out, no
    exchange.  Stand-ins replace the GPU pieces (conditioning, one sampled window) by per-clip deterministic functions; what is
    under test is the clip shard of every raw tensor, the per-rank seed offset, the window algebra on a shard and that the shards'
    results are exactly the corresponding clips of the unsharded rollout."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from . import eval as ev
    r, w, _ = init_from_env('gloo')
    B, T, past, P, K = 64, 14, 10, 8, 2
    g = torch.Generator().manual_seed(1)
    raw = dict(body_pose=torch.randn(T, B, 66, generator=g), hand_pose=torch.randn(T, B, 90, generator=g), body_trans=torch.randn(T, B, 3, generator=g),
               obj_angles=torch.randn(T, B, 3, generator=g), obj_trans=torch.randn(T, B, 3, generator=g), beta=torch.randn(T, B, 10, generator=g),
               obj_points=torch.randn(B, P, 3, generator=g))

    def fake_batch_from_raw(model, rw, past_len=10):
        return dict(rw, gt=rw['body_trans'].permute(1, 2, 0)[:, None].contiguous())

    def fake_sample(model, diffusion, correction, bt, past_len, noise=None, seed=None, **kw):
        # "prediction" = a per-clip function of the window's inputs (so that a wrong shard or a wrong re-centring shows)
        Tn, Bn = bt['body_pose'].shape[:2]
        drift = torch.arange(Tn, dtype=torch.float32)[:, None, None] * 0.01
        body = torch.cat([bt['body_pose'] * 0.5, bt['hand_pose'], bt['body_trans'] + drift], dim=2)
        obj = torch.cat([bt['obj_angles'] * 0.5, bt['obj_trans'] - drift], dim=2)
        pelvis = bt['body_trans'] + 0.1
        verts = pelvis[:, :, None, :].expand(Tn, Bn, 3, 3) + 0.0
        return obj, body, verts, verts.clone(), pelvis
    keep = ev.batch_from_raw, ev.sample_once_proj, ev._x_T
    ev.batch_from_raw, ev.sample_once_proj, ev._x_T = fake_batch_from_raw, fake_sample, (lambda gt, sd: None)

    class Corr:
        smpl = None
    try:
        whole = ev.sample_long(None, None, Corr(), raw, K, past, seed=5)
        sl, mine = ev.sample_long_sharded(None, None, Corr(), raw, K, past, seed=5)
    finally:
        ev.batch_from_raw, ev.sample_once_proj, ev._x_T = keep
    assert (sl.start, sl.stop) == (8 * r, 8 * r + 8)
    for a, b in zip(whole, mine):
        assert a.shape[0] == T + K * (T - past) and torch.equal(a[:, sl], b), (a.shape, b.shape)
    barrier()
    shutdown()
