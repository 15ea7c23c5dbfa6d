"""Worker functions of the multi-process ``gloo`` tests (tests/test_abi_and_host.py): module-level so that ``mp.spawn`` can pickle them.
They exercise interdiff_amd.dist / eval on CPU ranks; the GPU pieces are replaced by per-clip deterministic stand-ins."""
import os
import sys
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from interdiff_amd.dist import (METRIC_KEYS, init_from_env, shard_slice, shard_batch, gather_metrics, gather_scalar,        # noqa: E402
                                max_over_ranks, barrier, shutdown)


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
    from interdiff_amd import eval as ev
    r, w, _ = init_from_env('gloo')
    B, T, P, div = (5 if world <= 4 else 64), 12, 8, 3               # 2 ranks: uneven shards on purpose; 8 ranks: BASELINE config #4's 64 clips = 8 per rank
    batch = dict(gt=torch.arange(B, dtype=torch.float32)[:, None, None, None].expand(B, 1, 144, T).contiguous(),
                 cond=torch.zeros(10, B, 256), hand_pose=torch.zeros(T, B, 90), beta=torch.zeros(T, B, 10), obj_points=torch.zeros(B, P, 3))
    calls = []

    def fake_evaluate_batch(model, diffusion, correction, local, past_len, mode, diverse_samples, seed=None, shard=None, **kw):
        calls.append((local['gt'].shape[0], seed, diverse_samples, shard))
        assert local['cond'].shape[1] == local['hand_pose'].shape[1] == local['beta'].shape[1] == local['obj_points'].shape[0] == local['gt'].shape[0]
        clip = local['gt'][:, 0, 0, 0]
        return {k: clip * 10 + i + (0.001 * seed if k == 'penetrate' else 0.0) for i, k in enumerate(METRIC_KEYS)}
    real, ev.evaluate_batch = ev.evaluate_batch, fake_evaluate_batch
    try:
        full, means = ev.evaluate_sharded(None, None, None, batch, 10, 'correction', div, seed=100)
    finally:
        ev.evaluate_batch = real
    sl = shard_slice(B, r, w)
    # every rank runs the draws under the SAME base seed, its clips at their global position (shard): the sharded run is the
    # unsharded one bit for bit (the GPU side of that statement: tests/test_hip_parity.py test_emulated_ranks_equal_unsharded_*)
    assert calls == [(sl.stop - sl.start, 100, div, (sl.start, B))], calls
    clip = torch.arange(B, dtype=torch.float32)
    seeds = torch.full((B,), 100.0)
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
    assert full2['global_mpjpe'].numel() == B * w and calls[-1] == (B, 7, 1, (r * B, w * B))
    # ragged shards without counts must be refused, not hang the collective
    try:
        gather_metrics({k: torch.zeros(1 + r) for k in METRIC_KEYS}, w)
        raise AssertionError('gather_metrics accepted world > 1 without counts')
    except ValueError:
        pass
    barrier()
    shutdown()


def _selftest_long_worker(rank, world, port):
    """eval.sample_long_sharded on CPU ranks (gloo): BASELINE config #4's partitioning -- every rank rolls its own clips out, no
    exchange.  Stand-ins replace the GPU pieces (conditioning, one sampled window) by per-clip deterministic functions; what is
    under test is the clip shard of every raw tensor, the per-rank seed offset, the window algebra on a shard and that the shards'
    results are exactly the corresponding clips of the unsharded rollout."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from interdiff_amd import eval as ev
    r, w, _ = init_from_env('gloo')
    B, T, past, P, K = 64, 14, 10, 8, 2
    g = torch.Generator().manual_seed(1)
    raw = dict(body_pose=torch.randn(T, B, 66, generator=g), hand_pose=torch.randn(T, B, 90, generator=g), body_trans=torch.randn(T, B, 3, generator=g),
               obj_angles=torch.randn(T, B, 3, generator=g), obj_trans=torch.randn(T, B, 3, generator=g), beta=torch.randn(T, B, 10, generator=g),
               obj_points=torch.randn(B, P, 3, generator=g))

    def fake_batch_from_raw(model, rw, past_len=10, batch_clips=None):
        assert batch_clips in (None, B)
        return dict(rw, gt=rw['body_trans'].permute(1, 2, 0)[:, None].contiguous())

    def fake_sample(model, diffusion, correction, bt, past_len, noise=None, seed=None, shard=None, **kw):
        seen.append((seed, shard))
        # "prediction" = a per-clip function of the window's inputs (so that a wrong shard or a wrong re-centring shows)
        Tn, Bn = bt['body_pose'].shape[:2]
        drift = torch.arange(Tn, dtype=torch.float32)[:, None, None] * 0.01
        body = torch.cat([bt['body_pose'] * 0.5, bt['hand_pose'], bt['body_trans'] + drift], dim=2)
        obj = torch.cat([bt['obj_angles'] * 0.5, bt['obj_trans'] - drift], dim=2)
        pelvis = bt['body_trans'] + 0.1
        verts = pelvis[:, :, None, :].expand(Tn, Bn, 3, 3) + 0.0
        return obj, body, verts, verts.clone(), pelvis
    keep = ev.batch_from_raw, ev.sample_once_proj, ev._x_T
    ev.batch_from_raw, ev.sample_once_proj, ev._x_T = fake_batch_from_raw, fake_sample, (lambda gt, sd, shard=None: None)
    seen = []

    class Corr:
        smpl = None
    try:
        whole = ev.sample_long(None, None, Corr(), raw, K, past, seed=5)
        n_whole = len(seen)
        sl, mine = ev.sample_long_sharded(None, None, Corr(), raw, K, past, seed=5)
    finally:
        ev.batch_from_raw, ev.sample_once_proj, ev._x_T = keep
    assert (sl.start, sl.stop) == (8 * r, 8 * r + 8)
    assert [sd for sd, _ in seen[:n_whole]] == [sd for sd, _ in seen[n_whole:]] == [5 + k for k in range(K + 1)]          # same seeds as the unsharded rollout ...
    assert all(sh is None for _, sh in seen[:n_whole]) and all(sh == (8 * r, B) for _, sh in seen[n_whole:])                 # ... at the shard's global position
    for a, b in zip(whole, mine):
        assert a.shape[0] == T + K * (T - past) and torch.equal(a[:, sl], b), (a.shape, b.shape)
    barrier()
    shutdown()
