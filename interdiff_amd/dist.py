"""Multi-GPU layer: clips are independent through the whole path, so the batch dimension is
sharded over ranks (one process per GPU), weights / body model are replicated, and the ONLY collective is one
all-gather of the six per-clip metric vectors per eval batch (RCCL over xGMI: backend "nccl" on ROCm; payload
<= 6*B_local*4 bytes, latency-bound).  ``gloo`` exercises the same code on CPU in the tests."""
import os
import torch
import torch.distributed as dist

METRIC_KEYS = ('global_mpjpe', 'local_mpjpe', 'body_translation', 'obj_translation', 'obj_rot_error', 'penetrate')


def init_from_env(backend=None, single_rank_group=False):
    """torchrun-style env (RANK, WORLD_SIZE, LOCAL_RANK, MASTER_ADDR, MASTER_PORT). Returns (rank, world, local_rank).
    ``single_rank_group``: create the process group even for world == 1 (MASTER_ADDR / MASTER_PORT default to 127.0.0.1 / 29500), so
    that one GPU can push the path's collective through RCCL for real (tests, bench.py)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world == 1 and single_rank_group:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
    if (world > 1 or single_rank_group) and not dist.is_initialized():
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


def gather_metrics(local, world, counts=None, return_header=False, force_collective=False, check_header=False):
    """local: dict name -> [B_local] tensor.  Returns dict name -> [B_total] (rank order) on every rank.

    ONE fixed-size all-gather and no host synchronisation: every rank's shard size is known by construction (``counts``: the
    ``shard_slice`` sizes of the global batch -- REQUIRED when world > 1: a rank cannot know the others' sizes, and mismatched
    buffer sizes in an all-gather hang RCCL), so the buffer is [7, max(counts)] floats -- row 0 carries the sender's own count in slot
    0 (a header the receiver can verify: ``return_header`` hands the device tensor [world] back, ``check_header`` compares it with
    ``counts`` on the host -- a sync, for tests and debugging, never on the hot path), rows 1..6 the six metric vectors, zero padded.
    ``force_collective``: issue the all-gather even in a one-rank group (so that a single GPU exercises RCCL for real)."""
    stacked = torch.stack([local[k].float() for k in METRIC_KEYS])                 # [6, B_local]
    n = stacked.shape[1]
    group = dist.is_available() and dist.is_initialized()
    if not group or (world == 1 and not force_collective):
        out = {k: stacked[i] for i, k in enumerate(METRIC_KEYS)}
        return (out, torch.tensor([float(n)], device=stacked.device)) if return_header else out
    if dist.get_world_size() != world:
        raise ValueError('world = %d but the process group has %d ranks' % (world, dist.get_world_size()))
    if counts is None:
        if world > 1:
            raise ValueError('gather_metrics needs every rank\'s shard size (counts) when world > 1')
        counts = [n]
    counts = [int(c) for c in counts]
    cap = max(max(counts), 1)
    if len(counts) != world or counts[dist.get_rank()] != n:
        raise ValueError('counts must list every rank\'s shard size (this rank holds %d, counts = %r)' % (n, counts))
    buf = torch.zeros(7, cap, device=stacked.device)
    buf[0, 0] = float(n)
    buf[1:, :n] = stacked
    bufs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf)                                                      # the one collective of the path
    header = torch.stack([b[0, 0] for b in bufs])
    if check_header and header.tolist() != [float(c) for c in counts]:
        raise RuntimeError('shard sizes disagree across ranks: headers %r, counts %r' % (header.tolist(), counts))
    full = torch.cat([b[1:, :c] for b, c in zip(bufs, counts)], dim=1)
    out = {k: full[i] for i, k in enumerate(METRIC_KEYS)}
    return (out, header) if return_header else out


def collective_backend_info():
    """What the all-gather runs on: backend name of the default group and, for "nccl" on ROCm, the RCCL version torch was built against
    (bench.py records it next to the measurement)."""
    info = {'backend': dist.get_backend() if (dist.is_available() and dist.is_initialized()) else None, 'rccl_version': None}
    try:
        if torch.cuda.is_available():
            v = torch.cuda.nccl.version()
            info['rccl_version'] = '.'.join(str(x) for x in v) if isinstance(v, tuple) else str(v)
    except Exception as e:                       # informational only
        info['rccl_version'] = 'unavailable (%s)' % type(e).__name__
    return info


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
