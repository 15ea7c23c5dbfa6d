"""Evaluation glue of eval_smpl_short.py (rows E1/E2 of SURVEY.md §8) on the HIP path.

Mirrors ``sample_once_proj`` (:133-177), ``sample_once`` (:179-215), ``get_gt`` (:225-250), ``metrics`` (:24-81) and
``smooth`` (:217-223).  The reference feeds a dataset batch (dict of per-frame lists) through the encoder
(``_get_embeddings``: a "next" row); here the clip batch is the tensor schema of SURVEY.md §8(d):

    gt [B,1,144,T]   cond [10,B,256]   hand_pose [T,B,90] (GT hands, NOT padded)   beta [T,B,10]   obj_points [B,P,3]

torch is used for views / concatenation; every arithmetic step of the scored path (rot6d -> axis-angle, SMPL, nearest
neighbours, the metric reductions) runs in libinterdiff_hip.so.  The only tensor arithmetic left to torch is a handful of
elementwise adds outside it: the rendering-only ``smooth`` and the translation re-centring between the windows of
``sample_long``.
"""
import ctypes as C
import torch
from . import _lib, dist, io, transforms as tr
from .dist import METRIC_KEYS

SMPL_DIM = 132          # 22 joints x rot6d (eval_smpl_short.py:416)


def idx_pad(past_len, T):
    """eval_smpl_short.py:407."""
    return list(range(past_len)) + [past_len - 1] * (T - past_len)


def split_tokens(x):
    """[B,1,144,T] -> body [T,B,135], obj [T,B,9]  (:154-155)."""
    xt = x.squeeze(1).permute(2, 0, 1).contiguous()
    return xt[..., :SMPL_DIM + 3], xt[..., SMPL_DIM + 3:]


def model_kwargs_for(batch, past_len):
    """model_kwargs['y'] as sample_once_proj builds it (:136-150), minus 'smpl'/'obj_model' which the HipCorrection
    object owns on this path."""
    gt = batch['gt']
    T = gt.shape[-1]
    mask = torch.ones_like(gt, dtype=torch.bool)
    mask[..., past_len:] = False
    return dict(cond=batch['cond'], inpainted_motion=gt, inpainting_mask=mask,
                hand_pose=batch['hand_pose'][idx_pad(past_len, T)].contiguous(), beta=batch['beta'], obj_points=batch['obj_points'])


def _to_pose(tokens_body, tokens_obj, hands, smpl, beta):
    """rot6d tokens -> (obj [T,B,6] axis-angle|trans, body [T,B,159], verts, jtr) (:156-173)."""
    T, B, _ = tokens_body.shape
    body_rot = tr.rotation_6d_to_axis_angle(tokens_body[..., :SMPL_DIM].reshape(T, B, -1, 6)).reshape(T, B, -1)
    obj_rot = tr.rotation_6d_to_axis_angle(tokens_obj[..., :6])
    body = torch.cat([body_rot, hands, tokens_body[..., -3:]], dim=2)
    flat = body.reshape(T * B, -1)
    verts, jtr, _, _ = smpl(flat[:, :-3], th_betas=beta.reshape(T * B, -1), th_trans=flat[:, -3:], want_v_posed=False)
    obj = torch.cat([obj_rot, tokens_obj[..., -3:]], dim=2)
    return obj, body, verts.reshape(T, B, -1, 3), jtr.reshape(T, B, -1, 3)


def finalize(sample, batch, smpl, past_len):
    body, obj = split_tokens(sample)
    T = body.shape[0]
    obj_pred, body_pred, verts, jtr = _to_pose(body, obj, batch['hand_pose'][idx_pad(past_len, T)], smpl, batch['beta'])
    return obj_pred, body_pred, verts, jtr, jtr[:, :, 0, :]


def _x_T(gt, seed, shard=None):
    """x_T ~ N(0, I) (:153 ``torch.randn``): from torch's global generator, or from ``seed`` when the caller wants a reproducible draw.
    ``shard=(first_clip, total_clips)``: ``gt`` is clips [first, first + B) of a larger batch; with a seed the WHOLE batch's tensor is
    drawn and this shard's slice returned ("sliced from one global tensor for parity with a 1-GPU run")."""
    if seed is None:
        return torch.randn(*gt.shape, device=gt.device)
    gen = torch.Generator(device=gt.device).manual_seed(int(seed))
    if shard is None or (shard[0] == 0 and shard[1] == gt.shape[0]):
        return torch.randn(*gt.shape, device=gt.device, generator=gen)
    first, total = shard
    return torch.randn(total, *gt.shape[1:], device=gt.device, generator=gen)[first:first + gt.shape[0]].contiguous()


def as_clip_batch(model, batch, past_len=10):
    """The reference hands its entry points the DataLoader's batch -- a dict of per-frame lists (data/dataset_smpl.py:182-204;
    ``sample_once_proj(batch)``, eval_smpl_short.py:133-150).  Such a batch goes through ``io.batch_from_dataset`` (the stacks of
    model/diffusion_smpl.py:197-201) and the HIP conditioning path (``batch_from_raw``); a clip batch of this module's tensor schema
    passes through."""
    if io.is_dataset_batch(batch):
        return batch_from_raw(model, io.batch_from_dataset(batch, device=model.device), past_len)
    return batch


def sample_once_proj(model, diffusion, correction, batch, past_len=10, noise=None, **loop_kw):
    """Full InterDiff: diffusion + correction hook.  Returns (obj_pred [T,B,6], body_pred [T,B,159], verts [T,B,V,3],
    jtr [T,B,J,3], pelvis [T,B,3]) like the reference (:177).  ``noise`` / ``step_noise`` / ``seed`` make it deterministic.
    ``batch``: a clip batch (module docstring) or the DataLoader's dict-of-lists batch, as the reference passes it."""
    batch = as_clip_batch(model, batch, past_len)
    gt = batch['gt']
    if noise is None:
        noise = _x_T(gt, loop_kw.get('seed'), loop_kw.get('shard'))
    sample = diffusion.p_sample_loop(model, tuple(gt.shape), clip_denoised=False, noise=noise,
                                     model_kwargs={'y': model_kwargs_for(batch, past_len)}, denoised_fn=correction, **loop_kw)
    return finalize(sample, batch, correction.smpl if correction is not None else loop_kw['smpl'], past_len)


def sample_once(model, diffusion, smpl, batch, past_len=10, noise=None, **loop_kw):
    """Diffusion only (mode no_correction, :179-215)."""
    batch = as_clip_batch(model, batch, past_len)
    gt = batch['gt']
    if noise is None:
        noise = _x_T(gt, loop_kw.get('seed'), loop_kw.get('shard'))
    y = model_kwargs_for(batch, past_len)
    sample = diffusion.p_sample_loop(model, tuple(gt.shape), clip_denoised=False, noise=noise, model_kwargs={'y': y}, **loop_kw)
    return finalize(sample, batch, smpl, past_len)


def get_gt(batch, smpl):
    """:225-250 -- GT hands are used as they are (no padding).  Returns (obj_gt, jtr_gt, body_gt, faces)."""
    body, obj = split_tokens(batch['gt'])
    obj_gt, body_gt, _, jtr = _to_pose(body, obj, batch['hand_pose'], smpl, batch['beta'])
    return obj_gt, jtr, body_gt, smpl.th_faces


def smooth(obj, body, verts, jtrs, pelvis, future_len):
    """:217-223 (rendering only; applied after the metrics)."""
    f = future_len
    out = []
    for a in (obj, body, verts, jtrs, pelvis):
        a[-f:] = a[-f:] + (2 * a[-f - 1] - a[-f - 2] - a[-f])
        out.append(a)
    return tuple(out)


def batch_from_raw(model, raw, past_len=10, batch_clips=None):
    """Dataset-side quantities -> the clip batch of this module, through the HIP conditioning path.
    raw: body_pose [T,B,66] axis-angle, hand_pose [T,B,90], body_trans [T,B,3], obj_angles [T,B,3] axis-angle,
    obj_trans [T,B,3], beta [T,B,10], obj_points [B,P,3].  ``batch_clips``: clips of the whole batch when ``raw`` is a shard of it."""
    cond, gt = model._get_embeddings(raw['body_pose'], raw['body_trans'], raw['obj_angles'], raw['obj_trans'], raw['obj_points'], past_len,
                                     **({} if batch_clips is None else {'batch_clips': batch_clips}))
    return dict(gt=gt.permute(1, 2, 0).unsqueeze(1).contiguous(), cond=cond, hand_pose=raw['hand_pose'].contiguous(),
                beta=raw['beta'].contiguous(), obj_points=raw['obj_points'].contiguous())


def next_window_raw(body, obj, pelvis, raw, future_len):
    """``get_batch`` (eval_smpl_long.py:26-84), for every clip what it does for its one clip: the ``past`` predicted frames body
    [past,B,159] / obj [past,B,6] / pelvis [past,B,3] become the next window's raw inputs, translated so that the pelvis of their
    first frame is the origin (:35-44,:56-57; rotation = I, so orientations are untouched -- upstream only re-expresses them as
    canonical rotation vectors, :51-54,:58-63, the same rotations), the ``future_len`` future frames are copies of the last past
    frame (:74).  Returns (raw dict, centroid [B,3]).  Pinned against the reference's own function (tests/golden/long.npz)."""
    centroid = pelvis[0].clone()                                                # [B,3] origin of the next window
    pad = lambda a: torch.cat([a, a[-1:].expand(future_len, *a.shape[1:])], dim=0).contiguous()
    nxt = dict(body_pose=pad(body[..., :66]), hand_pose=pad(body[..., 66:156]), body_trans=pad(body[..., -3:] - centroid),
               obj_angles=pad(obj[..., :3]), obj_trans=pad(obj[..., 3:] - centroid), beta=raw['beta'], obj_points=raw['obj_points'])
    return nxt, centroid


def sample_long(model, diffusion, correction, raw, windows, past_len=10, mode='correction', seed=None, x_T=None, step_noise=None, shard=None,
                **loop_kw):
    """Autoregressive long-horizon forecasting (eval_smpl_long.py:26-84,273-285; BASELINE config #4).

    Upstream this path is unreleased/broken (``denormalize`` / ``correct`` are undefined, ``get_batch`` copies clip 0 into
    every clip and ``--autoregressive`` is never passed: SURVEY.md §2 row 17), so the semantics are fixed here, per clip,
    following what ``get_batch`` does for its one clip: the last ``past_len`` predicted frames become the next window's
    past, translated so that the pelvis of their first frame is the origin (orientation untouched: rotation = I upstream),
    future frames padded with the last past frame, new conditioning through ``_get_embeddings``, sample, translate back,
    append the window's future frames.  Every clip keeps its chain on its own GPU: no exchange between ranks.
    ``x_T(k)`` / ``step_noise(k)`` (optional callables) inject window k's initial noise / per-step noise callable for deterministic
    parity with oracle/long_horizon.py; by default window k draws both from ``seed + k`` (``seed=None``: a fresh base seed per call).
    ``shard=(first_clip, total_clips)``: ``raw`` is a clip shard of a larger rollout; every window then draws the larger batch's noise at
    this shard's elements (diffusion.py), so the shard's rollout equals the same clips of the unsharded one bit for bit.
    Returns (obj [T+K*F,B,6], body [T+K*F,B,159], verts, jtr, pelvis) in the first window's coordinate frame."""
    smpl = correction.smpl
    T = raw['body_pose'].shape[0]
    fut = T - past_len
    if seed is None:                     # like p_sample_loop / evaluate_batch: every call draws its own noise unless the caller pins a seed
        from .diffusion import fresh_seed
        seed = fresh_seed()
    clips_kw = {} if shard is None else {'batch_clips': shard[1]}
    if shard is not None:
        loop_kw = dict(loop_kw, shard=shard)
    def run(bt, k):
        sd = seed + k
        nz = x_T(k) if x_T is not None else _x_T(bt['gt'], sd, shard)
        kw = dict(loop_kw, step_noise=step_noise(k)) if step_noise is not None else loop_kw
        if mode == 'correction':
            return sample_once_proj(model, diffusion, correction, bt, past_len, noise=nz, seed=sd, **kw)
        return sample_once(model, diffusion, smpl, bt, past_len, noise=nz, seed=sd, **kw)
    obj, body, verts, jtr, pelvis = run(batch_from_raw(model, raw, past_len, **clips_kw), 0)
    for k in range(windows):
        nxt, centroid = next_window_raw(body[-past_len:], obj[-past_len:], pelvis[-past_len:], raw, fut)
        o, b_, v, j, p = run(batch_from_raw(model, nxt, past_len, **clips_kw), 1 + k)
        o, b_ = o.clone(), b_.clone()
        o[..., 3:] += centroid
        b_[..., -3:] += centroid
        v, j, p = v + centroid[None, :, None, :], j + centroid[None, :, None, :], p + centroid
        obj, body = torch.cat([obj, o[past_len:]], dim=0), torch.cat([body, b_[past_len:]], dim=0)
        verts, jtr, pelvis = torch.cat([verts, v[past_len:]], dim=0), torch.cat([jtr, j[past_len:]], dim=0), torch.cat([pelvis, p[past_len:]], dim=0)
    return obj, body, verts, jtr, pelvis


class Metrics:
    """``metrics(obj_pred, body_jtr, body, obj_gt, body_jtr_gt, body_gt, verts, faces, obj_points)`` (:24-81) on
    ``interdiff_metrics``; owns its workspace.  ``correction`` supplies the mesh topology / packed SMPL handle."""

    def __init__(self, correction):
        self.lib = _lib.load()
        self.c = correction
        self._ws = None

    def __call__(self, obj_pred, body_jtr, body, obj_gt, body_jtr_gt, body_gt, verts, faces, obj_points):
        T, B, J, _ = body_jtr_gt.shape
        if obj_points.shape[1] != self.c.ctx.n_points:
            raise ValueError('obj_points must have %d points' % self.c.ctx.n_points)
        f32 = lambda a: a.contiguous().float()
        op, og, jt, jg = f32(obj_pred), f32(obj_gt), f32(body_jtr), f32(body_jtr_gt)
        bt, bg, vv, pts = f32(body[..., -3:]), f32(body_gt[..., -3:]), f32(verts), f32(obj_points)
        need = self.lib.interdiff_metrics_workspace_bytes(C.byref(self.c.ctx), B, T)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=op.device)
        out = torch.empty(6, B, dtype=torch.float32, device=op.device)
        _lib.check(self.lib.interdiff_metrics(C.byref(self.c.ctx), _lib.dptr(op), _lib.dptr(jt), _lib.dptr(bt), _lib.dptr(og),
                                              _lib.dptr(jg), _lib.dptr(bg), _lib.dptr(vv), _lib.dptr(pts), B, T, J, _lib.dptr(out),
                                              _lib.dptr(self._ws), self._ws.numel(), _lib.stream()), 'metrics')
        return {k: out[i] for i, k in enumerate(METRIC_KEYS)}


def evaluate_batch(model, diffusion, correction, batch, past_len=10, mode='correction', diverse_samples=1, noise=None, seed=None,
                   **loop_kw):
    """One iteration of the reference's outer eval loop (:252-296) for a clip batch: sample ``diverse_samples`` times,
    score each against the ground truth on the future frames, keep the per-clip minimum (:291-296).  Returns the six
    [B] metric vectors (ready for dist.gather_metrics).

    Every draw is an independent ancestral sample like upstream's fresh ``randn_like`` per step (:275-279,
    gaussian_diffusion.py:532): draw j runs the in-kernel noise generator under its own seed -- ``seed + j`` when a base seed is
    given (reproducible), otherwise one fresh 62-bit seed per draw from torch's global generator.  A fixed ``noise`` (x_T) with
    ``diverse_samples > 1`` therefore still gives different samples."""
    smpl = correction.smpl
    batch = as_clip_batch(model, batch, past_len)
    obj_gt, jtr_gt, body_gt, faces = get_gt(batch, smpl)
    met = Metrics(correction)
    best = None
    for j in range(diverse_samples):
        sd = None if seed is None else int(seed) + j
        if mode == 'correction':
            obj, body, verts, jtr, _ = sample_once_proj(model, diffusion, correction, batch, past_len, noise=noise, seed=sd, **loop_kw)
        else:
            obj, body, verts, jtr, _ = sample_once(model, diffusion, smpl, batch, past_len, noise=noise, seed=sd, **loop_kw)
        m = met(obj[past_len:], jtr[past_len:], body[past_len:], obj_gt[past_len:], jtr_gt[past_len:], body_gt[past_len:],
                verts[past_len:], faces, batch['obj_points'])
        best = m if best is None else {k: torch.minimum(best[k], m[k]) for k in m}
    return best


BATCH_DIMS = dict(gt=0, cond=1, hand_pose=1, beta=1, obj_points=0)        # clip dimension of every tensor of a clip batch


def evaluate_sharded(model, diffusion, correction, batch, past_len=10, mode='correction', diverse_samples=1, seed=None,
                     presharded=False, rank=None, world=None, collate=True, **loop_kw):
    """One eval batch of the reference's outer loop (:265-296) over ALL ranks: every rank takes a contiguous shard of the clips
    (they are independent through the whole path), runs ``evaluate_batch`` on it and the six per-clip metric vectors are
    collated with ONE fixed-size all-gather (RCCL over xGMI on GPUs, gloo in the CPU tests; shard sizes are known by construction, so
    nothing is exchanged or synchronised before it) -- the only collective of the path.
    Returns (per-clip metrics {name: [B_total]} in clip order on every rank, their means {name: float} = the numbers upstream
    accumulates at :291-296).

    A sharded run IS the unsharded run, bit for bit (given ``seed``): draw j of every rank uses the same seed ``seed + j`` and the rank's
    clips sit at their GLOBAL position in that draw's noise stream (``shard=(first_clip, total_clips)`` down to the Philox counters
    of every step and the slice of x_T, diffusion.py), and the feed-forward tile class follows the global batch.  The reference draws
    one ``randn_like`` tensor per step for the whole batch (gaussian_diffusion.py:532); SURVEY.md §8(e) names this option.
    ``presharded``: ``batch`` is this rank's shard already and every rank holds the same number of clips (bench.py builds its clips
    per rank); ragged pre-sharded batches must go through the unsharded form.
    ``rank`` / ``world``: override the process group's (a test that emulates N ranks one after another on one GPU);
    ``collate=False``: skip the all-gather and return this rank's metrics only (same emulation)."""
    r0, w0 = dist.get_rank_world()
    rank, world = r0 if rank is None else int(rank), w0 if world is None else int(world)
    if presharded:                                                  # `batch` already is this rank's shard: equal sizes on every rank
        B = batch['gt'].shape[0]
        total, sl, local = world * B, slice(rank * B, (rank + 1) * B), batch
        counts = [B] * world
    else:
        total = batch['gt'].shape[0]
        sl = dist.shard_slice(total, rank, world)
        local = dist.shard_batch(batch, rank, world, BATCH_DIMS)
        counts = [dist.shard_slice(total, q, world).stop - dist.shard_slice(total, q, world).start for q in range(world)]
    if sl.stop > sl.start:
        m = evaluate_batch(model, diffusion, correction, local, past_len, mode, diverse_samples, seed=seed,
                           **dict(loop_kw, shard=(sl.start, total)))
    else:                                                           # more ranks than clips: contribute an empty shard
        m = {k: torch.empty(0, device=batch['gt'].device) for k in METRIC_KEYS}
    if not collate:
        return m, {k: float(v.mean()) for k, v in m.items() if v.numel()}
    full = dist.gather_metrics(m, world, counts=counts)                 # ONE fixed-size all-gather, no host sync before it
    return full, {k: float(v.mean()) for k, v in full.items()}


RAW_DIMS = dict(body_pose=1, hand_pose=1, body_trans=1, obj_angles=1, obj_trans=1, beta=1, obj_points=0)      # clip dimension of the raw (dataset-side) tensors


def sample_long_sharded(model, diffusion, correction, raw, windows, past_len=10, mode='correction', seed=None, rank=None, world=None, **kw):
    """BASELINE config #4's partitioning (eval_smpl_long.py, B = 64 over 8 GPUs): every rank rolls out ITS clips -- the
    autoregressive chain of a clip never leaves its GPU, there is no exchange between ranks.  Every rank runs the windows under the
    SAME seeds (``seed + k``) with its clips at their global position in the noise streams (``shard``), so the ranks' rollouts are the
    clips of the unsharded rollout bit for bit.  ``rank`` / ``world`` override the process group's (one-GPU emulation of N ranks).
    Returns (this rank's clip slice of the global batch, ``sample_long``'s tuple for those clips)."""
    r0, w0 = dist.get_rank_world()
    rank, world = r0 if rank is None else int(rank), w0 if world is None else int(world)
    B = raw['body_pose'].shape[1]
    sl = dist.shard_slice(B, rank, world)
    local = dist.shard_batch(raw, rank, world, RAW_DIMS)
    return sl, sample_long(model, diffusion, correction, local, windows, past_len, mode, seed=seed, shard=(sl.start, B), **kw)
