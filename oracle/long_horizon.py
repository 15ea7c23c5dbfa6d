"""Long-horizon autoregressive forecasting ("next" row N3, BASELINE config #4), restated on CPU tensors.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows eval_smpl_long.py:26-84 (``get_batch``: how the last ``past_len`` predicted frames become the next window's past) and
:273-285 (the rollout loop).  Upstream this script is unreleased / broken -- ``denormalize`` and ``correct`` (:279,:286) are not
defined anywhere, ``get_batch`` reads clip 0 and repeats it over the batch (:34,:44,:52,:56-63 index ``[i, 0]`` / ``.repeat(B, 1)``)
and ``--autoregressive`` is never passed.  PINNED as far as upstream code exists: ``next_window_raw`` is checked against the reference's own
``get_batch`` (tests/golden/long.npz: recorded from the function's own source with its one non-executable method chain removed, see
make_golden.py gen_long); ``denormalize`` / ``correct`` have no upstream definition, so for them the restatement below DEFINES the
contract.  The three upstream defects are fixed in the one way the surrounding code implies:

  * ``get_batch`` is applied to EVERY clip with the arithmetic it applies to clip 0: origin of the next window = pelvis of the first
    of the ``past_len`` frames (:35-38), ``rotation = rotation_v = I`` (:37-38) so the translation line (:40-44) collapses to
    ``trans - centroid``; root / object orientation go through scipy's ``Rotation.from_rotvec(..).as_rotvec()`` (:51-54,:58-63:
    same rotation, canonical angle in [0, pi]); pose[3:] copied (:55); the ``future_len`` future frames are copies of the last past
    frame (:74);
  * ``denormalize`` = the inverse of that re-centring (add the window's centroid back to both translations, the vertices, the joints);
  * ``correct`` = identity.
"""
import numpy as np
import torch
from scipy.spatial.transform import Rotation
from . import correction as ocor, denoiser as oden, diffusion as odf


def _canon_rotvec(aa):
    """Rotation.from_rotvec(aa).as_rotvec() (eval_smpl_long.py:51-54,58-63 with rotation = I), any leading shape."""
    flat = aa.reshape(-1, 3).double().numpy()
    return torch.from_numpy(Rotation.from_rotvec(flat).as_rotvec().reshape(aa.shape)).to(aa.dtype)


def next_window_raw(body, obj, pelvis, raw, past_len, future_len):
    """get_batch, per clip.  body [past,B,159] (66 axis-angle | 90 hands | 3 trans), obj [past,B,6] (axis-angle | trans),
    pelvis [past,B,3] -> (raw inputs of the next window [past+future, B, .], centroid [B,3])."""
    centroid = pelvis[0].clone()                                                  # :35-36
    pose = body[..., :156].clone()
    pose[..., :3] = _canon_rotvec(body[..., :3])                                  # :51-54
    trans = body[..., -3:] - centroid                                             # :40-44 with rotation = I
    o_ang = _canon_rotvec(obj[..., :3])                                           # :58-63
    o_tr = obj[..., 3:6] - centroid                                               # :56-57
    pad = lambda a: torch.cat([a, a[-1:].expand(future_len, *a.shape[1:])], dim=0).contiguous()      # :74
    return dict(body_pose=pad(pose[..., :66]), hand_pose=pad(pose[..., 66:156]), body_trans=pad(trans), obj_angles=pad(o_ang),
                obj_trans=pad(o_tr), beta=raw['beta'], obj_points=raw['obj_points']), centroid


def sample_window(sd, smpl, objproj_sd, raw, past_len, sched, x_T, step_noise, mode='correction'):
    """One window: conditioning (MDM._get_embeddings) -> sampler (+ hook) -> poses, like eval_smpl_short.sample_once(_proj)."""
    T = raw['body_pose'].shape[0]
    cond, gt = oden.get_embeddings(sd, raw['body_pose'], raw['body_trans'], raw['obj_angles'], raw['obj_trans'], raw['obj_points'], past_len)
    gt4 = gt.permute(1, 2, 0).unsqueeze(1).contiguous()
    mask = torch.ones_like(gt4, dtype=torch.bool)
    mask[..., past_len:] = False
    y = dict(cond=cond, inpainted_motion=gt4, inpainting_mask=mask, hand_pose=raw['hand_pose'][ocor.idx_pad(past_len, T)], beta=raw['beta'],
             obj_points=raw['obj_points'], smpl=smpl, obj_model=objproj_sd)
    hook = (lambda x, t, kw: ocor.denoised_fn(x, t, kw, past_len=past_len)) if mode == 'correction' else None
    x = odf.p_sample_loop(lambda x, t, y: oden.mdm_forward(sd, x, t, y['cond']), tuple(gt4.shape), sched, x_T, step_noise, {'y': y}, denoised_fn=hook)
    obj, body, verts, jtr = ocor.finalize(x, gt4, raw['hand_pose'], raw['beta'], smpl, past_len)
    return obj, body, verts, jtr, jtr[:, :, 0]


def rollout(sd, smpl, objproj_sd, raw, windows, past_len, sched, x_T, step_noise, mode='correction'):
    """eval_smpl_long.py:273-285 for one draw.  ``x_T(k)`` / ``step_noise(k)`` give window k's initial noise / per-step noise
    callable.  Returns (obj [T+K*F,B,6], body [T+K*F,B,159], verts, jtr, pelvis) in the first window's coordinate frame."""
    T = raw['body_pose'].shape[0]
    fut = T - past_len
    obj, body, verts, jtr, pelvis = sample_window(sd, smpl, objproj_sd, raw, past_len, sched, x_T(0), step_noise(0), mode)
    for k in range(windows):
        nxt, centroid = next_window_raw(body[-past_len:], obj[-past_len:], pelvis[-past_len:], raw, past_len, fut)      # :276
        o, b, v, j, p = sample_window(sd, smpl, objproj_sd, nxt, past_len, sched, x_T(k + 1), step_noise(k + 1), mode)  # :277
        o, b = o.clone(), b.clone()                                                                                     # denormalize (:279)
        o[..., 3:] += centroid
        b[..., -3:] += centroid
        v, j, p = v + centroid[None, :, None, :], j + centroid[None, :, None, :], p + centroid
        obj, body = torch.cat([obj, o[past_len:]], dim=0), torch.cat([body, b[past_len:]], dim=0)                       # :281-284
        verts, jtr, pelvis = torch.cat([verts, v[past_len:]], dim=0), torch.cat([jtr, j[past_len:]], dim=0), torch.cat([pelvis, p[past_len:]], dim=0)
    return obj, body, verts, jtr, pelvis
