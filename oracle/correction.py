"""The ``denoised_fn`` correction hook and the eval glue (rows C1, E1, E2).

Follows eval_smpl_short.py:84-130 (denoised_fn), :133-177 (sample_once_proj),
:225-250 (get_gt) and :24-81 (metrics).  The batch dict-of-lists of the
reference dataset is replaced by plain tensors (SURVEY.md §8(d) schema):
gt [B,1,144,T], cond [M,B,256], hand_pose [T,B,90] (GT hands, NOT yet padded),
beta [T,B,10], obj_points [B,P,3].
"""
import torch
from . import rotations as R
from .smpl import smpl_forward
from .geometry import vertex_normals, point2point_signed
from .objprojector import objprojector_sample

SMPL_DIM = 132          # 22 joints x rot6d  (eval_smpl_short.py:416)
MARKERS67 = [3470, 3171, 3327, 857, 1812, 628, 182, 3116, 3040, 239,
             1666, 1725, 0, 2174, 1568, 1368, 3387, 2112, 1053, 1058,
             3336, 3346, 1323, 2108, 3122, 3314, 1252, 1082, 1861, 1454,
             850, 2224, 3233, 1769, 6728, 4343, 5273, 4116, 3694, 6399,
             6540, 6488, 3749, 5135, 5194, 3512, 5635, 5210, 4360, 4841,
             6786, 5573, 4538, 4544, 6736, 6747, 4804, 5568, 6544, 6682,
             5322, 4927, 5686, 4598, 6633, 3506, 3508]      # data/utils.py:232-238


def idx_pad(past_len, T):
    return list(range(past_len)) + [past_len - 1] * (T - past_len)


def split_tokens(x):
    """[B,1,144,T] -> body [T,B,135], obj [T,B,9]."""
    xt = x.squeeze(1).permute(2, 0, 1).contiguous()
    return xt[..., :SMPL_DIM + 3], xt[..., SMPL_DIM + 3:]


def correction_gate(t0):
    """eval_smpl_short.py:85."""
    return not (t0 > 500 or t0 % 50 != 0)


def correction_terms(x, y, past_len, markers_idx=MARKERS67):
    """Everything denoised_fn computes before the blend; returned for per-stage parity."""
    body, obj = split_tokens(x)
    body_gt, obj_gt = split_tokens(y['inpainted_motion'])
    T, B, _ = body.shape
    obj_R = R.rotation_6d_to_matrix(obj[..., :6])                                   # :89
    body_rot = R.matrix_to_axis_angle(R.rotation_6d_to_matrix(body[..., :SMPL_DIM].reshape(T, B, -1, 6))).reshape(T, B, -1)
    pose = torch.cat([body_rot, y['hand_pose']], dim=2).reshape(T * B, -1)          # :92-96
    verts, jtr, _ = smpl_forward(y['smpl'], pose, y['beta'].reshape(T * B, -1), body[..., -3:].reshape(T * B, 3))
    markers = verts[:, markers_idx].reshape(T, B, -1, 3)                             # :102-103
    pts = torch.matmul(y['obj_points'][None], obj_R.transpose(-1, -2)) + obj[:, :, None, -3:]   # :107
    normals = vertex_normals(verts, y['smpl']['faces'])
    o2h = point2point_signed(verts, pts.reshape(T * B, -1, 3), x_normals=normals)[0]             # [T*B,P]
    w = torch.where(o2h < 0, 20.0, 0.0).to(x.dtype)                                  # :113-117
    loss = (o2h.abs() * w).reshape(T, B, -1)
    md = torch.sqrt(((markers[:, :, None] - pts[:, :, :, None]) ** 2).sum(-1))      # [T,B,P,67]
    distance = md.min(dim=3)[0].min(dim=2)[0].mean(dim=0)                            # :120
    condition = ~((loss[past_len:].mean(dim=2).mean(dim=0) < 0.002) & (distance < 0.02))
    contact = (md < 0.02).any(dim=2)[past_len:].sum(dim=0)                           # [B,67] int64
    return dict(body=body, obj=obj, obj_gt=obj_gt, verts=verts, jtr=jtr, markers=markers, pts=pts,
                normals=normals, o2h=o2h, loss=loss, distance=distance, condition=condition,
                contact=contact, body_rot=body_rot)


def denoised_fn(x, t, model_kwargs, past_len=10, total_steps_const=1000):
    """x [B,1,144,T]; mutates and returns x like the reference (:129-130)."""
    t0 = int(t[0])
    if not correction_gate(t0):
        return x
    y = model_kwargs['y']
    s = correction_terms(x, y, past_len)
    proj = objprojector_sample(y['obj_model'], s['obj_gt'][..., :6], s['obj_gt'][..., 6:], s['markers'],
                               s['contact'], past_len)
    x_ = torch.cat([s['body'], proj], dim=2).permute(1, 2, 0).unsqueeze(1).contiguous()
    a = t0 / total_steps_const                                                       # :128 hard-coded 1000
    x_ = a * x + (1 - a) * x_
    x[s['condition']] = x_[s['condition']]
    return x


def finalize(sample, gt, hand_pose, beta, smpl, past_len):
    """sample_once_proj after the loop (eval_smpl_short.py:154-177)."""
    body, obj = split_tokens(sample)
    T, B, _ = body.shape
    pad = idx_pad(past_len, T)
    body_rot = R.matrix_to_axis_angle(R.rotation_6d_to_matrix(body[..., :SMPL_DIM].reshape(T, B, -1, 6))).reshape(T, B, -1)
    obj_rot = R.matrix_to_axis_angle(R.rotation_6d_to_matrix(obj[..., :6]))
    body_pred = torch.cat([body_rot, hand_pose[pad], body[..., -3:]], dim=2)         # [T,B,159]
    verts, jtr, _ = smpl_forward(smpl, body_pred.reshape(T * B, -1)[:, :-3], beta.reshape(T * B, -1),
                                 body_pred.reshape(T * B, -1)[:, -3:])
    obj_pred = torch.cat([obj_rot, obj[..., -3:]], dim=2)
    return obj_pred, body_pred, verts.reshape(T, B, -1, 3), jtr.reshape(T, B, -1, 3)


def ground_truth(gt, hand_pose, beta, smpl):
    """get_gt (eval_smpl_short.py:225-250): GT hands are NOT padded here."""
    body, obj = split_tokens(gt)
    T, B, _ = body.shape
    body_rot = R.matrix_to_axis_angle(R.rotation_6d_to_matrix(body[..., :SMPL_DIM].reshape(T, B, -1, 6))).reshape(T, B, -1)
    obj_rot = R.matrix_to_axis_angle(R.rotation_6d_to_matrix(obj[..., :6]))
    body_gt = torch.cat([body_rot, hand_pose, body[..., -3:]], dim=2)
    _, jtr, _ = smpl_forward(smpl, body_gt.reshape(T * B, -1)[:, :-3], beta.reshape(T * B, -1),
                             body_gt.reshape(T * B, -1)[:, -3:])
    return torch.cat([obj_rot, obj[..., -3:]], dim=2), jtr.reshape(T, B, -1, 3), body_gt


def metrics(obj_pred, body_jtr, body, obj_gt, body_jtr_gt, body_gt, verts, faces, obj_points):
    """eval_smpl_short.py:24-81.  All inputs already sliced to the future frames."""
    T, B = body_jtr_gt.shape[:2]
    Rm = R.axis_angle_to_matrix(obj_pred[..., :3])
    pts = torch.matmul(obj_points[None], Rm.transpose(-1, -2)) + obj_pred[:, :, None, -3:]
    vflat = verts.reshape(T * B, -1, 3)
    normals = vertex_normals(vflat, faces)
    o2h = point2point_signed(vflat, pts.reshape(T * B, -1, 3), x_normals=normals)[0]
    penetrate = (o2h < 0).reshape(T, B, -1).to(verts.dtype).mean(dim=2).mean(dim=0)
    nrm = lambda v: torch.sqrt((v * v).sum(-1))
    g_mpjpe = nrm(body_jtr - body_jtr_gt).mean(dim=2).mean(dim=0)
    l_mpjpe = nrm((body_jtr - body_jtr[:, :, 0:1]) - (body_jtr_gt - body_jtr_gt[:, :, 0:1])).mean(dim=2).mean(dim=0)
    body_tr = nrm(body[..., -3:] - body_gt[..., -3:]).mean(dim=0)
    obj_tr = nrm(obj_pred[..., -3:] - obj_gt[..., -3:]).mean(dim=0)
    q, qg = R.axis_angle_to_quaternion(obj_pred[..., :3]), R.axis_angle_to_quaternion(obj_gt[..., :3])
    rot = torch.minimum((q - qg).abs().sum(-1), (q + qg).abs().sum(-1)).mean(dim=0)
    return dict(global_mpjpe=g_mpjpe, local_mpjpe=l_mpjpe, body_translation=body_tr,
                obj_translation=obj_tr, obj_rot_error=rot, penetrate=penetrate)
