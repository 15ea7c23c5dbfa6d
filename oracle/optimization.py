"""TEST ORACLE ONLY -- CPU restatement (torch autograd) of the physics post-optimisation, "next" row N4 of SURVEY.md
§8(f): optimization.py:19-173 of the reference.  Never imported by the product path (interdiff_amd/optimize.py runs the
hand-written backward kernels of csrc/optimize.hip).

Pinned by tests/golden/optim.npz: the reference's own ``optimize()`` run through refshim on a synthetic clip (its
printed losses, the gradients its Adam sees at the first iteration, and the parameters it returns).

The clip is T frames of one sequence.  Optimised variables (optimization.py:123-136): rotation MATRICES of the global
orientation [T,1,3,3], the 21 body joints, the 30 hand joints and the object [T,3,3], plus the body / object
translations [T,3]; Adam(lr 1e-3), 200 iterations, the iterate with the lowest loss after iteration 150 is returned.
"""
import torch
from . import rotations as rot
from .smpl import smpl_forward
from .geometry import vertex_normals, nn_argmin

LEFT_FOOT, RIGHT_FOOT = 10, 11
STATIC_THRESHOLD = 0.008            # optimization.py:51-52
CONTACT_RADIUS = 0.5                # :75
N_ITERS, SAVE_AFTER, RATIO_DEN = 200, 150, 350      # :139,140,147


def static_feet(jtr_gt):
    """optimization.py:47-52: frames whose foot joint moved < 8 mm in the ground (xz) plane."""
    out = []
    for j in (LEFT_FOOT, RIGHT_FOOT):
        f = jtr_gt[:, j][:, [0, 2]]
        out.append((torch.norm(f[1:] - f[:-1], dim=1) + 1e-6) < STATIC_THRESHOLD)
    return out


def initial_state(model, pose, trans, obj_angles, obj_trans, betas):
    """optimization.py:27-52,123-134: the constants of the loss and the starting point."""
    T = pose.shape[0]
    R = rot.axis_angle_to_matrix(pose.reshape(T, 52, 3))
    verts_gt, jtr_gt, _ = smpl_forward(model, pose, betas, trans)
    left_static, right_static = static_feet(jtr_gt)
    init = dict(glo=R[:, :1].clone(), body=R[:, 1:22].clone(), hand=R[:, 22:].clone(), transl=trans.clone(),
                obj_transl=obj_trans.clone(), obj_rot=rot.axis_angle_to_matrix(obj_angles))
    return init, dict(verts_gt=verts_gt, left_static=left_static, right_static=right_static)


def _second(x):
    return (x[1:-1] - x[:-2]) - (x[2:] - x[1:-1])


def calc_loss(model, p, init, const, betas, obj_points, ratio):
    """optimization.py:54-121.  p / init: dicts glo [T,1,3,3], body [T,21,3,3], hand [T,30,3,3], transl, obj_transl [T,3],
    obj_rot [T,3,3].  Returns (loss, [total, collision, reg, reg_v])."""
    T = p['transl'].shape[0]
    pose = rot.matrix_to_axis_angle(torch.cat([p['glo'], p['body'], p['hand']], dim=1)).reshape(T, -1)
    verts, jtr, _ = smpl_forward(model, pose, betas, p['transl'])
    pts = torch.matmul(obj_points[None], p['obj_rot'].permute(0, 2, 1)) + p['obj_transl'][:, None]
    normals = vertex_normals(verts, model['faces'])
    # point2point_signed (tools.py:45-66): indices carry no gradient, the vectors do
    yidx = nn_argmin(pts.detach(), verts.detach())
    xidx = nn_argmin(verts.detach(), pts.detach())
    g3 = lambda src, idx: torch.gather(src, 1, idx[..., None].expand(-1, -1, 3))
    y2x = pts - g3(verts, yidx)
    o2h_signed = y2x.norm(dim=2) * (g3(normals, yidx) * y2x).sum(-1).sign()
    w = torch.zeros_like(o2h_signed)
    w[o2h_signed < 0] = 20 * ratio if ratio < 1 else 20                                  # :66-70
    near = (verts - g3(pts, xidx)).norm(dim=2) < CONTACT_RADIUS                          # == (distance < 0.5).any(dim=1), :74-75
    w_verts = torch.full_like(near, 1e-2, dtype=verts.dtype)
    w_verts[near] = 0
    loss_verts_reg = ((verts - const['verts_gt']).abs().sum(2) * w_verts).sum(dim=1).mean()
    loss_dist_o = (o2h_signed.abs() * w).sum(dim=1).mean()

    def foot(j, static):
        if not bool(static.any()):
            return 0
        f = jtr[:, j][:, [0, 2]]
        return torch.mean((f[1:] - f[:-1])[static] ** 2)
    loss_feet = foot(LEFT_FOOT, const['left_static']) + foot(RIGHT_FOOT, const['right_static'])
    reg = (0.1 * (p['obj_transl'] - init['obj_transl']).abs().mean() + 0.1 * (p['obj_rot'] - init['obj_rot']).abs().mean()
           + 0.005 * (p['body'] - init['body']).abs().sum(dim=2).sum(dim=1).mean() + 0.1 * (p['transl'] - init['transl']).abs().mean()
           + 0.1 * (p['glo'] - init['glo']).abs().mean() + loss_verts_reg)
    sm = lambda x, a, b: a * torch.mean(_second(x) ** 2) + b * torch.mean((x[1:] - x[:-1]) ** 2)
    body_v = (1000 * torch.mean((_second(p['body']) ** 2).sum(dim=2).sum(dim=1))
              + 100 * torch.mean(((p['body'][1:] - p['body'][:-1]) ** 2).sum(dim=2).sum(dim=1)) + 1000 * loss_feet)
    reg_v = (sm(p['hand'], 50, 50) + sm(p['obj_transl'], 1000, 100) + sm(p['obj_rot'], 1000, 100) + body_v
             + sm(p['transl'], 10, 10) + sm(p['glo'], 5, 5))
    loss = loss_dist_o + reg + reg_v
    return loss, torch.stack([loss.detach(), loss_dist_o.detach(), reg.detach(), reg_v.detach()])


def loss_and_grads(model, params, pose, trans, obj_angles, obj_trans, betas, obj_points, ii):
    """One evaluation of calc_loss + backward at the given parameters (dict of the six tensors), iteration number ii."""
    with torch.no_grad():
        init, const = initial_state(model, pose, trans, obj_angles, obj_trans, betas)
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    with torch.enable_grad():
        loss, parts = calc_loss(model, p, init, const, betas, obj_points, ii / RATIO_DEN)
    loss.backward()
    return parts, {k: v.grad.detach() for k, v in p.items()}


PARAM_ORDER = ('body', 'transl', 'glo', 'obj_transl', 'obj_rot', 'hand')          # optimization.py:136


def optimize(model, pose, trans, obj_angles, obj_trans, betas, obj_points, iters=None, lr=1e-3, record_grads=False):
    """optimization.py:123-172.  pose [T,156] axis-angle, ...; ``iters`` = the iteration numbers ii to run (default
    range(200)).  Returns dict(pose [T,156], trans, obj_angles, obj_trans [T,3], losses [K,4], grads)."""
    T = pose.shape[0]
    iters = list(range(N_ITERS)) if iters is None else list(iters)
    with torch.no_grad():
        init, const = initial_state(model, pose, trans, obj_angles, obj_trans, betas)
    p = {k: v.clone().requires_grad_(True) for k, v in init.items()}
    opt = torch.optim.Adam([p[k] for k in PARAM_ORDER], lr=lr)
    best, saved, losses, grads = 1e7, None, [], None
    for ii in iters:
        opt.zero_grad()
        with torch.enable_grad():
            loss, parts = calc_loss(model, p, init, const, betas, obj_points, ii / RATIO_DEN)
        loss.backward()
        if record_grads and grads is None:
            grads = {k: p[k].grad.detach().clone() for k in PARAM_ORDER}
        opt.step()
        losses.append(parts)
        if ii > SAVE_AFTER and float(loss.detach()) < best:
            best = float(loss.detach())
            saved = {k: v.detach().clone() for k, v in p.items()}
    out = dict(losses=torch.stack(losses), grads=grads, params={k: v.detach().clone() for k, v in p.items()})
    if saved is not None:
        with torch.no_grad():
            aa = rot.matrix_to_axis_angle(torch.cat([saved['glo'], saved['body'], saved['hand']], dim=1)).reshape(T, -1)
            out.update(pose=aa, trans=saved['transl'], obj_angles=rot.matrix_to_axis_angle(saved['obj_rot']),
                       obj_trans=saved['obj_transl'])
    return out
