"""SMPL-H forward kinematics + linear blend skinning (rows B1-B3).

Follows libsmpl/smplpytorch/pytorch/smpl_layer.py:72-175 with
rodrigues_layer.py / tensutils.py.  ``model`` is a dict of the seven buffers
the reference registers (smpl_layer.py:47-69): v_template[V,3],
shapedirs[V,3,10], posedirs[V,3,9*(J-1)], J_regressor[J,V], weights[V,J],
faces[F,3] (int64), parents[J] (parents[0] unused).
"""
import torch
from .rotations import rodrigues_smpl


def smpl_forward(model, pose, betas, trans):
    """pose [N,3J] axis-angle, betas [N,10], trans [N,3] ->
    (verts [N,V,3], jtr [N,J,3], v_posed [N,V,3])."""
    N = pose.shape[0]
    J = model['weights'].shape[1]
    R = rodrigues_smpl(pose.reshape(N, J, 3))                       # [N,J,3,3]
    eye = torch.eye(3, dtype=pose.dtype)
    pose_map = (R[:, 1:] - eye).reshape(N, 9 * (J - 1))             # :89-92
    v_shaped = model['v_template'][None] + torch.einsum('vck,nk->nvc', model['shapedirs'], betas)   # :102
    joints = torch.einsum('jv,nvc->njc', model['J_regressor'], v_shaped)                           # :103
    v_posed = v_shaped + torch.einsum('vcp,np->nvc', model['posedirs'], pose_map)                  # :106-107
    # kinematic chain (:117-130)
    parents = [int(p) for p in model['parents']]
    G = [None] * J

    def rigid(Rj, tj):
        top = torch.cat([Rj, tj[..., None]], dim=-1)                # [N,3,4]
        bot = torch.tensor([0.0, 0.0, 0.0, 1.0], dtype=pose.dtype).expand(N, 1, 4)
        return torch.cat([top, bot], dim=-2)
    G[0] = rigid(R[:, 0], joints[:, 0])
    for i in range(1, J):
        G[i] = G[parents[i]] @ rigid(R[:, i], joints[:, i] - joints[:, parents[i]])
    Gs = torch.stack(G, dim=1)                                      # [N,J,4,4]
    jtr = Gs[:, :, :3, 3]
    # remove rest pose (:135-142): A_i = G_i - [0 | G_i [J_i;0]]
    jh = torch.cat([joints, torch.zeros(N, J, 1, dtype=pose.dtype)], dim=-1)
    corr = torch.einsum('njab,njb->nja', Gs, jh)                    # [N,J,4]
    A = Gs.clone()
    A[..., 3] = A[..., 3] - corr
    # skinning (:144-152)
    Tm = torch.einsum('vj,njab->nvab', model['weights'], A)         # [N,V,4,4]
    vh = torch.cat([v_posed, torch.ones(N, v_posed.shape[1], 1, dtype=pose.dtype)], dim=-1)
    verts = torch.einsum('nvab,nvb->nva', Tm, vh)[..., :3]
    return verts + trans[:, None], jtr + trans[:, None], v_posed
