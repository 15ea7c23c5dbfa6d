"""Contact-frame correction predictor ``ObjProjector.sample`` (rows D1, D2),
restated functionally from a state_dict with the reference module's key names
(``st_gcnns_relative.{0..3}.*``, ``st_gcnns.{0..3}.*``, ``st_gcnns_all.{0..3}.*``).

Follows model/correction_smpl.py:55-67 (DCT matrices, fp64 numpy then .float()),
:79-138 (sample, eval branch = argmax), model/layers.py:271-345 (ST_GCNN_layer,
eval-mode BatchNorm) and model/sublayers.py:378-419,464-516 (graph convs).
Hand marker ids: data/utils.py:249-260 (left_hand_ids + right_hand_ids).
"""
import numpy as np
import torch

HAND_MARKERS = [10, 11, 14, 31, 13, 17, 23, 28, 27] + [60, 43, 44, 47, 62, 46, 51, 57]


def dct_matrices(N):
    """correction_smpl.py:55-67."""
    k = np.arange(N)[:, None].astype(np.float64)
    i = np.arange(N)[None, :].astype(np.float64)
    w = np.full((N, 1), np.sqrt(2.0 / N))
    w[0, 0] = np.sqrt(1.0 / N)
    d = w * np.cos(np.pi * (i + 0.5) * k / N)
    return d, np.linalg.inv(d)


def _bn(x, sd, p, eps=1e-5):
    sh = (1, -1, 1, 1)
    return ((x - sd[p + '.running_mean'].reshape(sh)) / torch.sqrt(sd[p + '.running_var'].reshape(sh) + eps)
            * sd[p + '.weight'].reshape(sh) + sd[p + '.bias'].reshape(sh))


def _conv1x1(x, sd, p):
    W = sd[p + '.weight'][:, :, 0, 0]
    return torch.einsum('oc,nctv->notv', W, x) + sd[p + '.bias'].reshape(1, -1, 1, 1)


def st_gcnn_layer(x, sd, p):
    """layers.py:339-345.  x [n,c,t,v]."""
    if (p + '.residual.0.weight') in sd:
        res = _bn(_conv1x1(x, sd, p + '.residual.0'), sd, p + '.residual.1')
    else:
        res = x
    Tm = sd[p + '.gcn.T']
    if Tm.dim() == 2:                                   # version 0: shared over nodes
        g = torch.einsum('nctv,tq->ncqv', x, Tm)
    else:                                               # version 2
        g = torch.einsum('nctv,vtq->ncqv', x, Tm)
        g = torch.einsum('nctv,tvw->nctw', g, sd[p + '.gcn.A'])
    h = _bn(_conv1x1(g, sd, p + '.tcn.0'), sd, p + '.tcn.1') + res
    a = sd[p + '.prelu.weight']
    return torch.where(h >= 0, h, a * h)


def _stack(x, sd, name):
    for i in range(4):
        x = st_gcnn_layer(x, sd, '%s.%d' % (name, i))
    return x


def objprojector_sample(sd, obj_angles, obj_trans, human_verts, contact, past_len, n_pre=10):
    """obj_angles [T,B,6], obj_trans [T,B,3], human_verts [T,B,P,>=3], contact [B,P]
    -> [T,B,9]  (eval branch, initialize=False)."""
    hv = human_verts[..., :3]
    T, B, P, _ = hv.shape
    dt = obj_angles.dtype
    dct64, idct64 = dct_matrices(T)
    dct = torch.from_numpy(dct64).to(dt)[:n_pre]            # [n_pre, T]
    idct = torch.from_numpy(idct64).to(dt)[:, :n_pre]       # [T, n_pre]
    idx_pad = list(range(past_len)) + [past_len - 1] * (T - past_len)

    rel = torch.cat([obj_angles[:, :, None, :].expand(T, B, P, 6),
                     obj_trans[:, :, None, :] - hv], dim=3)[idx_pad]            # [T,B,P,9]
    rel = torch.einsum('kt,tbpc->bckp', dct, rel)                                # [B,9,n_pre,P]
    rel = rel + _stack(rel, sd, 'st_gcnns_relative')
    hdct = torch.einsum('kt,tbpc->bckp', dct, hv)                                # [B,3,n_pre,P]
    multi = torch.cat([rel[:, :6], rel[:, 6:9] + hdct], dim=1)

    og = torch.cat([obj_angles, obj_trans], dim=2)[idx_pad]                      # [T,B,9]
    o = torch.einsum('kt,tbc->bck', dct, og)[..., None]                          # [B,9,n_pre,1]
    o = o + _stack(o, sd, 'st_gcnns')

    allx = torch.cat([o, multi], dim=3)                                          # [B,9,n_pre,P+1]
    allx = allx + _stack(allx, sd, 'st_gcnns_all')
    res = torch.einsum('tk,bckp->tbpc', idct, allx)                              # [T,B,P+1,9]

    csum = contact.sum(dim=1)
    score = contact.to(dt).clone()
    score[:, HAND_MARKERS] = score[:, HAND_MARKERS] + 0.5
    pick = torch.where(csum > 0, 1 + torch.argmax(score, dim=1), torch.zeros_like(csum, dtype=torch.int64))
    return res[:, torch.arange(B), pick, :]
