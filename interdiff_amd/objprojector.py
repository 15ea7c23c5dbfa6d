"""Corrector seam: ``ObjProjector.sample(obj_angles, obj_trans, human_verts, contact)``
(model/correction_smpl.py:79-138, eval branch) on ``interdiff_objprojector_sample``.

``pack_objprojector`` takes the reference module's state_dict (``checkpoints/correction.ckpt`` keys with
the ``model.`` prefix stripped) and folds, on the host in float64:
  * eval-mode BatchNorm into the preceding 1x1 convolution (tcn.0/tcn.1 and residual.0/residual.1);
  * the idx_pad frame repetition into ``dct_pad`` [n_pre, past_len];
  * DCT / IDCT matrices exactly as get_dct_matrix builds them (fp64, inverse by numpy) -> fp32.
Arena layer block (see csrc/objproj.hip): Tm | (A^T padded to 80x80 per coefficient) | Wt bt Wr br (zero-padded to
multiples of 16 channels: MFMA operands) | prelu.
"""
import ctypes as C
import numpy as np
import torch
from . import _lib

HAND_MARKERS = [10, 11, 14, 31, 13, 17, 23, 28, 27] + [60, 43, 44, 47, 62, 46, 51, 57]   # data/utils.py:249-260
STACKS = ('st_gcnns_relative', 'st_gcnns', 'st_gcnns_all')
VP = 80                                  # nodes padded to 5 MFMA tiles (csrc/objproj.hip)


def dct_matrices(N):
    k = np.arange(N)[:, None].astype(np.float64)
    i = np.arange(N)[None, :].astype(np.float64)
    w = np.full((N, 1), np.sqrt(2.0 / N))
    w[0, 0] = np.sqrt(1.0 / N)
    d = w * np.cos(np.pi * (i + 0.5) * k / N)
    return d, np.linalg.inv(d)


def _np(a):
    return (a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)).astype(np.float64)


def _fold(sd, conv, bn, eps=1e-5):
    W, b = _np(sd[conv + '.weight'])[:, :, 0, 0], _np(sd[conv + '.bias'])
    g, beta = _np(sd[bn + '.weight']), _np(sd[bn + '.bias'])
    mu, var = _np(sd[bn + '.running_mean']), _np(sd[bn + '.running_var'])
    s = g / np.sqrt(var + eps)
    return W * s[:, None], (b - mu) * s + beta


def pack_objprojector(sd, T, past_len, device, n_pre=10, P=67):
    parts, n = [], [0]

    def add(a):
        a = np.ascontiguousarray(a, dtype=np.float32).ravel()
        off = n[0]
        pad = (-a.size) % 16
        parts.append(a)
        if pad:
            parts.append(np.zeros(pad, np.float32))
        n[0] += a.size + pad
        return off
    op = _lib.ObjProj()
    op.T, op.past_len, op.P, op.n_pre = T, past_len, P, n_pre
    dct, idct = dct_matrices(T)
    d = dct[:n_pre]
    dpad = d[:, :past_len].copy()
    dpad[:, past_len - 1] = d[:, past_len - 1:].sum(axis=1)
    op.dct_pad, op.dct, op.idct = add(dpad), add(d), add(idct[:, :n_pre])
    bonus = np.zeros(P)
    bonus[HAND_MARKERS] = 0.5
    op.hand_bonus = add(bonus)
    for s, name in enumerate(STACKS):
        for l in range(4):
            p = '%s.%d' % (name, l)
            Wt, bt = _fold(sd, p + '.tcn.0', p + '.tcn.1')
            Wr, br = _fold(sd, p + '.residual.0', p + '.residual.1')
            cout, cin = Wt.shape
            cinp, coutp = -(-cin // 16) * 16, -(-cout // 16) * 16

            def padw(W):
                out = np.zeros((coutp, cinp))
                out[:cout, :cin] = W
                return out.ravel()

            def padb(b):
                out = np.zeros(coutp)
                out[:cout] = b
                return out
            Tm = _np(sd[p + '.gcn.T']).ravel()
            if s == 2:
                A = _np(sd[p + '.gcn.A'])                                  # [n_pre, nodes, nodes] : y[w] = sum_v x[v] A[t][v][w]
                AT = np.zeros((n_pre, VP, VP))
                AT[:, :A.shape[2], :A.shape[1]] = A.transpose(0, 2, 1)     # [t][w][v], zero padded to 80 x 80
                blk = [Tm, AT.ravel()]
            else:
                blk = [Tm, np.zeros(12)]                                   # keep the weights 16-byte aligned
            blk += [padw(Wt), padb(bt), padw(Wr), padb(br), _np(sd[p + '.prelu.weight']).ravel()]
            op.layer[s * 4 + l] = add(np.concatenate(blk))
            op.cout[s * 4 + l], op.cin[s * 4 + l] = cout, cin
    arena = torch.from_numpy(np.concatenate(parts)).to(device)
    op.arena = arena.data_ptr()
    return op, arena


class ObjProjector:
    def __init__(self, state_dict, T, past_len=10, device='cuda', n_pre=10):
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.T, self.past_len = T, past_len
        self.cop, self.arena = pack_objprojector(state_dict, T, past_len, self.device, n_pre=n_pre)

    def eval(self):
        return self

    def sample(self, obj_angles, obj_trans, human_verts, contact, initialize=False):
        if initialize:
            raise NotImplementedError('initialize=True (mean over nodes) is a training-time option')
        T, B = obj_angles.shape[:2]
        if T != self.T:
            raise ValueError('ObjProjector was packed for T=%d' % self.T)
        hv = human_verts[..., :3].contiguous().float()
        oa, ot = obj_angles.contiguous().float(), obj_trans.contiguous().float()
        ct = contact.to(torch.int32).contiguous()
        out = torch.empty(T, B, 9, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.interdiff_objprojector_sample(C.byref(self.cop), _lib.dptr(oa), _lib.dptr(ot), _lib.dptr(hv),
                                                          _lib.dptr(ct, torch.int32), B, _lib.dptr(out), _lib.stream()),
                   'objprojector_sample')
        return out
