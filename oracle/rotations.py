"""Rotation conversions the reference takes from ``pytorch3d.transforms`` 0.7.2.

pytorch3d is NOT vendored under /root/reference (environment.yml:65 pins
0.7.2) -> this is a restatement of its published algorithms; parity for this
file is UNPINNED upstream (no reference tests) and pinned here by round-trip
identities (tests/test_oracle_golden.py).  Quaternions are (w, x, y, z).

Reference call sites: eval_smpl_short.py:18,33,65-66,90-91,157-162;
model/diffusion_smpl.py:4,212-213; model/correction_smpl.py:4,71.

Also: ``rodrigues_smpl`` = the reference's own axis-angle -> matrix used inside
SMPL_Layer (libsmpl/smplpytorch/pytorch/rodrigues_layer.py:13-52), which is
NOT the pytorch3d one (it has the ``+1e-8`` quirk and renormalises the
quaternion).
"""
import torch


def _unit(v, eps=1e-12):
    # torch.nn.functional.normalize semantics: v / max(||v||, eps)
    n = torch.sqrt((v * v).sum(-1, keepdim=True))
    return v / torch.clamp(n, min=eps)


def rotation_6d_to_matrix(d6):
    """Gram-Schmidt on the two 3-vectors; rows of the result are (b1,b2,b3)."""
    a1, a2 = d6[..., 0:3], d6[..., 3:6]
    b1 = _unit(a1)
    b2 = _unit(a2 - (b1 * a2).sum(-1, keepdim=True) * b1)
    b3 = torch.stack([
        b1[..., 1] * b2[..., 2] - b1[..., 2] * b2[..., 1],
        b1[..., 2] * b2[..., 0] - b1[..., 0] * b2[..., 2],
        b1[..., 0] * b2[..., 1] - b1[..., 1] * b2[..., 0],
    ], dim=-1)
    return torch.stack([b1, b2, b3], dim=-2)


def matrix_to_rotation_6d(m):
    """First two ROWS, flattened."""
    return m[..., 0:2, :].reshape(m.shape[:-2] + (6,)).clone()


def _sinc_half(angle):
    """sin(angle/2)/angle with the 0.5 - angle^2/48 series below 1e-6."""
    small = angle.abs() < 1e-6
    safe = torch.where(small, torch.ones_like(angle), angle)
    return torch.where(small, 0.5 - angle * angle / 48.0, torch.sin(safe * 0.5) / safe)


def axis_angle_to_quaternion(aa):
    ang = torch.norm(aa, p=2, dim=-1, keepdim=True)            # torch.norm: zero subgradient at the origin (autograd users)
    return torch.cat([torch.cos(ang * 0.5), aa * _sinc_half(ang)], dim=-1)


def quaternion_to_matrix(q):
    r, i, j, k = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    s2 = 2.0 / (q * q).sum(-1)
    rows = [
        1 - s2 * (j * j + k * k), s2 * (i * j - k * r), s2 * (i * k + j * r),
        s2 * (i * j + k * r), 1 - s2 * (i * i + k * k), s2 * (j * k - i * r),
        s2 * (i * k - j * r), s2 * (j * k + i * r), 1 - s2 * (i * i + j * j),
    ]
    return torch.stack(rows, dim=-1).reshape(q.shape[:-1] + (3, 3))


def axis_angle_to_matrix(aa):
    return quaternion_to_matrix(axis_angle_to_quaternion(aa))


def _sqrt_positive_part(x):
    """sqrt(max(0, x)) with a ZERO subgradient where x <= 0 (pytorch3d's helper of the same name; matters only for
    autograd -- oracle/optimization.py -- the forward values equal sqrt(clamp(x, 0)))."""
    ret = torch.zeros_like(x)
    pos = x > 0
    ret[pos] = torch.sqrt(x[pos])
    return ret


def matrix_to_quaternion(m):
    """Four-candidate method of 0.7.2 (floor 0.1, argmax pick, NO sign
    standardisation of w)."""
    m00, m01, m02 = m[..., 0, 0], m[..., 0, 1], m[..., 0, 2]
    m10, m11, m12 = m[..., 1, 0], m[..., 1, 1], m[..., 1, 2]
    m20, m21, m22 = m[..., 2, 0], m[..., 2, 1], m[..., 2, 2]
    tr = torch.stack([1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22,
                      1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22], dim=-1)
    q_abs = _sqrt_positive_part(tr)
    cand = torch.stack([
        torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], dim=-1),
        torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], dim=-1),
        torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], dim=-1),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], dim=-1),
    ], dim=-2)
    cand = cand / (2.0 * torch.clamp(q_abs, min=0.1)[..., None])
    pick = q_abs.argmax(dim=-1)
    idx = pick[..., None, None].expand(pick.shape + (1, 4))
    return torch.gather(cand, -2, idx).squeeze(-2)


def quaternion_to_axis_angle(q):
    n = torch.norm(q[..., 1:], p=2, dim=-1, keepdim=True)
    half = torch.atan2(n, q[..., 0:1])
    ang = 2.0 * half
    return q[..., 1:] / _sinc_half(ang)


def matrix_to_axis_angle(m):
    return quaternion_to_axis_angle(matrix_to_quaternion(m))


def rodrigues_smpl(aa):
    """aa[..., 3] -> R[..., 3, 3] exactly as SMPL_Layer does it
    (rodrigues_layer.py:41-52 batch_rodrigues + :13-38 quat2mat)."""
    ang = torch.sqrt(((aa + 1e-8) * (aa + 1e-8)).sum(-1, keepdim=True))
    axis = aa / ang
    half = ang * 0.5
    q = torch.cat([torch.cos(half), torch.sin(half) * axis], dim=-1)
    q = q / torch.sqrt((q * q).sum(-1, keepdim=True))
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    w2, x2, y2, z2 = w * w, x * x, y * y, z * z
    wx, wy, wz, xy, xz, yz = w * x, w * y, w * z, x * y, x * z, y * z
    rows = [w2 + x2 - y2 - z2, 2 * xy - 2 * wz, 2 * wy + 2 * xz,
            2 * wz + 2 * xy, w2 - x2 + y2 - z2, 2 * yz - 2 * wx,
            2 * xz - 2 * wy, 2 * wx + 2 * yz, w2 - x2 - y2 + z2]
    return torch.stack(rows, dim=-1).reshape(aa.shape[:-1] + (3, 3))
