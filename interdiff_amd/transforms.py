"""pytorch3d.transforms names the reference imports (eval_smpl_short.py:18, model/diffusion_smpl.py:4,
model/correction_smpl.py:4), running on the HIP kernels of csrc/rotations.hip.  Device tensors only."""
import torch
from . import _lib


def _run(fn_name, x, nin_shape, nout_shape):
    lib = _lib.load()
    lead = x.shape[:len(x.shape) - len(nin_shape)]
    if tuple(x.shape[len(lead):]) != tuple(nin_shape):
        raise ValueError('%s: trailing shape must be %s, got %s' % (fn_name, nin_shape, tuple(x.shape)))
    xc = x.contiguous().float()
    out = torch.empty(tuple(lead) + tuple(nout_shape), dtype=torch.float32, device=x.device)
    n = 1
    for s in lead:
        n *= int(s)
    _lib.check(getattr(lib, fn_name)(_lib.dptr(xc, torch.float32), _lib.dptr(out), n, _lib.stream()), fn_name)
    return out


def rotation_6d_to_matrix(d6):
    return _run('interdiff_rotation_6d_to_matrix', d6, (6,), (3, 3))


def matrix_to_rotation_6d(m):
    return _run('interdiff_matrix_to_rotation_6d', m, (3, 3), (6,))


def matrix_to_axis_angle(m):
    return _run('interdiff_matrix_to_axis_angle', m, (3, 3), (3,))


def axis_angle_to_matrix(aa):
    return _run('interdiff_axis_angle_to_matrix', aa, (3,), (3, 3))


def axis_angle_to_quaternion(aa):
    return _run('interdiff_axis_angle_to_quaternion', aa, (3,), (4,))


def rotation_6d_to_axis_angle(d6):
    """Fused matrix_to_axis_angle(rotation_6d_to_matrix(.)) (eval_smpl_short.py:91,157-162)."""
    return _run('interdiff_rotation_6d_to_axis_angle', d6, (6,), (3,))
