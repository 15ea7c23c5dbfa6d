"""Import the REFERENCE's own source (read-only, /root/reference/interdiff) so
that ``make_golden.py`` can run it and record golden vectors.

Only usable in the build container: /root/reference does not exist on the GPU
box, and nothing at test/bench run time imports this module.

The reference's third-party deps are absent here (pytorch3d, local_attention,
pointnet2_ops, chamfer_distance, torchvision, chumpy, cv2, smplx, ...); they
are replaced by ``sys.modules`` stubs.  Where a stub has to COMPUTE something
(pytorch3d.transforms, LocalAttention, stochastic_depth) it uses the oracle's
restatement -- so goldens recorded through those stubs pin the reference's own
code AROUND them, not the third-party arithmetic itself ("parity unpinned"
pieces, see oracle/__init__.py).
"""
import os
import sys
import types
import importlib

REF = '/root/reference/interdiff'
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))


def available():
    return os.path.isdir(REF)


def install():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import torch
    from oracle import rotations, local_attn

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    p3d = mod('pytorch3d')
    p3d.transforms = mod('pytorch3d.transforms', **{k: getattr(rotations, k) for k in (
        'axis_angle_to_matrix', 'matrix_to_axis_angle', 'rotation_6d_to_matrix',
        'matrix_to_rotation_6d', 'axis_angle_to_quaternion', 'quaternion_to_matrix',
        'matrix_to_quaternion', 'quaternion_to_axis_angle')})
    mod('local_attention', LocalAttention=local_attn.LocalAttention)
    tv = mod('torchvision')
    tv.ops = mod('torchvision.ops',
                 stochastic_depth=lambda x, p, mode, training=True: x if (p == 0.0 or not training) else (_ for _ in ()).throw(NotImplementedError()))

    # --- pointnet2_ops.pointnet2_modules.PointnetSAModuleMSG (CUDA-only third party, "next" row N1): the module
    # structure / parameter names of pointnet2_ops 3.0.0 (mlps = ModuleList of Sequential(Conv2d 1x1 no bias,
    # BatchNorm2d, ReLU) x 3, +3 input channels for use_xyz) with torch's own conv / batch-norm doing the MLP, while the
    # sampling / grouping indices come from the oracle's restatement (oracle/pointnet2.py: parity unpinned).
    from oracle import pointnet2 as opn

    class _SA(torch.nn.Module):
        def __init__(self, npoint, radii, nsamples, mlps, use_xyz=True, bn=True):
            super().__init__()
            self.npoint, self.radii, self.nsamples = npoint, list(radii), list(nsamples)
            self.mlps = torch.nn.ModuleList()
            for spec in mlps:
                spec = list(spec)
                if use_xyz:
                    spec[0] += 3
                layers = []
                for i in range(1, len(spec)):
                    layers += [torch.nn.Conv2d(spec[i - 1], spec[i], kernel_size=1, bias=not bn), torch.nn.BatchNorm2d(spec[i]),
                               torch.nn.ReLU(True)]
                self.mlps.append(torch.nn.Sequential(*layers))

        def forward(self, xyz, features):
            B = xyz.shape[0]
            fidx = opn.furthest_point_sample(xyz, self.npoint)
            new_xyz = torch.gather(xyz, 1, fidx[:, :, None].expand(B, self.npoint, 3))
            outs = []
            for r, ns, mlp in zip(self.radii, self.nsamples, self.mlps):
                idx = opn.ball_query(r, ns, xyz, new_xyz)
                bi = torch.arange(B)[:, None, None]
                g = torch.cat([xyz[bi, idx] - new_xyz[:, :, None, :], features.permute(0, 2, 1)[bi, idx]], dim=3).permute(0, 3, 1, 2)
                outs.append(mlp(g.contiguous()).max(dim=3)[0])
            return new_xyz, torch.cat(outs, dim=1)
    pn = mod('pointnet2_ops')
    pn.pointnet2_modules = mod('pointnet2_ops.pointnet2_modules', PointnetSAModuleMSG=_SA)
    mod('smplx')
    ch = mod('chumpy', Ch=object)          # loader-only dep; class body never runs
    ch.ch = mod('chumpy.ch', MatVecMult=object)
    mod('cv2')

    # --- nearest neighbour CUDA op (tools.py:9,45-47): indices from the oracle's brute force
    from oracle import geometry

    class ChamferDistance:
        def __call__(self, x, y, x_normals=None, y_normals=None):
            return None, None, geometry.nn_argmin(x, y).int(), geometry.nn_argmin(y, x).int()
    mod('chamfer_distance', ChamferDistance=ChamferDistance)
    p3d.loss = mod('pytorch3d.loss')
    p3d.ops = mod('pytorch3d.ops', cot_laplacian=None)
    p3d.structures = mod('pytorch3d.structures', Meshes=None)
    hbp = mod('human_body_prior')
    hbp.tools = mod('human_body_prior.tools', tgm_conversion=None)

    # --- everything eval_smpl_short.py imports at module level but the hot path never touches
    pl = mod('pytorch_lightning', seed_everything=lambda *a, **k: None, LightningModule=torch.nn.Module)
    pl.loggers = mod('pytorch_lightning.loggers')
    ps = mod('psbody')
    ps.mesh = mod('psbody.mesh', Mesh=None)
    mod('data.dataset_smpl', Dataset=None, OBJECT_PATH='', MODEL_PATH='')
    rd = mod('render')
    rd.mesh_viz = mod('render.mesh_viz', visualize_body_obj=None)
    mod('train_correction_smpl', LitInteraction=None)
    mod('train_diffusion_smpl', LitInteraction=None)
    if REF not in sys.path:
        sys.path.insert(0, REF)


def load(name):
    """e.g. load('diffusion.gaussian_diffusion'), load('model.sublayers')."""
    install()
    return importlib.import_module(name)
