"""PointNet++ set abstraction as InterDiff's ``PointNet2Encoder`` uses it (encoder side, "next" row N1 of SURVEY.md §8(f)).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  The arithmetic lives in the third-party ``pointnet2_ops`` 3.0.0 CUDA
extension, which is neither under /root/reference nor installable here: **parity unpinned** -- this file restates the
published algorithm and the reference's call sites pin everything around it:
``model/layers.py:111-175`` (PointNet2Encoder: SA1 npoint 1024, radii {0.05, 0.1}, nsample {16, 32}, MLPs [1+3,16,16,32] /
[1+3,32,32,64]; SA2 npoint 1, radii {0.1, 0.2}, nsample {16, 32}, MLPs [96+3,64,64,128] / [96+3,64,96,128]; Linear 256->253
concatenated after the key-point xyz) and ``model/diffusion_smpl.py:14,210-211`` (input feature = ||p||).

Restated semantics
  * furthest_point_sample: starts at index 0, running min-distance initialised to 1e10, candidates with squared norm <= 1e-3
    are skipped (their running distance is not updated either), the next point is the arg-max of the running distance --
    lowest index on ties (the CUDA block reduction's tie order is implementation-defined);
  * ball_query(r, nsample): scan the source points in index order, a point is inside when d^2 < r^2, the first hit pre-fills
    all nsample slots, stop after nsample hits; no hit leaves zeros;
  * grouping: [xyz_j - centre | feature_j] (use_xyz=True); shared MLP = (1x1 conv without bias -> eval BatchNorm -> ReLU) x 3;
    max over the samples; scales concatenated.
All squared distances are (dx*dx + dy*dy) + dz*dz in fp32, evaluated with separate roundings.
"""
import torch

SA1 = dict(npoint=1024, radii=(0.05, 0.1), nsamples=(16, 32))
SA2 = dict(npoint=1, radii=(0.1, 0.2), nsamples=(16, 32))


def _d2(a, b):
    """a [...,3], b [...,3] broadcastable -> squared distance with the restated rounding order."""
    dx, dy, dz = a[..., 0] - b[..., 0], a[..., 1] - b[..., 1], a[..., 2] - b[..., 2]
    return (dx * dx + dy * dy) + dz * dz


def furthest_point_sample(xyz, npoint):
    """xyz [B,N,3] -> int64 [B,npoint]."""
    B, N, _ = xyz.shape
    out = torch.zeros(B, npoint, dtype=torch.int64)
    mag = (xyz[..., 0] * xyz[..., 0] + xyz[..., 1] * xyz[..., 1]) + xyz[..., 2] * xyz[..., 2]
    valid = mag > 1e-3
    for b in range(B):
        temp = torch.full((N,), 1e10, dtype=xyz.dtype)
        old = 0
        for j in range(1, npoint):
            d = _d2(xyz[b], xyz[b, old])
            d2 = torch.minimum(d, temp)
            temp = torch.where(valid[b], d2, temp)
            cand = torch.where(valid[b], d2, torch.full_like(d2, -1.0))
            best = cand.max()
            old = int(torch.nonzero(cand == best)[0]) if best > -1.0 else 0
            out[b, j] = old
    return out


def ball_query(radius, nsample, xyz, new_xyz):
    """xyz [B,N,3], new_xyz [B,M,3] -> int64 [B,M,nsample]."""
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    d2 = _d2(xyz[:, None, :, :], new_xyz[:, :, None, :])                 # [B,M,N]
    inside = d2 < radius * radius
    rank = torch.cumsum(inside.to(torch.int64), dim=2) - 1               # rank of every hit among the hits
    idx = torch.zeros(B, M, nsample, dtype=torch.int64)
    first = torch.where(inside.any(dim=2), inside.to(torch.int64).argmax(dim=2), torch.zeros(B, M, dtype=torch.int64))
    idx[:] = first[:, :, None]
    sel = inside & (rank < nsample)
    bb, mm, nn = torch.nonzero(sel, as_tuple=True)
    idx[bb, mm, rank[bb, mm, nn]] = nn
    return idx


def shared_mlp(x, sd, prefix, n_layers=3, eps=1e-5):
    """x [B,C,M,S]; keys '<prefix>.{3l}.weight' (conv [out,in,1,1]) and '<prefix>.{3l+1}.{weight,bias,running_mean,running_var}'."""
    for l in range(n_layers):
        W = sd['%s.%d.weight' % (prefix, 3 * l)][:, :, 0, 0]
        x = torch.einsum('oc,bcms->boms', W, x)
        bn = '%s.%d' % (prefix, 3 * l + 1)
        sh = (1, -1, 1, 1)
        x = (x - sd[bn + '.running_mean'].reshape(sh)) / torch.sqrt(sd[bn + '.running_var'].reshape(sh) + eps) \
            * sd[bn + '.weight'].reshape(sh) + sd[bn + '.bias'].reshape(sh)
        x = torch.relu(x)
    return x


def sa_module_msg(xyz, features, sd, prefix, npoint, radii, nsamples):
    """xyz [B,N,3], features [B,C,N] -> (new_xyz [B,npoint,3], new_features [B,sum(Cout),npoint])."""
    B = xyz.shape[0]
    fidx = furthest_point_sample(xyz, npoint)
    new_xyz = torch.gather(xyz, 1, fidx[:, :, None].expand(B, npoint, 3))
    outs = []
    for s, (r, ns) in enumerate(zip(radii, nsamples)):
        idx = ball_query(r, ns, xyz, new_xyz)                            # [B,npoint,ns]
        bi = torch.arange(B)[:, None, None]
        g_xyz = xyz[bi, idx] - new_xyz[:, :, None, :]                    # [B,npoint,ns,3]
        g_feat = features.permute(0, 2, 1)[bi, idx]                      # [B,npoint,ns,C]
        x = torch.cat([g_xyz, g_feat], dim=3).permute(0, 3, 1, 2)        # [B,3+C,npoint,ns]
        outs.append(shared_mlp(x, sd, '%s.mlps.%d' % (prefix, s)).max(dim=3)[0])
    return new_xyz, torch.cat(outs, dim=1)


def pointnet2_encode(sd, obj_points, prefix='pcEmbedding'):
    """obj_points [B,P,3] -> [B,256]  (model/diffusion_smpl.py:210-211 + PointNet2Encoder.forward with one key point)."""
    nrm = torch.sqrt((obj_points * obj_points).sum(dim=2, keepdim=True))     # obj_points.norm(dim=2, keepdim=True)
    xyz, feats = obj_points.contiguous(), nrm.permute(0, 2, 1).contiguous()
    xyz1, f1 = sa_module_msg(xyz, feats, sd, prefix + '.SA_modules.0', **SA1)
    xyz2, f2 = sa_module_msg(xyz1, f1, sd, prefix + '.SA_modules.1', **SA2)
    lin = f2.permute(0, 2, 1) @ sd[prefix + '.Linear.weight'].T + sd[prefix + '.Linear.bias']     # [B,1,253]
    return torch.cat([xyz2, lin], dim=2).reshape(obj_points.shape[0], -1)
