"""Which stage of the correction hook is sensitive to concurrent kernels on another stream?  (not product code)"""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import fixtures as fx                                                   # noqa: E402
from tests.test_hip_parity import make_correction, dev                             # noqa: E402
from interdiff_amd.mdm import MDM, ffn_parts                                       # noqa: E402
from interdiff_amd.smpl import SMPL_Layer                                          # noqa: E402
from interdiff_amd import transforms as tr                                         # noqa: E402

torch.set_grad_enabled(False)
mdm = MDM(fx.mdm_weights(), device='cuda')
smpl = SMPL_Layer(fx.smpl_model(), device='cuda')
B = 8
T, P = fx.TIMED_T, fx.TIMED_P
bt, y = fx.timed_inputs(B)
y = dev(y)
corr = make_correction(smpl, T, P)
A, Bs = torch.cuda.Stream(), torch.cuda.Stream()
x2 = torch.randn(800, 256, device='cuda')
parts = torch.empty(5, 800, 256, device='cuda')
big = torch.randn(2048, 2048, device='cuda')
g = torch.Generator().manual_seed(0)
N = T * B
pose = (0.3 * torch.randn(N, 156, generator=g)).cuda()
betas = torch.randn(N, 10, generator=g).cuda()
trans = (0.3 * torch.randn(N, 3, generator=g)).cuda()
objR = tr.axis_angle_to_matrix((0.3 * torch.randn(T, B, 3, generator=g)).cuda())
objT = (0.3 * torch.randn(T, B, 3, generator=g)).cuda()
verts_ref = smpl(pose, th_betas=betas, th_trans=trans, want_v_posed=False)[0].clone()
torch.cuda.synchronize()


def load_kernels(kind):
    with torch.cuda.stream(Bs):
        for i in range(300 if kind else 0):
            if kind == 'torch':
                torch.mm(big, big)
            else:
                mdm.ffn_math, mdm.ffn_rows = kind[0], kind[1]
                ffn_parts(mdm, x2, i % 8, out=parts)
    mdm.ffn_rows = 0


def stages(kind):
    torch.cuda.synchronize()
    load_kernels(kind)
    with torch.cuda.stream(A):
        v, j, _, _ = smpl(pose, th_betas=betas, th_trans=trans, want_v_posed=False)
        v, j = v.clone(), j.clone()
    torch.cuda.synchronize()
    load_kernels(kind)
    with torch.cuda.stream(A):
        o2h, idx = corr.contact_nn(verts_ref.reshape(T, B, -1, 3), y['obj_points'], objR, objT)
        o2h, idx = o2h.clone(), idx.clone()
    torch.cuda.synchronize()
    load_kernels(kind)
    with torch.cuda.stream(A):
        mk = verts_ref.reshape(T, B, -1, 3)[:, :, corr.markers_idx.long()].contiguous()
        contact = torch.zeros(B, 67, dtype=torch.int32, device='cuda')
        contact[:, 5] = 3
        pr = corr.objproj.sample(torch.zeros(T, B, 6, device='cuda') + 0.1, objT, mk, contact).clone()
    torch.cuda.synchronize()
    return dict(verts=v, jtr=j, o2h=o2h, idx=idx, proj=pr)


ref = stages(None)
for kind in (None, ('split', 16), ('split', 32), ('split', 16), ('split', 32), ('split', 16), ('split', 32)):
    for rep in range(4):
        got = stages(kind)
        bad = {k: float((ref[k].float() - got[k].float()).abs().max()) for k in ref if not torch.equal(ref[k], got[k])}
        print('load', kind, 'rep', rep, bad if bad else 'identical', flush=True)
        if 'jtr' in bad:
            nb = (ref['jtr'] != got['jtr']).any(dim=2).nonzero()
            print('   jtr frames', sorted(set(nb[:, 0].tolist()))[:12], 'joints', sorted(set(nb[:, 1].tolist()))[:20], 'n', nb.shape[0], flush=True)
