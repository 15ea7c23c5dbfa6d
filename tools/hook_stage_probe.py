"""Which stage of the correction hook is sensitive to concurrent kernels on another stream?  (not product code)

    python tools/hook_stage_probe.py                 the product's own split-f16 feed-forward kernel as the neighbour (exclusive CU since round 4: expected identical)
    python tools/hook_stage_probe.py aggr [reps]     the library's diagnostic f16-MFMA + streaming-load kernel (interdiff_debug_f16_aggressor: small LDS / register
                                                     needs, so it DOES share CUs) beside the hook's stages -- the victims of round 5's co-residency question:
                                                     smpl_pose_kernel + smpl_blend_skin_kernel (verts, jtr), corr_contact_kernel (o2h, idx), objproj_kernel (proj)
    INTERDIFF_HIP_LIB=build_ab/noslp/libinterdiff_hip.so python tools/hook_stage_probe.py aggr
                                                     the same with the library built with -fno-slp-vectorize (no compiler-formed packed-fp32 instruction in any victim)
"""
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


AGGR_SRC = torch.zeros(64 << 20, device='cuda')          # 256 MB to stream through
AGGR_SINK = torch.zeros(16, device='cuda')


def load_kernels(kind):
    if kind in ('aggr', 'aggr_noloads'):            # 60 000 iterations: several milliseconds, longer than any stage below (the contact scan is 1.3 ms)
        from interdiff_amd import _lib
        with torch.cuda.stream(Bs):
            _lib.check(_lib.load().interdiff_debug_f16_aggressor(_lib.dptr(AGGR_SRC), AGGR_SRC.numel(), _lib.dptr(AGGR_SINK), 60000, 1024, 1 if kind == 'aggr' else 0, _lib.stream()), 'aggressor')
        return
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
if len(sys.argv) > 1 and sys.argv[1] == 'aggr':
    from interdiff_amd import _lib
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    print('library', _lib.LIB_PATH)
    kinds = (None, 'aggr_noloads', 'aggr')
else:
    reps, kinds = 4, (None, ('split', 16), ('split', 32), ('split', 16), ('split', 32), ('split', 16), ('split', 32))
tally = {}
for kind in kinds:
    for rep in range(reps):
        got = stages(kind)
        bad = {k: float((ref[k].float() - got[k].float()).abs().max()) for k in ref if not torch.equal(ref[k], got[k])}
        print('load', kind, 'rep', rep, bad if bad else 'identical', flush=True)
        for k in ref:
            tally[(str(kind), k)] = tally.get((str(kind), k), 0) + (1 if k in bad else 0)
        if 'jtr' in bad:
            nb = (ref['jtr'] != got['jtr']).any(dim=2).nonzero()
            print('   jtr frames', sorted(set(nb[:, 0].tolist()))[:12], 'joints', sorted(set(nb[:, 1].tolist()))[:20], 'n', nb.shape[0], flush=True)
print('summary (runs of %d in which a stage output differs from the stage run alone):' % reps)
for kind in kinds:
    print('  neighbour %-14s' % (kind,), {k: tally[(str(kind), k)] for k in ref})
