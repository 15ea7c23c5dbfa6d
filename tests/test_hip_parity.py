"""GPU parity tests: every HIP entry point (through the C-ABI, via the ctypes host layer) against the CPU
oracle on the same seeded inputs and against the golden fixtures recorded from the reference's own source.
Tolerance: north_star's 1e-4 relative fp32 (max|delta| / max|ref|), tighter where the op allows; bit-exact for
indices and decisions.  Run with:  python -m pytest tests -m gpu"""
import os
import numpy as np
import pytest
import torch
from tests import fixtures as fx
from oracle import diffusion as odf, denoiser as oden, smpl as osmpl, geometry as ogeo
from oracle import objprojector as oobj, correction as ocor, rotations as R

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
DEV = 'cuda'


def rel(a, b):
    a, b = torch.as_tensor(a).detach().cpu().double(), torch.as_tensor(b).detach().cpu().double()
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


def close(a, b, tol, what=''):
    e = rel(a, b)
    assert e <= tol, '%s: rel err %.3e > %.1e' % (what, e, tol)
    return e


def dev(x):
    return {k: dev(v) for k, v in x.items()} if isinstance(x, dict) else (x.to(DEV) if isinstance(x, torch.Tensor) else x)


@pytest.fixture(scope='module')
def mdm(lib):
    from interdiff_amd.mdm import MDM
    return MDM(fx.mdm_weights(), device=DEV)


@pytest.fixture(scope='module')
def smpl(lib):
    from interdiff_amd.smpl import SMPL_Layer
    return SMPL_Layer(fx.smpl_model(), device=DEV)


def make_correction(smpl, T, P):
    from interdiff_amd.objprojector import ObjProjector
    from interdiff_amd.correction import HipCorrection
    op = ObjProjector(fx.objproj_weights(), T=T, past_len=fx.PAST, device=DEV)
    return HipCorrection(smpl, op, n_points=P, past_len=fx.PAST, device=DEV)


# ------------------------------------------------------------------------------------------ rotations (C2)
def test_rotations(lib):
    from interdiff_amd import transforms as tr
    g = torch.Generator().manual_seed(3)
    aa = torch.randn(4000, 3, generator=g)
    aa[0] = 0.0
    aa[1] = torch.tensor([1e-8, 0, 0])
    aa[2] = torch.tensor([3.1, 0.01, -0.02])                      # near pi
    d6 = torch.randn(7, 11, 6, generator=g)
    M = R.axis_angle_to_matrix(aa)
    close(tr.axis_angle_to_matrix(aa.to(DEV)), M, 1e-6, 'aa->matrix')
    close(tr.axis_angle_to_quaternion(aa.to(DEV)), R.axis_angle_to_quaternion(aa), 1e-6, 'aa->quat')
    close(tr.rotation_6d_to_matrix(d6.to(DEV)), R.rotation_6d_to_matrix(d6), 1e-6, '6d->matrix')
    close(tr.matrix_to_rotation_6d(M.to(DEV)), R.matrix_to_rotation_6d(M), 0, 'matrix->6d')
    # axis-angle representative can legitimately differ near pi / by quaternion branch: compare the ROTATION
    got = tr.matrix_to_axis_angle(M.to(DEV)).cpu()
    close(R.axis_angle_to_matrix(got), M, 2e-6, 'matrix->aa (as rotation)')
    same_branch = (got - R.matrix_to_axis_angle(M)).abs().max(dim=1)[0] < 1e-3
    assert same_branch.float().mean() > 0.99
    got = tr.rotation_6d_to_axis_angle(d6.to(DEV)).cpu()
    close(R.axis_angle_to_matrix(got), R.rotation_6d_to_matrix(d6), 2e-6, '6d->aa (as rotation)')


# ------------------------------------------------------------------------------------------ denoiser (A1-A4)
@pytest.mark.parametrize('tag,B,T', [('a', 2, 12), ('b', 3, 35)])
def test_mdm_forward_golden(mdm, tag, B, T):
    x, ts, cond = fx.mdm_inputs(B, T)
    got = mdm(x.to(DEV), ts.to(DEV), y={'cond': cond.to(DEV)})
    close(got, fx.golden('mdm.npz')['out_' + tag], 1e-4, 'vs reference golden')
    close(got, oden.mdm_forward(fx.mdm_weights(), x, ts, cond), 1e-4, 'vs oracle')


def test_mdm_forward_bench_shape(mdm):
    """BASELINE config #2 shape (B=16, T=100) against the oracle; also ragged T and B=1."""
    for B, T in ((16, 100), (1, 30), (5, 37)):
        x, ts, cond = fx.mdm_inputs(B, T)
        got = mdm(x.to(DEV), ts.to(DEV), y={'cond': cond.to(DEV)})
        close(got, oden.mdm_forward(fx.mdm_weights(), x, ts, cond), 1e-4, 'B=%d T=%d' % (B, T))


def test_mdm_forward_split_f16_vs_exact_and_fp64(lib):
    """The whole denoiser forward under both arithmetics (MDM.ffn_math: feed-forward block, QKV projection and the row block's three
    contractions as split-f16 products vs exact fp32 MFMA) against the oracle run in float64: the split form must be as close to the fp64
    answer as the exact form is (within 2x; both are a few 1e-7), at the bench shape, the reference's default clip length and a ragged one."""
    from interdiff_amd.mdm import MDM
    sd = fx.mdm_weights()
    sd64 = {k: torch.as_tensor(v).double() for k, v in sd.items()}
    models = {}
    for math in ('exact', 'split'):
        models[math] = MDM(sd, device=DEV)
        models[math].ffn_math = math
    for B, T in ((16, 100), (3, 35), (2, 13), (1, 208)):
        x, ts, cond = fx.mdm_inputs(B, T)
        ref = oden.mdm_forward(sd64, x.double(), ts, cond.double())
        err = {}
        for math, m in models.items():
            got = m(x.to(DEV), ts.to(DEV), y={'cond': cond.to(DEV)}).cpu().double()
            err[math] = float((got - ref).abs().max() / ref.abs().max())
        fx.record_parity('mdm_forward_vs_fp64_B%d_T%d' % (B, T), exact=err['exact'], split=err['split'])
        assert err['split'] <= max(2 * err['exact'], 2e-6), (B, T, err)
        assert err['exact'] <= 1e-5, (B, T, err)


def test_mdm_forward_with_either_self_attention_kernel(lib):
    """The split-f16 self-attention (csrc/attn_h2.h: the default of the split arithmetic since round 5) and the fp32 kernel it replaced (tune[IDF_TUNE_MISC] = 6; also what a
    clip longer than 192 frames takes) against the oracle in float64 at the bench shape, the longest clip of either kernel, the reference's default clip length and clips shorter
    than a key tile: the split-f16 kernel is as close to fp64 as the fp32 one (within 2x)."""
    from interdiff_amd import _lib
    from interdiff_amd.mdm import MDM
    sd = fx.mdm_weights()
    sd64 = {k: torch.as_tensor(v).double() for k, v in sd.items()}
    m = MDM(sd, device=DEV)
    for B, T in ((16, 100), (1, 208), (1, 192), (2, 129), (3, 35), (2, 13), (2, 1)):
        x, ts, cond = fx.mdm_inputs(B, T)
        ref = oden.mdm_forward(sd64, x.double(), ts, cond.double())
        err = {}
        for name, misc in (('fp32_attention', 6), ('split_attention', 0)):
            m.w.tune[_lib.TUNE['misc']] = misc
            got = m(x.to(DEV), ts.to(DEV), y={'cond': cond.to(DEV)}).cpu().double()
            err[name] = float((got - ref).abs().max() / ref.abs().max())
        m.w.tune[_lib.TUNE['misc']] = 0
        fx.record_parity('mdm_forward_split_attention_vs_fp64_B%d_T%d' % (B, T), **err)
        assert err['split_attention'] <= max(2 * err['fp32_attention'], 2e-6), (B, T, err)
        assert err['fp32_attention'] <= 2e-6, (B, T, err)


def test_mdm_no_rotary_switch(lib):
    from interdiff_amd.mdm import MDM
    x, ts, cond = fx.mdm_inputs(2, 12)
    got = MDM(fx.mdm_weights(), device=DEV, rotary=False)(x.to(DEV), ts.to(DEV), y={'cond': cond.to(DEV)})
    close(got, oden.mdm_forward(fx.mdm_weights(), x, ts, cond, rotary=False), 1e-4, 'rotary off')


# ------------------------------------------------------------------------------------------ sampler update (S4-S5)
def test_posterior_step_and_inpaint(lib):
    from interdiff_amd import _lib
    g = torch.Generator().manual_seed(4)
    for n in (1, 7, 4096, 16 * 144 * 100):
        x, x0, eps = (torch.randn(n, generator=g) for _ in range(3))
        c1, c2, s = np.float32(0.3), np.float32(0.69), np.float32(0.05)
        xd, x0d, epsd = x.to(DEV), x0.to(DEV), eps.to(DEV)          # keep references: temporaries would alias
        _lib.check(lib.interdiff_posterior_step(_lib.dptr(xd), _lib.dptr(x0d), _lib.dptr(epsd), n, float(c1), float(c2), float(s), 0, 0, _lib.stream()))
        close(xd, c1 * x0 + c2 * x + s * eps, 1e-6, 'posterior n=%d' % n)
        gt, mask = torch.randn(n, generator=g), torch.rand(n, generator=g) < 0.3
        gtd, md = gt.to(DEV), mask.to(DEV).view(torch.uint8)
        _lib.check(lib.interdiff_inpaint(_lib.dptr(x0d), _lib.dptr(gtd), _lib.dptr(md), n, _lib.stream()))
        assert torch.equal(x0d.cpu(), torch.where(mask, gt, x0))


def test_inkernel_noise_is_standard_normal(lib):
    from interdiff_amd import _lib
    n = 1 << 22
    a, b = torch.empty(n, device=DEV), torch.empty(n, device=DEV)
    _lib.check(lib.interdiff_randn(_lib.dptr(a), n, 233, 5, _lib.stream()))
    _lib.check(lib.interdiff_randn(_lib.dptr(b), n, 233, 6, _lib.stream()))
    for v in (a, b):
        assert abs(v.mean().item()) < 3e-3 and abs(v.std().item() - 1) < 3e-3
        assert abs((v ** 4).mean().item() - 3) < 0.05 and v.abs().max().item() < 7
    assert abs((a * b).mean().item()) < 3e-3                         # different steps are independent streams
    c = torch.empty(n, device=DEV)
    _lib.check(lib.interdiff_randn(_lib.dptr(c), n, 233, 5, _lib.stream()))
    assert torch.equal(a, c)                                         # counter-based => reproducible
    # posterior step with generated noise == posterior step with that noise handed in
    x = torch.randn(n, device=DEV)
    x1, x2, x0 = x.clone(), x.clone(), torch.randn(n, device=DEV)
    _lib.check(lib.interdiff_posterior_step(_lib.dptr(x1), _lib.dptr(x0), None, n, 0.25, 0.5, 0.125, 233, 5, _lib.stream()))
    _lib.check(lib.interdiff_posterior_step(_lib.dptr(x2), _lib.dptr(x0), _lib.dptr(a), n, 0.25, 0.5, 0.125, 233, 5, _lib.stream()))
    assert torch.equal(x1, x2)


# ------------------------------------------------------------------------------------------ SMPL (B1-B3), normals (B4)
def test_smpl_forward_golden(smpl):
    z, sub = fx.golden('smpl.npz'), fx.vertex_subset()
    pose, betas, trans = fx.smpl_inputs(4)
    verts, jtr, v_posed, _ = smpl(pose.to(DEV), th_betas=betas.to(DEV), th_trans=trans.to(DEV))
    close(verts[:, sub], z['verts'], 1e-5, 'verts vs reference golden')
    close(jtr, z['jtr'], 1e-5, 'jtr')
    close(v_posed[:, sub], z['v_posed'], 1e-5, 'v_posed')
    from interdiff_amd.geometry import vertex_normals
    n1 = vertex_normals(verts, smpl.th_faces[None].repeat(4, 1, 1))
    close(n1[:, sub], z['normals'], 1e-4, 'normals vs reference golden')
    ref = ogeo.vertex_normals(verts.cpu(), fx.smpl_model()['faces'])
    close(n1, ref, 1e-5, 'normals vs oracle on the same verts')


def test_smpl_forward_ragged_and_identity(smpl):
    model = fx.smpl_model()
    for N in (1, 37):
        pose, betas, trans = fx.smpl_inputs(N)
        verts, jtr, v_posed, _ = smpl(pose.to(DEV), th_betas=betas.to(DEV), th_trans=trans.to(DEV))
        rv, rj, rp = osmpl.smpl_forward(model, pose, betas, trans)
        close(verts, rv, 1e-5, 'verts N=%d' % N)
        close(jtr, rj, 1e-5, 'jtr N=%d' % N)
        close(v_posed, rp, 1e-5, 'v_posed N=%d' % N)
    pose, betas, trans = fx.smpl_inputs(3)
    verts, _, v_posed, _ = smpl(torch.zeros_like(pose).to(DEV), th_betas=betas.to(DEV), th_trans=trans.to(DEV))
    close(verts, v_posed + trans.to(DEV)[:, None], 1e-5, 'identity pose => v_shaped + trans')


# ------------------------------------------------------------------------------------------ NN / signed distance (B5)
def test_point2point_signed_golden(lib):
    from interdiff_amd.geometry import point2point_signed
    z = fx.golden('p2p.npz')
    x, y, xn = fx.p2p_inputs()
    r = point2point_signed(x.to(DEV), y.to(DEV), x_normals=xn.to(DEV), return_vector=True)
    assert np.array_equal(r[2].cpu().numpy().astype(np.int64), z['yidx']), 'yidx must be bit-exact'
    assert np.array_equal(r[3].cpu().numpy().astype(np.int64), z['xidx']), 'xidx must be bit-exact'
    assert r[3][1, 40] == r[3][1, 3] and r[2][0, 5] == 17             # tie -> lowest index; zero distance
    for got, k in zip((r[0], r[1], r[4], r[5]), ('y2x_signed', 'x2y_signed', 'y2x', 'x2y')):
        close(got, z[k], 1e-6, k)
    assert len(point2point_signed(x.to(DEV), y.to(DEV))) == 4
    with pytest.raises(ValueError):
        point2point_signed(x.to(DEV), y[:2].to(DEV))


def test_nn_argmin_full_size_bit_exact(lib):
    """6890 x 2048 (the hot-path size) against the oracle: indices identical, both directions."""
    from interdiff_amd.geometry import nn_argmin
    g = torch.Generator().manual_seed(9)
    x, y = 0.5 * torch.randn(2, 6890, 3, generator=g), 0.5 * torch.randn(2, 2048, 3, generator=g)
    assert torch.equal(nn_argmin(y.to(DEV), x.to(DEV)).cpu().long(), ogeo.nn_argmin(y, x))
    assert torch.equal(nn_argmin(x.to(DEV), y.to(DEV)).cpu().long(), ogeo.nn_argmin(x, y))


def _posed_body(smpl_layer, n_frames, seed):
    g = torch.Generator().manual_seed(seed)
    pose = 0.35 * torch.randn(n_frames, 156, generator=g)
    return smpl_layer(pose.to(DEV), th_betas=torch.randn(n_frames, 10, generator=g).to(DEV), th_trans=0.1 * torch.randn(n_frames, 3, generator=g).to(DEV))[0]


def test_contact_scan_block_culling_is_exact(smpl):
    """The nearest-vertex scan of the hook (csrc/correction.hip corr_contact_kernel: Morton scan order, 16-vertex blocks culled by
    their boxes per wave) against the brute-force oracle: indices bit-identical with and without the scan order, on a posed body,
    with the object near / inside / far from the body, duplicated vertices (ties inside a block, across blocks, a 100-vertex clump
    -> the all-records tie fallback), the winner in the last (partial: 6890 % 16 = 10) block; and the culling really happens."""
    T, B, P = 3, 2, 2048
    g = torch.Generator().manual_seed(77)
    verts = _posed_body(smpl, T * B, 5).reshape(T, B, 6890, 3).clone()
    pts = torch.empty(B, P, 3)
    pts[0] = 0.25 * torch.randn(P, 3, generator=g) + torch.tensor([0.3, 0.1, 0.0])      # overlapping the body
    pts[1] = 0.2 * torch.rand(P, 3, generator=g) + torch.tensor([1.5, -0.8, 0.4])         # a box 1.5 m away
    corr_on, corr_off = make_correction(smpl, 14, P), None
    from interdiff_amd.objprojector import ObjProjector
    from interdiff_amd.correction import HipCorrection
    corr_off = HipCorrection(smpl, ObjProjector(fx.objproj_weights(), T=14, past_len=fx.PAST, device=DEV), n_points=P, past_len=fx.PAST, device=DEV,
                             scan_order=False)
    assert corr_on.ctx.vorder and not corr_off.ctx.vorder
    order = corr_on.topo.vorder.cpu().long()
    # adversarial vertices (frame (t, b)): exact duplicates
    verts[0, 0, 40] = verts[0, 0, 3]                                       # a pair (lowest index must win)
    verts[0, 0, 2000:2100] = verts[0, 0, 2000].clone()                     # a clump of 100 identical vertices
    verts[1, 0, int(order[-1])] = pts[0, 7].to(DEV)                        # the LAST scan position is somebody's nearest vertex (distance 0)
    verts[1, 1, 6889] = verts[1, 1, 0]                                     # duplicate of vertex 0 at the highest index
    pts[0, 11] = verts[0, 0, 3].cpu()                                      # queries sitting exactly on the duplicated vertices
    pts[0, 12] = verts[0, 0, 2050].cpu()
    pts[1, 5] = verts[1, 1, 0].cpu()
    eye, zero = torch.eye(3).expand(T, B, 3, 3).contiguous(), torch.zeros(T, B, 3)
    want = ogeo.nn_argmin(pts[None].expand(T, B, P, 3).reshape(T * B, P, 3), verts.cpu().reshape(T * B, 6890, 3)).reshape(T, B, P)
    assert want[0, 0, 11] == 3 and want[0, 0, 12] == 2000 and want[1, 0, 7] == int(order[-1]) and want[1, 1, 5] == 0
    frac = {}
    for name, corr in (('scan order', corr_on), ('identity order', corr_off)):
        o2h, idx, (scored, tested, boxtests, *_) = corr.contact_nn(verts, pts.to(DEV), eye.to(DEV), zero.to(DEV), want_stats=True)
        assert torch.equal(idx.cpu().long(), want), '%s: %d indices differ' % (name, (idx.cpu().long() != want).sum())
        frac[name] = scored / tested
        frac[name + ' box tests per block'] = boxtests / tested
        if name == 'scan order':
            keep = o2h
        else:
            assert torch.equal(o2h, keep), 'signed distances must not depend on the scan order'
    print('blocks scored / tested:', frac)
    assert frac['scan order'] < 0.5, frac                                  # overlapping + far object, synthetic (loosely skinned) body
    # with a rigid transform: distances against the oracle (indices can legitimately differ where the transformed point differs in
    # its last bit, so compare the unsigned distance)
    Rm = R.axis_angle_to_matrix(torch.randn(T, B, 3, generator=g))
    tr = 0.3 * torch.randn(T, B, 3, generator=g)
    o2h, idx = corr_on.contact_nn(verts, pts.to(DEV), Rm.to(DEV), tr.to(DEV))
    q = torch.matmul(pts[None], Rm.transpose(-1, -2)) + tr[:, :, None]
    near = torch.gather(verts.cpu(), 2, idx.cpu().long()[..., None].expand(-1, -1, -1, 3))
    dist_at_idx = (q - near).norm(dim=-1)
    ref = ogeo.nn_argmin(q.reshape(T * B, P, 3), verts.cpu().reshape(T * B, 6890, 3)).reshape(T, B, P)
    near_ref = torch.gather(verts.cpu(), 2, ref[..., None].expand(-1, -1, -1, 3))
    close(dist_at_idx, (q - near_ref).norm(dim=-1), 1e-6, 'distance to the nearest vertex under a rigid transform')
    close(o2h.abs(), dist_at_idx, 1e-6, '|o2h|')
    assert (idx.cpu().long() == ref).float().mean() > 0.999


def test_contact_scan_small_mesh_and_ragged_points(lib):
    """A body model with V = 1037 (not a multiple of 16, 65 blocks) and P = 1000 object points (fewer than a workgroup's 2048 slots):
    indices against the oracle, both orders."""
    from interdiff_amd import synthetic as syn
    from interdiff_amd.smpl import SMPL_Layer
    from interdiff_amd.objprojector import ObjProjector
    from interdiff_amd.correction import HipCorrection
    V, P, T, B = 1037, 1000, 2, 3
    np_model = syn.smplh_model(seed=11, V=V, F=2100)
    layer = SMPL_Layer({k: torch.from_numpy(v) for k, v in np_model.items()}, device=DEV)      # only its topology / rest pose are used here
    g = torch.Generator().manual_seed(3)
    # "posed" vertices = the rest pose bent smoothly (the body-model kernel itself only supports V = 6890-class tilings)
    vt = torch.from_numpy(np_model['v_template'])
    verts = torch.stack([vt + 0.1 * torch.sin(3.0 * vt.roll(1, dims=1) + i) for i in range(T * B)]).reshape(T, B, V, 3).to(DEV)
    pts = 0.3 * torch.randn(B, P, 3, generator=g)
    eye, zero = torch.eye(3).expand(T, B, 3, 3).contiguous(), torch.zeros(T, B, 3)
    want = ogeo.nn_argmin(pts[None].expand(T, B, P, 3).reshape(T * B, P, 3), verts.cpu().reshape(T * B, V, 3)).reshape(T, B, P)
    for so in (True, False):
        corr = HipCorrection(layer, ObjProjector(fx.objproj_weights(), T=14, past_len=fx.PAST, device=DEV), n_points=P, past_len=fx.PAST, device=DEV,
                             markers=list(range(67)), scan_order=so)
        o2h, idx = corr.contact_nn(verts, pts.to(DEV), eye.to(DEV), zero.to(DEV))
        assert torch.equal(idx.cpu().long(), want), 'scan_order=%s' % so
        assert torch.isfinite(o2h).all()


# ------------------------------------------------------------------------------------------ ObjProjector (D1-D2)
@pytest.mark.parametrize('tag,T,B', [('a', 35, 3), ('b', 100, 2)])
def test_objprojector_golden(lib, tag, T, B):
    from interdiff_amd.objprojector import ObjProjector
    oa, ot, hv, contact = fx.objproj_inputs(T, B)
    op = ObjProjector(fx.objproj_weights(), T=T, past_len=fx.PAST, device=DEV)
    got = op.sample(oa.to(DEV), ot.to(DEV), hv.to(DEV), contact.to(DEV))
    close(got, fx.golden('objproj.npz')['out_' + tag], 1e-4, 'vs reference golden (real checkpoint)')
    close(got, oobj.objprojector_sample(fx.objproj_weights(), oa, ot, hv, contact, fx.PAST), 1e-4, 'vs oracle')


# ------------------------------------------------------------------------------------------ denoised_fn (C1)
def test_denoised_fn_golden(smpl):
    z = fx.golden('denoised_fn.npz')
    T, B, P = fx.DFN_SHAPE
    corr = make_correction(smpl, T, P)
    corr.debug = {}
    x, y = fx.denoised_fn_inputs()
    terms = ocor.correction_terms(x.clone(), dict(y, smpl=fx.smpl_model()), fx.PAST)
    for tval in fx.DFN_TS:
        t = torch.full((B,), tval, dtype=torch.int64, device=DEV)
        got = corr(x.clone().to(DEV), t, {'y': dev(y)})
        close(got, z['out_t%d' % tval], 1e-4, 'denoised_fn t=%d vs reference golden' % tval)
    assert torch.equal(corr.debug['condition'].cpu().bool(), terms['condition'])
    assert torch.equal(corr.debug['contact'].cpu().long(), terms['contact'])
    close(corr.debug['distance'], terms['distance'], 1e-5, 'distance')
    close(corr.debug['loss'], terms['loss'][fx.PAST:].mean(dim=2).mean(dim=0), 1e-4, 'loss')


# ------------------------------------------------------------------------------------------ whole sampler (S1-S5 + everything)
def test_full_1000_step_loop_golden(mdm, smpl):
    """HIP sampler + HIP denoiser + HIP correction over the full 1000 steps (11 corrections) with injected
    noise, against the reference's own p_sample_loop / MDM / denoised_fn run (tests/golden/loop.npz)."""
    from interdiff_amd.diffusion import create_gaussian_diffusion
    z = fx.golden('loop.npz')
    T, B, P = fx.LOOP_SHAPE
    noise, y, stream = fx.loop_inputs()
    corr = make_correction(smpl, T, P)
    diff = create_gaussian_diffusion('cosine', 1000)
    dumps = diff.p_sample_loop(mdm, tuple(noise.shape), noise=noise.to(DEV), clip_denoised=False, model_kwargs={'y': dev(y)},
                               denoised_fn=corr, dump_steps=fx.LOOP_DUMPS, step_noise=lambda i, x: stream.next_like(x).to(DEV))
    worst = 0.0
    for s, d in zip(fx.LOOP_DUMPS, dumps):
        worst = max(worst, close(d, z['dump_%d' % s], 1e-4, 'loop index %d' % s))
    print('full-loop worst rel err %.2e' % worst)
    fx.record_parity('loop_T12_B2_P64_1000steps_vs_reference', worst_rel_err=worst, asserted=1e-4, dumps=list(fx.LOOP_DUMPS))


def _pick(contact):
    """The marker ObjProjector.sample selects from a contact-count row (model/correction_smpl.py:126-134): 0 = none."""
    from oracle.objprojector import HAND_MARKERS
    score = contact.astype(np.float64).copy()
    score[..., HAND_MARKERS] += 0.5
    return np.where(contact.sum(-1) > 0, 1 + score.argmax(-1), 0)


def _poses_fp64(sample, batch):
    """oracle.correction.finalize (eval_smpl_short.py:154-177) in float64 on a sampler state: the pose-level yardstick."""
    from oracle.correction import MARKERS67, split_tokens
    T, B, _ = fx.FULL_SHAPE
    x = torch.as_tensor(sample).double()
    smpl64 = {k: (v.double() if v.is_floating_point() else v) for k, v in fx.smpl_model().items()}
    obj, body, verts, jtr = ocor.finalize(x, batch['gt'].double(), batch['hand_pose'].double(), batch['beta'].double(), smpl64, fx.PAST)
    b6, o6 = split_tokens(x)
    return dict(obj_translation=obj[..., 3:], obj_rotation=R.rotation_6d_to_matrix(o6[..., :6]), body_translation_and_hands=body[..., 66:],
                body_rotations=R.rotation_6d_to_matrix(b6[..., :132].reshape(T, B, 22, 6)), markers=verts[:, :, MARKERS67], joints=jtr)


def _full_size_report(mdm, smpl, names=('full.npz', 'full64.npz'), dump_steps=None):
    """The measurements of test_full_size_end_to_end_golden (also run per arithmetic variant by tests/pose_error_spread.py).  ``names`` / ``dump_steps``:
    the fixture pair (reference run, fp64 twin) and its dump indices -- the well-conditioned twin fixture of round 5 is ('fullwc.npz', 'fullwc64.npz')."""
    from interdiff_amd import eval as ev
    from interdiff_amd.diffusion import create_gaussian_diffusion
    z, z64 = fx.golden(names[0]), fx.golden(names[1])
    dump_steps = dump_steps or fx.FULL_DUMPS
    T, B, P = fx.FULL_SHAPE
    past = fx.PAST
    batch, noise, stream = fx.full_inputs()
    bd = dev(batch)
    corr = make_correction(smpl, T, P)
    corr.debug = {}
    diff = create_gaussian_diffusion('cosine', fx.FULL_STEPS)
    dec = dict(t=[], condition=[], contact=[])

    def hook(x, t, kw):
        out = corr(x, t, kw)
        if corr.is_active(t.host_value):
            dec['t'].append(int(t.host_value))
            dec['condition'].append(corr.debug['condition'].cpu().numpy().astype(bool))
            dec['contact'].append(corr.debug['contact'].cpu().numpy())
        return out
    y = ev.model_kwargs_for(bd, past)
    dumps = diff.p_sample_loop(mdm, tuple(noise.shape), noise=noise.to(DEV), clip_denoised=False, model_kwargs={'y': y}, denoised_fn=hook,
                               dump_steps=dump_steps, step_noise=lambda i, x: stream.next_like(x).to(DEV))
    assert dec['t'] == list(z['corr_t']) == [500 - 50 * k for k in range(11)]
    cond, contact = np.stack(dec['condition']), np.stack(dec['contact'])
    n_dec = cond.size
    rep = dict(shape='B=16 T=100 P=2048, 1000 steps, 11 corrections, injected noise', target_north_star=1e-4)
    rep['condition_flips_vs_reference'] = float((cond != z['condition']).sum()) / n_dec
    rep['contact_marker_flips_vs_reference'] = float((_pick(contact) != _pick(z['contact'])).sum()) / n_dec
    rep['contact_count_rows_differing_vs_reference'] = float((contact != z['contact']).any(-1).sum()) / n_dec
    rep['reference_vs_fp64_condition_flips'] = float((z['condition'] != z64['condition']).sum()) / n_dec
    rep['reference_vs_fp64_marker_flips'] = float((_pick(z['contact']) != _pick(z64['contact'])).sum()) / n_dec
    rep['hip_vs_fp64_marker_flips'] = float((_pick(contact) != _pick(z64['contact'])).sum()) / n_dec
    per_dump = {}
    for s_, d in zip(dump_steps, dumps):
        k = 'dump_%d' % s_
        per_dump[str(s_)] = dict(hip_vs_reference=rel(d, z[k]), hip_vs_fp64=rel(d, z64[k]), reference_vs_fp64=rel(z[k], z64[k]))
    rep['sampler_state_rel_err_by_loop_index'] = per_dump
    # the final poses through the reference's own post-processing (sample_once_proj :154-177), ours on the HIP path
    obj, body, verts, jtr, pelvis = ev.finalize(dumps[-1], bd, smpl, past)
    from oracle.correction import MARKERS67
    fin = dict(obj_translation=rel(obj[..., 3:], z['obj'][..., 3:]),
               obj_rotation=rel(R.axis_angle_to_matrix(obj[..., :3].cpu()), R.axis_angle_to_matrix(torch.from_numpy(z['obj'][..., :3]))),
               body_translation_and_hands=rel(body[..., 66:], z['body'][..., 66:]),
               body_rotations=rel(R.axis_angle_to_matrix(body[..., :66].reshape(T, B, 22, 3).cpu()),
                                  R.axis_angle_to_matrix(torch.from_numpy(z['body'][..., :66]).reshape(T, B, 22, 3))),
               markers=rel(verts[:, :, MARKERS67], z['markers']), joints=rel(jtr, z['jtr']))
    rep['final_outputs_rel_err_vs_reference'] = fin
    # pose-level fp64 yardstick: the oracle's finalize in float64 on the fp64 twin's final sample (full64.npz stores it rounded to
    # fp32: 6e-8, far below what is measured here).  How far the REFERENCE's own fp32 end-to-end poses are from exact arithmetic
    # bounds how closely any fp32 implementation can be asked to match them.
    pose64 = _poses_fp64(z64['dump_999'], batch)
    ref_pose = dict(obj_translation=z['obj'][..., 3:], obj_rotation=R.axis_angle_to_matrix(torch.from_numpy(z['obj'][..., :3])),
                    body_translation_and_hands=z['body'][..., 66:],
                    body_rotations=R.axis_angle_to_matrix(torch.from_numpy(z['body'][..., :66]).reshape(T, B, 22, 3)), markers=z['markers'], joints=z['jtr'])
    hip_pose = dict(obj_translation=obj[..., 3:], obj_rotation=R.axis_angle_to_matrix(obj[..., :3].cpu()), body_translation_and_hands=body[..., 66:],
                    body_rotations=R.axis_angle_to_matrix(body[..., :66].reshape(T, B, 22, 3).cpu()), markers=verts[:, :, MARKERS67], joints=jtr)
    rep['final_outputs_reference_vs_fp64'] = {k: rel(ref_pose[k], pose64[k]) for k in ref_pose}
    rep['final_outputs_hip_vs_fp64'] = {k: rel(hip_pose[k], pose64[k]) for k in hip_pose}
    # the same post-processing applied to the REFERENCE's own final sample: separates the error of the conversion / body-model
    # kernels from the sensitivity of rot6d -> axis-angle -> SMPL to the 6e-5 that the two samples differ by (with a random-init
    # denoiser some joints' 6-D vectors are close to parallel, where Gram-Schmidt amplifies any input difference)
    o2, b2, v2, j2, _ = ev.finalize(torch.from_numpy(z['dump_999']).to(DEV), bd, smpl, past)
    fin_same = dict(obj_translation=rel(o2[..., 3:], z['obj'][..., 3:]),
                    obj_rotation=rel(R.axis_angle_to_matrix(o2[..., :3].cpu()), R.axis_angle_to_matrix(torch.from_numpy(z['obj'][..., :3]))),
                    body_rotations=rel(R.axis_angle_to_matrix(b2[..., :66].reshape(T, B, 22, 3).cpu()),
                                       R.axis_angle_to_matrix(torch.from_numpy(z['body'][..., :66]).reshape(T, B, 22, 3))),
                    markers=rel(v2[:, :, MARKERS67], z['markers']), joints=rel(j2, z['jtr']))
    rep['final_outputs_rel_err_on_the_reference_sample'] = fin_same
    # fp64 anchor of the rot6d -> matrix step on that same sample (oracle/rotations.py in double): how far the reference's OWN fp32
    # conversion is from the exact answer there, next to ours
    from oracle.correction import split_tokens
    from oracle import rotations as OR
    body64, _ = split_tokens(torch.from_numpy(z['dump_999']).double())
    m64 = OR.rotation_6d_to_matrix(body64[..., :132].reshape(T, B, 22, 6))
    m_ref = R.axis_angle_to_matrix(torch.from_numpy(z['body'][..., :66]).reshape(T, B, 22, 3)).double()
    m_hip = R.axis_angle_to_matrix(b2[..., :66].reshape(T, B, 22, 3).cpu()).double()
    rot_anchor = dict(hip_vs_fp64=rel(m_hip, m64), reference_vs_fp64=rel(m_ref, m64))
    rep['body_rotations_on_the_reference_sample_vs_fp64'] = rot_anchor
    obj_gt, jtr_gt, body_gt, faces = ev.get_gt(bd, smpl)
    m = ev.Metrics(corr)(obj[past:], jtr[past:], body[past:], obj_gt[past:], jtr_gt[past:], body_gt[past:], verts[past:], faces, bd['obj_points'])
    rep['metrics_rel_err_vs_reference'] = {k: rel(m[k], z['m_' + k]) for k in m}
    rep['metrics_mean_hip'] = {k: float(m[k].mean()) for k in m}
    rep['metrics_mean_reference'] = {k: float(z['m_' + k].mean()) for k in m}
    return rep, per_dump, fin, fin_same, rot_anchor


def test_full_size_end_to_end_golden(mdm, smpl):
    """BASELINE config #2 itself, end to end (SURVEY.md §8(d) parity step 4): B=16, T=100, P=2048, the full 1000 steps with 11
    corrections and injected noise, against the REFERENCE's own sample_once_proj / get_gt / metrics run (tests/golden/full.npz) and
    against the oracle's fp64 twin (full64.npz).  Reported (gpurun_out/parity_r03.json -> profiles/): worst relative error of the
    sampler state at every dump, of the final poses / joints / markers, of the six metrics, the fraction of (call, clip) hook
    decisions that differ, and the same distances of the reference's fp32 run from the fp64 twin -- the yardstick: after the
    decisions start to act (t <= 500) two fp32 implementations can only agree as well as fp32 agrees with exact arithmetic."""
    rep, per_dump, fin, fin_same, rot_anchor = _full_size_report(mdm, smpl)
    fx.record_parity('full_size_end_to_end', **rep)
    print(rep)
    # Gates: north_star's 1e-4 on the sampler state at EVERY dump, the final sample included (measured on MI355X: <= 5.2e-7 up to
    # loop index 949, 2.1e-5 after the last correction = the reference's own fp32 distance from the fp64 twin; ours is 1.7e-6 from it), no decision
    # of the hook may differ, the conversion / body-model kernels within 1e-4 on identical input, the metrics within 2e-4
    # (penetration ratio: a count of sign decisions over 2048 x 90 points per clip, 2e-3).
    for s_, e in per_dump.items():
        assert e['hip_vs_reference'] <= 1e-4, (s_, e)
    assert rep['condition_flips_vs_reference'] == 0 and rep['contact_marker_flips_vs_reference'] == 0
    for k, e in fin_same.items():
        if k != 'body_rotations':
            assert e <= 1e-4, (k, e)
    # body rotations: a random-init denoiser leaves some joints' two 3-vectors nearly parallel, where Gram-Schmidt amplifies fp32
    # rounding beyond 1e-4 for ANY fp32 implementation: the gate is the distance to the fp64 answer, ours no further than the
    # reference's own (+1e-5), and 1e-4 between the two wherever that is attainable
    assert rot_anchor['hip_vs_fp64'] <= max(1e-4, rot_anchor['reference_vs_fp64'] + 1e-5), rot_anchor
    assert fin_same['body_rotations'] <= max(1e-4, 2 * rot_anchor['reference_vs_fp64']), (fin_same, rot_anchor)
    for k, e in rep['metrics_rel_err_vs_reference'].items():
        assert e <= (2e-3 if k == 'penetrate' else 2e-4), (k, e)
    # final poses (what north_star names).  Two gates per quantity, with yard = the REFERENCE's own fp32 distance from the fp64 answer.  After 1000
    # steps, 11 corrections and rot6d -> matrix -> SMPL the distance of an fp32-grade run from the exact answer is a chaotic function of its roundings:
    # five variants of this denoiser that all compute a forward to 4-5e-7 (exact fp32 MFMA with the 32- / 16- / 64-row feed-forward tile, split-f16
    # feed-forward + QKV, split-f16 everywhere) land at 4.7e-4 / 9.3e-5 / 9.3e-5 / 4.7e-4 / 1.4e-3 on the body rotations, the reference's own run at
    # 1.0e-3 (profiles/r04_pose_error_spread.txt, `python -m tests.pose_error_spread`).  So both gates bound the SCALE of the error by the reference's
    # realisation, not the realisation itself:
    #   (i)  HIP vs the fp64 answer: within twice the reference's own distance                hip_vs_fp64 <= max(1e-4, 2 yard)
    #        (until round 4 this read yard + 1e-5, which the variants then shipped happened to meet; it is a coin flip between two fp32-grade runs)
    #   (ii) HIP vs the reference: 1e-4 wherever fp32 can deliver it, else inside the ball both fp32 runs live in around the exact answer (triangle
    #        inequality over (i) and the yardstick): <= max(1e-4, 3 yard)
    # What is NOT relaxed: the sampler state within 1e-4 of the reference at every dump, no decision flipped, the kernels on identical input within 1e-4,
    # the metrics; and per forward the split-f16 form must be as close to fp64 as the exact form (test_mdm_forward_split_f16_vs_exact_and_fp64).
    for k, e in fin.items():
        yard = rep['final_outputs_reference_vs_fp64'][k]
        assert rep['final_outputs_hip_vs_fp64'][k] <= max(1e-4, 2 * yard), ('final %s vs fp64' % k, rep['final_outputs_hip_vs_fp64'][k], yard)
        assert e <= max(1e-4, 3 * yard), ('final %s vs reference' % k, e, yard)


@pytest.mark.parametrize('math', ['split', 'exact'])
def test_full_size_end_to_end_well_conditioned_golden_flat_1e4(smpl, math):
    """North_star's tolerance on the quantities it names -- final poses and object trajectories within 1e-4 relative of the reference's own
    eval_smpl_short.py path -- at BASELINE config #2 (B=16, T=100, P=2048, 1000 steps, 11 corrections, injected noise), as a FLAT gate: no yardstick
    term, every pose quantity, under both arithmetics (shipped split-f16 and exact fp32 MFMA).  The fixture (tests/golden/fullwc.npz = the REFERENCE's
    sampler + MDM + denoised_fn + sample_once_proj / get_gt / metrics, fullwc64.npz = the oracle's fp64 twin; `make_golden.py fullwc`, `fullwc64`) differs
    from full.npz in ONE thing: the synthetic denoiser's output heads are well conditioned (tests/fixtures.py mdm_weights_wc: 0.05 x the random heads
    around a bias that is a valid pose), so the Gram-Schmidt step of rot6d -> matrix is nowhere near singular and fp32 CAN deliver 1e-4 -- on full.npz
    (random-init heads, |cos| of a joint's two 3-vectors up to 0.99998) the reference's own fp32 run is 1e-3 from the exact answer and no fp32
    implementation can be held to 1e-4 there (test_full_size_end_to_end_golden keeps that fixture for the sampler state, the decisions and the metrics)."""
    from interdiff_amd.mdm import MDM
    model = MDM(fx.mdm_weights_wc(), device=DEV)
    model.ffn_math = math
    if math == 'split':
        assert model.arithmetic_report()['all_split'], model.arithmetic_report()      # the fixture's weights pass every f16 range proof: nothing silently on the exact kernels
    rep, per_dump, fin, fin_same, rot_anchor = _full_size_report(model, smpl, names=('fullwc.npz', 'fullwc64.npz'), dump_steps=fx.FULLWC_DUMPS)
    rep['arithmetic'] = math
    fx.record_parity('full_size_end_to_end_well_conditioned_%s' % math, **rep)
    print(rep)
    FLAT = 1e-4
    for s_, e in per_dump.items():
        assert e['hip_vs_reference'] <= FLAT, ('sampler state', s_, e)
    assert rep['condition_flips_vs_reference'] == 0 and rep['contact_marker_flips_vs_reference'] == 0
    for k, e in fin.items():                       # obj translation / rotation, body translation + hands, body rotations, markers, joints: HIP vs the reference
        assert e <= FLAT, ('final %s vs reference' % k, e)
    for k, e in rep['final_outputs_hip_vs_fp64'].items():
        assert e <= FLAT, ('final %s vs fp64' % k, e)
    for k, e in fin_same.items():                  # conversion / body-model kernels on the reference's own final sample
        assert e <= FLAT, ('on the reference sample: %s' % k, e)
    # the fixture does what it was built for: the reference's own fp32 run is well inside 1e-4 of the exact answer on every pose quantity
    for k, e in rep['final_outputs_reference_vs_fp64'].items():
        assert e <= 5e-5, ('yardstick: reference vs fp64 %s' % k, e)
    for k, e in rep['metrics_rel_err_vs_reference'].items():
        assert e <= (2e-3 if k == 'penetrate' else 2e-4), (k, e)


def test_exclusive_cu_claims_are_verified_and_hold(mdm, smpl):
    """Every kernel that issues the f16 MFMA must own its CU (DESIGN.md "exclusive CU"): (i) the launchers' own verification -- occupancy query == 1,
    160 KiB of LDS, >= 256 registers allocated -- passes for every such kernel on this device (a kernel that fails runs as its fp32 counterpart: on MI355X none
    may); (ii) the effect the rule exists for does not show: the hook's SMPL stage (one-wave pose kernel + skinning) computes the same bits alone and while the
    split-f16 feed-forward kernel runs on a second stream (the reproducer of round 4, tools/hook_stage_probe.py, as a test)."""
    from interdiff_amd import _lib
    from interdiff_amd.mdm import ffn_parts
    txt, bad = _lib.exclusive_cu_report()
    print(txt)
    assert bad == 0, txt
    rows = [ln for ln in txt.strip().split('\n') if ln]
    assert len(rows) >= 16 and all('exclusive' in ln and 'NOT' not in ln for ln in rows), txt      # 3 + 2 + 1 + 2 + 8 kernel instantiations
    fx.record_parity('exclusive_cu_report', kernels=len(rows), not_exclusive=bad, table=rows)
    assert mdm.ffn_math == 'split'
    g = torch.Generator().manual_seed(0)
    N = 800
    pose, betas, trans = (0.3 * torch.randn(N, 156, generator=g)).to(DEV), torch.randn(N, 10, generator=g).to(DEV), (0.3 * torch.randn(N, 3, generator=g)).to(DEV)
    x2, parts = torch.randn(N, 256, generator=g).to(DEV), torch.empty(5, N, 256, device=DEV)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

    def victim(load):
        torch.cuda.synchronize()
        if load:
            with torch.cuda.stream(sb):
                for i in range(200):
                    mdm.ffn_rows = 16 if i & 1 else 32
                    ffn_parts(mdm, x2, i % 8, out=parts)
            mdm.ffn_rows = 0
        with torch.cuda.stream(sa):
            v, j, _, _ = smpl(pose, th_betas=betas, th_trans=trans, want_v_posed=False)
            v, j = v.clone(), j.clone()
        torch.cuda.synchronize()
        return v, j
    v0, j0 = victim(False)
    differing = 0
    for rep in range(8):
        v, j = victim(True)
        differing += 0 if (torch.equal(v, v0) and torch.equal(j, j0)) else 1
    fx.record_parity('smpl_stage_beside_split_f16_ffn', runs=8, runs_that_differ=differing)
    assert differing == 0, 'the SMPL stage computed different bits beside the split-f16 feed-forward kernel in %d of 8 runs' % differing


DENY_CASES = [
    # (deny patterns, contraction the report must call 'exact' [key, layer or None], what the pattern takes away)
    ('ffn_h2_kernel', ('ffn', 3), 'the fused feed-forward block -> ffn.h fp32 kernel (csrc/denoiser.hip idf_launch_layer_ffn)'),
    ('ln_linear_h2_kernel', ('qkv', 0), 'both QKV projections, planes route included -> ffn.h ln_linear_kernel; the attention then splits fp32 rows itself'),
    ('planes out', ('qkv_hands_over_planes', 0), 'only the plane-pair output of the projection -> fp32 rows out of the split projection, split attention'),
    ('self_attn_h2_kernel<planes in>', ('qkv_hands_over_planes', 7), 'only the planes-in attention -> fp32 rows out of the projection, attention splits them itself'),
    ('self_attn_h2_kernel', ('self_attention', 7), 'every split-f16 attention (the planes route with it) -> fp32 self-attention with the out-projection in its tail'),
    ('rowblock8_kernel', None, 'the eight-wave row block -> round 4\'s four-wave split kernel (still split-f16)'),
    ('rowblock', ('rowblock', 2), 'both split row blocks -> fp32 row block'),
    ('step_tail_h2_kernel', ('embedding_and_heads', None), 'the step tail -> fp32 embedding GEMM and heads GEMM (gemm.h), unchained steps'),
    ('ffn_h2_kernel,ln_linear_h2_kernel,self_attn_h2_kernel,rowblock,step_tail_h2_kernel', ('ffn', 0), 'everything: the whole forward on fp32 kernels although the split arithmetic is selected'),
]


@pytest.mark.parametrize('case', range(len(DENY_CASES)))
def test_fp32_fallback_behind_every_split_f16_launcher(lib, case):
    """VERDICT r05 'what is weak' 1: the fp32 kernels behind the split-f16 launchers (csrc/common.h idf_exclusive_cu -> IDF_NOT_EXCLUSIVE) never ran on the only device the
    tests run on, where every kernel gets its CU.  The debug deny list (interdiff_debug_deny_exclusive) refuses kernels by name; each case denies one family, checks that
    ``MDM.arithmetic_report()`` names the downgrade, and runs (i) a forward at the bench clip length and a ragged one against oracle/denoiser.py at 1e-4 and (ii) a 30-step
    chained plain-step window (graph route) against the eager route bit for bit and against the oracle's p_sample_loop at 1e-4 -- in the MIXED arithmetic that results."""
    from interdiff_amd import _lib
    from interdiff_amd.mdm import MDM
    from interdiff_amd.diffusion import create_gaussian_diffusion
    patterns, expect, _ = DENY_CASES[case]
    sd = fx.mdm_weights()
    try:
        _lib.debug_deny_exclusive(patterns)
        model = MDM(sd, device=DEV)
        assert model.ffn_math == 'split'
        x, ts, cond = fx.mdm_inputs(4, 100)
        got = model(x.to(DEV), ts.to(DEV), y={'cond': cond.to(DEV)})            # (launches first: the verdict table fills as launchers ask)
        rep = model.arithmetic_report()
        assert rep['not_exclusive'], rep
        assert all(any(p in n for p in patterns.split(',')) for n in rep['not_exclusive']), rep['not_exclusive']
        if expect is None:
            assert rep['all_split'], rep                                           # the four-wave split row block took over: still split-f16 everywhere
        else:
            key, layer = expect
            val = rep[key] if layer is None else rep['layers'][layer][key]
            assert val in ('exact', False), (key, layer, val, rep)
            assert not rep['all_split'] or key == 'qkv_hands_over_planes', rep
        e1 = close(got, oden.mdm_forward(sd, x, ts, cond), 1e-4, 'forward, denied: ' + patterns)
        x2, ts2, cond2 = fx.mdm_inputs(3, 37)
        e2 = close(model(x2.to(DEV), ts2.to(DEV), y={'cond': cond2.to(DEV)}), oden.mdm_forward(sd, x2, ts2, cond2), 1e-4, 'ragged forward, denied: ' + patterns)
        # a window of plain steps: graph route (chained steps where the tail still runs) == eager route fed the same Philox stream, and both vs the oracle
        diff = create_gaussian_diffusion('cosine', 1000)
        bt, y = fx.timed_inputs(2)
        yd, x_t = dev(y), bt['noise'].to(DEV)
        n, seed = 30, 99
        g = diff.p_sample_loop(model, tuple(x_t.shape), noise=x_t, clip_denoised=False, model_kwargs={'y': yd}, seed=seed, n_steps=n, first_t=fx.TIMED_FIRST_T)
        draw = _philox_step(lib, seed)
        stream = [draw(it, x_t) for it in range(n)]
        ea = diff.p_sample_loop(model, tuple(x_t.shape), noise=x_t, clip_denoised=False, model_kwargs={'y': yd}, step_noise=torch.stack(stream), n_steps=n, first_t=fx.TIMED_FIRST_T)
        assert torch.equal(g, ea), ('graph route != eager route, denied: ' + patterns, float((g - ea).abs().max()))
        ref = odf.p_sample_loop(lambda x, t, y: oden.mdm_forward(sd, x, t, y['cond']), tuple(x_t.shape), odf.make_schedule(1000),
                                bt['noise'].clone(), lambda i, xx: stream[i].cpu(), {'y': y}, n_steps=n, first_t=fx.TIMED_FIRST_T)
        e3 = close(g, ref, 1e-4, '30 plain steps, denied: ' + patterns)
        fx.record_parity('fallback_denied_%d' % case, denied=patterns, forward=e1, ragged_forward=e2, window_30_steps=e3, not_exclusive=rep['not_exclusive'])
    finally:
        _lib.debug_deny_exclusive('')
    txt, bad = _lib.exclusive_cu_report()
    assert bad == 0, txt                                                           # the list is cleared: every kernel owns its CU again


def test_one_layer_failing_its_f16_range_proof_runs_mixed(lib):
    """A model where ONE layer's feed-forward block and ANOTHER layer's row block fail their pack-time f16 range proofs (huge LayerNorm gains: mdm.py ffn_h2_range_ok /
    ln_h2_range_ok) keeps those two on the exact fp32 kernels and everything else split-f16: the 50-step chain in that mixed arithmetic against the oracle at 1e-4,
    graph route == eager route bit for bit (only all-split and all-exact were gated end to end before round 6)."""
    from interdiff_amd.mdm import MDM
    from interdiff_amd.diffusion import create_gaussian_diffusion
    sd = {k: torch.as_tensor(v).clone() for k, v in fx.mdm_weights().items()}
    sd['decoder.layers.3.norm2.weight'] *= 6000.0          # layer 3's feed-forward input bound 16 max|gamma| + max|beta| leaves the f16 range
    sd['decoder.layers.5.norm1.weight'] *= 6000.0          # layer 5's row-block operand (its own norm1 output) does too
    model = MDM(sd, device=DEV)
    rep = model.arithmetic_report()
    assert rep['layers'][3]['ffn'] == 'exact' and rep['layers'][5]['rowblock'] == 'exact' and not rep['all_split'], rep
    assert sum(d['ffn'] == 'split' for d in rep['layers']) == 7 and sum(d['rowblock'] == 'split' for d in rep['layers']) >= 6, rep
    x, ts, cond = fx.mdm_inputs(4, 100)
    e1 = close(model(x.to(DEV), ts.to(DEV), y={'cond': cond.to(DEV)}), oden.mdm_forward(sd, x, ts, cond), 1e-4, 'mixed-arithmetic forward')
    diff = create_gaussian_diffusion('cosine', 1000)
    bt, y = fx.timed_inputs(2)
    yd, x_t = dev(y), bt['noise'].to(DEV)
    n, seed = 50, 1234
    g = diff.p_sample_loop(model, tuple(x_t.shape), noise=x_t, clip_denoised=False, model_kwargs={'y': yd}, seed=seed, n_steps=n, first_t=fx.TIMED_FIRST_T)
    draw = _philox_step(lib, seed)
    stream = [draw(it, x_t) for it in range(n)]
    ea = diff.p_sample_loop(model, tuple(x_t.shape), noise=x_t, clip_denoised=False, model_kwargs={'y': yd}, step_noise=torch.stack(stream), n_steps=n, first_t=fx.TIMED_FIRST_T)
    assert torch.equal(g, ea), float((g - ea).abs().max())
    ref = odf.p_sample_loop(lambda x, t, y: oden.mdm_forward(sd, x, t, y['cond']), tuple(x_t.shape), odf.make_schedule(1000),
                            bt['noise'].clone(), lambda i, xx: stream[i].cpu(), {'y': y}, n_steps=n, first_t=fx.TIMED_FIRST_T)
    e2 = close(g, ref, 1e-4, '50 plain steps in mixed arithmetic vs oracle')
    fx.record_parity('mixed_arithmetic_one_layer_exact', forward=e1, window_50_steps=e2)


def test_coresidency_reproducer_and_the_integrator_rule(tmp_path):
    """The stand-alone reproducer of the co-residency effect (tools/coresidency_repro.hip; DESIGN.md "exclusive CU") as a test.
    ASSERTED -- the rule an integrator is given: (i) a victim with NO packed-fp32 instruction (-fno-slp-vectorize) computes the same bits beside every aggressor form;
    (ii) the controls of the default build (no aggressor, fp32 MFMA + loads, loads alone) are bit-stable.  RECORDED, not asserted (it is the hardware's behaviour, and a
    later firmware may well end it): how many launches of the packed-fp32 victim differ beside the two f16-MFMA aggressor forms."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('no hipcc on this box')
    src = os.path.join(os.path.dirname(__file__), '..', 'tools', 'coresidency_repro.hip')
    out = {}
    for name, extra in (('default', []), ('noslp', ['-fno-slp-vectorize'])):
        exe = str(tmp_path / ('repro_' + name))
        subprocess.run([hipcc, '--offload-arch=gfx950', '-O3'] + extra + [src, '-o', exe], check=True, timeout=300)
        txt = subprocess.run([exe, '2', '5000', '1024'], check=True, timeout=300, capture_output=True, text=True).stdout
        rows = dict((m.group(1).strip(), int(m.group(2))) for m in re.finditer(r'^(.*?)\s*: (\d+) of \d+ victim launches differ', txt, re.M))
        assert len(rows) == 5, txt
        out[name] = rows
    fx.record_parity('coresidency_reproducer', launches_per_form=48, **{k: v for k, v in out.items()})
    assert all(v == 0 for v in out['noslp'].values()), out['noslp']
    for control in ('no aggressor', 'fp32 MFMA + global loads', 'global loads only'):
        assert out['default'][control] == 0, out['default']


def test_evaluate_batch_and_sample_once(mdm, smpl):
    """The outer-loop entries (eval_smpl_short.py:179-215,252-296): ``sample_once`` (mode no_correction) against the oracle chain,
    ``evaluate_batch`` = min over independent draws of the per-clip metrics, and the seeding contract: two calls that pass no seed
    draw different per-step noise (upstream: a fresh randn_like per step per call), the same seed reproduces bit for bit."""
    from interdiff_amd import eval as ev
    from interdiff_amd.mdm import MDM
    from interdiff_amd.diffusion import create_gaussian_diffusion
    T, B, P = fx.EVAL_SHAPE
    past = fx.PAST
    batch, noise, stream = fx.eval_inputs()
    model = MDM(fx.mdm_weights(), device=DEV, n_steps=fx.EVAL_STEPS)
    corr = make_correction(smpl, T, P)
    diff = create_gaussian_diffusion('cosine', fx.EVAL_STEPS)
    bd = dev(batch)
    # sample_once (no hook) vs the oracle's loop + finalize, injected noise
    obj, body, verts, jtr, pelvis = ev.sample_once(model, diff, smpl, bd, past, noise=noise.to(DEV),
                                                   step_noise=lambda i, x: stream.next_like(x).to(DEV))
    _, _, stream2 = fx.eval_inputs()
    y = fx.model_kwargs_y(dict(batch, noise=noise), T)
    ref = odf.p_sample_loop(lambda x, t, y: oden.mdm_forward(fx.mdm_weights(), x, t, y['cond']), tuple(noise.shape),
                            odf.make_schedule(fx.EVAL_STEPS), noise.clone(), lambda i, x: stream2.next_like(x), {'y': y})
    o_ref, b_ref, v_ref, j_ref = ocor.finalize(ref, batch['gt'], batch['hand_pose'], batch['beta'], fx.smpl_model(), past)
    e = max(close(obj[..., 3:], o_ref[..., 3:], 1e-4, 'sample_once obj translation'), close(jtr, j_ref, 1e-4, 'sample_once joints'),
            close(verts, v_ref, 1e-4, 'sample_once verts'), close(body[..., 66:], b_ref[..., 66:], 1e-4, 'sample_once body'))
    fx.record_parity('sample_once_no_correction_50steps_vs_oracle', worst_rel_err=e, asserted=1e-4)
    # seeding contract
    nz = noise.to(DEV)
    kw = dict(clip_denoised=False, model_kwargs={'y': ev.model_kwargs_for(bd, past)}, denoised_fn=corr)
    a = diff.p_sample_loop(model, tuple(nz.shape), noise=nz, **kw)
    b = diff.p_sample_loop(model, tuple(nz.shape), noise=nz, **kw)
    assert not torch.equal(a, b), 'two unseeded calls must draw different per-step noise'
    c, d = diff.p_sample_loop(model, tuple(nz.shape), noise=nz, seed=5, **kw), diff.p_sample_loop(model, tuple(nz.shape), noise=nz, seed=5, **kw)
    assert torch.equal(c, d)
    # evaluate_batch: per-clip minimum over independent draws, each draw reproducible from (seed + j)
    m3 = ev.evaluate_batch(model, diff, corr, bd, past, 'correction', diverse_samples=3, seed=40)
    singles = [ev.evaluate_batch(model, diff, corr, bd, past, 'correction', diverse_samples=1, seed=40 + j) for j in range(3)]
    for k in m3:
        want = torch.stack([s_[k] for s_ in singles]).min(dim=0)[0]
        assert torch.equal(m3[k], want), k
        assert m3[k].shape == (B,) and torch.isfinite(m3[k]).all()
    assert any(not torch.equal(singles[0][k], singles[1][k]) for k in m3), 'draws must be independent samples'
    nc = ev.evaluate_batch(model, diff, corr, bd, past, 'no_correction', diverse_samples=1, seed=40)
    assert all(torch.isfinite(v).all() for v in nc.values())
    full, means = ev.evaluate_sharded(model, diff, corr, bd, past, 'correction', diverse_samples=1, seed=40)     # world size 1: degenerate gather
    assert all(torch.equal(full[k], singles[0][k]) for k in full) and abs(means['global_mpjpe'] - float(singles[0]['global_mpjpe'].mean())) < 1e-6


# ------------------------------------------------------------------------------------------ eval glue + metrics (E1, E2)
def test_eval_glue_and_metrics_golden(mdm, smpl):
    """HIP sample_once_proj / get_gt / metrics against the reference's own functions (tests/golden/eval.npz):
    50-step schedule, injected noise, correction hook at t = 0."""
    from interdiff_amd import eval as ev
    from interdiff_amd.mdm import MDM
    from interdiff_amd.diffusion import create_gaussian_diffusion
    z = fx.golden('eval.npz')
    T, B, P = fx.EVAL_SHAPE
    past = fx.PAST
    batch, noise, stream = fx.eval_inputs()
    model = MDM(fx.mdm_weights(), device=DEV, n_steps=fx.EVAL_STEPS)
    corr = make_correction(smpl, T, P)
    diff = create_gaussian_diffusion('cosine', fx.EVAL_STEPS)
    bd = dev(batch)
    obj, body, verts, jtr, pelvis = ev.sample_once_proj(model, diff, corr, bd, past, noise=noise.to(DEV),
                                                        step_noise=lambda i, x: stream.next_like(x).to(DEV))
    sub = fx.vertex_subset()
    close(obj[..., 3:], z['obj'][..., 3:], 1e-4, 'obj translation')
    close(R.axis_angle_to_matrix(obj[..., :3].cpu()), R.axis_angle_to_matrix(torch.from_numpy(z['obj'][..., :3])), 1e-4, 'obj rotation')
    close(body[..., 66:], z['body'][..., 66:], 1e-4, 'hands + translation')
    close(R.axis_angle_to_matrix(body[..., :66].reshape(T, B, 22, 3).cpu()),
          R.axis_angle_to_matrix(torch.from_numpy(z['body'][..., :66]).reshape(T, B, 22, 3)), 1e-4, 'body rotations')
    close(verts[:, :, sub], z['verts'], 1e-4, 'verts')
    close(jtr, z['jtr'], 1e-4, 'jtr')
    close(pelvis, z['pelvis'], 1e-4, 'pelvis')
    obj_gt, jtr_gt, body_gt, faces = ev.get_gt(bd, smpl)
    close(jtr_gt, z['jtr_gt'], 1e-5, 'jtr_gt')
    close(obj_gt[..., 3:], z['obj_gt'][..., 3:], 1e-6, 'obj_gt')
    assert torch.equal(faces.cpu(), fx.smpl_model()['faces'].long())
    met = ev.Metrics(corr)
    m = met(obj[past:], jtr[past:], body[past:], obj_gt[past:], jtr_gt[past:], body_gt[past:], verts[past:], faces, bd['obj_points'])
    for k in m:
        close(m[k], z['m_' + k], 2e-4 if k != 'penetrate' else 2e-2, 'metric %s vs reference golden' % k)
    # the metric kernel alone, on the reference's own sample: tight
    full = lambda k: torch.from_numpy(z[k]).to(DEV)
    m2 = met(full('obj')[past:], full('jtr')[past:], full('body')[past:], full('obj_gt')[past:], full('jtr_gt')[past:],
             full('body_gt')[past:], verts[past:], faces, bd['obj_points'])
    for k in m2:
        close(m2[k], z['m_' + k], 1e-5 if k != 'penetrate' else 2e-2, 'metric kernel %s' % k)
    # min over diverse samples + smooth (host logic)
    o2 = ev.smooth(obj.clone(), body.clone(), verts.clone(), jtr.clone(), pelvis.clone(), T - past)[0]
    ref = obj.clone()
    ref[-(T - past):] = ref[-(T - past):] + (2 * obj[-(T - past) - 1] - obj[-(T - past) - 2] - obj[-(T - past)])
    assert torch.equal(o2, ref)


# ------------------------------------------------------------------------------------------ hipGraph route of the sampler
def test_graph_replay_equals_eager(smpl):
    """The captured plain-step graph (device-side scalars) and the eager loop produce bit-identical samples, with and
    without the correction hook (50-step schedule: the hook fires at t = 0), twice in a row (graph reuse)."""
    from interdiff_amd.mdm import MDM
    from interdiff_amd.diffusion import create_gaussian_diffusion
    T, B, P = fx.EVAL_SHAPE
    batch, noise, _ = fx.eval_inputs()
    y = dev(fx.model_kwargs_y(dict(batch, noise=noise), T))
    model = MDM(fx.mdm_weights(), device=DEV, n_steps=fx.EVAL_STEPS)
    corr = make_correction(smpl, T, P)
    diff = create_gaussian_diffusion('cosine', fx.EVAL_STEPS)
    nz = noise.to(DEV)
    for hook in (None, corr):
        eager = diff.p_sample_loop(model, tuple(nz.shape), noise=nz, clip_denoised=False, model_kwargs={'y': y}, denoised_fn=hook,
                                   seed=77, use_graph=False)
        for rep in range(2):
            graph = diff.p_sample_loop(model, tuple(nz.shape), noise=nz, clip_denoised=False, model_kwargs={'y': y}, denoised_fn=hook,
                                       seed=77)
            assert torch.equal(eager, graph), 'graph route differs (hook=%s, rep %d): %g' % (hook is not None, rep, (eager - graph).abs().max())
        other = diff.p_sample_loop(model, tuple(nz.shape), noise=nz, clip_denoised=False, model_kwargs={'y': y}, denoised_fn=hook, seed=78)
        assert not torch.equal(other, eager)
    assert len(model._graph_cache) == 1          # one capture per (schedule, shape, mask, cond shape), living on the denoiser
    # plain steps with the update inside the denoiser's last GEMM (interdiff_mdm_forward_step, the default: T = 14 above took its
    # per-row form, T = 12 here its 16-byte form) against the two-call form and against the eager loop
    assert T % 4 != 0 and diff.fuse_plain_step
    bt = fx._clip(41, 2, 12, 64)
    y12, n12 = dev(fx.model_kwargs_y(bt, 12)), bt['noise'].to(DEV)
    run = lambda **kw: diff.p_sample_loop(model, tuple(n12.shape), noise=n12, clip_denoised=False, model_kwargs={'y': y12}, seed=3, **kw)
    fused = run()
    diff.fuse_plain_step = False
    two_call = run()
    diff.fuse_plain_step = True
    assert torch.equal(two_call, fused) and torch.equal(fused, run(use_graph=False))


# ------------------------------------------------------------------------------------------ the token GEMM as an op
def test_gemm_f32_epilogues_vs_torch_fp32(lib):
    """interdiff_gemm_f32 (LDS-DMA pipelined fp32-MFMA GEMM) against torch CPU in float64, every epilogue and every tile
    configuration, bench shape + ragged M; asymmetric operands catch transposes."""
    from interdiff_amd.mdm import linear
    g = torch.Generator().manual_seed(12)
    for M, N, K in ((1600, 256, 1024), (1600, 1024, 256), (37, 144, 256), (1, 16, 64)):
        x, w, b, r = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g), torch.randn(M, N, generator=g)
        ref = x.double() @ w.double().T + b.double()
        xd, wd, bd, rd = x.to(DEV), w.to(DEV), b.to(DEV), r.to(DEV)
        for cfg in range(0, 10):
            close(linear(xd, wd, bd, cfg=cfg), ref, 2e-6, 'bias M=%d N=%d K=%d cfg=%d' % (M, N, K, cfg))
        close(linear(xd, wd, bd, gelu=True), torch.nn.functional.gelu(ref), 2e-6, 'gelu')
        close(linear(xd, wd, bd, residual=rd), ref + r.double(), 2e-6, 'residual')
        close(linear(xd, wd), x.double() @ w.double().T, 2e-6, 'no bias')
    with pytest.raises(RuntimeError):
        linear(torch.randn(4, 48).to(DEV), torch.randn(16, 48).to(DEV))          # K % 64 != 0


@pytest.mark.parametrize('math', ['exact', 'split'])
@pytest.mark.parametrize('M', [1, 15, 17, 31, 33, 63, 65, 160, 800, 1600, 3200, 3265])
def test_fused_ffn_vs_torch_fp64(mdm, M, math):
    """interdiff_mdm_ffn (csrc/ffn.h: linear1 -> gelu -> linear2 in one launch, five partial slabs summed by the reader) against
    torch CPU float64 on the model's own weights: a decoder layer and an encoder layer, ragged and multi-round row counts, ALL THREE row
    tiles (the 32-row kernel, the 16-row one small batches take, the 64-row one large batches take) and the default pick between them;
    the 16- and 64-row kernels sum every tile in the same order and must agree bit for bit.
    ``math``: 'exact' = fp32 MFMA (csrc/ffn.h), 'split' = split-f16 MFMA (csrc/ffn_h2.h) -- the SAME 2e-6 gate, the measured errors of
    both go side by side into the parity log; the split kernel's three row tiles are bit-identical."""
    from interdiff_amd.mdm import ffn_parts
    g = torch.Generator().manual_seed(100 + M)
    x2 = torch.randn(M, 256, generator=g)
    sd = fx.mdm_weights()
    outs = {}
    keep_math, mdm.ffn_math = mdm.ffn_math, math
    worst = 0.0
    try:
        assert all(mdm.w.layer[l].ffn_pack_h2 != 0 for l in range(8)) and all(mdm.w.enc_layer[l].ffn_pack_h2 != 0 for l in range(8)), 'range proof must hold for the test weights'
        for rows in (32, 16, 64, 0):
            mdm.ffn_rows = rows
            for enc, layer, pre in ((False, 1, 'decoder.layers.1.'), (False, 7, 'decoder.layers.7.'), (True, 3, 'encoder.layers.3.')):
                parts = ffn_parts(mdm, x2.to(DEV), layer, encoder=enc)
                assert parts.shape == (5, M, 256)
                got = (((parts[0] + parts[1]) + parts[2]) + parts[3]) + parts[4]
                w1, b1 = sd[pre + 'linear1.weight'].double(), sd[pre + 'linear1.bias'].double()
                w2, b2 = sd[pre + 'linear2.weight'].double(), sd[pre + 'linear2.bias'].double()
                xd = x2.double()
                ref = xd + torch.nn.functional.gelu(xd @ w1.T + b1) @ w2.T + b2
                worst = max(worst, close(got, ref, 2e-6, 'fused FFN (%s) M=%d rows=%d %s' % (math, M, rows, pre)))
                outs[rows, pre] = parts
            again = ffn_parts(mdm, x2.to(DEV), 1)
            assert torch.equal(again, ffn_parts(mdm, x2.to(DEV), 1)), 'deterministic: no atomics, fixed summation order'
            assert torch.equal(again, ffn_parts(mdm, x2.to(DEV), 1, batch_rows=M)), 'batch_rows = M is the default'
        for pre in ('decoder.layers.1.', 'decoder.layers.7.', 'encoder.layers.3.'):                   # the default = the documented pick
            assert torch.equal(outs[0, pre], outs[mdm.ffn_tile_for_rows(M), pre])
            assert torch.equal(outs[16, pre], outs[64, pre]), '16- and 64-row kernels: same summation order'
            if math == 'split':
                assert torch.equal(outs[16, pre], outs[32, pre]), 'split-f16 kernel: no K split across waves, every row tile gives the same bits'
        mdm.ffn_rows = 0                                       # a chain of a larger batch takes the BATCH's tile
        assert mdm.ffn_tile_for_rows(2700) == 32 and mdm.ffn_tile_for_rows(6400) == 64
        assert torch.equal(ffn_parts(mdm, x2.to(DEV), 1, batch_rows=2700), outs[32, 'decoder.layers.1.'])
        assert torch.equal(ffn_parts(mdm, x2.to(DEV), 1, batch_rows=6400), outs[64, 'decoder.layers.1.'])
        assert torch.equal(ffn_parts(mdm, x2.to(DEV), 1, batch_rows=max(M, 600)), outs[mdm.ffn_tile_for_rows(max(M, 600)), 'decoder.layers.1.'])
        fx.record_parity('fused_ffn_vs_fp64_%s_M%d' % (math, M), worst_rel_err=worst, asserted=2e-6)
    finally:
        mdm.ffn_rows = 0
        mdm.ffn_math = keep_math


# ------------------------------------------------------------------------------------------ other BASELINE configurations
@pytest.mark.parametrize('B,T', [(32, 100), (32, 35), (1, 30)])
def test_other_configs_run_and_match_oracle_step(smpl, B, T):
    """BASELINE config #3 (B=32 with the correction predictor), the reference's default clip length (T=35) and a single
    clip: one plain step + one corrected step of the sampler against the oracle (the per-op tests cover the rest)."""
    from interdiff_amd.mdm import MDM
    from interdiff_amd.diffusion import create_gaussian_diffusion
    P = 2048 if B == 32 and T == 100 else 256
    bt = fx._clip(100 + B + T, B, T, P)
    y = fx.model_kwargs_y(bt, T)
    model = MDM(fx.mdm_weights(), device=DEV)
    corr = make_correction(smpl, T, P)
    x = bt['noise']
    ts = torch.full((B,), 250, dtype=torch.int64)
    x0 = model(x.to(DEV), ts.to(DEV), y={'cond': bt['cond'].to(DEV)})
    ref0 = oden.mdm_forward(fx.mdm_weights(), x, ts, bt['cond'])
    close(x0, ref0, 1e-4, 'denoiser B=%d T=%d' % (B, T))
    if B * T <= 200:                                          # the CPU oracle's SMPL + NN pass is slow: bound it
        m = y['inpainting_mask']
        xin = ref0 * (~m) + y['inpainted_motion'] * m
        got = corr(xin.clone().to(DEV), ts.to(DEV), {'y': dev(y)})
        ref = ocor.denoised_fn(xin.clone(), ts, {'y': dict(y, smpl=fx.smpl_model(), obj_model=fx.objproj_weights())}, past_len=fx.PAST)
        close(got, ref, 1e-4, 'correction B=%d T=%d' % (B, T))
    else:
        # benchmark-size hook call: the reference's own denoised_fn output was recorded offline (tests/golden/corr32.npz, B=32, T=100,
        # P=2048 -- its [T,B,2048,67,3] temporaries need ~6 GB and minutes of CPU); other big shapes run the sampler across a corrected step
        if (T, B, P) == fx.CORR32_SHAPE:
            z = fx.golden('corr32.npz')
            xin, y32 = fx.corr32_inputs()
            corr.debug = {}
            got = corr(xin.clone().to(DEV), torch.full((B,), fx.CORR32_T, dtype=torch.int64, device=DEV), {'y': dev(y32)})
            e = close(got, z['out'], 1e-4, 'correction B=32 T=100 P=2048 vs reference golden')
            assert np.array_equal(corr.debug['condition'].cpu().numpy().astype(bool), z['condition'])
            assert np.array_equal(corr.debug['contact'].cpu().numpy(), z['contact'])
            fx.record_parity('corrected_step_B32_T100_P2048_vs_reference', rel_err=e, asserted=1e-4, condition_flips=0, contact_flips=0)
        diff = create_gaussian_diffusion('cosine', 1000)
        out = diff.p_sample_loop(model, tuple(x.shape), noise=x.to(DEV), clip_denoised=False, model_kwargs={'y': dev(y)},
                                 denoised_fn=corr, seed=3, n_steps=501)           # crosses the first correction step (t = 500)
        assert torch.isfinite(out).all()


# ------------------------------------------------------------------------------------------ encoder side ("next" row N1)
def test_get_embeddings_golden(mdm):
    """HIP MDM._get_embeddings (PointNet++ object encoder + embeddings + 8-layer encoder) against the reference's own module
    run (tests/golden/embed.npz) and the oracle; then the conditioning drives a decoder forward."""
    from oracle import pointnet2 as opn
    z = fx.golden('embed.npz')
    ei = fx.embedding_inputs()
    d = dev(ei)
    cond, gt = mdm._get_embeddings(d['body_pose'], d['body_trans'], d['obj_angles'], d['obj_trans'], d['obj_points'], fx.PAST)
    close(gt, z['gt'], 1e-6, 'gt vs reference golden')
    close(cond, z['cond'], 1e-4, 'cond vs reference golden')
    # the object encoder alone, including a cloud with fewer than 2048 points and one whose balls are empty
    from interdiff_amd import _lib
    import ctypes as C
    for pts in (ei['obj_points'], ei['obj_points'][:, :700], 3.0 * ei['obj_points'][:2]):
        ref = opn.pointnet2_encode(fx.mdm_weights(), pts)
        got = torch.empty(pts.shape[0], 256, device=DEV)
        pd = pts.contiguous().to(DEV)
        _lib.check(mdm.lib.interdiff_pointnet2_encode(C.byref(mdm.pn), _lib.dptr(pd), pts.shape[0], pts.shape[1], _lib.dptr(got), _lib.stream()))
        close(got, ref, 1e-5, 'pointnet2 P=%d' % pts.shape[1])
    x, ts, _ = fx.mdm_inputs(3, 35)
    close(mdm(x.to(DEV), ts.to(DEV), y={'cond': cond}), oden.mdm_forward(fx.mdm_weights(), x, ts, torch.from_numpy(z['cond'])), 1e-4,
          'decoder on the HIP conditioning')


def test_long_horizon_rollout(mdm, smpl):
    """Autoregressive rollout (BASELINE config #4; upstream path is broken, semantics fixed in eval.sample_long): window 0 is
    the short-horizon sample, every appended window is consistent with the body model in the first window's frame, and the
    chain is translation-covariant from window 1 on (each window is re-centred on its own first pelvis)."""
    from interdiff_amd import eval as ev, synthetic as syn
    from interdiff_amd.mdm import MDM
    from interdiff_amd.diffusion import create_gaussian_diffusion
    T, B, P, past, steps, K = 14, 2, 128, fx.PAST, 20, 2
    ei = {k: torch.from_numpy(v).to(DEV) for k, v in syn.make_embedding_inputs(seed=5, B=B, T=T, n_points=P).items()}
    g = torch.Generator().manual_seed(2)
    raw = dict(ei, hand_pose=(0.1 * torch.randn(T, B, 90, generator=g)).to(DEV), beta=torch.randn(1, B, 10, generator=g).expand(T, B, 10).contiguous().to(DEV))
    model = MDM(fx.mdm_weights(), device=DEV, n_steps=steps)
    corr = make_correction(smpl, T, P)
    diff = create_gaussian_diffusion('cosine', steps)
    obj, body, verts, jtr, pelvis = ev.sample_long(model, diff, corr, raw, K, past, seed=11)
    F = T - past
    assert obj.shape == (T + K * F, B, 6) and body.shape == (T + K * F, B, 159) and verts.shape[0] == T + K * F
    assert torch.isfinite(verts).all()
    nz = torch.randn(B, 1, 144, T, device=DEV, generator=torch.Generator(device=DEV).manual_seed(11))
    o0, b0, v0, j0, p0 = ev.sample_once_proj(model, diff, corr, ev.batch_from_raw(model, raw, past), past, noise=nz, seed=11)
    assert torch.equal(obj[:T], o0) and torch.equal(verts[:T], v0)
    assert torch.equal(pelvis, jtr[:, :, 0])
    flat = body.reshape(-1, 159)
    v_chk, j_chk, _, _ = smpl(flat[:, :-3], th_betas=raw['beta'][:1].expand(T + K * F, B, 10).reshape(-1, 10), th_trans=flat[:, -3:])
    close(v_chk.reshape(verts.shape), verts, 1e-5, 'appended windows are SMPL(body) in the first window frame')
    # against the CPU restatement of the fixed get_batch semantics (oracle/long_horizon.py), injected x_T and per-step noise
    from oracle import long_horizon as olh

    def streams():
        rs = [np.random.RandomState(9100 + k) for k in range(K + 1)]
        xT = lambda k: torch.from_numpy(np.random.RandomState(9000 + k).standard_normal((B, 1, 144, T)).astype(np.float32))
        return xT, rs
    xT, rs = streams()
    got = ev.sample_long(model, diff, corr, raw, K, past, x_T=lambda k: xT(k).to(DEV),
                         step_noise=lambda k: (lambda i, x: torch.from_numpy(rs[k].standard_normal(tuple(x.shape)).astype(np.float32)).to(DEV)))
    xT, rs = streams()
    cpu = {k: v.cpu() for k, v in raw.items()}
    want = olh.rollout(fx.mdm_weights(), fx.smpl_model(), fx.objproj_weights(), cpu, K, past, odf.make_schedule(steps), xT,
                       lambda k: (lambda i, x: torch.from_numpy(rs[k].standard_normal(tuple(x.shape)).astype(np.float32))))
    names = ('obj', 'body', 'verts', 'jtr', 'pelvis')
    errs = {}
    for n, a, b in zip(names, got, want):
        if n in ('obj', 'body'):                         # axis-angle blocks: compare the rotations, not their representation
            nr = 3 if n == 'obj' else 66
            errs[n + '_rot'] = close(R.axis_angle_to_matrix(a[..., :nr].reshape(*a.shape[:2], -1, 3).cpu()),
                                     R.axis_angle_to_matrix(b[..., :nr].reshape(*b.shape[:2], -1, 3)), 1e-4, 'long-horizon %s rotations' % n)
            errs[n + '_rest'] = close(a[..., nr:], b[..., nr:], 1e-4, 'long-horizon %s' % n)
        else:
            errs[n] = close(a, b, 1e-4, 'long-horizon %s vs oracle/long_horizon.py' % n)
    fx.record_parity('long_horizon_K2_T14_B2_vs_oracle', asserted=1e-4, **errs)


def test_real_behave_clips_end_to_end(mdm, smpl):
    """Rows N2 -> N1 -> sampler -> correction -> metrics on REAL BEHAVE motion (three windows of the shipped sequence, stored in
    tests/golden/etl.npz): on-GPU pelvis for the ETL, canonicalised clips, HIP conditioning, a short diffusion with the
    correction hook, metrics against the clips' own ground truth."""
    from interdiff_amd import data as D, eval as ev, synthetic as syn
    from interdiff_amd.mdm import MDM
    from interdiff_amd.diffusion import create_gaussian_diffusion
    z = fx.golden('etl.npz')
    sel, starts = z['sel'], z['starts']
    F = int(sel.max()) + 1
    def full(a):
        out = np.zeros((F,) + a.shape[1:], a.dtype)
        out[sel] = a
        return out
    seq = dict(poses=full(z['poses']), betas=full(z['betas']), trans=full(z['trans']), obj_angles=full(z['obj_angles']), obj_trans=full(z['obj_trans']))
    pelvis = D.sequence_pelvis(seq, smpl, device=DEV)
    close(pelvis[sel], z['pelvis'], 1e-5, 'pelvis of the sequence (HIP SMPL) vs the oracle pelvis used for the golden')
    past, fut, P, steps = 10, 25, 256, 20
    clips = [D.canonicalize_clip(seq, pelvis, int(s0), past, fut) for s0 in starts]
    pts = syn.make_embedding_inputs(seed=3, B=1, T=2, n_points=P)['obj_points'][0]
    raw = D.collate_raw(clips, pts, device=DEV)
    model = MDM(fx.mdm_weights(), device=DEV, n_steps=steps)
    corr = make_correction(smpl, past + fut, P)
    diff = create_gaussian_diffusion('cosine', steps)
    batch = ev.batch_from_raw(model, raw, past)
    # tokens of the past frames must survive the round trip through the sampler's inpainting untouched
    obj, body, verts, jtr, pelvis_s = ev.sample_once_proj(model, diff, corr, batch, past, seed=4)
    obj_gt, jtr_gt, body_gt, faces = ev.get_gt(batch, smpl)
    close(jtr[:past], jtr_gt[:past], 1e-5, 'past frames are the ground truth')
    close(body_gt[..., 66:156], raw['hand_pose'], 0, 'hands pass through')
    m = ev.Metrics(corr)(obj[past:], jtr[past:], body[past:], obj_gt[past:], jtr_gt[past:], body_gt[past:], verts[past:], faces, raw['obj_points'])
    assert all(torch.isfinite(v).all() and v.shape == (3,) for v in m.values())
    assert (m['penetrate'] >= 0).all() and (m['penetrate'] <= 1).all()


# ---- "next" row N4: physics post-optimisation (optimization.py:19-173) ------------------------------------------------
@pytest.fixture(scope='module')
def phys(smpl):
    from interdiff_amd.optimize import PhysicsOptimizer
    return PhysicsOptimizer(smpl)


@pytest.mark.gpu
def test_optimize_loss_and_gradients_golden(phys):
    """calc_loss + the hand-written backward at the parameters the REFERENCE's Adam saw at each executed iteration
    (tests/golden/optim.npz, reference optimize() on the same clip) against the reference's own losses and gradients.
    The first iteration sits at the initial pose, where verts - verts_gt is rounding noise and the sign of it (the
    verts_reg gradient) is implementation noise: gradients are compared from the second one on."""
    from oracle import optimization as oo
    g = fx.golden('optim.npz')
    opt = phys
    inp = [a.cuda() for a in fx.optim_inputs()]
    # fp64 twin: the oracle's calc_loss + autograd in float64 at the SAME iterates -- how far the reference's own fp32 gradients are
    # from exact arithmetic (sums over 20 670 vertex coordinates, sign(verts - verts_gt) of near-zero differences) is the yardstick
    d64 = lambda v: v.double() if isinstance(v, torch.Tensor) and v.is_floating_point() else v
    model64, inp64 = {k: d64(v) for k, v in fx.smpl_model().items()}, [d64(a) for a in fx.optim_inputs()]
    rep = {}
    for k, ii in enumerate(fx.OPT_ITERS):
        params = {n: torch.from_numpy(g['before_' + n][k]).cuda() for n in oo.PARAM_ORDER}
        parts, grads = opt.loss_and_grads(params, *inp, ii)
        np.testing.assert_allclose(parts.cpu().numpy(), g['losses'][k], atol=6e-5, rtol=1e-4)       # printed with 4 decimals
        if k == 0:
            continue
        _, g64 = oo.loss_and_grads(model64, {n: torch.from_numpy(g['before_' + n][k]).double() for n in oo.PARAM_ORDER}, *inp64, ii)
        for n in oo.PARAM_ORDER:
            ref, got, exact = g['grad_' + n][k], grads[n].cpu().numpy(), g64[n].numpy()
            scale = np.abs(exact).max()
            e = dict(hip_vs_reference=float(np.abs(got - ref).max() / scale), hip_vs_fp64=float(np.abs(got - exact).max() / scale),
                     reference_vs_fp64=float(np.abs(ref - exact).max() / scale))
            rep['iter%d_%s' % (ii, n)] = e
            # gate (largest-entry relative), yard = the reference's own fp32 distance from the fp64 gradient: each fp32 implementation is
            # granted max(north_star's 1e-4, yard) around the exact gradient, so the two may be twice that apart (measured on MI355X: the
            # reference and the HIP path sit on opposite sides of the fp64 value, 0.95e-4 and 0.98e-4 away, at the entry that decides).
            # The distance to fp64 is recorded, not gated: sign(verts - verts_gt) of rounding-noise differences is a coin flip per
            # implementation -- fp64 included.
            yard = e['reference_vs_fp64']
            assert e['hip_vs_reference'] <= 2 * max(1e-4, yard), (n, ii, e)
            assert ((got == 0) == (ref == 0)).all(), (n, k)       # exact zeros (identity hand joints) stay exactly zero
    fx.record_parity('post_optimisation_gradients_vs_reference_and_fp64', **{k: v for k, v in rep.items()})


@pytest.mark.gpu
def test_optimize_culled_scans_equal_brute_force(smpl):
    """The post-optimisation's two nearest-neighbour questions (optimization.py:64-65,74-75) through the culled kernels of
    csrc/correction.hip -- nearest vertex per point by the hook's block-culled scan, "any point within 0.5 m" per vertex against the
    boxes of 64-point patches -- against the round-2 brute-force kernel (``scan_order=False``): indices and near flags bit for bit,
    on overlapping objects, an object 0.45 m away (the radius cuts through the body) and one 1.5 m away (nothing near)."""
    import ctypes as C
    from interdiff_amd import _lib, synthetic as syn
    from interdiff_amd.optimize import PhysicsOptimizer
    bt = syn.make_optim_batch(seed=3, B=4, T=5, n_points=2048)
    bt['obj_trans'][2] += np.float32(0.45)
    bt['obj_trans'][3] += np.float32(1.5)
    args = [torch.from_numpy(bt[k]).to(DEV) for k in ('pose', 'trans', 'obj_angles', 'obj_trans', 'betas', 'obj_points')]
    res = {}
    for so in (True, False):
        opt = PhysicsOptimizer(smpl, scan_order=so)
        assert bool(opt.geo.vorder) == so
        single, B, T, st, bufs = opt._init(args, 151, 4)
        assert bool(st.porder) == so
        _lib.check(opt.lib.interdiff_optimize_loss_grad(C.byref(opt.ctx), C.byref(st), _lib.stream()), 'loss_grad')
        res[so] = (bufs['yidx'].clone(), bufs['near'].clone(), bufs['loss'].clone())
    assert torch.equal(res[True][0], res[False][0]), 'nearest-vertex indices differ: %d' % (res[True][0] != res[False][0]).sum()
    assert torch.equal(res[True][1], res[False][1]), 'near flags differ: %d' % (res[True][1] != res[False][1]).sum()
    frac = res[True][1].float().reshape(4, 5, -1).mean(dim=(1, 2))
    assert frac[0] > 0.2 and 0.0 < frac[2] < frac[0] and frac[3] == 0.0, frac      # the three regimes are really exercised
    close(res[True][2], res[False][2], 1e-5, 'losses')


@pytest.mark.gpu
def test_optimize_loop_vs_oracle_and_reference(phys):
    """The Adam loop over the golden run's iteration numbers.  Adam's first update is lr*sign(g): elements whose
    gradient is rounding noise move by +-lr in an implementation-defined direction, so parameters agree to a few lr and
    the loss trajectory to a few per cent (the same spread separates the oracle from the reference)."""
    g = fx.golden('optim.npz')
    opt = phys
    inp = [a.cuda() for a in fx.optim_inputs()]
    res = opt.optimize(*inp, iters=fx.OPT_ITERS)
    losses = res['losses'].cpu().numpy()
    np.testing.assert_allclose(losses[0], g['losses'][0], atol=6e-5, rtol=1e-4)
    np.testing.assert_allclose(losses[:, 0], g['losses'][:, 0], rtol=0.03)
    K, lr = len(fx.OPT_ITERS), 1e-3
    for n in ('body', 'transl', 'glo', 'obj_transl', 'obj_rot', 'hand'):
        assert np.abs(res['params'][n].cpu().numpy() - g['param_' + n]).max() <= 2 * K * lr * 1.01, n
    assert bool(res['saved'].all())
    # returned record = the best iterate as axis-angle (optimization.py:150-172)
    assert np.abs(res['trans'].cpu().numpy() - g['trans']).max() <= 2 * K * lr * 1.01
    assert np.abs(res['obj_trans'].cpu().numpy() - g['obj_trans']).max() <= 2 * K * lr * 1.01
    assert np.isfinite(res['pose'].cpu().numpy()).all() and np.abs(res['pose'].cpu().numpy() - g['pose']).max() < 0.05


@pytest.mark.gpu
def test_optimize_batch_and_full_schedule(phys):
    """Two clips side by side == each clip alone; the full 200-iteration schedule reduces the penetration and returns an
    iterate saved after iteration 150."""
    opt = phys
    a = [x.cuda() for x in fx.optim_inputs()]
    b = [x.cuda() for x in fx.optim_inputs(seed=9001)]
    both = [torch.stack([x, y]) for x, y in zip(a, b)]
    ra = opt.optimize(*a, iters=range(151, 154))
    rb = opt.optimize(*b, iters=range(151, 154))
    rab = opt.optimize(*both, iters=range(151, 154))
    for k in ('pose', 'trans', 'obj_angles', 'obj_trans'):
        assert (rab[k][0] - ra[k]).abs().max().item() <= 1e-5 and (rab[k][1] - rb[k]).abs().max().item() <= 1e-5, k
    np.testing.assert_allclose(rab['losses'][:, 0].cpu().numpy(), ra['losses'].cpu().numpy(), rtol=1e-5)
    full = opt.optimize(*both)
    ls = full['losses'].cpu().numpy()                  # [200, 2, 4]
    assert ls.shape == (200, 2, 4) and np.isfinite(ls).all()
    w = 20.0 * np.arange(200) / 350.0                   # the penetration weight ramps up (optimization.py:70), so the total is not monotone:
    assert (ls[199, :, 1] / w[199] < 0.8 * ls[1, :, 1] / w[1]).all()      # the un-weighted penetration depth must have gone down
    assert bool(full['saved'].all())
    assert torch.isfinite(full['pose']).all() and torch.isfinite(full['obj_angles']).all()


@pytest.mark.gpu
def test_optimize_config5_named_per_gpu_batch(phys):
    """BASELINE config #5 at its named per-GPU share (optimization.py on generated HOIs, B = 128 over 8 GPUs = 16 clips per GPU) with clips of the reference's own
    length, T = 35 (10 past + 25 future; optimization.py:19-173 takes whatever T a clip has), P = 2048 object points: the sixteen clips side by side through the
    kernels -- one Adam step, then loss parts and the gradients of all six parameter groups at the stepped parameters -- against the autograd oracle
    (oracle/optimization.py, pinned to the reference's own optimize() by tests/golden/optim.npz) on two of the sixteen; and batch == alone for those two."""
    from oracle import optimization as oo
    B, T, P = 16, 35, 2048
    clips = [fx.optim_inputs(seed=9500 + i, T=T, P=P) for i in range(B)]
    batch = [torch.stack([c[k] for c in clips]).cuda() for k in range(6)]
    res = phys.optimize(*batch, iters=[151])
    assert res['losses'].shape == (1, B, 4) and torch.isfinite(res['losses']).all()
    parts, grads = phys.loss_and_grads(res['params'], *batch, 152)
    model = fx.smpl_model()
    worst = {}
    for i in (3, 12):
        alone = phys.optimize(*[a.cuda() for a in clips[i]], iters=[151])
        for k in ('pose', 'trans', 'obj_angles', 'obj_trans'):
            assert (res[k][i] - alone[k]).abs().max().item() <= 1e-5, (i, k)
        ref_parts, ref_grads = oo.loss_and_grads(model, {k: v[i].cpu() for k, v in res['params'].items()}, *clips[i], 152)
        np.testing.assert_allclose(parts[i].cpu().numpy(), ref_parts.numpy(), rtol=2e-4, atol=1e-5)
        for n in oo.PARAM_ORDER:
            ref = ref_grads[n].numpy()
            e = np.abs(grads[n][i].cpu().numpy() - ref).max() / max(np.abs(ref).max(), 1e-30)
            worst[n] = max(worst.get(n, 0.0), float(e))
            assert np.abs(grads[n][i].cpu().numpy() - ref).max() <= 5e-4 * np.abs(ref).max() + 1e-6, (i, n, e)
    fx.record_parity('post_optimisation_config5_16clips_T35_P2048_vs_autograd_oracle', clips_checked=2, **{'grad_' + k: v for k, v in worst.items()})


@pytest.mark.gpu
def test_config1_skeleton_plumbing_sampler(lib):
    """BASELINE config #1 (eval_skeleton_no_correction.py: HO-GCN skeleton tokens C = 63+36+7 = 106, B=1, T=20, a 50-step
    cosine schedule, identity denoised_fn :82-83).  The skeleton denoiser is not a kernel target;
    what config #1 exercises is the sampler with an odd channel count, so a deterministic stand-in takes the model's
    place on both sides: HIP inpaint / posterior kernels vs the oracle loop, all 50 steps, injected noise."""
    from interdiff_amd.diffusion import create_gaussian_diffusion
    B, C, T, steps = 1, 106, 20, 50
    rs = np.random.RandomState(106)
    w = fx._randn(rs, C, C) * (0.5 / np.sqrt(C))
    gt, noise = fx._randn(rs, B, 1, C, T), fx._randn(rs, B, 1, C, T)
    mask = torch.ones(B, 1, C, T, dtype=torch.bool)
    mask[..., fx.PAST:] = False
    stream_a, stream_b = fx.NoiseStream(61), fx.NoiseStream(61)

    def make_model(wd):
        def model(x, t, y=None):
            return torch.tanh(torch.einsum('dc,bgct->bgdt', wd, x)) * (1.0 + 0.01 * t.float().view(-1, 1, 1, 1) / steps)
        return model
    y = dict(inpainting_mask=mask, inpainted_motion=gt)
    ref = odf.p_sample_loop(make_model(w), (B, 1, C, T), odf.make_schedule(steps), noise.clone(), lambda i, x: stream_a.next_like(x),
                            {'y': y}, denoised_fn=lambda x, t, kw: x)
    diff = create_gaussian_diffusion('cosine', steps)
    got = diff.p_sample_loop(make_model(w.to(DEV)), (B, 1, C, T), noise=noise.to(DEV), clip_denoised=False, model_kwargs={'y': dev(y)},
                             denoised_fn=lambda x, t, kw: x, device=DEV, step_noise=lambda i, x: stream_b.next_like(x).to(DEV))
    close(got, ref, 1e-5, 'config #1 sampler chain')
    assert torch.equal(got.cpu()[..., :fx.PAST], gt[..., :fx.PAST]) or (got.cpu()[..., :fx.PAST] - ref[..., :fx.PAST]).abs().max() < 1e-6


@pytest.mark.gpu
def test_config1_skeleton_tokens_through_the_denoiser_kernels(lib):
    """BASELINE config #1 through the REAL kernels: the denoiser with the HO-GCN skeleton's token width C = 106
    (21 x 3 body keypoints | 12 x 3 object keypoints | 7-D object pose; model/diffusion_skeleton.py:7-13,236-253: ``bodyEmbedding`` reads the
    63 body channels, ``objEmbedding`` the 36 object keypoints -- the pose channels are not embedded: zero columns here --, the same
    8-layer decoder, ``bodyFinalLinear`` / ``objFinalLinear`` heads), synthetic weights, against oracle/denoiser.py: one forward at
    B = 1, T = 20 / 30 (odd M, T % 4 != 0 too) and the 50-step sampler loop of eval_skeleton_no_correction.py (identity denoised_fn :82-83,
    ``--diffusion_steps 50`` = a 50-step cosine schedule) on the eager AND the graph route (fused steps, in-kernel update over 106 channels).
    ``calc_obj_pred`` (:225-233, a rigid transform of the object's base keypoints by the predicted pose) is glue outside the
    denoiser and not restated."""
    from interdiff_amd import synthetic as syn
    from interdiff_amd.mdm import MDM
    from interdiff_amd.diffusion import create_gaussian_diffusion
    sd = {k: torch.from_numpy(v) for k, v in syn.mdm_state_dict(233).items()}
    g = torch.Generator().manual_seed(106)
    n_body, n_obj = 63, 43
    sd['bodyEmbedding.weight'] = torch.randn(256, n_body, generator=g) / n_body ** 0.5
    sd['objEmbedding.weight'] = torch.cat([torch.randn(256, 36, generator=g) / 6.0, torch.zeros(256, 7)], dim=1)
    sd['bodyFinalLinear.weight'], sd['bodyFinalLinear.bias'] = torch.randn(n_body, 256, generator=g) / 16.0, 0.1 * torch.randn(n_body, generator=g)
    sd['objFinalLinear.weight'], sd['objFinalLinear.bias'] = torch.randn(n_obj, 256, generator=g) / 16.0, 0.1 * torch.randn(n_obj, generator=g)
    steps = 50
    model = MDM(sd, device=DEV, n_steps=steps)
    assert model.w.C == 106
    for B, T in ((1, 20), (1, 30), (3, 21)):
        x, ts, cond = torch.randn(B, 1, 106, T, generator=g), torch.randint(0, steps, (B,), generator=g), torch.randn(10, B, 256, generator=g)
        got = model(x.to(DEV), ts.to(DEV), y={'cond': cond.to(DEV)})
        ref = oden.mdm_forward(sd, x, ts, cond, n_body=n_body)
        close(got, ref, 1e-4, 'C=106 denoiser forward B=%d T=%d' % (B, T))
    B, T = 1, 20
    gt, noise, cond = torch.randn(B, 1, 106, T, generator=g), torch.randn(B, 1, 106, T, generator=g), torch.randn(10, B, 256, generator=g)
    mask = torch.ones(B, 1, 106, T, dtype=torch.bool)
    mask[..., fx.PAST:] = False
    y = dict(cond=cond, inpainting_mask=mask, inpainted_motion=gt)
    sa, sb = fx.NoiseStream(62), fx.NoiseStream(62)
    ref = odf.p_sample_loop(lambda x, t, y: oden.mdm_forward(sd, x, t, y['cond'], n_body=n_body), (B, 1, 106, T), odf.make_schedule(steps), noise.clone(),
                            lambda i, x: sa.next_like(x), {'y': y}, denoised_fn=lambda x, t, kw: x)
    diff = create_gaussian_diffusion('cosine', steps)
    got = diff.p_sample_loop(model, (B, 1, 106, T), noise=noise.to(DEV), clip_denoised=False, model_kwargs={'y': dev(y)},
                             denoised_fn=lambda x, t, kw: x, step_noise=lambda i, x: sb.next_like(x).to(DEV))
    e = close(got, ref, 1e-4, 'config #1: 50-step chain through the C=106 denoiser kernels vs oracle')
    # graph route (fused plain steps with the update inside the last GEMM, in-kernel noise) == eager route fed the same Philox stream
    timed = diff.p_sample_loop(model, (B, 1, 106, T), noise=noise.to(DEV), clip_denoised=False, model_kwargs={'y': dev(y)}, seed=17)
    eager = diff.p_sample_loop(model, (B, 1, 106, T), noise=noise.to(DEV), clip_denoised=False, model_kwargs={'y': dev(y)}, use_graph=False,
                               step_noise=_philox_step(lib, 17))
    assert torch.equal(timed, eager), 'C=106 graph route differs from eager: %g' % (timed - eager).abs().max()
    fx.record_parity('config1_skeleton_C106_B1_T20_50steps_vs_oracle', worst_rel_err=e, asserted=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize('B,T', [(1, 11), (2, 16), (3, 17), (1, 49), (5, 64), (2, 208), (64, 20)])
def test_denoiser_edge_sizes(mdm, B, T):
    """Ragged tiles everywhere: T below / at / just above one 16-token row block, M = B*T not a multiple of any GEMM tile,
    the longest supported clip (ATTN_MAX_T = 208) and BASELINE config #4's batch on one GPU."""
    x, ts, cond = fx.mdm_inputs(B, T)
    got = mdm(x.to(DEV), ts.to(DEV), y={'cond': cond.to(DEV)})
    ref = oden.mdm_forward(fx.mdm_weights(), x, ts, cond)
    close(got, ref, 1e-4, 'denoiser B=%d T=%d' % (B, T))


@pytest.mark.gpu
@pytest.mark.parametrize('M,B,T', [(15, 3, 35), (10, 2, 300), (16, 2, 20), (1, 2, 17), (7, 1, 224)])
def test_denoiser_any_memory_length_and_long_clips(lib, M, B, T):
    """The two hard limits of rounds 1-4, lifted behind slower-but-correct paths (round 5):
      * memory length != 10 -- the reference takes --past_len from the CLI (eval_smpl_short.py:376-377; cond = the encoded past frames, model/diffusion_smpl.py:195-223);
        lengths 1..16 take the generic layout of the folded memory (one 16-column score tile per head, absent slots masked; csrc/denoiser.hip MemLay);
      * clips longer than 208 frames -- the reference's bound is PositionalEncoding(max_len=5000) (model/layers.py:10); they take the K/V-tiled self-attention.
    Against oracle/denoiser.py (1e-4) and its float64 run (the split-f16 form as close to fp64 as the exact form, like test_mdm_forward_split_f16_vs_exact_and_fp64),
    both arithmetics; then back to the default shape on the same handle (the compact layout must be undisturbed)."""
    from interdiff_amd.mdm import MDM
    sd = fx.mdm_weights()
    sd64 = {k: torch.as_tensor(v).double() for k, v in sd.items()}
    rs = np.random.RandomState(7000 + 100 * M + T)
    x = torch.from_numpy(rs.standard_normal((B, 1, 144, T)).astype(np.float32))
    ts = torch.from_numpy(rs.randint(0, 1000, size=B).astype(np.int64))
    cond = torch.from_numpy(rs.standard_normal((M, B, 256)).astype(np.float32))
    ref = oden.mdm_forward(sd, x, ts, cond)
    ref64 = oden.mdm_forward(sd64, x.double(), ts, cond.double())
    err = {}
    for math in ('exact', 'split'):
        m = MDM(sd, device=DEV)
        m.ffn_math = math
        got = m(x.to(DEV), ts.to(DEV), y={'cond': cond.to(DEV)})
        assert m.mem_len == M
        close(got, ref, 1e-4, 'M=%d B=%d T=%d %s vs oracle' % (M, B, T, math))
        err[math] = rel(got, ref64)
        # the same handle, back at the default memory length and a short clip: the compact layout / one-shot attention again, bit-identical to a fresh model
        x2, ts2, c2 = fx.mdm_inputs(2, 12)
        again = m(x2.to(DEV), ts2.to(DEV), y={'cond': c2.to(DEV)})
        assert m.mem_len == 10
        fresh = MDM(sd, device=DEV)
        fresh.ffn_math = math
        assert torch.equal(again, fresh(x2.to(DEV), ts2.to(DEV), y={'cond': c2.to(DEV)}))
    fx.record_parity('denoiser_M%d_B%d_T%d_vs_fp64' % (M, B, T), exact=err['exact'], split=err['split'])
    assert err['split'] <= max(2 * err['exact'], 2e-6) and err['exact'] <= 1e-5, err


@pytest.mark.gpu
def test_memory_length_travels_with_the_folded_buffer(lib):
    """the memory length used to be mutable state of the shared weights handle, set by whichever ``prepare_memory`` came last.  The sequence
    forward(cond of length 10) -> prepare_memory(cond of length 15, into=caller's buffer) -> forward(the SAME length-10 cond: cache hit, no re-fold) then read the
    compact length-10 buffer with the generic length-15 offsets.  The length now travels with the buffer (MDM._bind_memory): the third call must reproduce the first
    bit for bit, the caller's buffer must still serve a length-15 forward, and a refused ``into`` must leave the handle alone."""
    from interdiff_amd.mdm import MDM
    sd = fx.mdm_weights()
    m = MDM(sd, device=DEV)
    B, T = 3, 35
    rs = np.random.RandomState(88)
    x = torch.from_numpy(rs.standard_normal((B, 1, 144, T)).astype(np.float32)).to(DEV)
    ts = torch.from_numpy(rs.randint(0, 1000, size=B).astype(np.int64)).to(DEV)
    c10 = torch.from_numpy(rs.standard_normal((10, B, 256)).astype(np.float32)).to(DEV)
    c15 = torch.from_numpy(rs.standard_normal((15, B, 256)).astype(np.float32)).to(DEV)
    first = m(x, ts, y={'cond': c10}).clone()
    buf = torch.empty(m.memctx_floats(B, 15), dtype=torch.float32, device=DEV)
    m.prepare_memory(c15, into=buf)
    again = m(x, ts, y={'cond': c10})
    assert torch.equal(first, again)
    assert m.mem_len == 10
    got15 = m(x, ts, memctx=buf)
    close(got15, oden.mdm_forward(sd, x.cpu(), ts.cpu(), c15.cpu()), 1e-4, 'caller-owned length-15 memory')
    assert m.mem_len == 15
    assert torch.equal(first, m(x, ts, y={'cond': c10}))
    with pytest.raises(ValueError):
        m.prepare_memory(c15, into=torch.empty(7, dtype=torch.float32, device=DEV))
    assert m.mem_len == 10 and torch.equal(first, m(x, ts, y={'cond': c10}))


@pytest.mark.gpu
def test_sampler_with_a_longer_memory_and_a_long_clip(lib, smpl):
    """The lifted limits through the SAMPLER: 30 plain steps + inpainting at memory length 15 (T = 35, the reference's clip length with --past_len 15) on the graph
    route against the CPU oracle's p_sample_loop fed the materialised Philox stream (1e-4), graph route == eager route bit for bit; the same at T = 240 (K/V-tiled
    attention), memory length 10."""
    from interdiff_amd.mdm import MDM
    from interdiff_amd.diffusion import create_gaussian_diffusion
    diff = create_gaussian_diffusion('cosine', 1000)
    for M, B, T, n in ((15, 3, 35, 30), (10, 2, 240, 12)):
        model = MDM(fx.mdm_weights(), device=DEV)
        rs = np.random.RandomState(7100 + M + T)
        gt = torch.from_numpy(rs.standard_normal((B, 1, 144, T)).astype(np.float32))
        cond = torch.from_numpy(rs.standard_normal((M, B, 256)).astype(np.float32))
        noise = torch.from_numpy(rs.standard_normal((B, 1, 144, T)).astype(np.float32))
        mask = torch.zeros_like(gt, dtype=torch.bool)
        mask[..., :M] = True
        y = dict(cond=cond, inpainted_motion=gt, inpainting_mask=mask)
        kw = dict(noise=noise.to(DEV), clip_denoised=False, model_kwargs={'y': dev(y)}, n_steps=n, first_t=900)
        timed = diff.p_sample_loop(model, tuple(noise.shape), seed=23, **kw)
        eager = diff.p_sample_loop(model, tuple(noise.shape), use_graph=False, step_noise=_philox_step(lib, 23), **kw)
        assert torch.equal(timed, eager), 'M=%d T=%d: graph route differs from eager: %g' % (M, T, (timed - eager).abs().max())
        stream = _philox_step(lib, 23)
        ref = odf.p_sample_loop(lambda x, t, y: oden.mdm_forward(fx.mdm_weights(), x, t, y["cond"]), tuple(noise.shape), odf.make_schedule(1000), noise.clone(),
                                lambda i, x: stream(i, x.to(DEV)).cpu(), {'y': y}, n_steps=n, first_t=900)
        e = close(timed, ref, 1e-4, 'sampler M=%d T=%d vs oracle' % (M, T))
        fx.record_parity('sampler_M%d_T%d_%dsteps_vs_oracle' % (M, T, n), worst_rel_err=e, asserted=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize('B,T', [(3, 20), (2, 13), (3, 35), (5, 7)])
def test_forward_step_matches_two_call_form(mdm, B, T):
    """interdiff_mdm_forward_step (denoiser + inpaint + posterior + in-kernel noise + state advance in the denoiser's own launches)
    against interdiff_mdm_forward followed by interdiff_posterior_step_dev on copies of the same state, two consecutive steps, bit
    for bit.  T % 4 == 0 takes the 16-byte form of the update; T = 13 / 35 (the reference's default clip length) / 7 the per-row
    form, where a lane's four frames may sit in two clips and draw from three Philox groups (csrc/gemm.h post_prefetch)."""
    from interdiff_amd import _lib
    from interdiff_amd.diffusion import create_gaussian_diffusion
    lib = _lib.load()
    x, ts0, cond = fx.mdm_inputs(B, T)
    g = torch.Generator().manual_seed(9)
    gt, mask = torch.randn(x.shape, generator=g).to(DEV), (torch.rand(x.shape, generator=g) < 0.2).to(DEV).view(torch.uint8)
    table = create_gaussian_diffusion('cosine', 1000)._table(torch.device(DEV))
    y = {'cond': cond.to(DEV)}
    assert mdm.supports_forward_step

    def fresh(elem0=0):
        st = torch.tensor([700, 11, 1234567, 0, 0, 0, elem0, 0], dtype=torch.int64, device=DEV)
        return x.to(DEV).clone(), torch.full((B,), 700, dtype=torch.int64, device=DEV), st
    for elem0 in (0, 4 * 1237):                          # a chain of a split batch: its x starts elem0 elements into the sample's noise
        xa, tsa, sta = fresh(elem0)
        xb, tsb, stb = fresh(elem0)
        x0 = torch.empty_like(xb)
        for step in range(2):
            mdm.forward_step(xa, tsa, table, sta, gt=gt, mask=mask, y=y)
            mdm(xb, tsb, y=y, out=x0)
            _lib.check(lib.interdiff_posterior_step_dev(_lib.dptr(xb), _lib.dptr(x0), _lib.dptr(gt), _lib.dptr(mask), xb.numel(), _lib.dptr(table),
                                                        _lib.dptr(stb), _lib.dptr(tsb), B, _lib.stream()), 'posterior_step_dev')
            assert torch.equal(xa, xb), 'step %d: %g' % (step, (xa - xb).abs().max())
            assert torch.equal(tsa, tsb) and torch.equal(sta[:3], stb[:3]) and int(sta[0]) == 699 - step and int(sta[1]) == 12 + step
    # without inpainting operands (gt = mask = NULL)
    xa, tsa, sta = fresh()
    xb, tsb, stb = fresh()
    mdm.forward_step(xa, tsa, table, sta, y=y)
    mdm(xb, tsb, y=y, out=x0)
    _lib.check(lib.interdiff_posterior_step_dev(_lib.dptr(xb), _lib.dptr(x0), None, None, xb.numel(), _lib.dptr(table), _lib.dptr(stb), _lib.dptr(tsb), B,
                                                _lib.stream()), 'posterior_step_dev')
    assert torch.equal(xa, xb) and torch.equal(tsa, tsb)


@pytest.mark.gpu
@pytest.mark.parametrize('B,T', [(16, 100), (3, 20), (3, 35), (2, 13)])
def test_chained_plain_steps_equal_unchained(mdm, B, T):
    """interdiff_mdm_forward_step_ex: four consecutive plain steps with every step's last launch also computing the next step's embedding
    (csrc/tail_h2.h, MODE 3; the next call starts at its QKV projection) against the same four steps each run on its own -- x, the timesteps and the sampler
    state bit for bit, at the bench shape (rows straddle clips: 16-row tiles at T = 100), a ragged last tile and the per-row update form (T % 4 != 0)."""
    from interdiff_amd.diffusion import create_gaussian_diffusion
    x, ts0, cond = fx.mdm_inputs(B, T)
    g = torch.Generator().manual_seed(19)
    gt, mask = torch.randn(x.shape, generator=g).to(DEV), (torch.rand(x.shape, generator=g) < 0.2).to(DEV).view(torch.uint8)
    table = create_gaussian_diffusion('cosine', 1000)._table(torch.device(DEV))
    y = {'cond': cond.to(DEV)}
    keep = mdm.ffn_math, getattr(mdm, 'rowblock_math', 'split')
    try:
        for math in ('split', 'exact'):
            mdm.ffn_math = math
            assert mdm.step_chaining == (math == 'split')          # exact arithmetic: the flags are ignored (and must still give the same bits)

            def run(chained):
                xa = x.to(DEV).clone()
                tsa = torch.full((B,), 640, dtype=torch.int64, device=DEV)
                sta = torch.tensor([640, 5, 424242, 0, 0, 0, 4 * 311, 0], dtype=torch.int64, device=DEV)
                for i in range(4):
                    mdm.forward_step(xa, tsa, table, sta, gt=gt, mask=mask, y=y, embed_ready=chained and i > 0, embed_next=chained and i < 3)
                return xa, tsa, sta
            (xa, tsa, sta), (xb, tsb, stb) = run(False), run(True)
            assert torch.equal(xa, xb) and torch.equal(tsa, tsb) and torch.equal(sta, stb), (math, float((xa - xb).abs().max()))
            assert int(tsa[0]) == 636 and torch.isfinite(xa).all()
    finally:
        mdm.ffn_math, mdm.rowblock_math = keep


@pytest.mark.gpu
@pytest.mark.parametrize('T,B,P', [(11, 1, 1), (12, 2, 63), (13, 1, 1000), (11, 2, 2048)])
def test_correction_edge_sizes(smpl, T, B, P):
    """Correction hook with ragged point counts (a single point, below one wave, not a multiple of the workgroup, the
    maximum 2048) and a single future frame: output and the discrete decisions against the oracle."""
    corr = make_correction(smpl, T, P)
    corr.debug = {}
    bt = fx._clip(700 + P, B, T, P)
    y = fx.model_kwargs_y(bt, T)
    rs = np.random.RandomState(P)
    x = bt['gt'] + 0.05 * fx._randn(rs, *bt['gt'].shape)
    t = torch.full((B,), 250, dtype=torch.int64)
    got = corr(x.clone().to(DEV), t.to(DEV), {'y': dev(y)})
    ref = ocor.denoised_fn(x.clone(), t, {'y': dict(y, smpl=fx.smpl_model(), obj_model=fx.objproj_weights())}, past_len=fx.PAST)
    close(got, ref, 1e-4, 'correction T=%d B=%d P=%d' % (T, B, P))
    terms = ocor.correction_terms(x.clone(), dict(y, smpl=fx.smpl_model()), fx.PAST)
    assert torch.equal(corr.debug['condition'].cpu().bool(), terms['condition'])
    assert torch.equal(corr.debug['contact'].cpu().long(), terms['contact'])


@pytest.mark.gpu
def test_hook_with_the_predictor_inside_the_scan_launch_equals_the_serial_one(smpl):
    """Round 6: the contact-frame predictor's three stacks ride in the contact scan's launch as leading workgroups (they need the markers only) and a small kernel picks the node
    once the labels exist (csrc/correction.hip correction_impl, corr_contact_kernel<false, true>, csrc/objproj.h).  Same bits as the scan followed by the one-launch predictor
    (ctx.tune = 2), outputs and decisions, called eagerly, on a non-default stream, and replayed from a captured graph; repeated calls do not disturb each other."""
    T, B, P = 14, 3, 300
    bt = fx._clip(4242, B, T, P)
    y = dev(fx.model_kwargs_y(bt, T))
    rs = np.random.RandomState(7)
    xs = [(bt['gt'] + 0.05 * fx._randn(rs, *bt['gt'].shape)).to(DEV) for _ in range(3)]

    def run(tune, x):
        corr = make_correction(smpl, T, P)
        corr.ctx.tune = tune
        corr.debug = {}
        out = corr.apply(x.clone(), 250, y)
        return out, corr.debug['condition'].clone(), corr.debug['contact'].clone(), corr.debug['distance'].clone(), corr.debug['loss'].clone()

    for x in xs:
        a, b = run(0, x), run(2, x)
        for u, v in zip(a, b):
            assert torch.equal(u, v)
    # the device-scalar form (what the sampler's graphs capture): eager on a side stream, then captured and replayed twice
    table = torch.zeros(1000, 4, device=DEV)
    table[:, 3] = torch.linspace(0.1, 0.9, 1000, device=DEV)
    state = torch.zeros(8, dtype=torch.int64, device=DEV)
    state[0] = 250
    outs = {}
    for tune in (0, 2):
        corr = make_correction(smpl, T, P)
        corr.ctx.tune = tune
        ws = corr.workspace_for(B, T) if hasattr(corr, 'workspace_for') else None
        st = torch.cuda.Stream()
        res = []
        with torch.cuda.stream(st):
            ws = corr._workspace(B, T) if ws is None else ws
            x = xs[0].clone()
            corr.apply_dev(x, table, state, y, ws)           # eager (creates the side stream outside any capture)
            res.append(x.clone())
        st.synchronize()
        xg = xs[1].clone()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(st):
            with torch.cuda.graph(g, stream=st):
                corr.apply_dev(xg, table, state, y, ws)
        for k in (1, 2):
            xg.copy_(xs[k])
            g.replay()
            torch.cuda.synchronize()
            res.append(xg.clone())
        outs[tune] = res
    for u, v in zip(outs[0], outs[2]):
        assert torch.equal(u, v)
    assert not torch.equal(outs[0][1], outs[0][2])


@pytest.mark.gpu
@pytest.mark.parametrize('T,P', [(3, 5), (7, 300)])
def test_optimize_edge_sizes_vs_oracle(phys, T, P):
    """Shortest clip the smoothness terms are defined for (T = 3) and ragged point counts: loss parts and gradients of one
    evaluation away from the initial pose (two Adam steps of the oracle first) against the autograd oracle."""
    from oracle import optimization as oo
    inp = fx.optim_inputs(seed=9100 + T, T=T, P=P)
    model = fx.smpl_model()
    warm = oo.optimize(model, *inp, iters=[151, 152])
    parts, grads = oo.loss_and_grads(model, warm['params'], *inp, 153)
    got_parts, got = phys.loss_and_grads({k: v.cuda() for k, v in warm['params'].items()}, *[a.cuda() for a in inp], 153)
    np.testing.assert_allclose(got_parts.cpu().numpy(), parts.numpy(), rtol=2e-4, atol=1e-5)
    for n in oo.PARAM_ORDER:
        ref = grads[n].numpy()
        assert np.abs(got[n].cpu().numpy() - ref).max() <= 5e-4 * np.abs(ref).max() + 1e-6, n
    with pytest.raises((RuntimeError, ValueError)):
        phys.optimize(*[a[:2].cuda() if a.shape[0] == T else a.cuda() for a in inp])      # T = 2: smoothness undefined


@pytest.mark.gpu
def test_optimize_real_behave_motion(phys, smpl):
    """Row N4 on REAL BEHAVE motion (windows of the shipped sequence stored in tests/golden/etl.npz; root rotations near pi,
    real hand poses): three canonicalised 20-frame clips side by side through the full 200-iteration schedule.  One
    loss/gradient evaluation is checked against the autograd oracle on the first clip; the run must stay finite, save an
    iterate after iteration 150 and not move the poses further than 200 Adam steps can."""
    from interdiff_amd import data as D, synthetic as syn
    from oracle import optimization as oo
    z = fx.golden('etl.npz')
    sel, starts = z['sel'], z['starts']
    F = int(sel.max()) + 1

    def full(a):
        out = np.zeros((F,) + a.shape[1:], a.dtype)
        out[sel] = a
        return out
    seq = dict(poses=full(z['poses']), betas=full(z['betas']), trans=full(z['trans']), obj_angles=full(z['obj_angles']), obj_trans=full(z['obj_trans']))
    pelvis = D.sequence_pelvis(seq, smpl, device=DEV)
    T, P = 20, 512
    clips = [D.canonicalize_clip(seq, pelvis, int(s0), 10, 10) for s0 in starts]
    pts = syn.make_embedding_inputs(seed=3, B=1, T=2, n_points=P)['obj_points'][0]
    st = lambda k: torch.from_numpy(np.stack([np.asarray(c[k], dtype=np.float32) for c in clips]))
    args = [st('pose'), st('trans'), st('obj_angles'), st('obj_trans'), st('betas'), torch.from_numpy(np.repeat(pts[None], len(clips), 0))]
    assert args[0].shape == (3, T, 156)
    # one evaluation vs the oracle (clip 0, away from the initial pose)
    one = [a[0] for a in args]
    model = fx.smpl_model()
    warm = oo.optimize(model, *one, iters=[151])
    parts, grads = oo.loss_and_grads(model, warm['params'], *one, 152)
    got_parts, got = phys.loss_and_grads({k: v.cuda() for k, v in warm['params'].items()}, *[a.cuda() for a in one], 152)
    np.testing.assert_allclose(got_parts.cpu().numpy(), parts.numpy(), rtol=2e-4, atol=1e-5)
    for n in oo.PARAM_ORDER:
        ref = grads[n].numpy()
        assert np.abs(got[n].cpu().numpy() - ref).max() <= 5e-4 * np.abs(ref).max() + 1e-6, n
    res = phys.optimize(*[a.cuda() for a in args])
    assert bool(res['saved'].all()) and torch.isfinite(res['losses']).all()
    for k, a in (('pose', args[0]), ('trans', args[1]), ('obj_angles', args[2]), ('obj_trans', args[3])):
        assert torch.isfinite(res[k]).all()
    # 200 steps of at most ~lr each (Adam's m/sqrt(v) exceeds 1 only mildly and briefly)
    assert (res['trans'].cpu() - args[1]).abs().max() <= 0.3 and (res['obj_trans'].cpu() - args[3]).abs().max() <= 0.3


@pytest.mark.gpu
def test_optimize_writes_stay_inside_their_buffers(smpl):
    """Every device buffer of the optimiser state is re-homed between two guard bands filled with a sentinel; after a
    few iterations (and a loss/gradient evaluation) no band may have been touched."""
    from interdiff_amd import _lib
    from interdiff_amd.optimize import PhysicsOptimizer
    GUARD = 4096

    class Guarded(PhysicsOptimizer):
        def _alloc(self, B, T, P, max_iters):
            st, bufs = super()._alloc(B, T, P, max_iters)
            if getattr(self, '_guards', None) is not None and self._guards[0] == (B, T, P, max_iters):
                return st, bufs
            guards = {}
            for k in _lib._OPT_PTRS:
                t = bufs[k]
                nbytes = t.numel() * t.element_size()
                raw = torch.full((nbytes + 2 * GUARD,), 0xA5, dtype=torch.uint8, device=t.device)
                inner = raw[GUARD:GUARD + nbytes].view(t.dtype).view(t.shape)
                inner.copy_(t)
                bufs[k] = inner
                guards[k] = raw
                setattr(st, k, inner.data_ptr())
            self._guards = ((B, T, P, max_iters), guards)
            return st, bufs

    opt = Guarded(smpl)
    inp = [a.cuda() for a in fx.optim_inputs(seed=9200, T=7, P=130)]
    res = opt.optimize(*inp, iters=range(150, 154))
    assert torch.isfinite(res['losses']).all()
    for k, raw in opt._guards[1].items():
        assert bool((raw[:GUARD] == 0xA5).all()) and bool((raw[-GUARD:] == 0xA5).all()), 'out-of-bounds write next to %s' % k


def _guarded(nbytes, dev, guard=4096):
    raw = torch.full((nbytes + 2 * guard,), 0xA5, dtype=torch.uint8, device=dev)
    return raw, raw[guard:guard + nbytes]


def _intact(raw, guard=4096):
    return bool((raw[:guard] == 0xA5).all()) and bool((raw[-guard:] == 0xA5).all())


@pytest.mark.gpu
def test_hot_path_writes_stay_inside_their_buffers(mdm, smpl):
    """Guard bands around the denoiser workspace / memory context / output, the SMPL outputs and workspace and the correction
    workspace at a ragged size: the exact byte counts the library asks for must be enough, and nothing may write past them."""
    import ctypes as C
    from interdiff_amd import _lib
    B, T, P = 3, 37, 300
    x, ts, cond = fx.mdm_inputs(B, T)
    # denoiser: exact-size workspace, memctx and output between guards
    need_ws = mdm.lib.interdiff_mdm_workspace_bytes(B, T)
    raw_ws, ws = _guarded(need_ws, DEV)
    raw_mc, mc = _guarded(mdm.lib.interdiff_mdm_memctx_floats(B) * 4, DEV)
    raw_o, o = _guarded(B * 144 * T * 4, DEV)
    saved = (mdm._ws, mdm._ws_shape, mdm._mem_key, mdm._ws_pool.get((B, T)), mdm._ws_pool.get((B, 16)), mdm._memctx_pool.get(B))
    try:
        mdm._ws, mdm._ws_shape, mdm._mem_key = None, None, None
        mdm._ws_pool[(B, T)], mdm._memctx_pool[B] = ws, mc.view(torch.float32)
        mdm._ws_pool.pop((B, 16), None)                         # the memory fold takes its own (B, 16) workspace: let it allocate
        out = o.view(torch.float32).view(B, 1, 144, T)
        got = mdm(x.to(DEV), ts.to(DEV), y={'cond': cond.to(DEV)}, out=out)
        close(got, oden.mdm_forward(fx.mdm_weights(), x, ts, cond), 1e-4, 'guarded denoiser forward')
        assert mdm._memctx.data_ptr() == mc.data_ptr() and mdm._ws.data_ptr() == ws.data_ptr()
    finally:
        mdm._ws, mdm._ws_shape, mdm._mem_key = saved[0], saved[1], None
        for k, v in (((B, T), saved[3]), ((B, 16), saved[4])):
            if v is None:
                mdm._ws_pool.pop(k, None)
            else:
                mdm._ws_pool[k] = v
        if saved[5] is None:
            mdm._memctx_pool.pop(B, None)
        else:
            mdm._memctx_pool[B] = saved[5]
    assert _intact(raw_ws) and _intact(raw_mc) and _intact(raw_o)
    # body model: exact-size outputs and workspace
    N = 5
    pose, betas, trans = fx.smpl_inputs(N)
    V, J = smpl.cmodel.V, smpl.cmodel.J
    raws = [_guarded(n, DEV) for n in (N * V * 12, N * J * 12, N * V * 12, smpl.lib.interdiff_smpl_workspace_bytes(C.byref(smpl.cmodel), N))]
    verts, jtr, vp = (r[1].view(torch.float32) for r in raws[:3])
    pd, bd, td = pose.to(DEV), betas.to(DEV), trans.to(DEV)          # keep the device copies alive across the launch
    _lib.check(smpl.lib.interdiff_smpl_forward(C.byref(smpl.cmodel), _lib.dptr(pd), _lib.dptr(bd), _lib.dptr(td), N,
                                               _lib.dptr(verts), _lib.dptr(jtr), _lib.dptr(vp), _lib.dptr(raws[3][1]), raws[3][1].numel(),
                                               _lib.stream()), 'smpl_forward')
    ref = osmpl.smpl_forward(fx.smpl_model(), pose, betas, trans)
    close(verts.view(N, V, 3), ref[0], 1e-5, 'guarded smpl verts')
    assert all(_intact(r[0]) for r in raws)
    # correction hook: exact-size workspace
    corr = make_correction(smpl, T, P)
    raw_c, wsc = _guarded(corr.lib.interdiff_correction_workspace_bytes(C.byref(corr.ctx), B, T), DEV)
    corr._ws = {torch.cuda.current_stream().cuda_stream: wsc}          # the hook keeps one workspace per stream it is called on
    bt = fx._clip(800, B, T, P)
    y = fx.model_kwargs_y(bt, T)
    xin = bt['gt'].clone().to(DEV)
    raw_x, xg = _guarded(xin.numel() * 4, DEV)
    xg = xg.view(torch.float32).view(xin.shape)
    xg.copy_(xin)
    corr(xg, torch.full((B,), 250, dtype=torch.int64, device=DEV), {'y': dev(y)})
    assert torch.isfinite(xg).all() and _intact(raw_c) and _intact(raw_x)


@pytest.mark.gpu
def test_memory_fold_is_not_reused_for_a_recycled_cond_address(mdm):
    """A new ``cond`` of the same shape that happens to get the address of a freed one (its version counter is 0 as well: the
    encoder writes it through a raw pointer) must not be mistaken for the memory that was folded last."""
    B, T = 2, 12
    x, ts, cond = fx.mdm_inputs(B, T)
    xd, tsd = x.to(DEV), ts.to(DEV)
    c1 = cond.to(DEV)
    a = mdm(xd, tsd, y={'cond': c1}).clone()
    addr = c1.data_ptr()
    del c1
    from interdiff_amd import _lib
    c2 = torch.empty(cond.shape, dtype=torch.float32, device=DEV)            # same size class: the allocator may reuse the block
    _lib.check(mdm.lib.interdiff_randn(_lib.dptr(c2), c2.numel(), 77, 0, _lib.stream()), 'randn')   # raw-pointer write: version stays 0
    assert c2._version == 0
    b = mdm(xd, tsd, y={'cond': c2})
    ref = oden.mdm_forward(fx.mdm_weights(), x, ts, c2.cpu())
    close(b, ref, 1e-4, 'second memory (address %s)' % ('recycled' if c2.data_ptr() == addr else 'fresh'))
    assert rel(a, b) > 1e-3


@pytest.mark.gpu
def test_graph_cache_serves_new_samples_without_recapture(mdm, smpl):
    """Two samples of the same shape with DIFFERENT cond / gt / mask tensors: the second one replays the first one's captured
    graphs (no new capture) and still equals its own eager run bit for bit."""
    from interdiff_amd.diffusion import create_gaussian_diffusion
    T, B, P = 12, 2, 64
    diff = create_gaussian_diffusion('cosine', 1000)
    corr = make_correction(smpl, T, P)
    outs = []
    for seed in (31, 32):
        bt = fx._clip(seed, B, T, P)
        y = dev(fx.model_kwargs_y(bt, T))
        noise = bt['noise'].to(DEV)
        g = diff.p_sample_loop(mdm, tuple(noise.shape), noise=noise, clip_denoised=False, model_kwargs={'y': y}, denoised_fn=corr, seed=5,
                               n_steps=120, first_t=560)
        e = diff.p_sample_loop(mdm, tuple(noise.shape), noise=noise, clip_denoised=False, model_kwargs={'y': y}, denoised_fn=corr, seed=5,
                               n_steps=120, first_t=560, use_graph=False)
        assert torch.equal(g, e)
        outs.append(g)
        mine = {k: len(v.graphs) for k, v in mdm._graph_cache.items() if k[0] == diff._uid}
        if seed == 31:
            n_graphs = mine
    assert len(mine) == 1 and mine == n_graphs
    assert not torch.equal(outs[0], outs[1])


@pytest.mark.gpu
def test_two_chain_plain_steps_equal_single_chain_and_eager(mdm, smpl):
    """Even batches of 4..16 clips step their plain steps as two half-batch chains on two branches of one captured graph (own x / ts
    slices, state, folded memory, workspace; noise drawn at the whole batch's counters through state[6]); hook steps run on the whole
    batch in between.  Bit-identical to the single chain and to the eager loop, with and without mask / hook, and on graph reuse."""
    from interdiff_amd.diffusion import create_gaussian_diffusion
    diff = create_gaussian_diffusion('cosine', 1000)
    diff.split_min_rows = 0                  # by default batches of <= 800 token rows stay one chain: take the split at test size
    P = 64
    for B, seed, T in ((4, 51, 12), (6, 52, 12), (4, 53, 13), (5, 54, 12)):         # T = 13: the per-row form of the fused update, chain 1 at an odd element offset; B = 5: chains of 3 + 2 clips
        corr = make_correction(smpl, T, P)
        bt = fx._clip(seed, B, T, P)
        y = dev(fx.model_kwargs_y(bt, T))
        noise = bt['noise'].to(DEV)
        for hook in (corr, None):
            for yy in (y, {k: v for k, v in y.items() if k not in ('inpainting_mask',)}):
                run = lambda **kw: diff.p_sample_loop(mdm, tuple(noise.shape), noise=noise, clip_denoised=False, model_kwargs={'y': yy},
                                                      denoised_fn=hook, seed=5, n_steps=120, first_t=560, **kw)
                assert diff.split_chains
                two = run()
                st = [v for k, v in mdm._graph_cache.items() if k[0] == diff._uid and k[1] == tuple(noise.shape)]
                joined = lambda v: any((key[4] if key[0] == 'hook' else key[2]) for key in v.graphs if isinstance(key, tuple))                 # chains forked / joined inside every graph block
                apart = lambda v: all(getattr(ch, 'graphs', None) for ch in v.chains)                         # chains on their own streams, own graphs
                assert any(hasattr(v, 'chains') and (joined(v) or apart(v)) for v in st), 'split route not taken'
                assert torch.equal(two, run()), 'graph reuse'
                if hook is not None and B % 2 == 0:  # the staggered per-chain loop (equal chains only) (option: hook called per half batch, chains on their own streams) against the default
                    diff.stagger_steps = 7
                    staggered = run()
                    diff.stagger_steps = 0
                    assert any(hasattr(v, 'chains') and apart(v) for v in st), 'staggered route not taken'
                    assert torch.equal(two, staggered), 'staggered chains differ from joined chains: %g' % (two - staggered).abs().max()
                diff.split_chains = False
                one = run()
                diff.split_chains = True
                assert torch.equal(two, one), 'two chains differ from one: %g' % (two - one).abs().max()
                assert torch.equal(two, run(use_graph=False)), 'two chains differ from the eager loop'


@pytest.mark.gpu
def test_captured_graphs_survive_calls_with_other_shapes(mdm, smpl):
    """A sample is captured at one batch size, a bigger batch then runs through the same denoiser (new workspace and memory
    context), and the first shape's captured graphs are replayed: their baked-in buffers must still be theirs."""
    from interdiff_amd.diffusion import create_gaussian_diffusion
    diff = create_gaussian_diffusion('cosine', 1000)
    T, P = 12, 64
    small = fx._clip(41, 2, T, P)
    big = fx._clip(42, 5, T, P)

    def run(bt, use_graph):
        y = dev(fx.model_kwargs_y(bt, T))
        n = bt['noise'].to(DEV)
        return diff.p_sample_loop(mdm, tuple(n.shape), noise=n, clip_denoised=False, model_kwargs={'y': y}, seed=9, n_steps=60,
                                  use_graph=use_graph)
    a1 = run(small, True)
    junk = [torch.full((1 << 22,), float('nan'), device=DEV) for _ in range(8)]      # whatever gets freed is likely to land here
    b1 = run(big, True)
    del junk
    junk = [torch.full((1 << 22,), float('nan'), device=DEV) for _ in range(8)]
    a2 = run(small, True)
    assert torch.equal(a1, a2) and torch.equal(a1, run(small, False)) and torch.equal(b1, run(big, False))


# ------------------------------------------------------------------------------------------ the TIMED route at the TIMED shape
def _philox_step(lib_, seed):
    """step_noise callable that materialises the in-kernel generator's stream: draw ``it`` = interdiff_randn(seed, it) over the
    whole batch (the counters every chain of the graph route uses, csrc/philox.h)."""
    from interdiff_amd import _lib

    def draw(it, x):
        out = torch.empty_like(x)
        _lib.check(lib_.interdiff_randn(_lib.dptr(out), out.numel(), seed, it, _lib.stream()), 'randn')
        return out
    return draw


@pytest.mark.gpu
@pytest.mark.parametrize('B,T', [(8, 100), (12, 100), (16, 100), (32, 100), (17, 100), (25, 100), (32, 35)])
def test_timed_route_equals_eager_at_bench_shape(lib, mdm, smpl, B, T):
    """What bench.py times -- hipGraph blocks of fused plain steps (interdiff_mdm_forward_step), two half-batch chains whose 32-row
    tiles straddle clips at T = 100, in-kernel Philox, whole-batch hook steps in between -- against the EAGER route (one launch
    sequence per step, two-call update, noise INJECTED from the materialised Philox stream) at BASELINE configs #2 / #3:
    B = 16 / 32, T = 100, P = 2048, 120 steps from t = 560 (corrected steps t = 500 and t = 450 inside).  Bit for bit.
    B = 12: chains of 600 rows inside a batch of 1200 -- every launch must take the feed-forward tile picked for the BATCH (32 rows),
    not the one a 600-row launch would pick for itself (MDM._pick_ffn_tile).  B = 8 (config #4's share of a GPU): 800 rows, the
    16-row feed-forward kernel, one chain.  B = 17 / 25: ODD batches, chains of 9 + 8 / 13 + 12 clips (bench.py's batch-scaling leg).
    B = 32, T = 35: the reference's own default shape (eval_smpl_short.py:376-380; bench.py ``reference_default_B32_T35``): T % 4 != 0,
    the per-row form of the fused update, chains of 560 rows.  Reference loop: diffusion/gaussian_diffusion.py:663-736."""
    from interdiff_amd.diffusion import create_gaussian_diffusion
    diff = create_gaussian_diffusion('cosine', 1000)
    P = fx.TIMED_P
    bt, y = fx.timed_inputs(B, T)
    y, x_t = dev(y), bt['noise'].to(DEV)
    corr = make_correction(smpl, T, P)
    seed = 1234 + B
    run = lambda **kw: diff.p_sample_loop(mdm, tuple(x_t.shape), noise=x_t, clip_denoised=False, model_kwargs={'y': y}, denoised_fn=corr,
                                          n_steps=fx.TIMED_STEPS, first_t=fx.TIMED_FIRST_T, **kw)
    assert diff.fuse_plain_step and diff.split_chains
    timed = run(seed=seed)
    st = [v for k, v in mdm._graph_cache.items() if k[0] == diff._uid and k[1] == tuple(x_t.shape)]
    two = B * T > mdm.one_chain_max_rows() and B >= 4
    assert len(st) == 1 and (hasattr(st[0], 'chains') and len(st[0].chains) == 2) == two, 'two-chain route %s' % ('not taken' if two else 'taken')
    flags = [(key[3], key[4]) if key[0] == 'hook' else (key[1], key[2]) for key in st[0].graphs if isinstance(key, tuple)]       # (fused, split) of plain-step and hook-step graphs
    assert all(f and sp == two for f, sp in flags), 'fused%s graphs expected: %r' % (' + split' if two else '', list(st[0].graphs))
    assert any(key[0] == 'hook' and key[2] == 49 for key in st[0].graphs if isinstance(key, tuple)), 'the 49 plain steps before a corrected step and the corrected step are ONE graph'
    assert 'fwd' not in st[0].graphs, 'hook steps must be captured whole, not run eagerly'

    eager = run(step_noise=_philox_step(lib, seed), use_graph=False)
    assert torch.equal(timed, eager), 'timed route differs from the eager injected-noise route at B=%d: %g' % (B, (timed - eager).abs().max())
    assert torch.equal(timed, run(seed=seed)), 'graph reuse'
    if B % 2 == 0:
        diff.stagger_steps = 7               # the optional staggered form (equal chains on their own streams, hook per half batch): same bits
        assert torch.equal(timed, run(seed=seed)), 'staggered chains differ from the joined form (whole-batch hook steps)'
        diff.stagger_steps = 0
    assert torch.isfinite(timed).all()
    fx.record_parity('timed_route_vs_eager_B%d_T%d_P2048_120steps_from_t560' % (B, T), bit_identical=1.0, corrected_steps_inside=2)


@pytest.mark.gpu
def test_timed_route_whole_sample_equals_eager(lib, mdm, smpl):
    """One whole 1000-step sample (11 corrections) at B = 16, T = 100, P = 2048: the default graph route against the eager route fed
    the materialised Philox stream.  Bit for bit."""
    from interdiff_amd.diffusion import create_gaussian_diffusion
    diff = create_gaussian_diffusion('cosine', 1000)
    T, P = fx.TIMED_T, fx.TIMED_P
    bt, y = fx.timed_inputs(16)
    y, x_T = dev(y), bt['noise'].to(DEV)
    corr = make_correction(smpl, T, P)
    kw = dict(noise=x_T, clip_denoised=False, model_kwargs={'y': y}, denoised_fn=corr)
    timed = diff.p_sample_loop(mdm, tuple(x_T.shape), seed=99, **kw)
    eager = diff.p_sample_loop(mdm, tuple(x_T.shape), step_noise=_philox_step(lib, 99), use_graph=False, **kw)
    assert torch.equal(timed, eager), 'whole sample: %g' % (timed - eager).abs().max()
    fx.record_parity('timed_route_vs_eager_B16_whole_sample', bit_identical=1.0, corrected_steps_inside=11)


@pytest.mark.gpu
def test_timed_route_window_vs_oracle(lib, mdm):
    """The timed route (graph, fused step, two chains, in-kernel Philox) against the CPU oracle's p_sample_loop fed the SAME
    (materialised) noise stream: B = 16, T = 100, 50 plain steps from t = 560 (the oracle's hook costs minutes per call at this
    size; corrected steps are covered bit for bit by the tests above + the full-size golden).  North-star tolerance 1e-4."""
    from interdiff_amd.diffusion import create_gaussian_diffusion
    diff = create_gaussian_diffusion('cosine', 1000)
    bt, y = fx.timed_inputs(16)
    yd, x_t = dev(y), bt['noise'].to(DEV)
    n, seed = 50, 4321
    got = diff.p_sample_loop(mdm, tuple(x_t.shape), noise=x_t, clip_denoised=False, model_kwargs={'y': yd}, seed=seed, n_steps=n, first_t=fx.TIMED_FIRST_T)
    draw = _philox_step(lib, seed)
    stream = [draw(it, x_t).cpu() for it in range(n)]
    ref = odf.p_sample_loop(lambda x, t, y: oden.mdm_forward(fx.mdm_weights(), x, t, y['cond']), tuple(x_t.shape), odf.make_schedule(1000),
                            bt['noise'].clone(), lambda i, x: stream[i], {'y': y}, n_steps=n, first_t=fx.TIMED_FIRST_T)
    e = close(got, ref, 1e-4, 'timed route, 50 steps from t=560 at B=16,T=100 vs oracle')
    fx.record_parity('timed_route_vs_oracle_B16_T100_50steps_from_t560', worst_rel_err=e, asserted=1e-4)


# ------------------------------------------------------------------------------------------ row (e): sharded == unsharded, bit for bit
def _shard_y(y, sl):
    dims = dict(inpainted_motion=0, inpainting_mask=0, obj_points=0, hand_pose=1, beta=1, cond=1)
    return {k: (v[(slice(None),) * dims[k] + (sl,)].contiguous() if k in dims else v) for k, v in y.items()}


@pytest.mark.gpu
@pytest.mark.parametrize('B,world', [(64, 8), (16, 2), (24, 3)])
def test_emulated_ranks_equal_unsharded_sampler(lib, mdm, smpl, B, world):
    """SURVEY.md §8(e) / VERDICT r03 #1: ``world`` emulated ranks, run ONE AFTER ANOTHER on this GPU, each on its contiguous clip shard
    with ``shard=(first_clip, total_clips)`` == the unsharded batch, ``torch.equal`` on the sampler state after 120 steps from t = 560
    (corrected steps t = 500 and t = 450 inside), at T = 100, P = 2048 -- the timed route (graphs, fused steps, chains, in-kernel
    Philox at the GLOBAL counters, feed-forward tile class of the GLOBAL batch).  B = 64 over 8 ranks is BASELINE config #4's
    partitioning (global 64-row tile, ranks on the bit-identical 16-row grid); B = 16 over 2: the global batch takes the 32-row tile,
    so must the 800-row shards; B = 24 over 3: chains inside the shards as well.  Also the eager route on one shard.
    Reference seam: eval_smpl_short.py:252-296 (one randn_like tensor per step for the whole batch, gaussian_diffusion.py:532)."""
    from interdiff_amd.diffusion import create_gaussian_diffusion
    from interdiff_amd.dist import shard_slice
    diff = create_gaussian_diffusion('cosine', 1000)
    T, P = fx.TIMED_T, fx.TIMED_P
    bt, y = fx.timed_inputs(B)
    y, x_t = dev(y), bt['noise'].to(DEV)
    corr = make_correction(smpl, T, P)
    seed = 777 + B
    kw = dict(clip_denoised=False, denoised_fn=corr, n_steps=fx.TIMED_STEPS, first_t=fx.TIMED_FIRST_T)
    whole = diff.p_sample_loop(mdm, tuple(x_t.shape), noise=x_t, model_kwargs={'y': y}, seed=seed, **kw)
    parts = []
    for r in range(world):
        sl = shard_slice(B, r, world)
        xs = x_t[sl].contiguous()
        parts.append(diff.p_sample_loop(mdm, tuple(xs.shape), noise=xs, model_kwargs={'y': _shard_y(y, sl)}, seed=seed, shard=(sl.start, B), **kw))
    got = torch.cat(parts, dim=0)
    assert torch.equal(got, whole), 'sharded (%d ranks) differs from unsharded at B=%d: %g' % (world, B, (got - whole).abs().max())
    # a shard WITHOUT its global position is a different sample (the test would be vacuous otherwise)
    sl = shard_slice(B, world - 1, world)
    xs = x_t[sl].contiguous()
    alone = diff.p_sample_loop(mdm, tuple(xs.shape), noise=xs, model_kwargs={'y': _shard_y(y, sl)}, seed=seed, **kw)
    assert not torch.equal(alone, whole[sl])
    # eager route of the last shard, in-kernel generator at the global counters
    eager = diff.p_sample_loop(mdm, tuple(xs.shape), noise=xs, model_kwargs={'y': _shard_y(y, sl)}, seed=seed, shard=(sl.start, B), use_graph=False, **kw)
    assert torch.equal(eager, whole[sl]), 'eager shard: %g' % (eager - whole[sl]).abs().max()
    # x_T drawn in-kernel (noise=None): the shard draws its slice of the whole batch's tensor
    y2 = dict(y)
    w2 = diff.p_sample_loop(mdm, tuple(x_t.shape), model_kwargs={'y': y2}, seed=seed, clip_denoised=False, n_steps=3)
    s2 = diff.p_sample_loop(mdm, tuple(xs.shape), model_kwargs={'y': _shard_y(y2, sl)}, seed=seed, clip_denoised=False, n_steps=3, shard=(sl.start, B))
    assert torch.equal(s2, w2[sl])
    fx.record_parity('sharded_equals_unsharded_B%d_world%d_T100_P2048_120steps_from_t560' % (B, world), bit_identical=1.0, corrected_steps_inside=2)


@pytest.mark.gpu
def test_emulated_ranks_equal_unsharded_eval_and_rollout(lib, smpl):
    """The two sharded entry points themselves, 8 emulated ranks on one GPU against the unsharded call, bit for bit:
    ``evaluate_sharded`` (eval_smpl_short.py:252-296; B = 16 clips, T = 24, two draws, 50-step schedule with a corrected step: equal
    per-clip metric vectors) and ``sample_long_sharded`` (BASELINE config #4's partitioning; K = 2 windows: equal rollouts incl. the
    conditioning pass per window)."""
    from interdiff_amd import eval as ev, synthetic as syn
    from interdiff_amd.mdm import MDM
    from interdiff_amd.diffusion import create_gaussian_diffusion
    T, B, P, past, steps, K, world = 24, 16, 256, fx.PAST, 50, 2, 8
    ei = {k: torch.from_numpy(v).to(DEV) for k, v in syn.make_embedding_inputs(seed=8, B=B, T=T, n_points=P).items()}
    g = torch.Generator().manual_seed(4)
    raw = dict(ei, hand_pose=(0.1 * torch.randn(T, B, 90, generator=g)).to(DEV), beta=torch.randn(1, B, 10, generator=g).expand(T, B, 10).contiguous().to(DEV))
    model = MDM(fx.mdm_weights(), device=DEV, n_steps=steps)
    corr = make_correction(smpl, T, P)
    diff = create_gaussian_diffusion('cosine', steps)
    batch = ev.batch_from_raw(model, raw, past)
    full, means = ev.evaluate_sharded(model, diff, corr, batch, past, 'correction', 2, seed=31)          # world = 1: the unsharded run
    got = {k: [] for k in full}
    for r in range(world):
        m, _ = ev.evaluate_sharded(model, diff, corr, batch, past, 'correction', 2, seed=31, rank=r, world=world, collate=False)
        for k in m:
            got[k].append(m[k])
    for k in full:
        assert torch.equal(torch.cat(got[k]), full[k]), (k, torch.cat(got[k]), full[k])
    whole = ev.sample_long(model, diff, corr, raw, K, past, seed=5)
    for r in range(world):
        sl, mine = ev.sample_long_sharded(model, diff, corr, raw, K, past, seed=5, rank=r, world=world)
        for name, a, b in zip(('obj', 'body', 'verts', 'jtr', 'pelvis'), whole, mine):
            assert torch.equal(a[:, sl], b), 'rank %d %s: %g' % (r, name, (a[:, sl] - b).abs().max())
    fx.record_parity('sharded_equals_unsharded_evaluate_and_rollout_B16_world8', bit_identical=1.0)


@pytest.mark.gpu
def test_rccl_single_rank_all_gather_on_gpu(lib):
    """The path's one collective through RCCL for real on the one GPU there is: a 1-rank ``nccl`` process group, ``gather_metrics`` with
    the world == 1 shortcut bypassed (``force_collective``), header verified.  Runs in a child process so that the process group never
    leaks into the other tests."""
    import os
    import subprocess
    import sys
    code = r'''
import os, sys, json, torch
sys.path.insert(0, %r)
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29611', RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
from interdiff_amd import dist as idist
import torch.distributed as dist
r, w, l = idist.init_from_env('nccl', single_rank_group=True)
assert dist.is_initialized() and dist.get_backend() == 'nccl' and (r, w) == (0, 1)
local = {k: torch.arange(5, dtype=torch.float32, device='cuda') + 10 * i for i, k in enumerate(idist.METRIC_KEYS)}
out, header = idist.gather_metrics(local, 1, counts=[5], return_header=True, force_collective=True, check_header=True)
torch.cuda.synchronize()
assert header.tolist() == [5.0] and all(torch.equal(out[k], local[k]) for k in local)
assert out['penetrate'].data_ptr() != local['penetrate'].data_ptr()          # came back out of the all-gather buffers, not the shortcut
assert idist.max_over_ranks(1.5, 'cuda') == 1.5 and idist.gather_scalar(2.5, 'cuda') == [2.5]
info = idist.collective_backend_info()
idist.barrier(); idist.shutdown()
print('RCCL_OK ' + json.dumps(info))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    p = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0 and 'RCCL_OK' in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]
    import json
    info = json.loads(p.stdout.split('RCCL_OK ', 1)[1].splitlines()[0])
    assert info['backend'] == 'nccl' and info['rccl_version']
    fx.record_parity('rccl_single_rank_all_gather', ok=1.0, rccl_version=str(info['rccl_version']))


@pytest.mark.gpu
def test_dataset_batch_form_is_a_drop_in(mdm, smpl):
    """``sample_once_proj(batch)`` / ``sample_once(batch)`` with the DataLoader's dict-of-lists batch, as the reference calls them
    (eval_smpl_short.py:133-150,179-215; data/dataset_smpl.py:182-204), equal the stacked-tensor form bit for bit; ``evaluate_batch`` takes it too."""
    from interdiff_amd import eval as ev, synthetic as syn
    from interdiff_amd.mdm import MDM
    from interdiff_amd.diffusion import create_gaussian_diffusion
    T, B, P, past, steps = 16, 3, 128, fx.PAST, 20
    ei = {k: torch.from_numpy(v) for k, v in syn.make_embedding_inputs(seed=9, B=B, T=T, n_points=P).items()}
    g = torch.Generator().manual_seed(3)
    hand, beta = 0.1 * torch.randn(T, B, 90, generator=g), torch.randn(1, B, 10, generator=g).expand(T, B, 10).contiguous()
    frames = [dict(smplfit_params=dict(pose=torch.cat([ei['body_pose'][t], hand[t]], dim=1), betas=beta[t], trans=ei['body_trans'][t]),
                   objfit_params=dict(angle=ei['obj_angles'][t].double(), trans=ei['obj_trans'][t].double())) for t in range(T)]
    ds_batch = dict(frames=frames, obj_points=torch.cat([ei['obj_points'], torch.zeros(B, P, 3)], dim=2), gender=['male'] * B)
    raw = dev(dict(ei, hand_pose=hand, beta=beta))
    model = MDM(fx.mdm_weights(), device=DEV, n_steps=steps)
    corr = make_correction(smpl, T, P)
    diff = create_gaussian_diffusion('cosine', steps)
    clip = ev.batch_from_raw(model, raw, past)
    a = ev.sample_once_proj(model, diff, corr, clip, past, seed=4)
    b = ev.sample_once_proj(model, diff, corr, ds_batch, past, seed=4)
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    a = ev.sample_once(model, diff, smpl, clip, past, seed=5)
    b = ev.sample_once(model, diff, smpl, ds_batch, past, seed=5)
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    ma, mb = ev.evaluate_batch(model, diff, corr, clip, past, seed=6), ev.evaluate_batch(model, diff, corr, ds_batch, past, seed=6)
    assert all(torch.equal(ma[k], mb[k]) for k in ma)


@pytest.mark.gpu
def test_config4_long_horizon_named_shape_golden(mdm, smpl):
    """BASELINE config #4 at its NAMED per-GPU shape: 8 clips (64 sharded over 8 GPUs), T = 100, P = 2048, the first window
    + K = 2 autoregressive windows, a 50-step schedule with one corrected step per window, injected x_T and per-step noise
    (tests/golden/long4.npz, recorded offline by `make_golden.py long4` from oracle/long_horizon.py, 383 s of CPU; upstream's own
    rollout cannot run, see there) against ``eval.sample_long`` on the HIP path: conditioning pass (PointNet++ + 8-layer encoder) per
    window, sampler + hook, window algebra, at 800 token rows (the 16-row feed-forward tile), under both feed-forward arithmetics.  1e-4 on translations / joints /
    markers, rotations compared as rotation matrices (1e-3, see below)."""
    from interdiff_amd import eval as ev
    from interdiff_amd.mdm import MDM
    from interdiff_amd.diffusion import create_gaussian_diffusion
    T, B, P, K, steps = fx.LONG4_SHAPE
    raw, x_T, step_noise = fx.long4_inputs()
    g = fx.golden('long4.npz')
    model = MDM(fx.mdm_weights(), device=DEV, n_steps=steps)
    corr = make_correction(smpl, T, P)
    diff = create_gaussian_diffusion('cosine', steps)

    def sn(k):
        f = step_noise(k)
        return lambda i, x: f(i, x).to(DEV)
    worst = {}
    for math in ('exact', 'split'):
        model.ffn_math = math
        obj, body, verts, jtr, pelvis = ev.sample_long(model, diff, corr, dev(raw), K, fx.PAST, x_T=lambda k: x_T(k).to(DEV), step_noise=sn)
        assert obj.shape == (T + K * (T - fx.PAST), B, 6)
        errs = {}
        for n, a, nr in (('obj', obj, 3), ('body', body, 66)):
            b = torch.from_numpy(g[n])
            errs[n + '_rot'] = rel(R.axis_angle_to_matrix(a[..., :nr].reshape(*a.shape[:2], -1, 3).cpu()), R.axis_angle_to_matrix(b[..., :nr].reshape(*b.shape[:2], -1, 3)))
            errs[n + '_rest'] = rel(a[..., nr:], b[..., nr:])
        errs['jtr'], errs['pelvis'], errs['markers'] = rel(jtr, g['jtr']), rel(pelvis, g['pelvis']), rel(verts[:, :, ocor.MARKERS67], g['markers'])
        fx.record_parity('config4_long_horizon_B8_T100_P2048_K2_50steps_vs_offline_golden_%s' % math, asserted_translations_joints_markers=1e-4, asserted_rotations=1e-3, **errs)
        worst[math] = errs
    # translations, joints, markers (what the rotations are FOR): north_star's 1e-4 (measured 2e-6 .. 8e-5).  Rotation MATRICES: 1e-3 -- the
    # rot6d -> matrix Gram-Schmidt of a random-init denoiser's output amplifies an fp32-level difference of two samples ~50x, and the golden is
    # itself an fp32 run (the same effect, at the same size, as in test_full_size_end_to_end_golden, where the fp64 twin shows the REFERENCE's own
    # fp32 run 1.0e-3 from exact on body rotations); three chained windows compound it.  Measured: body 4.9e-4 (exact-fp32 feed-forward) /
    # 3.0e-4 (split-f16), object 1.6e-4 / 1.1e-4.
    for math, errs in worst.items():
        for k, v in errs.items():
            assert v <= (1e-3 if k.endswith('_rot') else 1e-4), 'config #4 (%s) %s: rel err %.3e' % (math, k, v)

