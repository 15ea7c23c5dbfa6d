"""CPU-only: the C-ABI library loads and exports every symbol include/interdiff_hip.h declares; host-side packing
logic (no GPU compute calls)."""
import os
import re
import subprocess
import sys
import numpy as np
import pytest
import torch
from tests import fixtures as fx

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))


def header_symbols():
    src = open(os.path.join(ROOT, 'include', 'interdiff_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(interdiff_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    from interdiff_amd import _lib
    lib = _lib.load()
    declared = header_symbols()
    assert len(declared) >= 20
    out = subprocess.run(['nm', '-D', '--defined-only', _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r' T (interdiff_[a-z0-9_]+)', out))
    assert set(declared) <= exported, sorted(set(declared) - exported)
    assert set(_lib.exported_symbols()) == set(declared), 'ctypes table out of sync with the header'
    assert lib.interdiff_abi_version() == _lib.ABI_VERSION
    assert b'gfx950' in lib.interdiff_build_info()


def test_no_cpu_fallback():
    from interdiff_amd import transforms, _lib
    with pytest.raises(ValueError):
        transforms.axis_angle_to_matrix(torch.zeros(2, 3))          # CPU tensor -> loud failure, not a silent fallback
    assert issubclass(_lib.HipLibraryMissing, RuntimeError)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'interdiff_amd')
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(('.py', '.hip', '.h')):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', txt, flags=re.M), f
                assert '/root/reference' not in txt, f


def test_schedule_scalars_match_oracle():
    from interdiff_amd.diffusion import create_gaussian_diffusion
    from oracle import diffusion as odf
    for steps in (1000, 50):
        d, s = create_gaussian_diffusion('cosine', steps), odf.make_schedule(steps)
        for k in ('betas', 'posterior_mean_coef1', 'posterior_mean_coef2', 'posterior_log_variance_clipped'):
            np.testing.assert_allclose(getattr(d, k), s[k], rtol=0, atol=1e-15)
        z = fx.golden('schedule.npz')
        np.testing.assert_allclose(d.posterior_mean_coef1, z['posterior_mean_coef1_%d' % steps], rtol=0, atol=1e-15)
    assert d._sigma.dtype == np.float32 and d._c1.dtype == np.float32


def test_vertex_adjacency_order():
    from interdiff_amd.geometry import build_vertex_adjacency
    faces = np.array([[0, 1, 2], [2, 1, 3], [1, 0, 3]])
    ptr, af, ac = build_vertex_adjacency(faces, 4)
    assert ptr.tolist() == [0, 2, 5, 7, 9]
    # vertex 1: corner-1 faces first (0, 1), then corner-2 (none), then corner-0 (face 2)
    assert af[ptr[1]:ptr[2]].tolist() == [0, 1, 2] and ac[ptr[1]:ptr[2]].tolist() == [1, 1, 0]
    # vertex 3: corner 2 of faces 1 and 2
    assert af[ptr[3]:ptr[4]].tolist() == [1, 2] and ac[ptr[3]:ptr[4]].tolist() == [2, 2]


def test_pack_smpl_model_host_side():
    from interdiff_amd.smpl import pack_smpl_model
    model = fx.smpl_model()
    m, bufs = pack_smpl_model(model, 'cpu')
    assert (m.V, m.J, m.n_betas, m.KB) == (6890, 52, 10, 480) and 1 <= m.S <= 4
    betas = torch.randn(10)
    J_ref = model['J_regressor'] @ (model['v_template'] + torch.einsum('vck,k->vc', model['shapedirs'], betas))
    J_got = bufs['jt'] + torch.einsum('jck,k->jc', bufs['js'], betas)
    assert (J_ref - J_got).abs().max() < 1e-5
    dense = torch.zeros(6890, 52)
    dense.scatter_add_(1, bufs['skin_idx'].long(), bufs['skin_w'])
    assert torch.equal(dense, model['weights'])


def test_pack_objprojector_folds():
    from interdiff_amd.objprojector import pack_objprojector, dct_matrices
    op, arena = pack_objprojector(fx.objproj_weights(), T=35, past_len=10, device='cpu')
    d, _ = dct_matrices(35)
    dpad = arena[op.dct_pad:op.dct_pad + 100].reshape(10, 10).double().numpy()
    pad = list(range(10)) + [9] * 25
    x = np.random.RandomState(0).randn(35)
    np.testing.assert_allclose(dpad @ x[:10], d[:10] @ x[pad], atol=1e-5)
    assert list(op.cin) == [9, 32, 16, 32] * 3 and list(op.cout) == [32, 16, 32, 9] * 3


def test_gloo_sharding_world2():
    """N>1 path on CPU: clips are sharded over ranks, metrics all-gathered (gloo stands in for RCCL)."""
    import torch.multiprocessing as mp
    from tests import dist_workers
    mp.spawn(dist_workers._selftest_worker, args=(2, 29517), nprocs=2, join=True)


def test_gloo_evaluate_sharded_world2():
    """The eval entry that issues the path's one collective (interdiff_amd/eval.py: evaluate_sharded, eval_smpl_short.py:265-296):
    uneven clip shards, one seed for all ranks + each shard's global position, all-gather, clip order of the collated per-clip metric vectors -- 2 gloo ranks."""
    import torch.multiprocessing as mp
    from tests import dist_workers
    mp.spawn(dist_workers._selftest_eval_worker, args=(2, 29531), nprocs=2, join=True)


def test_gloo_world8_config4_and_config5_shapes():
    """BASELINE configs #4 / #5 partitioning at the named shapes, 8 gloo ranks on CPU (no 8-GPU node was available to measure on):
    ``evaluate_sharded`` at B = 64 (8 clips per rank: shard, global positions, the ONE fixed-size all-gather, clip order) and
    ``sample_long_sharded`` at B = 64 (every rank rolls out its own clips, no exchange; equal to the unsharded rollout clip by clip)."""
    import torch.multiprocessing as mp
    from tests import dist_workers
    mp.spawn(dist_workers._selftest_eval_worker, args=(8, 29547), nprocs=8, join=True)
    mp.spawn(dist_workers._selftest_long_worker, args=(8, 29559), nprocs=8, join=True)


def test_behave_etl_matches_reference_dataset(tmp_path):
    """"Next" row N2: clip canonicalisation + file formats + windows of interdiff_amd/data.py against the reference's own
    Dataset.__getitem__ on three windows of the shipped BEHAVE sequence (tests/golden/etl.npz holds the raw frames and the
    reference's outputs)."""
    import numpy as np
    from interdiff_amd import data as D
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'etl.npz'))
    sel, starts = z['sel'], z['starts']
    F = int(sel.max()) + 1
    full = lambda a: _scatter(F, sel, a)
    seq = dict(poses=full(z['poses']), betas=full(z['betas']), trans=full(z['trans']), obj_angles=full(z['obj_angles']), obj_trans=full(z['obj_trans']))
    pelvis = full(z['pelvis'])
    # round trip through the on-disk format the loader expects
    d = tmp_path / 'Date01_Sub01_backpack_back'
    d.mkdir()
    np.savez(d / 'smpl_fit_all.npz', poses=seq['poses'], betas=seq['betas'], trans=seq['trans'])
    np.savez(d / 'object_fit_all.npz', angles=seq['obj_angles'], trans=seq['obj_trans'], frame_times=np.arange(F))
    loaded = D.load_behave_sequence(str(d))
    assert all(np.array_equal(loaded[k], seq[k]) for k in seq)
    assert D.test_windows(1408, 10, 25)[:3] == [0, 35, 70] and len(D.test_windows(1408, 10, 25)) == 40
    for w, s0 in enumerate(starts):
        c = D.canonicalize_clip(loaded, pelvis, int(s0), 10, 25)
        for ours, ref in (('pose', 'pose_%d'), ('trans', 'trans_%d'), ('obj_angles', 'angle_%d'), ('obj_trans', 'otrans_%d'), ('pelvis', 'pelvis_%d')):
            a, b = np.asarray(c[ours], np.float64), z[ref % w].astype(np.float64)
            assert np.abs(a - b).max() <= 1e-6 * max(1.0, np.abs(b).max()), (ours, w, np.abs(a - b).max())
        assert np.allclose(c['centroid'], z['centroid_%d' % w]) and np.allclose(c['rotation'], z['rotation_%d' % w])
        assert np.abs(c['pelvis'][0]).max() < 1e-6                                   # first pelvis is the origin
        R0 = __import__('scipy.spatial.transform', fromlist=['Rotation']).Rotation.from_rotvec(c['pose'][0, :3]).as_matrix()
        assert abs(R0[2, 0]) < 1e-5                                                   # yaw of the first frame removed
    # the contact-side records (contact.npz / info.json in the reference's on-disk format; data/dataset_smpl.py:48-56,152-180)
    import json
    cuts_o, cuts_h = np.cumsum(z['obj_contact_n'])[:-1], np.cumsum(z['human_contact_n'])[:-1]
    per_frame = lambda parts: [parts[list(sel).index(i)] if i in set(sel.tolist()) else np.zeros(0, np.int64) for i in range(F)]
    contact = dict(object_points=z['obj_points6'], object_contact_vertex_label=per_frame(np.split(z['obj_contact_idx'], cuts_o)),
                   human_contact_vertex_label=per_frame(np.split(z['human_contact_idx'], cuts_h)), foot_contact_joint_label=full(z['foot_label']).astype(np.int64))
    np.savez(d / 'contact.npz', contact)
    json.dump(dict(gender='male', cat='backpack'), open(d / 'info.json', 'w'))
    loaded = D.load_behave_sequence(str(d))
    assert loaded['gender'] == 'male' and loaded['obj_name'] == 'backpack' and np.array_equal(loaded['obj_points'], z['obj_points6'])
    lf, rf = full(z['left_foot']), full(z['right_foot'])
    for w, s0 in enumerate(starts):
        c = D.canonicalize_clip(loaded, pelvis, int(s0), 10, 25)
        lab = D.clip_labels(loaded, c, lf, rf, int(s0), 10, 25)
        ref = z['objpts_%d' % w].astype(np.float64)
        assert np.abs(lab['obj_points'] - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max()), (w, np.abs(lab['obj_points'] - ref).max())
        assert np.array_equal(lab['obj_points'][..., 6], ref[..., 6])
        assert np.array_equal(lab['ground_joint_label'], z['ground_%d' % w].astype(np.float64)), w
        assert np.array_equal(lab['contact_label'].sum(1), z['contact_count_%d' % w])
        first = np.array([int(np.nonzero(r)[0][0]) if r.any() else -1 for r in lab['contact_label']])
        assert np.array_equal(first, z['contact_first_%d' % w])
    clips = [D.canonicalize_clip(loaded, pelvis, int(s0), 10, 25) for s0 in starts]
    raw = D.collate_raw(clips, np.zeros((64, 3), np.float32), device='cpu')
    assert raw['body_pose'].shape == (35, 3, 66) and raw['hand_pose'].shape == (35, 3, 90) and raw['obj_points'].shape == (3, 64, 3)


def _scatter(F, sel, a):
    import numpy as np
    out = np.zeros((F,) + a.shape[1:], a.dtype)
    out[sel] = a
    return out


def test_joint_map_vjp_host_matches_autograd():
    """The derivative code of csrc/rot_dual.h (rotation -> matrix_to_axis_angle -> SMPL Rodrigues, "next" row N4) is
    host+device inline; its host instance (interdiff_debug_joint_map_vjp) must equal torch autograd of the oracle,
    including the exact-zero entries at identity rotations (Adam never moves those)."""
    from interdiff_amd import _lib
    from oracle import rotations as rot
    lib = _lib.load()
    rs = np.random.RandomState(0)
    n = 64
    aa = 0.8 * rs.standard_normal((n, 3)).astype(np.float32)
    aa[0] = 0
    aa[1] = [3.0, 0.5, -0.3]            # all four quaternion candidates get picked
    aa[2] = [0, 3.1, 0.2]
    aa[3] = [0.1, 3.0, 3.0]
    aa[4] = [1e-4, 0, 0]
    R = rot.axis_angle_to_matrix(torch.from_numpy(aa))
    R = R + 2e-3 * torch.from_numpy(rs.standard_normal((n, 3, 3)).astype(np.float32)) * (torch.arange(n) >= 5)[:, None, None]
    R = R.clone().requires_grad_(True)
    g = torch.from_numpy(rs.standard_normal((n, 3, 3)).astype(np.float32))
    with torch.enable_grad():
        rot.rodrigues_smpl(rot.matrix_to_axis_angle(R)).backward(g)
    ref = R.grad.numpy().reshape(n, 9)
    Rn, gn = np.ascontiguousarray(R.detach().numpy().reshape(n, 9)), np.ascontiguousarray(g.numpy().reshape(n, 9))
    out = np.zeros((n, 9), np.float32)
    assert lib.interdiff_debug_joint_map_vjp(Rn.ctypes.data, gn.ctypes.data, out.ctypes.data, n) == 0
    assert np.abs(out - ref).max() <= 1e-5 * np.abs(ref).max()
    assert ((out == 0) == (ref == 0)).all()


def test_ffn_pack_and_kernel_addressing_by_emulation():
    """The fused feed-forward kernel (csrc/ffn.h) restated lane by lane in numpy (tests/ffn_emulator.py) on the weight stream
    pack_ffn builds: sum of the five partial slabs == x2 + linear2(gelu(linear1(x2))) for a ragged row count, with the DMA applied
    at issue time and at the covering wait (ring-slot reuse hazards show up as a wrong answer in one of the two); both row tiles."""
    import numpy as np
    from interdiff_amd.mdm import pack_ffn, pad_ffn_bias, ffn_slices
    from tests.ffn_emulator import emulate_ffn, _gelu
    rs = np.random.RandomState(0)
    M = 37                                                 # two M tiles, the second ragged
    x2 = rs.standard_normal((M, 256)).astype(np.float32)
    w1 = (rs.standard_normal((1024, 256)) / 16).astype(np.float32)
    w2 = (rs.standard_normal((256, 1024)) / 32).astype(np.float32)
    b1, b2 = rs.standard_normal(1024).astype(np.float32), rs.standard_normal(256).astype(np.float32)
    pack = pack_ffn(w1, w2)
    assert [s for s in ffn_slices()] == [(0, 208), (208, 208), (416, 208), (624, 208), (832, 208)] and pack.size == 5 * 106496
    ref = x2.astype(np.float64) + _gelu(x2.astype(np.float64) @ w1.T.astype(np.float64) + b1) @ w2.T.astype(np.float64) + b2
    for bm in (32, 16, 64):                                # ffn_fused_kernel / ffn_fused16_kernel (three M tiles, the last ragged) / ffn_fused64_kernel (one ragged tile)
        for late in (False, True):
            parts = emulate_ffn(x2, pack, pad_ffn_bias(b1), b2, late, bm)
            err = np.abs(parts.sum(0) - ref).max()
            assert err < 1e-9, (bm, late, err)


def test_split_f16_ffn_pack_and_kernel_addressing_by_emulation():
    """The split-f16 feed-forward kernel (csrc/ffn_h2.h) restated lane by lane in numpy (tests/ffn_emulator.py emulate_ffn_h2) on the
    stream pack_ffn_h2 builds: the f16 planes (hi, residual x 2^11), the [tile][plane][lane][8] fragments, the K-step ring with the DMA
    applied at issue time and at the covering wait, the in-place split of the x2 rows, the plane addressing, weights as the MFMA's
    A operand, the GELU / split / zero-pad phase -- sum of the five slabs == x2 + linear2(gelu(linear1(x2))) to the split's own
    representation error (2^-23 per operand; sums exact in the emulator), all three row tiles, ragged rows.  Also the split itself and
    the pack-time range proof."""
    import numpy as np
    from interdiff_amd.mdm import pack_ffn_h2, pad_ffn_bias, split_f16, ffn_h2_range_ok, H2_SLICE_FLOATS
    from tests.ffn_emulator import emulate_ffn_h2, _gelu
    rs = np.random.RandomState(0)
    v = np.concatenate([rs.standard_normal(4096) * 10.0 ** rs.uniform(-12, 4, 4096), [0.0, -0.0, 2.0 ** -14, 2.0 ** -15, -2.0 ** -24, 65000.0, 1e-30]]).astype(np.float32)
    hi, lo = split_f16(v)
    back = hi.astype(np.float64) + lo.astype(np.float64) / 2048.0
    assert np.all(np.abs(back - v) <= np.maximum(np.abs(v.astype(np.float64)) * 2.0 ** -23, 2.0 ** -25)), 'v = hi + lo / 2048 to 2^-23 |v| (absolute 2^-25 at the bottom)'
    nz = hi != 0
    assert np.all(np.abs(hi[nz].astype(np.float32)) >= 2.0 ** -14), 'no subnormal in the hi plane'
    M = 37                                                 # ragged in every tile size
    x2 = rs.standard_normal((M, 256)).astype(np.float32)
    w1 = (rs.standard_normal((1024, 256)) / 16).astype(np.float32)
    w2 = (rs.standard_normal((256, 1024)) / 32).astype(np.float32)
    b1, b2 = rs.standard_normal(1024).astype(np.float32), rs.standard_normal(256).astype(np.float32)
    pack = pack_ffn_h2(w1, w2)
    assert pack.dtype == np.float32 and pack.size == 5 * H2_SLICE_FLOATS == 5 * 110592
    ref = x2.astype(np.float64) + _gelu(x2.astype(np.float64) @ w1.T.astype(np.float64) + b1) @ w2.T.astype(np.float64) + b2
    outs = []
    for bm in (32, 16, 64):
        for late in (False, True):
            parts = emulate_ffn_h2(x2, pack, pad_ffn_bias(b1), b2, late, bm)
            err = np.abs(parts.sum(0) - ref).max() / np.abs(ref).max()
            assert err < 2e-7, (bm, late, err)
            outs.append(parts)
    assert all(np.array_equal(outs[0], o) for o in outs[1:]), 'row tile and DMA timing must not change a single element'
    ones, zeros = np.ones(256, np.float32), np.zeros(256, np.float32)
    assert ffn_h2_range_ok(w1, b1, w2, ones, zeros)
    assert not ffn_h2_range_ok(w1 * 400, b1, w2, ones, zeros) and not ffn_h2_range_ok(w1, b1, w2 * 3e6, ones, zeros) and not ffn_h2_range_ok(w1, b1, w2, ones * 5000, zeros)


def test_out_projection_fragments_match_the_attention_kernel_addressing():
    """mdm.sa_out_fragments vs the loads of self_attn_kernel<true> (csrc/denoiser.hip): lane (kq, li) of wave `wave` reads, for head h,
    k-group s and column tile c, the float4 at ((((h*4+wave)*4+s)*4+c)*64+lane)*4 and uses element e as W_o[(wave*4+c)*16+li][h*64+16s+4kq+e];
    the per-head partial products it forms must add up to the plain out-projection."""
    from interdiff_amd.mdm import sa_out_fragments
    rs = np.random.RandomState(0)
    w = rs.randn(256, 256).astype(np.float32)
    f = sa_out_fragments(w)
    assert f.shape == (256 * 256,)
    h, wave, s, c, lane, e = np.meshgrid(*[np.arange(n) for n in (4, 4, 4, 4, 64, 4)], indexing='ij')
    li, kq = lane & 15, lane >> 4
    idx = ((((h * 4 + wave) * 4 + s) * 4 + c) * 64 + lane) * 4 + e
    assert np.array_equal(f[idx], w[(wave * 4 + c) * 16 + li, h * 64 + 16 * s + 4 * kq + e])
    assert np.array_equal(np.sort(idx.ravel()), np.arange(256 * 256))          # a permutation: every weight exactly once
    # the kernel's contraction, in numpy: out[:, n] = sum_h ctx[:, h*64:(h+1)*64] . W_o[n, h*64:(h+1)*64]
    ctx = rs.randn(32, 256)
    fr = f.reshape(4, 4, 4, 4, 4, 16, 4).astype(np.float64)                    # [h][wave][s][c][kq][li][e]
    out = np.zeros((32, 256))
    for hh in range(4):
        a = ctx[:, hh * 64:(hh + 1) * 64].reshape(32, 4, 4, 4)                 # [row][s][kq][e]
        out += np.einsum('rsqe,wscqle->rwcl', a, fr[hh]).reshape(32, 256)
    assert np.allclose(out, ctx @ w.astype(np.float64).T, rtol=1e-12, atol=1e-12)


def test_qkv_pack_and_kernel_addressing_by_emulation():
    """The LayerNorm+linear kernel of the QKV projection (csrc/ffn.h ln_linear_kernel) restated lane by lane on the stream
    pack_linear160 builds: LN(sum of the five slabs) . W^T + b for a ragged row count, DMA applied at issue and at the wait."""
    import numpy as np
    from interdiff_amd.mdm import pack_linear160
    from tests.ffn_emulator import emulate_ln_linear
    rs = np.random.RandomState(1)
    M, N = 37, 768
    slabs = rs.standard_normal((5, M, 256)).astype(np.float32)
    w = (rs.standard_normal((N, 256)) / 16).astype(np.float32)
    b, lnw, lnb = (rs.standard_normal(n).astype(np.float32) for n in (N, 256, 256))
    pack = pack_linear160(w)
    assert pack.size == 5 * 160 * 256                       # 768 rows in 5 slices of 160, the tail zero
    x = slabs.astype(np.float64).sum(0)
    xn = (x - x.mean(1, keepdims=True)) / np.sqrt(x.var(1, keepdims=True) + 1e-5) * lnw + lnb
    for late in (False, True):
        assert np.abs(emulate_ln_linear(slabs, lnw, lnb, pack, b, N, late) - (xn @ w.T.astype(np.float64) + b)).max() < 1e-9
    assert np.abs(emulate_ln_linear(slabs[:1], None, None, pack, b, N, True) - (slabs[0].astype(np.float64) @ w.T + b)).max() < 1e-9


def test_split_f16_qkv_pack_and_kernel_addressing_by_emulation():
    """The split-f16 LayerNorm + linear kernel (csrc/ffn_h2.h ln_linear_h2_kernel) restated lane by lane (tests/ffn_emulator.py) on the stream
    pack_linear160_h2 builds: LN(sum of the five slabs) . W^T + b for a ragged row count, the power-of-two ROW SCALE (rows of very different
    magnitude, incl. values far outside f16's range: the plane image must stay inside [0.5, 1)), DMA applied at issue and at the wait; also
    without LayerNorm (layer 0: the embedding output as it is)."""
    import numpy as np
    from interdiff_amd.mdm import pack_linear160_h2
    from tests.ffn_emulator import emulate_ln_linear_h2
    rs = np.random.RandomState(1)
    M, N = 37, 768
    slabs = rs.standard_normal((5, M, 256)).astype(np.float32)
    w = (rs.standard_normal((N, 256)) / 16).astype(np.float32)
    b, lnw, lnb = (rs.standard_normal(n).astype(np.float32) for n in (N, 256, 256))
    pack = pack_linear160_h2(w)
    assert pack.dtype == np.float32 and pack.size == 5 * 160 * 256                       # 768 rows in 5 slices of 160, the tail zero: the fp32 stream's size
    x = slabs.astype(np.float64).sum(0)
    xn = (x - x.mean(1, keepdims=True)) / np.sqrt(x.var(1, keepdims=True) + 1e-5) * lnw + lnb
    ref = xn @ w.T.astype(np.float64) + b
    for late in (False, True):
        got = emulate_ln_linear_h2(slabs, lnw, lnb, pack, b, N, late)
        assert np.abs(got - ref).max() / np.abs(ref).max() < 3e-7, np.abs(got - ref).max() / np.abs(ref).max()
    raw = slabs[:1] * (10.0 ** rs.uniform(-6, 7, (1, M, 1))).astype(np.float32)          # rows from 1e-6 to 1e7: no LayerNorm, row scale only
    ref0 = raw[0].astype(np.float64) @ w.T.astype(np.float64) + b
    got0 = emulate_ln_linear_h2(raw, None, None, pack, b, N, True)
    assert (np.abs(got0 - ref0).max(axis=1) / np.abs(ref0).max(axis=1)).max() < 3e-7


def test_split_f16_row_block_query_fragments_and_range_proof():
    """The split-f16 row block (csrc/denoiser.hip rowblock_kernel<.., H2>) reads the learned queries as 16-byte plane fragments: wave w (K quarter),
    K step s, tap j, lane (kq, li < NQ) loads word offset ((((2 w + s) 3 + j) 2 + plane) 4 + kq) NQ + li of mdm.py qan_fragments_h2 and contracts its 8
    halves with k = 64 w + 32 s + 8 kq .. + 7 of the token row.  Restated here: the three-product sum over those fragments equals the fp32
    contraction to 2^-21, and the LayerNorm range proof accepts ordinary weights and refuses ones that could leave the f16 range."""
    import numpy as np
    from interdiff_amd.mdm import qan_fragments_h2, split_f16, ln_h2_range_ok, NQ
    rs = np.random.RandomState(5)
    qc = (rs.standard_normal((NQ, 3, 256)) * 0.05).astype(np.float32)
    qc[0, 0, :4] = [1e-6, -3e-5, 6.2e-5, 0.0]                           # below the hi plane's flush threshold
    frag = qan_fragments_h2(qc)
    assert frag.dtype == np.float32 and frag.size == 16 * 3 * 4 * NQ * 4
    halves = frag.view(np.float16).reshape(-1, 8)                        # 16-byte units
    x = rs.standard_normal((18, 256)).astype(np.float32)
    xh, xl = split_f16(x)
    got = np.zeros((16, 3, NQ), np.float64)
    for w in range(4):
        for s in range(2):
            for kq in range(4):
                k0 = 64 * w + 32 * s + 8 * kq
                for j in range(3):
                    for li in range(NQ):
                        u = ((((2 * w + s) * 3 + j) * 2 + 0) * 4 + kq) * NQ + li
                        bh, bl = halves[u].astype(np.float64), halves[u + 4 * NQ].astype(np.float64)        # the lo' plane sits 4 kq x NQ units behind
                        ah, al = xh[j:j + 16, k0:k0 + 8].astype(np.float64), xl[j:j + 16, k0:k0 + 8].astype(np.float64)
                        got[:, j, li] += ah @ bh + (ah @ bl + al @ bh) / 2048.0
    ref = np.einsum('tjk,njk->tjn', np.stack([x[j:j + 16].astype(np.float64) for j in range(3)], 1), qc.astype(np.float64))
    assert np.abs(got - ref).max() <= 2.0 ** -21 * np.abs(ref).max() + 1e-9
    one, zero = np.ones(256, np.float32), np.zeros(256, np.float32)
    assert ln_h2_range_ok((one, zero), (3 * one, one))
    assert not ln_h2_range_ok((one, zero), (4000 * one, zero))           # 16 x 4000 > 60000
    assert not ln_h2_range_ok((one * np.float32(np.nan), zero))


def test_split_f16_step_tail_fragments():
    """csrc/tail_h2.h reads W_out / W_in as 16-byte plane fragments: output tile nt, K step s, plane, lane (li, kq) at 16-byte unit
    ((nt KS + s) 2 + plane) 64 + 16 kq + li of mdm.py pack_tail_h2, contracted with k = 32 s + 8 kq .. + 7 of the token row.  Restated: the three-product sum
    over the fragments equals the fp32 contraction to 2^-21 for both matrices (K = 144 zero-padded to 160 for the embedding)."""
    import numpy as np
    from interdiff_amd.mdm import pack_tail_h2, split_f16
    rs = np.random.RandomState(9)
    w_out = (rs.standard_normal((144, 256)) * 0.06).astype(np.float32)
    w_in = (rs.standard_normal((256, 144)) * 0.08).astype(np.float32)
    oh2, ih2 = pack_tail_h2(w_out, w_in)
    assert oh2.size == 9 * 8 * 2 * 64 * 4 and ih2.size == 16 * 5 * 2 * 64 * 4
    for frag, w, ntile, ks, K in ((oh2, w_out, 9, 8, 256), (ih2, w_in, 16, 5, 144)):
        halves = frag.view(np.float16).reshape(-1, 8)
        x = rs.standard_normal((16, 32 * ks)).astype(np.float32)
        x[:, K:] = 0
        xh, xl = split_f16(x, flush=False)
        got = np.zeros((16, 16 * ntile), np.float64)
        for nt in range(ntile):
            for s in range(ks):
                for kq in range(4):
                    k0 = 32 * s + 8 * kq
                    ah, al = xh[:, k0:k0 + 8].astype(np.float64), xl[:, k0:k0 + 8].astype(np.float64)
                    for li in range(16):
                        u = ((nt * ks + s) * 2) * 64 + 16 * kq + li
                        bh, bl = halves[u].astype(np.float64), halves[u + 64].astype(np.float64)
                        got[:, 16 * nt + li] += ah @ bh + (ah @ bl + al @ bh) / 2048.0
        ref = x[:, :K].astype(np.float64) @ w.astype(np.float64).T
        assert np.abs(got - ref).max() <= 2.0 ** -21 * np.abs(ref).max() + 1e-9


def test_scan_order_is_a_consistent_relabelling():
    """Host side of the exact block culling (interdiff_amd/geometry.py MeshTopology): vorder is a permutation, faces_scan / marker
    positions are the same mesh relabelled, and 16 consecutive scan positions of the rest pose are spatially compact."""
    import numpy as np
    import torch
    from interdiff_amd import synthetic as syn
    from interdiff_amd.geometry import MeshTopology, morton_order
    m = syn.smplh_model(seed=7)
    V = m['v_template'].shape[0]
    topo = MeshTopology(torch.from_numpy(m['faces']), V, device='cpu', rest_vertices=m['v_template'])
    order = topo.vorder.numpy().astype(np.int64)
    assert sorted(order.tolist()) == list(range(V))
    assert np.array_equal(order[topo.faces_scan.numpy()], m['faces'])
    # adjacency as (a, b) pairs: (v, a, b) is the incident face, rotated so that v comes first (orientation kept)
    ptr, fa = topo.adj_ptr.numpy(), topo.adj_face.numpy()
    pair = order[topo.adj_pair_scan.numpy()]
    for v in (0, 5, 1234, V - 1):
        for e in range(ptr[v], ptr[v + 1]):
            f = m['faces'][fa[e]].tolist()
            assert [v, pair[e, 0], pair[e, 1]] in (f, f[1:] + f[:1], f[2:] + f[:2]), (v, e)
    ids = [0, 17, 6889, 3470]
    assert np.array_equal(order[topo.scan_positions(ids, 'cpu').numpy()], ids)
    assert MeshTopology(torch.from_numpy(m['faces']), V, device='cpu').vorder is None
    vt = m['v_template'][order]
    nb = V // 16
    ext = np.linalg.norm(vt[:nb * 16].reshape(nb, 16, 3).max(1) - vt[:nb * 16].reshape(nb, 16, 3).min(1), axis=1)
    whole = np.linalg.norm(vt.max(0) - vt.min(0))
    assert np.median(ext) < 0.1 * whole, (np.median(ext), whole)
    pts = np.array([[0., 0, 0], [1, 1, 1], [0, 0, 0], [0.5, 0.5, 0.5]])
    assert morton_order(pts).tolist() == [0, 2, 3, 1]               # ties keep the lower index first


def test_feed_forward_tile_is_picked_from_the_batch_rows():
    """MDM._pick_ffn_tile (host logic, no GPU): the fused feed-forward block has 16-, 32- and 64-row kernels; the 32-row one agrees with
    the others to rounding only, so the tile handed to the library (tune[IDF_TUNE_FFN]: 2 / 1 / 3) follows the rows of the WHOLE batch
    -- the same rule as csrc/ffn.h ffn_tile_for_rows -- and an explicit ``ffn_rows`` wins (A/B runs)."""
    from types import SimpleNamespace
    from interdiff_amd import _lib
    from interdiff_amd.mdm import MDM
    k = _lib.TUNE['ffn']
    m = SimpleNamespace(w=_lib.MdmWeights(), ffn_rows=0, ffn_tile_for_rows=MDM.ffn_tile_for_rows, FFN16_MAX_ROWS=MDM.FFN16_MAX_ROWS)
    want = {1: 16, 800: 16, 801: 32, 1600: 32, 2400: 32, 2799: 32, 2800: 64, 3200: 64, 3264: 64, 3265: 64, 4800: 64, 6400: 64, 12800: 64}
    for rows, tile in want.items():
        assert MDM.ffn_tile_for_rows(rows) == tile, (rows, MDM.ffn_tile_for_rows(rows))
        MDM._pick_ffn_tile(m, rows)
        assert m.w.tune[k] == {16: 2, 32: 1, 64: 3}[tile], (rows, m.w.tune[k])
    # a shard / chain of a batch: the ROUNDING CLASS (32 vs 16 = 64) is the whole batch's, the tile inside the 16 / 64 class its own
    for whole, own, tile in ((6400, 800, 16), (6400, 1600, 64), (6400, 3200, 64), (1600, 800, 32), (1600, 400, 32), (800, 400, 16), (3200, 1600, 64), (2400, 1200, 32)):
        MDM._pick_ffn_tile(m, whole, own)
        assert m.w.tune[k] == {16: 2, 32: 1, 64: 3}[tile], (whole, own, m.w.tune[k])
        assert MDM.ffn_class_for_rows(whole) == (32 if tile == 32 else 16)
    for forced, code in ((32, 1), (16, 2), (64, 3)):
        m.ffn_rows = forced
        MDM._pick_ffn_tile(m, 100)
        assert m.w.tune[k] == code
    assert [m.w.tune[i] for i in range(8) if i != k] == [0] * 7          # nothing else is touched (ffn_math: 'exact' without the attribute)
    m.ffn_math = 'split'
    MDM._pick_ffn_tile(m, 100)
    assert m.w.tune[_lib.TUNE['ffn_math']] == 2                        # split-f16 feed-forward + QKV, exact row block (no rowblock_math attribute)
    m.rowblock_math = 'split'
    MDM._pick_ffn_tile(m, 100)
    assert m.w.tune[_lib.TUNE['ffn_math']] == 1                        # split-f16 everywhere
    src = open(os.path.join(ROOT, 'interdiff_amd', 'csrc', 'ffn.h')).read()
    assert 'constexpr int FFN16_MAX_ROWS = %d, FFN64_MIN_ROWS = %d;' % (MDM.FFN16_MAX_ROWS, MDM.FFN64_MIN_ROWS) in src


def test_io_lightning_checkpoint_reproduces_the_committed_fixture(tmp_path):
    """interdiff_amd.io.load_lightning_state_dict on the reference's REAL checkpoints/correction.ckpt (a pytorch-lightning 1.7 file,
    ``LitInteraction.load_from_checkpoint``: eval_smpl_short.py:425-426, train_correction_smpl.py:24-40) must give, array for array and
    byte for byte, what tests/golden/correction_ckpt.npz holds (the fixture every ObjProjector test and bench.py load).  Skipped where
    the reference tree is absent (the GPU box); a synthetic ``.ckpt`` of the same layout is always checked."""
    from interdiff_amd import io as iio
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'correction_ckpt.npz'))
    fake = str(tmp_path / 'fake.ckpt')
    torch.save({'state_dict': {'model.' + k: torch.from_numpy(z[k]) for k in z.files[:5]} | {'other.w': torch.zeros(2)}, 'hyper_parameters': {'x': 1},
                'pytorch-lightning_version': '1.7.1'}, fake)
    sd = iio.load_lightning_state_dict(fake)
    assert sorted(sd) == sorted(z.files[:5]) and all(np.array_equal(sd[k].numpy(), z[k]) for k in sd)
    with pytest.raises(ValueError):
        iio.load_lightning_state_dict(fake, prefix='nothing.')
    iio.state_dict_to_npz(sd, str(tmp_path / 'sd.npz'))
    back = iio.load_state_dict_npz(str(tmp_path / 'sd.npz'))
    assert all(torch.equal(back[k], sd[k]) for k in sd)
    # a checkpoint that smuggles an arbitrary object is refused (weights_only=True) unless the caller declares the file trusted
    class Payload:
        def __reduce__(self):
            return (list, ((1, 2),))
    evil = str(tmp_path / 'evil.ckpt')
    torch.save({'state_dict': {'model.w': torch.zeros(2)}, 'callbacks': Payload()}, evil)
    with pytest.raises(ValueError, match='trust_pickle'):
        iio.load_lightning_state_dict(evil)
    assert list(iio.load_lightning_state_dict(evil, trust_pickle=True)) == ['w']
    import argparse
    ns = str(tmp_path / 'ns.ckpt')                    # what lightning really writes next to the weights: an argparse.Namespace -- allow-listed
    torch.save({'state_dict': {'model.w': torch.ones(2)}, 'hyper_parameters': {'args': argparse.Namespace(dct=10)}}, ns)
    assert torch.equal(iio.load_lightning_state_dict(ns)['w'], torch.ones(2))
    real = '/root/reference/interdiff/checkpoints/correction.ckpt'
    if not os.path.exists(real):
        pytest.skip('reference checkpoint not on this box')
    sd = iio.load_lightning_state_dict(real)
    assert sorted(sd) == sorted(z.files) and len(sd) == 196
    for k in z.files:
        a = sd[k].numpy()
        assert a.dtype == z[k].dtype and a.shape == z[k].shape and a.tobytes() == z[k].tobytes(), k
    iio.state_dict_to_npz(sd, str(tmp_path / 'real.npz'))
    z2 = np.load(str(tmp_path / 'real.npz'))
    assert all(z2[k].tobytes() == z[k].tobytes() for k in z.files)


def test_io_smplh_npz_and_dataset_batch_adaptor(tmp_path):
    """``load_smplh_npz``: the seven SMPL_Layer buffers (smpl_layer.py:47-69) from an .npz, in this package's naming and in the official
    SMPL+H release's (``f``, ``kintree_table`` [2,J], 16 shape components, flat posedirs).  ``batch_from_dataset``: the DataLoader
    batch of data/dataset_smpl.py:182-204 (dict of per-frame lists) -> the stacks the reference builds itself
    (model/diffusion_smpl.py:197-201, eval_smpl_short.py:145-150)."""
    from interdiff_amd import io as iio
    from interdiff_amd import synthetic as syn
    m = syn.smplh_model(7)
    p1 = str(tmp_path / 'ours.npz')
    iio.save_smplh_npz(p1, m)
    a = iio.load_smplh_npz(p1)
    for k in iio.SMPLH_KEYS:
        assert np.array_equal(np.asarray(a[k]), np.asarray(m[k]).reshape(a[k].shape)), k
    V, J = a['v_template'].shape[0], a['weights'].shape[1]
    rs = np.random.RandomState(0)
    sd16 = np.concatenate([np.asarray(m['shapedirs'], np.float32), rs.standard_normal((V, 3, 6)).astype(np.float32)], axis=2)
    par = np.asarray(m['parents']).astype(np.int64)
    kt = np.stack([np.where(par < 0, 4294967295, par), np.arange(J)]).astype(np.uint32)
    p2 = str(tmp_path / 'official.npz')
    np.savez(p2, v_template=m['v_template'], shapedirs=sd16, posedirs=np.asarray(m['posedirs']).reshape(V * 3, -1), J_regressor=m['J_regressor'],
             weights=m['weights'], kintree_table=kt, f=np.asarray(m['faces']).astype(np.uint32))
    b = iio.load_smplh_npz(p2)
    for k in iio.SMPLH_KEYS:
        assert np.array_equal(b[k], a[k]), k
    with pytest.raises(ValueError):
        np.savez(str(tmp_path / 'bad.npz'), **{**{k: np.asarray(m[k]) for k in iio.SMPLH_KEYS}, 'weights': np.asarray(m['weights'])[:, :40]})
        iio.load_smplh_npz(str(tmp_path / 'bad.npz'))
    # the official release pickles J_regressor (a scipy sparse matrix): that one entry needs trust_pickle=True, nothing else is ever unpickled
    import scipy.sparse as sp
    jr_obj = np.empty((), dtype=object)
    jr_obj[()] = sp.csc_matrix(np.asarray(m['J_regressor']))
    p3 = str(tmp_path / 'official_sparse.npz')
    np.savez(p3, v_template=m['v_template'], shapedirs=sd16, posedirs=np.asarray(m['posedirs']).reshape(V * 3, -1), J_regressor=jr_obj,
             weights=m['weights'], kintree_table=kt, f=np.asarray(m['faces']).astype(np.uint32), extra_object=np.array({'a': 1}, dtype=object))
    with pytest.raises(ValueError, match='trust_pickle'):
        iio.load_smplh_npz(p3)
    c = iio.load_smplh_npz(p3, trust_pickle=True)
    assert np.array_equal(c['J_regressor'], a['J_regressor'])
    p4 = str(tmp_path / 'object_weights.npz')
    np.savez(p4, **{**{k: np.asarray(m[k]) for k in iio.SMPLH_KEYS}, 'weights': np.array({'w': 1}, dtype=object)})
    with pytest.raises(ValueError, match='pickled object'):
        iio.load_smplh_npz(p4, trust_pickle=True)
    # dataset batch: T records of collated tensors
    T, B, P = 5, 3, 7
    g = torch.Generator().manual_seed(0)
    frames = [dict(smplfit_params=dict(pose=torch.randn(B, 156, generator=g).double(), betas=torch.randn(B, 10, generator=g), trans=torch.randn(B, 3, generator=g)),
                   objfit_params=dict(angle=torch.randn(B, 3, generator=g).double(), trans=torch.randn(B, 3, generator=g).double()),
                   pelvis=torch.zeros(B, 3)) for _ in range(T)]
    batch = dict(frames=frames, obj_points=torch.randn(B, P, 6, generator=g).double(), gender=['male'] * B)
    assert iio.is_dataset_batch(batch) and not iio.is_dataset_batch(dict(gt=0))
    raw = iio.batch_from_dataset(batch)
    pose = torch.cat([f['smplfit_params']['pose'].unsqueeze(0) for f in frames], dim=0).float()           # the reference's own expressions
    assert torch.equal(raw['body_pose'], pose[..., :66]) and torch.equal(raw['hand_pose'], pose[..., 66:])
    assert torch.equal(raw['body_trans'], torch.cat([f['smplfit_params']['trans'].unsqueeze(0) for f in frames], dim=0).float())
    assert torch.equal(raw['obj_angles'], torch.cat([f['objfit_params']['angle'].unsqueeze(0) for f in frames], dim=0).float())
    assert torch.equal(raw['obj_trans'], torch.cat([f['objfit_params']['trans'].unsqueeze(0) for f in frames], dim=0).float())
    assert torch.equal(raw['beta'], torch.stack([f['smplfit_params']['betas'] for f in frames], dim=0))
    assert torch.equal(raw['obj_points'], batch['obj_points'][:, :, :3].float())
    assert all(v.dtype == torch.float32 and v.is_contiguous() for v in raw.values())
    assert raw['body_pose'].shape == (T, B, 66) and raw['beta'].shape == (T, B, 10) and raw['obj_points'].shape == (B, P, 3)


def test_documents_quote_the_measurement_files():
    """DESIGN.md / BASELINE.md / README.md carry their measured tables between GENERATED markers, rendered by tools/render_tables.py from profiles/r06_bench.json,
    parity_r06.json and r06_kernel_stats_bench.txt: the committed documents must be exactly what the committed measurement files render to (no hand-typed numbers)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'render_tables.py'), '--check'], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    for doc in ('DESIGN.md', 'BASELINE.md', 'README.md'):
        txt = open(os.path.join(ROOT, doc)).read()
        assert txt.count('<!-- BEGIN GENERATED') == txt.count('<!-- END GENERATED') >= 2, doc


def test_transposing_lds_read_address_map_of_the_attention_kernel():
    """csrc/attn_h2.h reads the P.V operand out of ROW-MAJOR V planes with gfx950's ds_read_b64_tr_b16.  What the instruction returns was measured
    (tools/tr_read_probe.hip): inside a 16-lane group, lane c names four contiguous halves (its "chunk"), and lane i receives element i & 3 of the chunks
    (i >> 2) + 4 j, j = 0..3.  This test restates that mapping in numpy and checks that the kernel's address formula -- lane (li, kq) of K step s and head-dim tile dt
    names the chunk at key 32 s + 8 kq + 4 h + (li >> 2), dims 16 dt + 4 (li & 3) ..+3, h = 0, 1 -- hands every lane the MFMA B fragment it must hold:
    V[32 s + 8 kq + 0..7][16 dt + li]."""
    VRS, T = 72, 128
    rng = np.random.RandomState(0)
    V = rng.randint(0, 1 << 15, size=(T, 64)).astype(np.int64)
    lds = np.zeros(T * VRS, np.int64)
    for k in range(T):
        lds[k * VRS:k * VRS + 64] = V[k]

    def tr_read(addr):                      # addr[16]: element index each lane of the group names -> out[16][4]
        chunks = np.stack([lds[a:a + 4] for a in addr])                  # [lane c][4 halves]
        return np.array([[chunks[(i >> 2) + 4 * j][i & 3] for j in range(4)] for i in range(16)])
    for s in range(T // 32):
        for dt in range(4):
            for kq in range(4):
                ko = 32 * s + 8 * kq
                frag = np.zeros((16, 8), np.int64)
                for h in range(2):
                    addr = [(ko + 4 * h + (li >> 2)) * VRS + dt * 16 + 4 * (li & 3) for li in range(16)]
                    assert all(a % 4 == 0 for a in addr)                 # 8-byte aligned: a misaligned transposing read returns the aligned address's data
                    frag[:, 4 * h:4 * h + 4] = tr_read(addr)
                want = np.stack([V[ko:ko + 8, dt * 16 + li] for li in range(16)])
                assert np.array_equal(frag, want), (s, dt, kq)


def test_qkv_plane_bounds_hold_and_keep_the_split_in_range():
    """The QKV kernel writes the self-attention's f16 planes under a per-row power of two derived WITHOUT looking at the output (csrc/ffn_h2.h ln_linear_h2_kernel<.., PLANES>):
    every input row is divided by 2^e (|x'| < 1), so |out_c| <= 2^e ||W_c||_1 + |b_c|; mdm.py qkv_bounds supplies max ||W_c||_1 and max |b_c| per q / k / v block.  Restated
    here in numpy on rows of very different magnitude: the bound holds for every output, the scaled outputs stay below 2^15 (f16 range), and the split (hi + lo' 2^-11) of the
    scaled outputs reproduces them to 2^-21 of the row's bound."""
    from interdiff_amd.mdm import qkv_bounds, split_f16
    rng = np.random.RandomState(5)
    W = (rng.randn(768, 256) * 0.08).astype(np.float32)
    b = (rng.randn(768) * 0.3).astype(np.float32)
    bnd = qkv_bounds(W, b)
    assert bnd.shape == (8,) and bnd.dtype == np.float32
    x = (rng.randn(40, 256) * np.exp(rng.uniform(-12, 12, size=(40, 1)))).astype(np.float32)
    x[7] = 0.0
    out = x.astype(np.float64) @ W.astype(np.float64).T + b
    for r in range(x.shape[0]):
        amax = np.abs(x[r]).max()
        e = int(np.frexp(amax)[1]) if amax > 0 else 0              # |x| 2^-e in [0.5, 1)
        for t in range(3):
            bound = np.float32(np.float32(2.0 ** e) * bnd[t] + bnd[3 + t])
            o = out[r, 256 * t:256 * (t + 1)]
            assert np.abs(o).max() <= bound, (r, t)
            E = int(np.frexp(max(float(bound), 1e-30))[1])           # bound < 2^E (frexp: bound = m 2^E, m in [0.5, 1))
            scaled = (o * 2.0 ** (15 - E)).astype(np.float32)
            assert np.abs(scaled).max() < 2.0 ** 15
            hi, lo = split_f16(scaled)
            back = hi.astype(np.float64) + lo.astype(np.float64) * 2.0 ** -11
            assert np.abs(back - scaled).max() <= 2.0 ** 15 * 2.0 ** -21
    # the per-TYPE bound is only taken when no (type, head) group sits far below it (mdm.py qkv_bounds_usable): one outlier head -> the layer keeps fp32 rows (qkv_bounds_ok = 0)
    from interdiff_amd.mdm import qkv_bounds_usable
    assert qkv_bounds_usable(W, b, bnd)
    W2 = W.copy()
    W2[256 + 64 * 2:256 + 64 * 3] *= 200.0                          # head 2 of the key block: every other head's k planes would be scaled ~200x too low
    assert not qkv_bounds_usable(W2, b, qkv_bounds(W2, b))
    W3 = W.copy()
    W3[5] *= 8.0                                                   # one outlier row inside a head: 8x is within the slack that is tolerated (3 of 22 bits)
    assert qkv_bounds_usable(W3, b, qkv_bounds(W3, b))
    assert not qkv_bounds_usable(W * np.float32(1e25), b, qkv_bounds(W * np.float32(1e25), b))      # far from overflow, as before


def test_marker_contact_threshold_constant_is_the_exact_one():
    """csrc/correction.hip tests d2 < MARK_D2 instead of sqrtf(d2) < 0.02f (eval_smpl_short.py:110-112) -- valid because a correctly rounded square root is monotone:
    MARK_D2 must be THE smallest float whose root is >= 0.02f.  Re-derived here from the definition and compared with the literal in the source; a neighbourhood of
    floats around it is checked element by element."""
    import re
    src = open(os.path.join(os.path.dirname(__file__), '..', 'interdiff_amd', 'csrc', 'correction.hip')).read()
    lit = np.float32(float(re.search(r'constexpr float MARK_D2 = ([0-9.eE+-]+)f;', src).group(1)))
    t = np.float32(0.02)
    x = np.float32(t * t)
    while np.sqrt(x, dtype=np.float32) >= t:
        x = np.nextafter(x, np.float32(0), dtype=np.float32)
    theta = np.nextafter(x, np.float32(1), dtype=np.float32)
    assert lit.view(np.uint32) == theta.view(np.uint32) == 0x39d1b716
    bits = np.arange(int(theta.view(np.uint32)) - 2000, int(theta.view(np.uint32)) + 2000, dtype=np.uint32)
    d2 = bits.view(np.float32)
    assert np.array_equal(np.sqrt(d2, dtype=np.float32) < t, d2 < theta)
    assert np.array_equal(np.sqrt(d2.astype(np.float64)).astype(np.float32) < t, d2 < theta)      # (a double root rounded to float is the correctly rounded float root)
