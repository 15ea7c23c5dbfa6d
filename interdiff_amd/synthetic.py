"""Deterministic synthetic stand-ins for the assets the reference cannot ship
(``diffusion.ckpt`` missing, SMPL-H ``.pkl`` licensed) and for
BEHAVE-shaped clips.

numpy-only (legacy ``RandomState`` => bit-identical across machines), so that
the build container (golden generation against the reference), the tests and
``bench.py`` on the GPU box all see exactly the same weights and inputs.
Nothing here is on the hot path.
"""
import numpy as np

# SMPL-H kinematic tree (public model metadata; reference reads it from
# kintree_table[0], smpl_layer.py:67-69).  parents[0] is the root marker.
SMPLH_PARENTS = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19,
                 20, 22, 23, 20, 25, 26, 20, 28, 29, 20, 31, 32, 20, 34, 35,
                 21, 37, 38, 21, 40, 41, 21, 43, 44, 21, 46, 47, 21, 49, 50]

N_LAYERS, QAN_LAYERS = 8, (1, 2, 3, 4, 5, 6)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ----------------------------------------------------------------------------
# denoiser weights (key names + shapes of the reference MDM, decoder path only:
# model/diffusion_smpl.py:13-17,73-120,177-179)
# ----------------------------------------------------------------------------
def mdm_state_dict(seed=233, d=256, ff=1024, n_body=135, n_obj=9, n_queries=10):
    rs = np.random.RandomState(seed)
    sd = {}

    def linear(name, out_f, in_f):
        b = 1.0 / np.sqrt(in_f)
        sd[name + '.weight'] = _f32(rs.uniform(-b, b, (out_f, in_f)))
        sd[name + '.bias'] = _f32(rs.uniform(-b, b, (out_f,)))

    def norm(name):
        sd[name + '.weight'] = _f32(1.0 + 0.1 * rs.standard_normal(d))
        sd[name + '.bias'] = _f32(0.1 * rs.standard_normal(d))

    def mha(name):
        b = np.sqrt(6.0 / (d + 3 * d))
        sd[name + '.in_proj_weight'] = _f32(rs.uniform(-b, b, (3 * d, d)))
        sd[name + '.in_proj_bias'] = _f32(0.02 * rs.standard_normal(3 * d))
        linear(name + '.out_proj', d, d)

    linear('bodyEmbedding', d, n_body)
    linear('objEmbedding', d, n_obj)
    linear('embedTimeStep.time_embed.0', d, d)
    linear('embedTimeStep.time_embed.2', d, d)
    for l in range(N_LAYERS):
        p = 'decoder.layers.%d' % l
        if l in QAN_LAYERS:
            sd[p + '.queries'] = _f32(rs.normal(-1.0 / np.sqrt(d), 1.0 / np.sqrt(d), (n_queries, d)))
            sd[p + '.wk'] = _f32(rs.normal(-1.0 / np.sqrt(n_queries), 1.0 / np.sqrt(n_queries), (n_queries, 1)))
        else:
            mha(p + '.self_attn')
        mha(p + '.multihead_attn')
        linear(p + '.linear1', ff, d)
        linear(p + '.linear2', d, ff)
        for k in (1, 2, 3):
            norm(p + '.norm%d' % k)
    linear('bodyFinalLinear', n_body, d)
    linear('objFinalLinear', n_obj, d)
    # ---- encoder side ("next" row N1; drawn AFTER everything above so the decoder weights keep their values):
    # encoder.layers.{0..7} (std: self_attn, QaN: queries/wk; linear1/2, norm1/2) and pcEmbedding = PointNet2Encoder
    # (model/layers.py:111-140; pointnet2_ops 3.0.0 build_shared_mlp naming: mlps.<scale>.{0,3,6} conv, {1,4,7} BatchNorm)
    for l in range(N_LAYERS):
        p = 'encoder.layers.%d' % l
        if l in QAN_LAYERS:
            sd[p + '.queries'] = _f32(rs.normal(-1.0 / np.sqrt(d), 1.0 / np.sqrt(d), (n_queries, d)))
            sd[p + '.wk'] = _f32(rs.normal(-1.0 / np.sqrt(n_queries), 1.0 / np.sqrt(n_queries), (n_queries, 1)))
        else:
            mha(p + '.self_attn')
        linear(p + '.linear1', ff, d)
        linear(p + '.linear2', d, ff)
        for k in (1, 2):
            norm(p + '.norm%d' % k)
    for s_, specs in enumerate(PC_MLPS):
        for m_, spec in enumerate(specs):
            for l_ in range(3):
                cin, cout = spec[l_], spec[l_ + 1]
                q = 'pcEmbedding.SA_modules.%d.mlps.%d' % (s_, m_)
                sd['%s.%d.weight' % (q, 3 * l_)] = _f32(rs.standard_normal((cout, cin, 1, 1)) * np.sqrt(2.0 / cin))
                sd['%s.%d.weight' % (q, 3 * l_ + 1)] = _f32(1.0 + 0.1 * rs.standard_normal(cout))
                sd['%s.%d.bias' % (q, 3 * l_ + 1)] = _f32(0.1 * rs.standard_normal(cout))
                sd['%s.%d.running_mean' % (q, 3 * l_ + 1)] = _f32(0.1 * rs.standard_normal(cout))
                sd['%s.%d.running_var' % (q, 3 * l_ + 1)] = _f32(rs.uniform(0.5, 1.5, cout))
    linear('pcEmbedding.Linear', d - 3, 256)
    return sd


# shared-MLP channel plans of PointNet2Encoder(c_in=1, c_out=256, num_keypoints=1): +3 = use_xyz (model/layers.py:118-138)
PC_MLPS = (((1 + 3, 16, 16, 32), (1 + 3, 32, 32, 64)), ((96 + 3, 64, 64, 128), (96 + 3, 64, 96, 128)))


def make_embedding_inputs(seed=77, B=3, T=35, n_points=2048):
    """Inputs of MDM._get_embeddings in tensor form: axis-angle body pose [T,B,66], translations, object pose, and an
    object point cloud that lives on a ~0.4 m box surface-ish shell (so that the 5-20 cm ball queries find neighbours)."""
    rs = np.random.RandomState(seed)
    walk = lambda shape0, scale, step: scale * rs.standard_normal((1,) + shape0) + np.cumsum(step * rs.standard_normal((T,) + shape0), axis=0)
    pts = rs.uniform(-0.2, 0.2, (B, n_points, 3))
    ax = rs.randint(0, 3, size=(B, n_points))
    np.put_along_axis(pts, ax[..., None], np.sign(np.take_along_axis(pts, ax[..., None], axis=2)) * 0.2, axis=2)   # onto the faces of the box
    pts[:, 7] = 1e-3 * rs.standard_normal((B, 3))             # a point with |p|^2 <= 1e-3: furthest-point sampling must skip it
    return dict(body_pose=_f32(walk((B, 66), 0.3, 0.02)), body_trans=_f32(walk((B, 3), 0.1, 0.016)),
                obj_angles=_f32(walk((B, 3), 1.0, 0.03)), obj_trans=_f32(walk((B, 3), 0.3, 0.01)), obj_points=_f32(pts))


# ----------------------------------------------------------------------------
# SMPL-H shaped body model (buffers of smpl_layer.py:47-69)
# ----------------------------------------------------------------------------
def smplh_model(seed=7, V=6890, F=13776, n_betas=10, coherent=False):
    """SMPL-H shaped stand-in.  ``coherent=False`` (the model every golden fixture was recorded with) gives each vertex a fourth bone
    drawn at random from the whole skeleton, so neighbouring vertices can be dragged apart by unrelated joints; ``coherent=True``
    keeps all of a vertex's bones in the kinematic neighbourhood of its primary joint (parent, grand-parent, a child or sibling),
    which is how the real SMPL-H skinning weights look (smooth over the surface).  Both draw the same random numbers."""
    rs = np.random.RandomState(seed)
    parents = list(SMPLH_PARENTS)
    J = len(parents)
    # rest skeleton: child = parent + offset, hands get short bones
    rest = np.zeros((J, 3))
    for j in range(1, J):
        scale = 0.03 if j >= 22 else 0.18
        rest[j] = rest[parents[j]] + scale * rs.standard_normal(3)
    # vertices clustered around a primary joint
    prim = rs.randint(0, J, size=V)
    prim[:J] = np.arange(J)                                    # every joint owns >= 1 vertex
    v_template = rest[prim] + 0.04 * rs.standard_normal((V, 3))
    # skinning weights: primary + up to 3 relatives (parent, grand-parent, random)
    weights = np.zeros((V, J))
    par = np.array([max(p, 0) for p in parents])
    third = rs.randint(0, J, size=V)
    if coherent:
        kids = [[c for c in range(1, J) if parents[c] == j] for j in range(J)]
        near = [kids[j] + [c for c in kids[par[j]] if c != j] + [int(par[j])] for j in range(J)]     # children, then siblings, then the parent
        third = np.array([near[prim[v]][third[v] % len(near[prim[v]])] for v in range(V)])
    others = np.stack([par[prim], par[par[prim]], third], axis=1)
    w = np.concatenate([rs.uniform(0.5, 1.0, (V, 1)), rs.uniform(0.0, 0.3, (V, 3))], axis=1)
    w[rs.uniform(size=(V, 4)) < 0.15] = 0.0                     # some vertices have < 4 bones
    w[:, 0] = np.maximum(w[:, 0], 0.2)
    for c, col in enumerate([prim, others[:, 0], others[:, 1], others[:, 2]]):
        np.add.at(weights, (np.arange(V), col), w[:, c])
    weights /= weights.sum(1, keepdims=True)
    # joint regressor: each joint from 32 vertices of its own cluster (falls back to random)
    Jreg = np.zeros((J, V))
    for j in range(J):
        own = np.nonzero(prim == j)[0]
        pick = own[:32] if len(own) >= 8 else rs.randint(0, V, size=32)
        Jreg[j, pick] = rs.uniform(0.5, 1.5, size=len(pick))
    Jreg /= Jreg.sum(1, keepdims=True)
    # random but valid (non-degenerate index) triangles touching every vertex
    faces = np.stack([rs.permutation(F) % V, rs.randint(0, V, F), rs.randint(0, V, F)], axis=1)
    bad = (faces[:, 0] == faces[:, 1]) | (faces[:, 1] == faces[:, 2]) | (faces[:, 0] == faces[:, 2])
    faces[bad] = np.stack([faces[bad, 0], (faces[bad, 0] + 1) % V, (faces[bad, 0] + 2) % V], axis=1)
    return dict(
        v_template=_f32(v_template),
        shapedirs=_f32(0.01 * rs.standard_normal((V, 3, n_betas))),
        posedirs=_f32(2e-3 * rs.standard_normal((V, 3, 9 * (J - 1)))),
        J_regressor=_f32(Jreg),
        weights=_f32(weights),
        faces=np.ascontiguousarray(faces, dtype=np.int64),
        parents=np.array(parents, dtype=np.int64),
    )


# ----------------------------------------------------------------------------
# clips
# ----------------------------------------------------------------------------
def aa_to_matrix(aa):
    """Rodrigues, float64 numpy; aa [...,3] -> [...,3,3] (host-side data prep only)."""
    th = np.linalg.norm(aa, axis=-1, keepdims=True)
    k = aa / np.maximum(th, 1e-12)
    K = np.zeros(aa.shape[:-1] + (3, 3))
    K[..., 0, 1], K[..., 0, 2] = -k[..., 2], k[..., 1]
    K[..., 1, 0], K[..., 1, 2] = k[..., 2], -k[..., 0]
    K[..., 2, 0], K[..., 2, 1] = -k[..., 1], k[..., 0]
    s, c = np.sin(th)[..., None], np.cos(th)[..., None]
    return np.eye(3) + s * K + (1 - c) * (K @ K)


def matrix_to_6d(m):
    return m[..., :2, :].reshape(m.shape[:-2] + (6,))


def make_clip_batch(seed=233, B=16, T=100, past_len=10, n_points=2048, d=256, n_mem=10):
    """BEHAVE-shaped synthetic batch; all float32 numpy.

    gt [B,1,144,T]  tokens: 22x rot6d | body trans | obj rot6d | obj trans
    cond [n_mem,B,d], hand_pose [T,B,90], beta [T,B,10] (constant over time),
    obj_points [B,n_points,3], noise [B,1,144,T]."""
    rs = np.random.RandomState(seed)

    def walk(shape0, scale, step):
        x0 = scale * rs.standard_normal((1,) + shape0)
        return x0 + np.cumsum(step * rs.standard_normal((T,) + shape0), axis=0)
    body_aa = walk((B, 22, 3), 0.3, 0.02)
    body_tr = walk((B, 3), 0.1, 0.016)
    obj_aa = walk((B, 3), 1.0, 0.03)
    off = 0.35 * rs.standard_normal((1, B, 3))
    obj_tr = body_tr + off + np.cumsum(0.01 * rs.standard_normal((T, B, 3)), axis=0)
    gt = np.concatenate([matrix_to_6d(aa_to_matrix(body_aa)).reshape(T, B, 132), body_tr,
                         matrix_to_6d(aa_to_matrix(obj_aa)), obj_tr], axis=2)         # [T,B,144]
    beta = np.repeat(rs.standard_normal((1, B, 10)), T, axis=0)
    return dict(
        gt=_f32(gt.transpose(1, 2, 0)[:, None]),
        cond=_f32(rs.standard_normal((n_mem, B, d))),
        hand_pose=_f32(0.1 * rs.standard_normal((T, B, 90))),
        beta=_f32(beta),
        obj_points=_f32(rs.uniform(-0.2, 0.2, (B, n_points, 3))),
        noise=_f32(rs.standard_normal((B, 1, 144, T))),
        past_len=past_len,
    )


def make_optim_batch(seed=1, B=16, T=20, n_points=2048):
    """Clips in optimization.py's schema ("next" row N4; :216 uses past 10 + future 10 frames): pose [B,T,156] axis-angle,
    trans / obj_angles / obj_trans [B,T,3], betas [B,T,10], obj_points [B,P,3] -- slow motion, the object overlapping the
    body so that the penetration term is active."""
    rs = np.random.RandomState(seed)
    pose = 0.3 * rs.standard_normal((B, 1, 156)) + np.cumsum(0.01 * rs.standard_normal((B, T, 156)), axis=1)
    trans = 0.1 * rs.standard_normal((B, 1, 3)) + np.cumsum(0.005 * rs.standard_normal((B, T, 3)), axis=1)
    obj_angles = rs.standard_normal((B, 1, 3)) + np.cumsum(0.02 * rs.standard_normal((B, T, 3)), axis=1)
    obj_trans = trans + 0.25 * rs.standard_normal((B, 1, 3)) + np.cumsum(0.004 * rs.standard_normal((B, T, 3)), axis=1)
    betas = np.repeat(rs.standard_normal((B, 1, 10)), T, axis=1)
    return dict(pose=_f32(pose), trans=_f32(trans), obj_angles=_f32(obj_angles), obj_trans=_f32(obj_trans), betas=_f32(betas),
                obj_points=_f32(rs.uniform(-0.25, 0.25, (B, n_points, 3))))
