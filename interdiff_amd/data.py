"""BEHAVE clip ETL ("next" row N2 of SURVEY.md §8(f)): on-disk sequence -> canonicalised clip batch in the tensor schema the
HIP path consumes (``eval.batch_from_raw``).

Restates data/dataset_smpl.py:44-56 (files: ``smpl_fit_all.npz`` {poses [F,156], betas [F,10], trans [F,3]},
``object_fit_all.npz`` {angles [F,3], trans [F,3], frame_times}), :93-100 (test windows: consecutive fragments of
past_len + future_len frames) and :105-204 (``__getitem__``: every clip is expressed in the frame of its first pose --
origin at the first pelvis, yaw of the first global orientation removed).  Host-side numpy/scipy like the reference (this is
dataset plumbing, not the hot path); the one heavy step, the per-frame pelvis = SMPL joint 0 over the whole sequence
(:57,68), runs on the GPU through ``SMPL_Layer``.  The contact-side records of a clip -- the object cloud of every frame with
normals and contact labels, foot-ground labels, human contact labels (:48-50,152-180; read by the reference's training code, never
by the sampler, the hook or the metrics) -- come from ``clip_labels`` when the sequence has a ``contact.npz``; the per-vertex
``human_verts`` records and rendering inputs are not produced."""
import os
import numpy as np
import torch
from scipy.spatial.transform import Rotation


def load_behave_sequence(seq_dir):
    """data/dataset_smpl.py:44-56: the two fit files, and -- when the sequence directory has them (the shipped sample does not) --
    ``contact.npz`` (object cloud [P,6] = xyz | normal, per-frame contact vertex lists of object and body, first-frame foot labels)
    and ``info.json`` (gender, object category)."""
    import json
    with np.load(os.path.join(seq_dir, 'object_fit_all.npz'), allow_pickle=True) as f:
        obj_angles, obj_trans = f['angles'], f['trans']
    with np.load(os.path.join(seq_dir, 'smpl_fit_all.npz'), allow_pickle=True) as f:
        poses, betas, trans = f['poses'], f['betas'], f['trans']
    n = min(len(poses), len(obj_angles))
    seq = dict(poses=poses[:n], betas=betas[:n], trans=trans[:n], obj_angles=obj_angles[:n], obj_trans=obj_trans[:n], seq_name=os.path.basename(seq_dir))
    cpath, ipath = os.path.join(seq_dir, 'contact.npz'), os.path.join(seq_dir, 'info.json')
    if os.path.isfile(cpath):
        with np.load(cpath, allow_pickle=True) as f:
            d = f['arr_0'].item()
        seq.update(obj_points=d['object_points'], obj_contact_label=d['object_contact_vertex_label'], contact_label=d['human_contact_vertex_label'],
                   ground_joint_label=d['foot_contact_joint_label'])
    if os.path.isfile(ipath):
        info = json.load(open(ipath))
        seq.update(gender=info['gender'], obj_name=info['cat'])
    return seq


def clip_labels(seq, clip, left_foot, right_foot, start, past_len, future_len, sample_rate=1, n_verts=6890):
    """The contact-side records of ``Dataset.__getitem__`` (data/dataset_smpl.py:152-180) for the frames of a canonicalised ``clip``:
    obj_points [T,P,7] = the object cloud in the clip's frame (R_t p + t_t | R_t n | 1 where the frame's object contact list names the
    point), ground_joint_label [T,2] (left / right foot moved < 1 cm since the PREVIOUS sequence frame; the sequence's very first frame
    takes the label stored in the file), contact_label [T,V] (body vertices in contact).  ``left_foot`` / ``right_foot`` [F,3]: joints
    10 / 11 over the sequence (like ``sequence_pelvis`` for joint 0)."""
    T = past_len + future_len
    idx = start + sample_rate * np.arange(T)
    pts = np.asarray(seq['obj_points'], np.float64)
    P = pts.shape[0]
    R = Rotation.from_rotvec(np.asarray(clip['obj_angles'], np.float64)).as_matrix()                       # [T,3,3] canonicalised object rotation
    out = np.zeros((T, P, 7))
    out[..., :3] = np.einsum('pc,tdc->tpd', pts[:, :3], R) + np.asarray(clip['obj_trans'], np.float64)[:, None, :]
    out[..., 3:6] = np.einsum('pc,tdc->tpd', pts[:, 3:6], R)
    contact = np.zeros((T, n_verts), bool)
    ground = np.zeros((T, 2))
    for k, i in enumerate(idx):
        out[k, np.asarray(seq['obj_contact_label'][i], np.int64), 6] = 1
        contact[k, np.asarray(seq['contact_label'][i], np.int64)] = True
        if i > 0:
            ground[k, 0] = np.linalg.norm(left_foot[i] - left_foot[i - 1]) < 0.01
            ground[k, 1] = np.linalg.norm(right_foot[i] - right_foot[i - 1]) < 0.01
        else:
            ground[k, int(seq['ground_joint_label'][i]) - 10] = 1
    return dict(obj_points=out, ground_joint_label=ground, contact_label=contact)


def load_ply_vertices(path):
    """Vertices of an ASCII or binary-little-endian PLY (e.g. objects/backpack/backpack_f1000.ply), centred like
    prepare_behave.py:89-94 centres the template (mean of the vertices)."""
    with open(path, 'rb') as f:
        header, fmt, nv, props = [], None, 0, []
        in_vertex = False
        while True:
            line = f.readline().decode('ascii', 'ignore').strip()
            header.append(line)
            if line.startswith('format'):
                fmt = line.split()[1]
            elif line.startswith('element'):
                in_vertex = line.split()[1] == 'vertex'
                if in_vertex:
                    nv = int(line.split()[2])
            elif line.startswith('property') and in_vertex:
                props.append((line.split()[1], line.split()[2]))
            elif line == 'end_header':
                break
        if fmt == 'ascii':
            rows = [f.readline().split() for _ in range(nv)]
            v = np.array([[float(r[0]), float(r[1]), float(r[2])] for r in rows], dtype=np.float64)
        else:
            np_t = {'float': '<f4', 'float32': '<f4', 'double': '<f8', 'float64': '<f8', 'uchar': 'u1', 'uint8': 'u1', 'int': '<i4', 'uint': '<u4',
                    'short': '<i2', 'ushort': '<u2', 'char': 'i1'}
            dt = np.dtype([(name, np_t[t]) for t, name in props])
            rec = np.frombuffer(f.read(nv * dt.itemsize), dtype=dt, count=nv)
            v = np.stack([rec['x'], rec['y'], rec['z']], axis=1).astype(np.float64)
    return v - v.mean(0)


def sample_points(vertices, n, seed=0):
    """n object points from the template vertices (the reference samples the SURFACE with trimesh at preparation time,
    prepare_behave.py:89-94,130 -- not reproducible here; vertices are repeated / sub-sampled deterministically instead)."""
    rs = np.random.RandomState(seed)
    idx = rs.permutation(len(vertices))[:n] if len(vertices) >= n else np.concatenate([np.arange(len(vertices)), rs.randint(0, len(vertices), n - len(vertices))])
    return vertices[idx].astype(np.float32)


def sequence_pelvis(seq, smpl, device='cuda', chunk=512):
    """Joint 0 of SMPL over every frame (data/dataset_smpl.py:57,68), on the GPU."""
    out = []
    for s in range(0, len(seq['poses']), chunk):
        pose = torch.from_numpy(np.asarray(seq['poses'][s:s + chunk], dtype=np.float32)).to(device)
        betas = torch.from_numpy(np.asarray(seq['betas'][s:s + chunk], dtype=np.float32)).to(device)
        trans = torch.from_numpy(np.asarray(seq['trans'][s:s + chunk], dtype=np.float32)).to(device)
        out.append(smpl(pose, th_betas=betas, th_trans=trans, want_v_posed=False)[1][:, 0].cpu().numpy())
    return np.float32(np.concatenate(out))


def test_windows(n_frames, past_len, future_len, sample_rate=1):
    """Start frames of the test split (data/dataset_smpl.py:93-96: one clip per fragment, offset 0)."""
    fragment = (past_len + future_len) * sample_rate
    return [i * fragment for i in range(n_frames // fragment)]


def canonicalize_clip(seq, pelvis, start, past_len, future_len, sample_rate=1):
    """The clip [start, start + T*rate) expressed in the frame of its first pose (data/dataset_smpl.py:105-160): origin at the first
    frame's pelvis, the first frame's heading (yaw of the global orientation) removed.  Whole clip at once: one gather per array, one
    batched rotation composition, one matrix product per translation.
    Returns dict(pose [T,156], trans [T,3], betas [T,10], obj_angles [T,3], obj_trans [T,3], pelvis [T,3], centroid, rotation)."""
    T = past_len + future_len
    idx = start + sample_rate * np.arange(T)
    pose, trans = np.array(seq['poses'][idx]), np.array(seq['trans'][idx])
    angle, otrans, pel = np.array(seq['obj_angles'][idx]), np.array(seq['obj_trans'][idx]), np.array(pelvis[idx])
    centroid = pel[0].copy()
    # heading of the first frame: the body's x axis projected on the ground plane (x, z) -> a rotation about y that undoes it
    x_axis = Rotation.from_rotvec(pose[0, :3]).as_matrix()[:, 0]
    c, s = x_axis[[0, 2]] / np.hypot(x_axis[0], x_axis[2])
    yaw = np.array([[c, 0, -s], [0, 1, 0], [s, 0, c]], dtype=np.float32)
    rotation = np.linalg.inv(yaw).astype(np.float32)
    unyaw = Rotation.from_matrix(rotation)
    # SMPL rotates about the pelvis of its own (unposed) frame, so the translation that moves the posed pelvis by `rotation` is
    # R (t + p0) - p0 with p0 = posed pelvis - t, all relative to the centroid
    t_rel, p_rel = trans - centroid, pel - centroid
    p0 = p_rel - t_rel
    pose[:, :3] = (unyaw * Rotation.from_rotvec(pose[:, :3])).as_rotvec()
    return dict(pose=pose, trans=(t_rel + p0) @ rotation.T - p0, betas=np.array(seq['betas'][idx]),
                obj_angles=(unyaw * Rotation.from_rotvec(angle)).as_rotvec(), obj_trans=(otrans - centroid) @ rotation.T,
                pelvis=p_rel @ rotation.T, centroid=centroid, rotation=rotation)


def collate_raw(clips, obj_points, device='cuda'):
    """List of canonicalised clips (same T) + object points [P,3] or [B,P,3] -> the ``raw`` dict of eval.batch_from_raw:
    body_pose [T,B,66], hand_pose [T,B,90], body_trans, obj_angles, obj_trans [T,B,3], beta [T,B,10], obj_points [B,P,3]."""
    st = lambda k: torch.from_numpy(np.stack([np.asarray(c[k], dtype=np.float32) for c in clips], axis=1)).to(device)
    pose = st('pose')
    pts = np.asarray(obj_points, dtype=np.float32)
    if pts.ndim == 2:
        pts = np.repeat(pts[None], len(clips), axis=0)
    return dict(body_pose=pose[..., :66].contiguous(), hand_pose=pose[..., 66:].contiguous(), body_trans=st('trans'), obj_angles=st('obj_angles'),
                obj_trans=st('obj_trans'), beta=st('betas'), obj_points=torch.from_numpy(pts).to(device))
