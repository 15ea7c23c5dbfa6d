"""File loaders and the dataset-batch adaptor: what turns the reference's own ASSETS and BATCHES into this package's inputs.

  * ``load_lightning_state_dict`` -- ``LitInteraction.load_from_checkpoint(path, args=args)`` (eval_smpl_short.py:425-426,
    train_diffusion_smpl.py:34-44, train_correction_smpl.py:24-40): a pytorch-lightning ``.ckpt`` is a pickled dict whose
    ``'state_dict'`` holds the LightningModule's parameters under the attribute prefix ``model.``; ``MDM(state_dict)`` /
    ``ObjProjector(state_dict)`` take the reference's key names WITHOUT that prefix.
  * ``load_smplh_npz`` / ``save_smplh_npz`` -- the seven buffers ``SMPL_Layer.__init__`` registers from the licensed ``.pkl``
    (libsmpl/smplpytorch/pytorch/smpl_layer.py:47-69), as an ``.npz``.  Accepts this package's own key names and
    the names of the official SMPL+H ``.npz`` release (``f``, ``kintree_table``, 16 shape components of which the layer uses 10).
  * ``batch_from_dataset`` -- the DataLoader batch of data/dataset_smpl.py:105-204 (a dict of per-frame lists) to the stacked
    tensors the reference itself builds at model/diffusion_smpl.py:195-208 and eval_smpl_short.py:145-150; ``eval.sample_once_proj``
    / ``sample_once`` / ``get_gt`` accept either form.
Host-side plumbing: numpy / torch only, no kernel."""
import numpy as np
import torch

SMPLH_KEYS = ('v_template', 'shapedirs', 'posedirs', 'J_regressor', 'weights', 'parents', 'faces')


def _torch_load_weights(path, map_location, trust_pickle):
    """torch.load restricted to tensors and plain containers (``weights_only=True``); a lightning checkpoint's ``hyper_parameters`` hold an
    ``argparse.Namespace``, which is allow-listed for this one call (it is data: attribute names -> values).  Anything else a file wants to
    unpickle is code execution by whoever wrote the file, and is refused unless the caller says the file is trusted."""
    import argparse
    import pickle
    import contextlib
    try:                                              # torch >= 2.5: a scoped allow-list; 2.4: a process-wide one; older: plain weights_only=True (a Namespace in the file is then refused)
        from torch.serialization import safe_globals
        allow = safe_globals([argparse.Namespace])
    except ImportError:
        allow = contextlib.nullcontext()
        add = getattr(torch.serialization, 'add_safe_globals', None)
        if add is not None:
            add([argparse.Namespace])
    try:
        with allow:
            return torch.load(path, map_location=map_location, weights_only=True)
    except (pickle.UnpicklingError, RuntimeError, AttributeError, TypeError) as e:      # what an unpickler refusing a global raises; OSError (no such file, ...) propagates as it is
        if not trust_pickle:
            raise ValueError('%s does not load with weights_only=True (%s: %s); pass trust_pickle=True only for a file whose origin you trust -- '
                             'a pickle runs code' % (path, type(e).__name__, str(e).splitlines()[0] if str(e) else '')) from e
        return torch.load(path, map_location=map_location, weights_only=False)


def load_lightning_state_dict(path, prefix='model.', map_location='cpu', trust_pickle=False):
    """``.ckpt`` (pytorch-lightning) or a plain ``torch.save``d state_dict -> {reference key name: tensor}.  Keys that do not start
    with ``prefix`` (optimizer-side entries of other attributes) are dropped; with ``prefix=''`` everything is kept.  Also returns
    nothing else: hyper-parameters travel on the command line in the reference (``args.dct`` is absent from correction.ckpt's own
    hparams and comes from the CLI default, eval_smpl_short.py:394).  The file is read with ``weights_only=True`` (tensors, containers,
    ``argparse.Namespace``); ``trust_pickle=True`` falls back to a full unpickle for checkpoints that carry other objects."""
    ck = _torch_load_weights(path, map_location, trust_pickle)
    sd = ck['state_dict'] if isinstance(ck, dict) and 'state_dict' in ck else ck
    if not isinstance(sd, dict) or not sd:
        raise ValueError('%s holds no state_dict' % path)
    out = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    if not out:
        raise ValueError('no key of %s starts with %r (keys look like %r)' % (path, prefix, next(iter(sd))))
    return out


def state_dict_to_npz(state_dict, path):
    """A state_dict as plain arrays (how tests/golden/correction_ckpt.npz was made: tests/golden/make_golden.py)."""
    np.savez(path, **{k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in state_dict.items()})


def load_state_dict_npz(path):
    with np.load(path) as z:
        return {k: torch.from_numpy(z[k]) for k in z.files}


def load_smplh_npz(path, n_betas=10, trust_pickle=False):
    """-> dict(v_template [V,3], shapedirs [V,3,n_betas], posedirs [V,3,9(J-1)], J_regressor [J,V], weights [V,J], parents [J] (root -1),
    faces [F,3]) as numpy arrays: what ``SMPL_Layer(model)`` takes.  Plain arrays are read without pickle.  The official SMPL+H release stores
    ``J_regressor`` as a pickled scipy sparse matrix: that ONE entry is unpickled, and only with ``trust_pickle=True`` (a pickle runs code);
    every other entry of the file is still read as a plain array."""
    with np.load(path, allow_pickle=False) as z:
        a = {}
        for k in z.files:
            try:
                a[k] = z[k]
            except ValueError:                     # an object array: needs pickle
                if k != 'J_regressor':
                    if k in ('v_template', 'shapedirs', 'posedirs', 'weights', 'parents', 'kintree_table', 'faces', 'f'):
                        raise ValueError('%s: entry %r is a pickled object; only J_regressor may be (the official release\'s sparse matrix)' % (path, k))
                    continue                       # entries the layer does not use are skipped, not unpickled
                if not trust_pickle:
                    raise ValueError('%s stores J_regressor as a pickled object (the official SMPL+H release does); pass trust_pickle=True only for a '
                                     'file whose origin you trust -- a pickle runs code' % path)
                with np.load(path, allow_pickle=True) as zp:
                    a[k] = zp[k]
    m = {}
    m['v_template'] = np.asarray(a['v_template'], np.float32).reshape(-1, 3)
    V = m['v_template'].shape[0]
    sd = np.asarray(a['shapedirs'], np.float32)
    if sd.ndim != 3 or sd.shape[:2] != (V, 3) or sd.shape[2] < n_betas:
        raise ValueError('shapedirs must be [V,3,>=%d], got %r' % (n_betas, sd.shape))
    m['shapedirs'] = np.ascontiguousarray(sd[:, :, :n_betas])                     # the layer uses the first 10 components (smpl_layer.py:49)
    m['weights'] = np.asarray(a['weights'], np.float32)
    J = m['weights'].shape[1]
    pd = np.asarray(a['posedirs'], np.float32)
    m['posedirs'] = pd.reshape(V, 3, -1)
    jr = a['J_regressor']
    if jr.dtype == object:                                                         # scipy sparse matrix pickled into the official file
        jr = jr.item().toarray()
    m['J_regressor'] = np.asarray(jr, np.float32)
    if 'parents' in a:
        par = np.asarray(a['parents']).astype(np.int64).reshape(-1)
    elif 'kintree_table' in a:
        par = np.asarray(a['kintree_table']).astype(np.int64)[0].copy()
    else:
        raise ValueError('neither parents nor kintree_table in %s' % path)
    par[0] = -1
    m['parents'] = par
    m['faces'] = np.asarray(a['faces'] if 'faces' in a else a['f']).astype(np.int64).reshape(-1, 3)
    if m['posedirs'].shape != (V, 3, 9 * (J - 1)) or m['J_regressor'].shape != (J, V) or m['weights'].shape != (V, J) or par.shape != (J,):
        raise ValueError('inconsistent SMPL-H buffers in %s' % path)
    return m


def save_smplh_npz(path, model):
    np.savez(path, **{k: (model[k].detach().cpu().numpy() if isinstance(model[k], torch.Tensor) else np.asarray(model[k])) for k in SMPLH_KEYS})


def is_dataset_batch(batch):
    return isinstance(batch, dict) and 'frames' in batch


def batch_from_dataset(batch, device=None):
    """DataLoader batch (data/dataset_smpl.py:182-204 after default collation: ``batch['frames'][t]['smplfit_params']['pose'|'betas'|'trans']``
    [B,...], ``['objfit_params']['angle'|'trans']`` [B,3], ``batch['obj_points']`` [B,P,>=3]) -> the raw tensor dict of
    ``eval.batch_from_raw``: body_pose [T,B,66], hand_pose [T,B,90] (GT hands, not padded), body_trans, obj_angles, obj_trans [T,B,3],
    beta [T,B,10], obj_points [B,P,3] -- the same stacks the reference builds at model/diffusion_smpl.py:197-201 and
    eval_smpl_short.py:145,148,150."""
    t = lambda a: torch.as_tensor(np.asarray(a) if not isinstance(a, torch.Tensor) else a).float()
    frames = batch['frames']
    pose = torch.stack([t(f['smplfit_params']['pose']) for f in frames], dim=0)                       # [T,B,156]
    raw = dict(body_pose=pose[..., :66], hand_pose=pose[..., 66:],
               body_trans=torch.stack([t(f['smplfit_params']['trans']) for f in frames], dim=0),
               obj_angles=torch.stack([t(f['objfit_params']['angle']) for f in frames], dim=0),
               obj_trans=torch.stack([t(f['objfit_params']['trans']) for f in frames], dim=0),
               beta=torch.stack([t(f['smplfit_params']['betas']) for f in frames], dim=0),
               obj_points=t(batch['obj_points'])[:, :, :3])
    if raw['body_pose'].shape[-1] != 66 or raw['hand_pose'].shape[-1] != 90:
        raise ValueError('smplfit_params.pose must be [B,156] per frame')
    return {k: (v.contiguous().to(device) if device is not None else v.contiguous()) for k, v in raw.items()}
