"""Correction-hook seam: ``denoised_fn(x, t, model_kwargs) -> x`` (eval_smpl_short.py:84-130) on the fused
``interdiff_correction`` entry point.  ``HipCorrection`` owns the packed SMPL model, mesh adjacency,
ObjProjector and workspace; calling it mutates and returns ``x`` like the reference does (:129-130)."""
import ctypes as C
import os
import itertools
import numpy as np
import torch
from . import _lib
from .geometry import MeshTopology

MARKERS67 = [3470, 3171, 3327, 857, 1812, 628, 182, 3116, 3040, 239,
             1666, 1725, 0, 2174, 1568, 1368, 3387, 2112, 1053, 1058,
             3336, 3346, 1323, 2108, 3122, 3314, 1252, 1082, 1861, 1454,
             850, 2224, 3233, 1769, 6728, 4343, 5273, 4116, 3694, 6399,
             6540, 6488, 3749, 5135, 5194, 3512, 5635, 5210, 4360, 4841,
             6786, 5573, 4538, 4544, 6736, 6747, 4804, 5568, 6544, 6682,
             5322, 4927, 5686, 4598, 6633, 3506, 3508]      # markerset_ssm67_smplh, data/utils.py:232-238


_UID = itertools.count(1)


def correction_gate(t0):
    """eval_smpl_short.py:85: the hook only acts for t <= 500 and t % 50 == 0."""
    return not (t0 > 500 or t0 % 50 != 0)


class HipCorrection:
    def __init__(self, smpl_layer, objprojector, n_points=2048, past_len=10, markers=MARKERS67, device='cuda', scan_order=True):
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.smpl, self.objproj, self.past_len = smpl_layer, objprojector, past_len
        self.topo = MeshTopology(smpl_layer.th_faces, smpl_layer.cmodel.V, self.device, rest_vertices=getattr(smpl_layer, 'v_template', None) if scan_order else None)
        self.markers_idx = torch.tensor(list(markers), dtype=torch.int32, device=self.device)
        ctx = _lib.CorrectionCtx()
        ctx.smpl = C.pointer(smpl_layer.cmodel)
        ctx.objproj = C.pointer(objprojector.cop)
        ctx.faces, ctx.adj_ptr = self.topo.faces.data_ptr(), self.topo.adj_ptr.data_ptr()
        ctx.adj_face, ctx.adj_corner = self.topo.adj_face.data_ptr(), self.topo.adj_corner.data_ptr()
        ctx.markers_idx = self.markers_idx.data_ptr()
        ctx.n_markers, ctx.n_points, ctx.past_len = len(markers), n_points, past_len
        if self.topo.vorder is not None:       # scan order of the exact nearest-vertex scan (block culling); results do not depend on it
            self.markers_scan = self.topo.scan_positions(markers, self.device)
            ctx.vorder, ctx.faces_scan, ctx.markers_scan = self.topo.vorder.data_ptr(), self.topo.faces_scan.data_ptr(), self.markers_scan.data_ptr()
            ctx.adj_pair_scan = self.topo.adj_pair_scan.data_ptr()
            ctx.vrank = self.topo.vrank.data_ptr()
        ctx.tune = 2 if os.environ.get('INTERDIFF_HOOK_ONE_STREAM') == '1' else 0      # A/B only (tools/): the one-launch predictor after the scan instead of inside its launch
        self.ctx = ctx
        self._ws = {}
        self.debug = None            # set to {} to receive condition/contact/distance/loss of the last call
        self._uid = next(_UID)       # names this hook in the sampler's per-shape graph cache (never reused, unlike id())

    graph_capturable = True          # apply_dev() enqueues kernels only, every launch parameter independent of the timestep: the sampler captures whole hook steps

    def workspace_for(self, B, T):
        """A workspace of the hook for (B, T) that the CALLER owns (the sampler's graph cache: its address is baked into captured graphs)."""
        return torch.empty(self.lib.interdiff_correction_workspace_bytes(C.byref(self.ctx), B, T), dtype=torch.uint8, device=self.device)

    def apply_dev(self, x, table, state, y, ws):
        """``apply`` with the blend weight t / 1000 read on the device from the sampler's coefficient table at its current timestep
        (``table[state[0]][3]``, interdiff_correction_dev) and a caller-owned workspace: hipGraph-capturable, one capture serves every
        corrected step.  ``y``: dict(inpainted_motion, hand_pose, beta, obj_points) of contiguous float tensors."""
        B, _, Cc, T = x.shape
        if y['obj_points'].shape[1] != self.ctx.n_points:
            raise ValueError('obj_points must have %d points' % self.ctx.n_points)
        _lib.check(self.lib.interdiff_correction_dev(C.byref(self.ctx), _lib.dptr(x, torch.float32), _lib.dptr(y['inpainted_motion'], torch.float32),
                                                     _lib.dptr(y['hand_pose'], torch.float32), _lib.dptr(y['beta'], torch.float32),
                                                     _lib.dptr(y['obj_points'], torch.float32), B, T, _lib.dptr(table), _lib.dptr(state),
                                                     _lib.dptr(ws), ws.numel(), _lib.stream()), 'correction_dev')
        return x

    def _workspace(self, B, T):
        """One workspace per STREAM the hook is called on: the sampler calls it for the two halves of a batch on two streams at once."""
        need = self.lib.interdiff_correction_workspace_bytes(C.byref(self.ctx), B, T)
        key = torch.cuda.current_stream(self.device).cuda_stream
        ws = self._ws.pop(key, None)                  # (re-inserted below: the dict is kept in least-recently-used order)
        if ws is None or ws.numel() < need:
            ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        self._ws[key] = ws
        if len(self._ws) > self.MAX_STREAM_WORKSPACES:      # stream handles are recycled and a workspace is > 100 MB at the bench shape: keep a few, drop
            torch.cuda.synchronize(self.device)       # the stalest (nothing may still be running on it: rare path, one sync)
            self._ws.pop(next(iter(self._ws)))
        return ws

    MAX_STREAM_WORKSPACES = 4

    @staticmethod
    def slice_kwargs(model_kwargs, sl):
        """``model_kwargs`` of the clips ``sl`` of the batch (every entry of ``y`` the hook reads is per clip, eval_smpl_short.py:88-106):
        what lets the sampler call the hook per half batch.  Contiguous copies, made once per sample."""
        y = model_kwargs['y']
        per_clip_dim = dict(inpainted_motion=0, inpainting_mask=0, obj_points=0, hand_pose=1, beta=1, cond=1)
        ys = {k: (v[(slice(None),) * per_clip_dim[k] + (sl,)].contiguous() if k in per_clip_dim and isinstance(v, torch.Tensor) else v) for k, v in y.items()}
        return dict(model_kwargs, y=ys)

    def apply(self, x, t0, y):
        """Run the correction unconditionally for timestep value t0 (host int); x [B,1,144,T] in place."""
        B, _, Cc, T = x.shape
        if not x.is_contiguous():
            raise ValueError('x must be contiguous (it is updated in place)')
        if y['obj_points'].shape[1] != self.ctx.n_points:
            raise ValueError('obj_points must have %d points' % self.ctx.n_points)
        gt = y['inpainted_motion'].contiguous()
        hp, beta, pts = y['hand_pose'].contiguous().float(), y['beta'].contiguous().float(), y['obj_points'].contiguous().float()
        ws = self._workspace(B, T)
        dbg = [None] * 4
        if self.debug is not None:
            dbg = [torch.empty(B, dtype=torch.uint8, device=self.device), torch.empty(B, len(MARKERS67), dtype=torch.int32, device=self.device),
                   torch.empty(B, device=self.device), torch.empty(B, device=self.device)]
        blend_t = float(np.float32(t0) / np.float32(1000))                      # hard-coded 1000, :128
        _lib.check(self.lib.interdiff_correction(C.byref(self.ctx), _lib.dptr(x, torch.float32), _lib.dptr(gt, torch.float32),
                                                 _lib.dptr(hp), _lib.dptr(beta), _lib.dptr(pts), B, T, blend_t,
                                                 *[_lib.dptr(d, allow_none=True) for d in dbg], _lib.dptr(ws), ws.numel(),
                                                 _lib.stream()), 'correction')
        if self.debug is not None:
            self.debug.update(condition=dbg[0], contact=dbg[1], distance=dbg[2], loss=dbg[3])
        return x

    def contact_nn(self, verts, obj_points, objR, objT, want_stats=False):
        """Signed object->human distances and nearest vertices (tools.point2point_signed's o2h half, tools.py:45-76, with the object
        transform of eval_smpl_short.py:107 fused): verts [T,B,V,3], obj_points [B,P,3], objR [T,B,3,3], objT [T,B,3] ->
        (o2h [T,B,P], idx int32 [T,B,P][, (blocks scored, blocks there are, box tests made), summed over waves])."""
        T, B = verts.shape[:2]
        P = self.ctx.n_points
        o2h = torch.empty(T, B, P, device=self.device)
        idx = torch.empty(T, B, P, dtype=torch.int32, device=self.device)
        stats = torch.zeros(12, dtype=torch.int64, device=self.device) if want_stats else None
        need = self.lib.interdiff_contact_nn_workspace_bytes(C.byref(self.ctx), B, T)
        ws = getattr(self, '_nn_ws', None)
        if ws is None or ws.numel() < need:
            ws = self._nn_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        v, pts, R_, t_ = (a.contiguous().float() for a in (verts, obj_points, objR, objT))
        _lib.check(self.lib.interdiff_contact_nn(C.byref(self.ctx), _lib.dptr(v), _lib.dptr(pts), _lib.dptr(R_), _lib.dptr(t_), B, T, _lib.dptr(o2h),
                                                 _lib.dptr(idx), _lib.dptr(stats, allow_none=True), _lib.dptr(ws), ws.numel(), _lib.stream()), 'contact_nn')
        return (o2h, idx, tuple(int(x) for x in stats.cpu())) if want_stats else (o2h, idx)

    def is_active(self, t0):
        """Host-side gate (eval_smpl_short.py:85) -- lets the sampler replay its captured plain-step graph otherwise."""
        return correction_gate(int(t0))

    def __call__(self, x, t, model_kwargs):
        t0 = getattr(t, 'host_value', None)
        if t0 is None:
            t0 = int(t[0])                      # device sync, like the reference's `t[0] > 500`
        if not correction_gate(t0):
            return x
        return self.apply(x, t0, model_kwargs['y'])
