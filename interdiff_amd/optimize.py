"""Physics post-optimisation seam ("next" row N4 of SURVEY.md §8(f)): ``optimize(index, data)`` of the reference
(optimization.py:19-173) on the hand-written backward kernels of csrc/optimize.hip.

The reference optimises ONE clip per call with torch autograd + ``optim.Adam``; ``PhysicsOptimizer.optimize`` takes the
clip (or a batch of clips, optimised side by side) as tensors and returns the record fields the reference writes back
into ``data['frames']`` (:167-172).  torch allocates the device buffers and owns the stream -- all arithmetic, the Adam
update and the best-iterate bookkeeping run in the HIP library (no autograd, no torch.optim).
"""
import ctypes as C
import numpy as np
import torch
from . import _lib
from .geometry import MeshTopology

N_ITERS, NP = 200, _lib.OPT_NP
SEGMENTS = dict(glo=(0, 9), body=(9, 198), hand=(198, 468), transl=(468, 471), obj_transl=(471, 474), obj_rot=(474, 483))
_SHAPES = dict(glo=(1, 3, 3), body=(21, 3, 3), hand=(30, 3, 3), transl=(3,), obj_transl=(3,), obj_rot=(3, 3))
KSLICE = 1024


class PhysicsOptimizer:
    def __init__(self, smpl_layer, device='cuda', scan_order=True):
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.smpl = smpl_layer
        cm = smpl_layer.cmodel
        self.V, self.J, self.KB = cm.V, cm.J, cm.KB
        self.topo = MeshTopology(smpl_layer.th_faces, cm.V, self.device, rest_vertices=getattr(smpl_layer, 'v_template', None) if scan_order else None)
        geo = _lib.CorrectionCtx()
        geo.smpl = C.pointer(cm)
        geo.faces, geo.adj_ptr = self.topo.faces.data_ptr(), self.topo.adj_ptr.data_ptr()
        geo.adj_face, geo.adj_corner = self.topo.adj_face.data_ptr(), self.topo.adj_corner.data_ptr()
        geo.adj_pair = self.topo.adj_pair.data_ptr()
        if self.topo.vorder is not None:       # scan order of the exact nearest-vertex scan (block culling, csrc/correction.hip); results do not depend on it
            self._markers_scan = self.topo.scan_positions([0], self.device)
            geo.vorder, geo.faces_scan, geo.adj_pair_scan = self.topo.vorder.data_ptr(), self.topo.faces_scan.data_ptr(), self.topo.adj_pair_scan.data_ptr()
            geo.markers_scan = self._markers_scan.data_ptr()
            geo.vrank = self.topo.vrank.data_ptr()
        self.geo = geo
        # constants of the backward pass: the blend basis transposed (k-major rows, zero padded to the split-K slice) and
        # the skinning weights regrouped by joint
        self.K3P = (3 * cm.V + KSLICE - 1) // KSLICE * KSLICE
        blend = smpl_layer._bufs['blend_rows']                              # [3V][KB]
        self.blendT = torch.zeros(cm.KB, self.K3P, dtype=torch.float32, device=self.device)
        self.blendT[:, :3 * cm.V] = blend.t()
        idx, w = smpl_layer._bufs['skin_idx'].cpu().numpy(), smpl_layer._bufs['skin_w'].cpu().numpy()
        vtx = np.repeat(np.arange(cm.V, dtype=np.int32)[:, None], idx.shape[1], axis=1)
        keep = w != 0
        order = np.lexsort((vtx[keep], idx[keep]))                           # by joint, then ascending vertex
        jj, vv, ww = idx[keep][order], vtx[keep][order], w[keep][order]
        ptr = np.zeros(cm.J + 1, np.int32)
        ptr[1:] = np.cumsum(np.bincount(jj, minlength=cm.J))
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
        self.jv_ptr, self.jv_vtx, self.jv_w = t(ptr), t(vv.astype(np.int32)), t(ww.astype(np.float32))
        ctx = _lib.OptCtx()
        ctx.geo = C.pointer(self.geo)
        ctx.blendT, ctx.jv_ptr, ctx.jv_vtx, ctx.jv_w = (x.data_ptr() for x in (self.blendT, self.jv_ptr, self.jv_vtx, self.jv_w))
        ctx.K3P = self.K3P
        self.ctx = ctx
        self._state = None

    # ---- buffers ----------------------------------------------------------------------------------------------------
    def _alloc(self, B, T, P, max_iters):
        key = (B, T, P, max_iters)
        if self._state is not None and self._state[0] == key:
            return self._state[1], self._state[2]
        N, V, J = B * T, self.V, self.J
        f = lambda *s: torch.zeros(*s, dtype=torch.float32, device=self.device)
        i = lambda *s: torch.zeros(*s, dtype=torch.int32, device=self.device)
        bufs = dict(betas=f(N, 10), obj_points=f(B, P, 3), param=f(N, NP), init=f(N, NP), grad=f(N, NP), m=f(N, NP), v=f(N, NP), best=f(N, NP),
                    pose=f(N, 156), tr=f(N, 3), verts=f(N, V, 3), vposed=f(N, V, 3), verts_gt=f(N, V, 3), gv=f(N, V, 3),
                    jtr=f(N, J, 3), pts=f(N, P, 3), y2x=f(N, P, 3), y2x_signed=f(N, P), yidx=i(N, P), near=i(N, V),
                    dvposed=f(N, self.K3P), dA=f(N, J, 12), dfeat=f(self.K3P // KSLICE, N, self.KB), gtr=f(N, 3), lossf=f(N, _lib.OPT_NLOSS),
                    loss=f(B, 4), loss_hist=f(max_iters, B, 4), best_loss=f(B), flag=i(B),
                    foot_static=torch.zeros(N, 2, dtype=torch.uint8, device=self.device), foot_cnt=i(B, 2), ctl=i(4))
        need = self.lib.interdiff_smpl_workspace_bytes(C.byref(self.smpl.cmodel), N)
        bufs['smpl_ws'] = torch.empty(max(need, 256), dtype=torch.uint8, device=self.device)
        st = _lib.OptState()
        st.B, st.T, st.P, st.max_iters = B, T, P, max_iters
        for k in _lib._OPT_PTRS:
            setattr(st, k, bufs[k].data_ptr())
        st.smpl_ws_bytes = bufs['smpl_ws'].numel()
        if self.topo.vorder is not None and P <= 2048:           # scratch of the culled nearest-neighbour kernels
            bufs.update(porder=i(B, P), psort=f(N, 2048, 4), pbox=f(N, 32, 2, 4))
            st.porder, st.psort, st.pbox = (bufs[k].data_ptr() for k in ('porder', 'psort', 'pbox'))
        self._state = (key, st, bufs)
        return st, bufs

    @staticmethod
    def _batched(pose, trans, obj_angles, obj_trans, betas, obj_points):
        single = pose.dim() == 2
        if single:
            pose, trans, obj_angles, obj_trans, betas, obj_points = (a[None] for a in (pose, trans, obj_angles, obj_trans, betas, obj_points))
        B, T = pose.shape[:2]
        if pose.shape[2] != 156 or T < 3:
            raise ValueError('pose must be [T,156] or [B,T,156] with T >= 3')
        if obj_points.shape[0] != B or obj_points.dim() != 3:
            raise ValueError('obj_points must be [P,3] or [B,P,3]')
        flat = [a.reshape(B * T, -1).contiguous().float() for a in (pose, trans, obj_angles, obj_trans, betas)]
        return single, B, T, flat, obj_points[..., :3].contiguous().float()

    def _init(self, args, first_iter, max_iters):
        single, B, T, (pose, trans, obj_angles, obj_trans, betas), pts = self._batched(*args)
        st, bufs = self._alloc(B, T, pts.shape[1], max_iters)
        bufs['betas'].copy_(betas)
        bufs['obj_points'].copy_(pts)
        _lib.check(self.lib.interdiff_optimize_init(C.byref(self.ctx), C.byref(st), _lib.dptr(pose), _lib.dptr(trans), _lib.dptr(obj_angles),
                                                    _lib.dptr(obj_trans), int(first_iter), _lib.stream()), 'optimize_init')
        return single, B, T, st, bufs

    @staticmethod
    def _split(rows, B, T, single):
        out = {}
        for k, (a, b) in SEGMENTS.items():
            v = rows[:, a:b].reshape((B, T) + _SHAPES[k])
            out[k] = v[0] if single else v
        return out

    # ---- API --------------------------------------------------------------------------------------------------------
    def loss_and_grads(self, params, pose, trans, obj_angles, obj_trans, betas, obj_points, ii):
        """calc_loss + backward (optimization.py:54-121,143) at ``params`` (dict glo/body/hand/transl/obj_transl/obj_rot)
        for iteration number ii.  Returns ([total, collision, reg, reg_v] per clip, dict of gradients)."""
        single, B, T, st, bufs = self._init((pose, trans, obj_angles, obj_trans, betas, obj_points), ii, 1)
        rows = torch.cat([params[k].reshape(B * T, -1).float() for k in ('glo', 'body', 'hand', 'transl', 'obj_transl', 'obj_rot')], dim=1)
        bufs['param'].copy_(rows)
        _lib.check(self.lib.interdiff_optimize_loss_grad(C.byref(self.ctx), C.byref(st), _lib.stream()), 'optimize_loss_grad')
        loss = bufs['loss'].clone()
        return (loss[0] if single else loss), self._split(bufs['grad'].clone(), B, T, single)

    def optimize(self, pose, trans, obj_angles, obj_trans, betas, obj_points, iters=None):
        """optimization.py:123-172.  ``iters``: the iteration numbers ii to run, consecutive (default range(200)).
        Returns dict(pose [.,T,156], trans, obj_angles, obj_trans [.,T,3], losses [K,(B,)4], params, saved [B] bool)."""
        iters = list(range(N_ITERS)) if iters is None else [int(i) for i in iters]
        if not iters or iters != list(range(iters[0], iters[0] + len(iters))):
            raise ValueError('iters must be consecutive iteration numbers')
        single, B, T, st, bufs = self._init((pose, trans, obj_angles, obj_trans, betas, obj_points), iters[0], len(iters))
        stream = _lib.stream()
        for _ in iters:
            _lib.check(self.lib.interdiff_optimize_step(C.byref(self.ctx), C.byref(st), stream), 'optimize_step')
        N = B * T
        o = [torch.empty(N, d, dtype=torch.float32, device=self.device) for d in (156, 3, 3, 3)]
        _lib.check(self.lib.interdiff_optimize_finish(C.byref(self.ctx), C.byref(st), *(_lib.dptr(a) for a in o), stream), 'optimize_finish')
        shp = (lambda a: a.reshape(T, -1)) if single else (lambda a: a.reshape(B, T, -1))
        losses = bufs['loss_hist'].clone()
        return dict(pose=shp(o[0]), trans=shp(o[1]), obj_angles=shp(o[2]), obj_trans=shp(o[3]), losses=losses[:, 0] if single else losses,
                    params=self._split(bufs['param'].clone(), B, T, single), saved=bufs['best_loss'] < 1e7)
