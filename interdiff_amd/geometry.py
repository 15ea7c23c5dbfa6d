"""Geometry seam: ``vertex_normals(v, f)`` (data/tools.py:4) and ``point2point_signed(x, y, ...)``
(tools.py:11) on the HIP kernels of csrc/geometry.hip."""
import numpy as np
import torch
from . import _lib

_ADJ_CACHE = {}


def build_vertex_adjacency(faces, V):
    """vertex -> incident (face, corner) list in the reference's accumulation order
    (index_add over corner 1, then corner 2, then corner 0; ascending face index inside each)."""
    f = np.asarray(faces, dtype=np.int64)
    F = f.shape[0]
    order = []
    for rank, corner in enumerate((1, 2, 0)):
        order.append(np.stack([f[:, corner], np.full(F, rank), np.arange(F), np.full(F, corner)], axis=1))
    rec = np.concatenate(order)
    rec = rec[np.lexsort((rec[:, 2], rec[:, 1], rec[:, 0]))]
    ptr = np.zeros(V + 1, np.int64)
    np.add.at(ptr, rec[:, 0] + 1, 1)
    return np.cumsum(ptr).astype(np.int32), rec[:, 2].astype(np.int32), rec[:, 3].astype(np.int32)


def morton_order(points, bits=10):
    """Indices that sort ``points`` [n,3] along a Morton (Z-order) curve of ``bits`` bits per axis inside their bounding box; ties
    keep the lower index first.  Host-side, once per mesh."""
    p = np.asarray(points, dtype=np.float64).reshape(-1, 3)
    lo, ext = p.min(0), np.maximum(p.max(0) - p.min(0), 1e-30)
    q = np.clip(((p - lo) / ext * (2 ** bits - 1)).astype(np.int64), 0, 2 ** bits - 1)
    code = np.zeros(len(p), dtype=np.int64)
    for b in range(bits):
        for a in range(3):
            code |= ((q[:, a] >> b) & 1) << (3 * b + a)
    return np.argsort(code, kind='stable')


class MeshTopology:
    """Device copies of faces + adjacency for one mesh (built once).  With ``rest_vertices`` [V,3] it also carries the SCAN ORDER of
    the exact nearest-vertex kernels (csrc/correction.hip): ``vorder`` = Morton order of the rest pose (skinning is spatially smooth,
    so 16 consecutive scan positions stay a compact clump under any pose and whole blocks can be culled by their bounding box),
    ``faces_scan`` = faces in scan positions.  Results never depend on the order -- only how many vertex blocks a scan can skip."""

    def __init__(self, faces, V, device='cuda', rest_vertices=None):
        f = faces.detach().cpu().numpy() if isinstance(faces, torch.Tensor) else np.asarray(faces)
        if f.ndim == 3:
            f = f[0]
        ptr, adj_face, adj_corner = build_vertex_adjacency(f, V)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        self.V = V
        self.faces, self.adj_ptr, self.adj_face, self.adj_corner = t(f.astype(np.int32)), t(ptr), t(adj_face), t(adj_corner)
        self.vorder = self.faces_scan = self.rank = self.adj_pair_scan = self.vrank = None
        # per adjacency entry (vertex v, face, corner c) the other two corners (a, b) with normal contribution (a - v) x (b - v):
        # data/tools.py:27-39 accumulates cross(v2 - v1, v0 - v1) at corner 1, cross(v0 - v2, v1 - v2) at 2, cross(v1 - v0, v2 - v0) at 0
        fa, co = adj_face.astype(np.int64), adj_corner.astype(np.int64)
        a_of, b_of = np.array([1, 2, 0]), np.array([2, 0, 1])
        pair = np.stack([f[fa, a_of[co]], f[fa, b_of[co]]], axis=1)
        self.adj_pair = t(pair.astype(np.int32))
        if rest_vertices is not None:
            order = morton_order(np.asarray(rest_vertices).reshape(V, 3))
            rank = np.empty(V, np.int64)
            rank[order] = np.arange(V)
            self.rank = rank                                        # vertex -> scan position (host)
            self.vorder, self.faces_scan = t(order.astype(np.int32)), t(rank[f].astype(np.int32))
            self.adj_pair_scan = t(rank[pair].astype(np.int32))
            self.vrank = t(rank.astype(np.int32))                    # vertex -> scan position on the device: the contact scan reads a frame's vertices coalesced and scatters them

    def scan_positions(self, vertex_ids, device):
        """Scan positions of the given vertices (e.g. the marker set) as a device int32 tensor."""
        ids = np.asarray(list(vertex_ids), dtype=np.int64)
        return torch.from_numpy((self.rank[ids] if self.rank is not None else ids).astype(np.int32)).to(device)


def _topology(faces, V, device):
    """Topology of the mesh `faces` describes, from a small cache keyed by CONTENT: the reference hands in a freshly repeated faces
    tensor on every call (eval_smpl_short.py:98), so an address is neither stable nor unique.  One device compare per call; the hot
    path holds a MeshTopology and never comes here."""
    f0 = faces[0] if faces.dim() == 3 else faces
    key = (tuple(f0.shape), V, str(device))
    f32 = f0.to(device=device, dtype=torch.int32)
    for topo in _ADJ_CACHE.get(key, []):
        if torch.equal(topo.faces, f32):
            return topo
    topo = MeshTopology(f0, V, device)
    _ADJ_CACHE.setdefault(key, []).append(topo)
    if len(_ADJ_CACHE[key]) > 4:
        _ADJ_CACHE[key].pop(0)
    return topo


def vertex_normals(vertices, faces):
    """vertices [N,V,3], faces [N,F,3] (the reference repeats them per frame) or [F,3] -> [N,V,3]."""
    assert vertices.ndimension() == 3 and vertices.shape[2] == 3
    if isinstance(faces, MeshTopology):
        topo = faces
    else:
        assert faces.shape[-1] == 3
        if faces.ndimension() == 3:
            assert vertices.shape[0] == faces.shape[0]
        topo = _topology(faces, vertices.shape[1], vertices.device)
    lib = _lib.load()
    N, V, _ = vertices.shape
    v = vertices.contiguous().float()
    out = torch.empty_like(v)
    _lib.check(lib.interdiff_vertex_normals(_lib.dptr(v, torch.float32), N, V, _lib.dptr(topo.faces), _lib.dptr(topo.adj_ptr),
                                            _lib.dptr(topo.adj_face), _lib.dptr(topo.adj_corner), _lib.dptr(out), _lib.stream()),
               'vertex_normals')
    return out


def point2point_signed(x, y, x_normals=None, y_normals=None, return_vector=False):
    """Same outputs, order and error behaviour as tools.py:11-76 (indices are int32 like the CUDA op's)."""
    N, P1, D = x.shape
    P2 = y.shape[1]
    if y.shape[0] != N or y.shape[2] != D:
        raise ValueError("y does not have the correct shape.")
    if D != 3:
        raise ValueError("only 3-D points are supported")
    lib = _lib.load()
    dev = x.device
    xc, yc = x.contiguous().float(), y.contiguous().float()
    xn = x_normals.contiguous().float() if x_normals is not None else None
    yn = y_normals.contiguous().float() if y_normals is not None else None
    y2x_s = torch.empty(N, P2, device=dev)
    x2y_s = torch.empty(N, P1, device=dev)
    yidx = torch.empty(N, P2, dtype=torch.int32, device=dev)
    xidx = torch.empty(N, P1, dtype=torch.int32, device=dev)
    y2x = torch.empty(N, P2, 3, device=dev) if return_vector else None
    x2y = torch.empty(N, P1, 3, device=dev) if return_vector else None
    _lib.check(lib.interdiff_point2point_signed(_lib.dptr(xc), P1, _lib.dptr(yc), P2, N, _lib.dptr(xn, allow_none=True),
                                                _lib.dptr(yn, allow_none=True), _lib.dptr(y2x_s), _lib.dptr(x2y_s), _lib.dptr(yidx),
                                                _lib.dptr(xidx), _lib.dptr(y2x, allow_none=True), _lib.dptr(x2y, allow_none=True),
                                                _lib.stream()), 'point2point_signed')
    if not return_vector:
        return y2x_s, x2y_s, yidx, xidx
    return y2x_s, x2y_s, yidx, xidx, y2x, x2y


def nn_argmin(q, r):
    lib = _lib.load()
    N, Pq, _ = q.shape
    idx = torch.empty(N, Pq, dtype=torch.int32, device=q.device)
    qc, rc = q.contiguous().float(), r.contiguous().float()       # hold references until the launch is enqueued
    _lib.check(lib.interdiff_nn_argmin(_lib.dptr(qc), Pq, _lib.dptr(rc), r.shape[1], N, _lib.dptr(idx), _lib.stream()), 'nn_argmin')
    return idx
