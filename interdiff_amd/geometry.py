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


class MeshTopology:
    """Device copies of faces + adjacency for one mesh (built once)."""

    def __init__(self, faces, V, device='cuda'):
        f = faces.detach().cpu().numpy() if isinstance(faces, torch.Tensor) else np.asarray(faces)
        if f.ndim == 3:
            f = f[0]
        ptr, adj_face, adj_corner = build_vertex_adjacency(f, V)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        self.V = V
        self.faces, self.adj_ptr, self.adj_face, self.adj_corner = t(f.astype(np.int32)), t(ptr), t(adj_face), t(adj_corner)


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
