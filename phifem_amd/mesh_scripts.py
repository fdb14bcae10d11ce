"""Host mirror of `phifem.mesh_scripts` (src/phifem/mesh_scripts.py) over the C ABI.

Same entry point, argument meaning, return shape and error behaviour as the reference's
`compute_tags_measures` (mesh_scripts.py:571-653); the classification itself runs in HIP.
"""
import ctypes as C
import os
import warnings

import numpy as np

from . import _lib as L
from .mesh import LazyMeshTags, Mesh, MeshTags

# mesh_scripts.py:22-25
debug_mode = os.environ.get("MODE", "") == "debug"

_ZERO_DEN_MSG = ("The detection function is zero everywhere on a cell. We mark it as 'cut' but "
                 "this can be incorrect and should be carefully checked.")


class NodalFunction:
    """A P1 function on a mesh given by its vertex values (stands in for a
    `dolfinx.fem.Function` in a first-order Lagrange space).  `values` may be a numpy array or
    a torch tensor living on the mesh's GPU."""

    def __init__(self, values, degree=1):
        """degree 1: one value per vertex.  degree 2: vertex values then edge-midpoint values
        (`mesh.p2_dof_points()`; quadrilaterals: `mesh.q2_dof_points()`), numpy or a torch tensor on the
        mesh's GPU; evaluated at the detection points by the library on the device
        (`phx_levelset_eval_points`) and classified there (PHX_PHI_POINTS)."""
        if degree not in (1, 2):
            raise NotImplementedError("level-set functions of degree 1 and 2 are implemented")
        self.values = values
        self.degree = degree


class DeviceExpression:
    """A level-set EXPRESSION evaluated on the device: `f(x)` receives a torch tensor of shape (gdim, npoints) on
    the mesh's GPU -- the physical detection points, produced by the library (`phx_detection_points_physical`) --
    and returns a float64 tensor of npoints values.  The device-side counterpart of passing a plain callable
    (which is evaluated by numpy on the host, like the reference's UFL-expression mode,
    tests/test_compute_meshtags.py:159-161)."""

    def __init__(self, f):
        if not callable(f):
            raise TypeError("DeviceExpression wraps a callable x -> phi on torch tensors")
        self.f = f


class Quadric:
    """phi(x) = sum_a (s_a x_a - c_a)^2 + c0, evaluated on the device at the detection points
    (the closed form behind `gen_levelset` of tests/test_compute_meshtags.py:18-25 and the
    spherical level-sets of the BASELINE configurations)."""

    def __init__(self, centre, scale, c0):
        p = np.zeros(7)
        p[0:len(centre)] = centre
        p[3:3 + len(scale)] = scale
        p[6] = c0
        self.params = p


class BoundaryMeasure:
    """Stand-in for the `ufl.Measure("ds", subdomain_data=...)` the reference returns
    (mesh_scripts.py:631-633,644): `measure(100)` / `measure(101)` give the flat int32
    [cell, local facet, ...] integration entities; on a sub-mesh the measure covers every
    exterior facet."""

    def __init__(self, mesh, box_mode):
        self._mesh = mesh
        self._box = box_mode
        self._cache = {}

    def __call__(self, tag):
        if not self._box:
            return self._mesh.boundary_facets.reshape(-1)
        if tag not in self._cache:
            n = C.c_int64(0)
            L.check(L.lib.phx_integration_entities(self._mesh._h, tag, None, C.byref(n)))
            out = np.empty(2 * n.value, dtype=np.int32)
            if n.value:
                L.check(L.lib.phx_integration_entities(
                    self._mesh._h, tag, out.ctypes.data_as(C.c_void_p), C.byref(n)))
            self._cache[tag] = out
        return self._cache[tag]


def _levelset_args(mesh, levelset, degree):
    """-> (phi_kind, pointer, loc, keepalive)."""
    if isinstance(levelset, NodalFunction) and levelset.degree == 2:
        return _device_points(mesh, degree, nodal=levelset.values)
    if isinstance(levelset, DeviceExpression):
        return _device_points(mesh, degree, expr=levelset.f)
    if isinstance(levelset, NodalFunction):
        v = levelset.values
        if hasattr(v, "data_ptr"):
            # the kernels read the tensor as a raw double*: it has to be one
            import torch
            if v.dtype != torch.float64 or not v.is_contiguous() or v.numel() != mesh.nv:
                raise ValueError("a nodal level-set tensor must be contiguous float64 with one value per mesh vertex")
            if v.is_cuda and v.device.index != mesh.device:
                raise ValueError(f"nodal level-set lives on cuda:{v.device.index}, the mesh on cuda:{mesh.device}")
        else:
            v = np.ascontiguousarray(v, dtype=np.float64)
            if v.shape[0] != mesh.nv:
                raise ValueError("nodal level-set must have one value per mesh vertex")
        p, loc = L.ptr(v)
        return L.PHI_NODAL_P1, p, loc, v
    if isinstance(levelset, Quadric):
        p, loc = L.ptr(levelset.params)
        return L.PHI_QUADRIC, p, loc, levelset.params
    if callable(levelset):
        # "UFL expression" mode (tests/test_compute_meshtags.py:159-161): the host evaluates
        # the callable at the physical detection points, the device does the classification.
        vals = _evaluate_callable(mesh, levelset, degree)
        p, loc = L.ptr(vals)
        return L.PHI_POINTS, p, loc, vals
    raise TypeError("discrete_levelset must be a NodalFunction, a Quadric or a callable x -> phi")


def _device_points(mesh, degree, nodal=None, expr=None):
    """PHX_PHI_POINTS values produced on the device: a degree-2 nodal level-set tabulated by the library, or a
    caller's torch expression at the physical detection points."""
    import torch
    dev = torch.device("cuda", mesh.device)
    cnt = C.c_int64(0)
    L.check(L.lib.phx_levelset_points_count(mesh._h, degree, C.byref(cnt)))
    out = torch.empty(cnt.value, dtype=torch.float64, device=dev)
    if nodal is not None:
        if mesh.cell_type == "quadrilateral":
            want, what = mesh.nv + mesh.nf + mesh.nc, "a Q2 level-set has one value per vertex, per facet and per cell"
        elif mesh.cell_type in _EDGE_VERTS:
            want, what = mesh.nv + mesh.ne, "a P2 level-set has one value per vertex and per edge"
        else:
            raise NotImplementedError("P2 level-sets are implemented on simplices and quadrilaterals")
        v = nodal
        if hasattr(v, "data_ptr"):
            if v.dtype != torch.float64 or not v.is_contiguous():
                raise ValueError("a nodal level-set tensor must be contiguous float64")
            if v.is_cuda and v.device.index != mesh.device:
                raise ValueError(f"nodal level-set lives on cuda:{v.device.index}, the mesh on cuda:{mesh.device}")
            n = v.numel()
        else:
            v = np.ascontiguousarray(v, dtype=np.float64)
            n = v.shape[0]
        if n != want:
            raise ValueError(what)
        p, loc = L.ptr(v)
        L.sync_torch_stream(dev)        # `out` was allocated (and possibly recycled) on torch's stream
        L.check(L.lib.phx_levelset_eval_points(mesh._h, degree, p, loc, C.c_void_p(out.data_ptr())))
        return L.PHI_POINTS, C.c_void_p(out.data_ptr()), L.DEVICE, (out, v)
    xq = torch.empty((cnt.value, mesh.gdim), dtype=torch.float64, device=dev)
    L.sync_torch_stream(dev)
    L.check(L.lib.phx_detection_points_physical(mesh._h, degree, C.c_void_p(xq.data_ptr())))
    mesh.synchronize()                  # ... and torch's stream does not wait for the mesh stream either
    vals = expr(xq.t())
    if not hasattr(vals, "data_ptr") or not vals.is_cuda or vals.numel() != cnt.value:
        raise ValueError("a DeviceExpression must return one value per point as a tensor on the mesh's GPU")
    out.copy_(vals.reshape(-1).to(torch.float64))
    L.sync_torch_stream(dev)            # `f` and the copy ran on torch's stream, the tagging kernels will not
    return L.PHI_POINTS, C.c_void_p(out.data_ptr()), L.DEVICE, out


def _ref_points(cell_type, degree, which):
    n = C.c_int64(0)
    ct = L.CELL_TYPES[cell_type]
    L.check(L.lib.phx_detection_points(ct, degree, which, None, C.byref(n)))
    dim = {"triangle": 2, "quadrilateral": 2, "tetrahedron": 3}[cell_type] - which
    out = np.empty((n.value, dim))
    L.check(L.lib.phx_detection_points(ct, degree, which, out.ctypes.data_as(C.c_void_p), C.byref(n)))
    return out


def _shape(kind, pts):
    x = pts[:, 0]
    if kind == "interval":
        return np.stack([1.0 - x, x], axis=1)
    y = pts[:, 1]
    if kind == "triangle":
        return np.stack([(1.0 - x) - y, x, y], axis=1)
    if kind == "quadrilateral":
        return np.stack([(1.0 - x) * (1.0 - y), x * (1.0 - y), (1.0 - x) * y, x * y], axis=1)
    z = pts[:, 2]
    return np.stack([((1.0 - x) - y) - z, x, y, z], axis=1)


_FACET_VERTS = {
    "triangle": np.array([[1, 2], [0, 2], [0, 1]]),
    "quadrilateral": np.array([[0, 1], [0, 2], [1, 3], [2, 3]]),
    "tetrahedron": np.array([[1, 2, 3], [0, 2, 3], [0, 1, 3], [0, 1, 2]]),
}


def _push(N, xv):
    """x_q = sum_i N[q,i] x_i, accumulated left to right (same order as the device code)."""
    acc = N[None, :, 0, None] * xv[:, None, 0, :]
    for i in range(1, N.shape[1]):
        acc = acc + N[None, :, i, None] * xv[:, None, i, :]
    return acc  # (n, npts, gdim)


_EDGE_VERTS = {
    "triangle": np.array([[1, 2], [0, 2], [0, 1]]),
    "tetrahedron": np.array([[2, 3], [1, 3], [1, 2], [0, 3], [0, 2], [0, 1]]),
}


def _p2_tab(cell_type, lam):
    """P2 basis at barycentric points lam (npts, n): vertex functions then edge functions."""
    n = lam.shape[1]
    ev = _EDGE_VERTS[cell_type]
    N = np.empty((lam.shape[0], n + ev.shape[0]))
    for i in range(n):
        N[:, i] = lam[:, i] * (2.0 * lam[:, i] - 1.0)
    for k, (a, b) in enumerate(ev):
        N[:, n + k] = 4.0 * lam[:, a] * lam[:, b]
    return N


def _l3(t):
    return np.stack([2 * (t - 0.5) * (t - 1), 4 * t * (1 - t), 2 * t * (t - 0.5)], axis=1)


_Q2_IDX = [(0, 0), (2, 0), (0, 2), (2, 2), (1, 0), (0, 1), (2, 1), (1, 2), (1, 1)]


def _q2_tab(pts):
    """Q2 basis at reference points of the unit square: 4 vertices, 4 edge midpoints (local facet order), centre."""
    Lx, Ly = _l3(pts[:, 0]), _l3(pts[:, 1])
    return np.stack([Lx[:, a] * Ly[:, b] for a, b in _Q2_IDX], axis=1)


def _evaluate_q2(mesh, nodal, degree):
    """phi_h in Q2 on quadrilaterals (nodal layout: vertices, edge midpoints by facet id, cell centres) at the cell
    detection points and at the facet detection points of the background-boundary facets."""
    if nodal.shape[0] != mesh.nv + mesh.nf + mesh.nc:
        raise ValueError("a Q2 level-set has one value per vertex, per facet and per cell")
    cells, c2f = mesh.cells, mesh.c2f
    cell_dofs = np.concatenate([cells, mesh.nv + c2f, (mesh.nv + mesh.nf + np.arange(mesh.nc))[:, None]], axis=1)
    vc = nodal[cell_dofs] @ _q2_tab(_ref_points("quadrilateral", degree, 0)).T
    bf = mesh.boundary_facets
    s = _ref_points("quadrilateral", degree, 1)[:, 0]          # parameter along the facet, first -> second vertex
    vf = np.empty((bf.shape[0], s.shape[0]))
    for lf, (ax, val) in enumerate([(1, 0.0), (0, 0.0), (0, 1.0), (1, 1.0)]):
        sel = np.flatnonzero(bf[:, 1] == lf)
        if sel.size == 0:
            continue
        pts = np.stack([np.full_like(s, val) if ax == 0 else s, np.full_like(s, val) if ax == 1 else s], axis=1)
        vf[sel] = nodal[cell_dofs[bf[sel, 0]]] @ _q2_tab(pts).T
    return np.ascontiguousarray(np.concatenate([vc.reshape(-1), vf.reshape(-1)]))


def _evaluate_p2(mesh, nodal, degree):
    """phi_h in P2 (vertex + edge values) at the cell detection points and at the facet detection
    points of the background-boundary facets (layout of PHX_PHI_POINTS)."""
    if mesh.cell_type == "quadrilateral":
        return _evaluate_q2(mesh, nodal, degree)
    if mesh.cell_type not in _EDGE_VERTS:
        raise NotImplementedError("P2 level-sets are implemented on simplices and quadrilaterals")
    cells, c2e = mesh.cells, mesh.c2e
    if nodal.shape[0] != mesh.nv + mesh.ne:
        raise ValueError("a P2 level-set has one value per vertex and per edge")
    cell_dofs = np.concatenate([cells, mesh.nv + c2e], axis=1)
    lam_c = _shape(mesh.cell_type, _ref_points(mesh.cell_type, degree, 0))      # P1 shape = barycentric
    vc = nodal[cell_dofs] @ _p2_tab(mesh.cell_type, lam_c).T                      # (nc, npts)
    bf = mesh.boundary_facets
    ftype = "interval" if mesh.tdim == 2 else "triangle"
    mu = _shape(ftype, _ref_points(mesh.cell_type, degree, 1))                   # (nq, nvpf)
    fv = _FACET_VERTS[mesh.cell_type]
    vf = np.empty((bf.shape[0], mu.shape[0]))
    for lf in range(fv.shape[0]):
        sel = np.flatnonzero(bf[:, 1] == lf)
        if sel.size == 0:
            continue
        lam = np.zeros((mu.shape[0], cells.shape[1]))
        lam[:, fv[lf]] = mu
        vf[sel] = nodal[cell_dofs[bf[sel, 0]]] @ _p2_tab(mesh.cell_type, lam).T
    return np.ascontiguousarray(np.concatenate([vc.reshape(-1), vf.reshape(-1)]))


def _evaluate_callable(mesh, f, degree):
    x = mesh.x
    cells = mesh.cells
    Nc = _shape(mesh.cell_type, _ref_points(mesh.cell_type, degree, 0))
    xq = _push(Nc, x[cells])
    with np.errstate(all="ignore"):
        vc = np.asarray(f(xq.reshape(-1, mesh.gdim).T), dtype=np.float64).reshape(-1)
    bf = mesh.boundary_facets
    ftype = "interval" if mesh.tdim == 2 else "triangle"
    Nf = _shape(ftype, _ref_points(mesh.cell_type, degree, 1))
    fv = _FACET_VERTS[mesh.cell_type][bf[:, 1]]
    xf = _push(Nf, x[np.take_along_axis(cells[bf[:, 0]], fv, axis=1)])
    with np.errstate(all="ignore"):
        vf = np.asarray(f(xf.reshape(-1, mesh.gdim).T), dtype=np.float64).reshape(-1)
    return np.ascontiguousarray(np.concatenate([vc, vf]))


def _reshape_map(offsets, array):
    """mesh_scripts.py:195-214 on a plain adjacency (offsets[n+1], array): dense (n, max_links)
    table padded with -1, the links of an entity stored in REVERSE order.  The device kernels read
    the CSR adjacencies directly; this host helper exists for callers of the reference's API."""
    offsets = np.asarray(offsets, dtype=np.int64)
    array = np.asarray(array)
    num = np.diff(offsets)
    width = int(num.max()) if num.size else 0
    emap = -np.ones((num.size, width), dtype=np.int64)
    for k in range(width):
        has = num > k
        emap[has, k] = array[offsets[1:][has] - k - 1]
    return emap, width


def _tag_cells(mesh, levelset, detection_degree, single_layer_cut=False):
    """mesh_scripts.py:284-390."""
    mesh._flush_lazy_tags()     # MeshTags of an earlier call keep the state they were created in
    kind, p, loc, keep = _levelset_args(mesh, levelset, detection_degree)
    warn = C.c_int(0)
    L.check(L.lib.phx_tag_cells(mesh._h, kind, p, loc, detection_degree,
                                1 if single_layer_cut else 0, C.byref(warn)))
    if warn.value:
        warnings.warn(_ZERO_DEN_MSG, RuntimeWarning)  # mesh_scripts.py:129-133
    return (kind, p, loc, keep)


def _tag_facets(mesh, staged, detection_degree):
    """mesh_scripts.py:393-558."""
    mesh._flush_lazy_tags()
    kind, p, loc, keep = staged
    L.check(L.lib.phx_tag_facets(mesh._h, kind, p, loc, detection_degree))
    if mesh.nbf < mesh.nc:
        # the reference's `ds` detection has a zero denominator on every cell without a
        # background-boundary facet, so its RuntimeWarning always fires here
        warnings.warn(_ZERO_DEN_MSG, RuntimeWarning)


def _meshtags(mesh, facets):
    return LazyMeshTags(mesh.tdim - 1 if facets else mesh.tdim, mesh, facets)


def compute_tags_measures(mesh, discrete_levelset, detection_degree, box_mode=False,
                          single_layer_cut=False, overwrite_tags={}):
    """Compute the mesh (cells and facets) tags as well as the discrete boundary measures.

    Mirrors src/phifem/mesh_scripts.py:571-653.

    Args:
        mesh: a `phifem_amd.Mesh`.
        discrete_levelset: `NodalFunction` (P1 values), `Quadric`, or a callable x -> phi in
            the reference's numpy convention (x[0], x[1], ...), evaluated like a UFL expression.
        detection_degree: degree of the boundary-point detection rule.
        box_mode: False -> tags on the sub-mesh of cells tagged 1/2; True -> on the input mesh.
        single_layer_cut: force a single layer of cut cells.
        overwrite_tags: {"cells": MeshTags, "facets": MeshTags} user tags that win.

    Returns (cells_tags, facets_tags, submesh|None, boundaries_measure, submesh_maps|None).
    """
    staged = _tag_cells(mesh, discrete_levelset, detection_degree, single_layer_cut)
    if debug_mode:
        cv = mesh.cell_tag_values()
        if not np.any(cv == 1):
            raise ValueError("No interior cells (1)!")          # mesh_scripts.py:361-362
        if not np.any(cv == 2):
            print("WARNING: no cut cells computed in the partition.")
    _tag_facets(mesh, staged, detection_degree)
    if debug_mode:
        fv = mesh.facet_tag_values()
        if not np.any(fv == 1):
            raise ValueError("No interior facets (1)!")         # mesh_scripts.py:500-501
        if not np.any(fv == 2):
            print("WARNING: no cut facet computed in the partition.")
        if not np.any(fv == 4):
            raise ValueError("No boundary facets (4)!")         # mesh_scripts.py:504-505

    for key, is_facet in (("cells", 0), ("facets", 1)):
        if key in overwrite_tags:
            ow = overwrite_tags[key]
            idx = np.ascontiguousarray(ow.indices, dtype=np.int32)
            val = np.ascontiguousarray(ow.values, dtype=np.int32)
            mesh._flush_lazy_tags()
            L.check(L.lib.phx_overwrite_tags(mesh._h, is_facet, idx.size,
                                             idx.ctypes.data_as(C.c_void_p),
                                             val.ctypes.data_as(C.c_void_p)))

    if box_mode:
        return (_meshtags(mesh, False), _meshtags(mesh, True), None,
                BoundaryMeasure(mesh, True), None)

    h = C.c_void_p()
    L.check(L.lib.phx_submesh_create(mesh._h, C.byref(h)))
    sub = Mesh(h, parent=mesh)
    c_map = np.empty(sub.nc, dtype=np.int32)
    v_map = np.empty(sub.nv, dtype=np.int32)
    L.check(L.lib.phx_submesh_maps(sub._h, c_map.ctypes.data_as(C.c_void_p),
                                   v_map.ctypes.data_as(C.c_void_p)))
    return (_meshtags(sub, False), _meshtags(sub, True), sub, BoundaryMeasure(sub, False),
            [c_map, v_map, v_map.copy()])
