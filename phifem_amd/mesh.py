"""Mesh handle resident in HBM + MeshTags view (host side of the C ABI)."""
import ctypes as C

import weakref

import numpy as np

from . import _lib as L


class MeshTags:
    """The part of dolfinx.mesh.MeshTags the reference relies on
    (src/phifem/mesh_scripts.py:386-388,425-427,562-567): `.indices` (int32, strictly
    increasing), `.values` (int32), `.dim`, `.find(v)`."""

    def __init__(self, dim, indices, values):
        self.dim = int(dim)
        self.indices = np.ascontiguousarray(indices, dtype=np.int32)
        self.values = np.ascontiguousarray(values, dtype=np.int32)
        if self.indices.size > 1 and np.any(np.diff(self.indices) <= 0):
            raise ValueError("MeshTags entities must be sorted and unique")

    def find(self, v):
        return self.indices[self.values == v]


class LazyMeshTags(MeshTags):
    """MeshTags whose host arrays are fetched from the device on first use.  The tags of a 256^3 box are 3e8 bytes
    on the device; `compute_tags_measures` must return MeshTags objects (src/phifem/mesh_scripts.py:647-653), but a
    caller that goes on to assemble and solve never looks at them.  The mesh flushes every outstanding object
    before its tags change again (`Mesh._flush_lazy_tags`), so what is read is always the state at creation."""

    def __init__(self, dim, mesh, facets):
        self.dim = int(dim)
        self._mesh, self._facets = mesh, facets
        self._idx = self._val = None
        mesh._lazy_tags.add(self)

    def _load(self):
        if self._idx is None:
            m = self._mesh
            vals = m.facet_tag_values() if self._facets else m.cell_tag_values()
            idx = np.flatnonzero(vals > 0).astype(np.int32)
            self._idx, self._val = idx, np.ascontiguousarray(vals[idx], dtype=np.int32)
            self._mesh = None

    @property
    def indices(self):
        self._load()
        return self._idx

    @property
    def values(self):
        self._load()
        return self._val


class Mesh:
    """Owns a `phx_mesh*`.  Arrays stay on the GPU; accessors copy on demand."""

    def __init__(self, handle, parent=None, device=None):
        self._h = C.c_void_p(handle) if not isinstance(handle, C.c_void_p) else handle
        self.parent = parent
        self._lazy_tags = weakref.WeakSet()
        self._tag_generation = 0        # bumped by every Python entry that changes the tags of this mesh
        self.device = int(device) if device is not None else (parent.device if parent is not None else 0)
        cnt = (C.c_int64 * 6)()
        L.check(L.lib.phx_mesh_counts(self._h, cnt))
        self.gdim, ct, self.nv, self.nc, self.nf, self.nbf = (int(v) for v in cnt)
        self.cell_type = L.CELL_NAMES[ct]
        self.tdim = self.gdim
        self.nvpc = {"triangle": 3, "quadrilateral": 4, "tetrahedron": 4}[self.cell_type]
        self.nfpc = self.nvpc

    # --- construction -------------------------------------------------------------------
    @classmethod
    def from_arrays(cls, cell_type, x, cells, device=0):
        """Unstructured mesh (stands in for XDMFFile.read_mesh,
        tests/test_compute_meshtags.py:136-137).  Quadrilaterals in tensor-product order."""
        if cell_type not in L.CELL_TYPES:
            raise NotImplementedError(
                "Mesh tags computation does not support other cell types than "
                "'triangle', 'quadrilateral' or 'tetrahedron'")  # mesh_scripts.py:326-329
        x = np.ascontiguousarray(x, dtype=np.float64)
        cells = np.ascontiguousarray(cells, dtype=np.int32)
        h = C.c_void_p()
        L.check(L.lib.phx_mesh_create(x.shape[1], L.CELL_TYPES[cell_type], x.shape[0],
                                      x.ctypes.data_as(C.c_void_p), cells.shape[0],
                                      cells.ctypes.data_as(C.c_void_p), device, C.byref(h)))
        return cls(h, device=device)

    def __del__(self):
        try:
            if self._h:
                L.lib.phx_mesh_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # --- accessors ----------------------------------------------------------------------
    def _flush_lazy_tags(self):
        """Materialise the MeshTags handed out so far.  Called by every Python entry that is about to change the
        tags of this mesh (`_tag_cells`, `_tag_facets`, the overwrite path), so a MeshTags object always shows
        the state it was created in; callers that change tags through the raw C ABI call it themselves."""
        self._tag_generation += 1
        for t in list(self._lazy_tags):
            t._load()

    def _get(self, which, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        L.check(L.lib.phx_mesh_get_array(self._h, which, out.ctypes.data_as(C.c_void_p), L.HOST))
        return out

    @property
    def x(self):
        return self._get(L.ARR_COORDS, (self.nv, self.gdim), np.float64)

    @property
    def cells(self):
        return self._get(L.ARR_CELLS, (self.nc, self.nvpc), np.int32)

    @property
    def c2f(self):
        return self._get(L.ARR_C2F, (self.nc, self.nfpc), np.int32)

    @property
    def f2c(self):
        return self._get(L.ARR_F2C, (self.nf, 2), np.int32)

    @property
    def boundary_facets(self):
        """(cell, local facet) of the background-boundary facets, ascending facet index."""
        return self._get(L.ARR_BFACETS, (self.nbf, 2), np.int32)

    @property
    def ne(self):
        n = C.c_int64(0)
        L.check(L.lib.phx_mesh_edge_count(self._h, C.byref(n)))
        return n.value

    @property
    def c2e(self):
        """cell -> edges (basix local edge order); builds the edge numbering on first use."""
        nepc = 6 if self.cell_type == "tetrahedron" else 3
        self.ne
        return self._get(L.ARR_C2E, (self.nc, nepc), np.int32)

    @property
    def edges(self):
        """(ne, 2) vertex pairs, ascending."""
        return self._get(L.ARR_EDGES, (self.ne, 2), np.int32)

    def p2_dof_points(self):
        """Coordinates of the P2 nodes: the vertices, then the edge midpoints."""
        x, e = self.x, self.edges
        return np.concatenate([x, 0.5 * (x[e[:, 0]] + x[e[:, 1]])], axis=0)

    def q2_dof_points(self):
        """Coordinates of the Q2 nodes of a quadrilateral mesh: the vertices, the edge midpoints in FACET order,
        the cell centres (the nodal layout `NeumannRobinSolver` and `NodalFunction(degree=2)` use on quadrilaterals)."""
        if self.cell_type != "quadrilateral":
            raise ValueError("q2_dof_points is for quadrilateral meshes")
        x, cells, c2f = self.x, self.cells, self.c2f
        fverts = np.empty((self.nf, 2), dtype=np.int64)
        lfv = np.array([[0, 1], [0, 2], [1, 3], [2, 3]])       # local facet -> local vertices (tensor-product order)
        for lf in range(4):
            fverts[c2f[:, lf]] = cells[:, lfv[lf]]
        return np.concatenate([x, 0.5 * (x[fverts[:, 0]] + x[fverts[:, 1]]), x[cells].mean(axis=1)], axis=0)

    def cell_tag_values(self):
        return self._get(L.ARR_CELL_TAGS, (self.nc,), np.int32)

    def facet_tag_values(self):
        return self._get(L.ARR_FACET_TAGS, (self.nf,), np.int32)

    def synchronize(self):
        L.check(L.lib.phx_mesh_synchronize(self._h))

    def timings(self):
        t = (C.c_double * 8)()
        L.check(L.lib.phx_last_timings(self._h, t))
        return {"tag_cells": t[0], "tag_facets": t[1], "assemble": t[2], "solve": t[3],
                "spmv_avg": t[4]}


def create_box(lo, hi, n, device=0, offset=None, n_global=None):
    """Kuhn simplicial box generated on the device (dolfinx.mesh.create_box / create_rectangle,
    demo/weak-dirichlet/flower/main.py:45-46)."""
    lo = np.ascontiguousarray(lo, dtype=np.float64)
    hi = np.ascontiguousarray(hi, dtype=np.float64)
    n = np.ascontiguousarray(n, dtype=np.int64)
    gdim = lo.size
    off = None if offset is None else np.ascontiguousarray(offset, dtype=np.int64)
    ng = None if n_global is None else np.ascontiguousarray(n_global, dtype=np.int64)
    h = C.c_void_p()
    L.check(L.lib.phx_mesh_create_box(
        gdim, lo.ctypes.data_as(C.c_void_p), hi.ctypes.data_as(C.c_void_p),
        n.ctypes.data_as(C.c_void_p),
        None if off is None else off.ctypes.data_as(C.c_void_p),
        None if ng is None else ng.ctypes.data_as(C.c_void_p), device, C.byref(h)))
    return Mesh(h, device=device)


def create_rectangle(bbox, n, device=0):
    """dolfinx.mesh.create_rectangle(comm, [[x0,y0],[x1,y1]], [nx,ny]) with diagonal 'right'."""
    return create_box(bbox[0], bbox[1], n, device=device)
