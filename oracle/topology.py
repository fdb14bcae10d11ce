"""Mesh topology on plain arrays (oracle; test infrastructure only).

Stands in for the dolfinx connectivities the reference asks for at
`src/phifem/mesh_scripts.py:151-153,308-315,419-422,430` [3P: libdolfinx].

Numbering contract shared with the HIP library's host builder (`phx_mesh_create`):
  * cells and vertices keep the caller's order;
  * facets are numbered by the lexicographic rank of their sorted vertex tuple;
  * local facet i of a cell follows `points.FACET_VERTS`;
  * f2c lists the incident cells in ascending cell index, -1 padded.
"""
import numpy as np

from .points import FACET_VERTS


def vtk_quads_to_tensor(cells):
    """XDMF/VTK quadrilaterals are cyclic (a,b,c,d); basix wants (a,b,d,c)."""
    cells = np.asarray(cells)
    return cells[:, [0, 1, 3, 2]]


class Topology:
    def __init__(self, cell_type, cells, num_vertices=None):
        cells = np.ascontiguousarray(cells, dtype=np.int64)
        self.cell_type = cell_type
        self.cells = cells
        self.nc, self.nvpc = cells.shape
        self.nv = int(cells.max()) + 1 if num_vertices is None else int(num_vertices)
        fv = FACET_VERTS[cell_type]
        self.nfpc, self.nvpf = fv.shape
        allf = np.sort(cells[:, fv].reshape(-1, self.nvpf), axis=1)
        uniq, inv = np.unique(allf, axis=0, return_inverse=True)
        self.facet_vertices = uniq  # sorted tuple per facet
        self.nf = uniq.shape[0]
        self.c2f = inv.reshape(self.nc, self.nfpc).astype(np.int64)
        # f2c, ascending cell index
        order = np.argsort(self.c2f.reshape(-1), kind="stable")
        fsorted = self.c2f.reshape(-1)[order]
        csorted = order // self.nfpc
        start = np.searchsorted(fsorted, np.arange(self.nf))
        cnt = np.diff(np.append(start, fsorted.size))
        assert cnt.max() <= 2, "non-manifold facet"
        f2c = -np.ones((self.nf, 2), dtype=np.int64)
        f2c[:, 0] = csorted[start]
        two = cnt == 2
        f2c[two, 1] = csorted[start[two] + 1]
        self.f2c = f2c
        self.boundary_facets = np.flatnonzero(cnt == 1)

    def facet_key_map(self):
        """sorted-vertex-tuple -> facet id (numbering-free comparisons in tests)."""
        return {tuple(int(x) for x in v): i for i, v in enumerate(self.facet_vertices)}
