"""Detection point sets on reference cells (oracle; test infrastructure only).

Restates `src/phifem/mesh_scripts.py:28-92` (a1) and extends the same construction to
tetrahedra (the reference is 2-D only, `mesh_scripts.py:322-329`).

Arithmetic contract shared with the HIP library (so that both sides are bit-identical):
  * the 1-D lattice is numpy.linspace(0, 1, N+1):  t_i = i * (1.0 / N), t_N = 1.0
  * "1 - t" is one subtraction in double
  * shape functions are evaluated as written in `shape_functions` below, left to right,
    with no fused multiply-add.
"""
import numpy as np

# local facet -> local vertices (basix/dolfinx convention: simplex facet i is opposite
# vertex i; quadrilateral vertices are in tensor-product order)
FACET_VERTS = {
    "interval": np.array([[0], [1]], dtype=np.int32),
    "triangle": np.array([[1, 2], [0, 2], [0, 1]], dtype=np.int32),
    "quadrilateral": np.array([[0, 1], [0, 2], [1, 3], [2, 3]], dtype=np.int32),
    "tetrahedron": np.array([[1, 2, 3], [0, 2, 3], [0, 1, 3], [0, 1, 2]], dtype=np.int32),
}
FACET_TYPE = {"triangle": "interval", "quadrilateral": "interval", "tetrahedron": "triangle"}
TDIM = {"interval": 1, "triangle": 2, "quadrilateral": 2, "tetrahedron": 3}


def lattice_1d(N):
    """numpy.linspace(0,1,N+1) spelled out (mesh_scripts.py:37,52,77)."""
    t = np.arange(N + 1, dtype=np.float64) * (1.0 / N)
    t[N] = 1.0
    return t


def segment_points(N):
    """mesh_scripts.py:28-40: N+1 evenly spaced points, midpoint for N == 0."""
    if N > 0:
        return lattice_1d(N).reshape(-1, 1)
    return np.array([[0.5]])


def triangle_boundary_points(N):
    """mesh_scripts.py:43-65: 3N points walking (0,0)->(1,0)->(0,1)->(0,0)."""
    if N <= 0:
        return np.array([[1.0 / 3.0, 1.0 / 3.0]])
    t = lattice_1d(N)
    pts = [(t[i], 0.0) for i in range(N + 1)]
    pts += [(1.0 - t[i], t[i]) for i in range(1, N + 1)]
    pts += [(0.0, 1.0 - t[i]) for i in range(1, N)]
    return np.array(pts, dtype=np.float64)


def square_boundary_points(N):
    """mesh_scripts.py:68-92: 4N points walking the unit square counter-clockwise."""
    if N <= 0:
        return np.array([[0.5, 0.5]])
    t = lattice_1d(N)
    pts = [(t[i], 0.0) for i in range(N + 1)]
    pts += [(1.0, t[i]) for i in range(1, N + 1)]
    pts += [(1.0 - t[i], 1.0) for i in range(1, N + 1)]
    pts += [(0.0, 1.0 - t[i]) for i in range(1, N)]
    return np.array(pts, dtype=np.float64)


def triangle_lattice_points(N):
    """All degree-N lattice points of the closed reference triangle (3-D facet detection;
    the 3-D analogue of `segment_points`, which is the closed lattice of a 2-D facet).
    Order: j (second coordinate) outer, i inner."""
    if N <= 0:
        return np.array([[1.0 / 3.0, 1.0 / 3.0]])
    t = lattice_1d(N)
    return np.array([(t[i], t[j]) for j in range(N + 1) for i in range(N + 1 - j)],
                    dtype=np.float64)


def tetrahedron_boundary_points(N):
    """Degree-N lattice points on the BOUNDARY of the reference tetrahedron (design
    decision, no reference: in 2-D the cell detection points are the union of the closed
    facet lattices, mesh_scripts.py:43-65; this is the same union for the 4 faces).
    N=1: 4 vertices; N=2: 10; N=3: 20; barycentre for N == 0.
    Order: k outer, j middle, i inner, interior lattice points skipped."""
    if N <= 0:
        return np.array([[0.25, 0.25, 0.25]])
    t = lattice_1d(N)
    pts = []
    for k in range(N + 1):
        for j in range(N + 1 - k):
            for i in range(N + 1 - k - j):
                l = N - i - j - k
                if i == 0 or j == 0 or k == 0 or l == 0:
                    pts.append((t[i], t[j], t[k]))
    return np.array(pts, dtype=np.float64)


def cell_detection_points(cell_type, N):
    if cell_type == "triangle":
        return triangle_boundary_points(N)
    if cell_type == "quadrilateral":
        return square_boundary_points(N)
    if cell_type == "tetrahedron":
        return tetrahedron_boundary_points(N)
    # mesh_scripts.py:326-329
    raise NotImplementedError(
        "Mesh tags computation does not support other cell types than "
        "'triangle', 'quadrilateral' or 'tetrahedron'")


def facet_detection_points(cell_type, N):
    """Points on the reference FACET (mesh_scripts.py:434)."""
    if FACET_TYPE[cell_type] == "interval":
        return segment_points(N)
    return triangle_lattice_points(N)


def shape_functions(cell_type, pts):
    """First-order (vertex) shape functions at reference points -> (npts, nvpc).
    Order of operations is part of the contract (see module docstring)."""
    pts = np.asarray(pts, dtype=np.float64)
    if cell_type == "interval":
        x = pts[:, 0]
        return np.stack([1.0 - x, x], axis=1)
    if cell_type == "triangle":
        x, y = pts[:, 0], pts[:, 1]
        return np.stack([(1.0 - x) - y, x, y], axis=1)
    if cell_type == "tetrahedron":
        x, y, z = pts[:, 0], pts[:, 1], pts[:, 2]
        return np.stack([((1.0 - x) - y) - z, x, y, z], axis=1)
    if cell_type == "quadrilateral":
        x, y = pts[:, 0], pts[:, 1]
        return np.stack([(1.0 - x) * (1.0 - y), x * (1.0 - y), (1.0 - x) * y, x * y], axis=1)
    raise NotImplementedError(cell_type)


def facet_to_cell_points(cell_type, lf, fpts):
    """Map reference-facet points to reference-cell coordinates of local facet `lf`
    (affine through the facet's vertices in FACET_VERTS order)."""
    ref_verts = {
        "triangle": np.array([[0., 0.], [1., 0.], [0., 1.]]),
        "quadrilateral": np.array([[0., 0.], [1., 0.], [0., 1.], [1., 1.]]),
        "tetrahedron": np.array([[0., 0., 0.], [1., 0., 0.], [0., 1., 0.], [0., 0., 1.]]),
    }[cell_type]
    fv = ref_verts[FACET_VERTS[cell_type][lf]]
    N = shape_functions(FACET_TYPE[cell_type], fpts)
    return N @ fv
