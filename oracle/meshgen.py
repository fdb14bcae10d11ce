"""Structured simplicial background meshes (oracle; test infrastructure only).

Stands in for `dolfinx.mesh.create_rectangle` (demo/weak-dirichlet/flower/main.py:45-46,
diagonal "right") and `create_box` [3P].  Numbering contract shared with the HIP library's
device generator (`phx_mesh_create_box`):
  vertex (i,j,k)  -> i + (nx+1)*(j + (ny+1)*k)
  cube   (i,j,k)  -> i + nx*(j + ny*k);  its simplices are 2*cube+t (2-D), 6*cube+t (3-D)
  simplex t is the Kuhn/Freudenthal path  o, o+e_a, o+e_a+e_b(, o+e_a+e_b+e_c)  for the
  t-th axis permutation in lexicographic order.
"""
import itertools

import numpy as np

PERMS2 = [(0, 1), (1, 0)]
PERMS3 = list(itertools.permutations((0, 1, 2)))


def create_box(lo, hi, n, offset=None, n_global=None):
    """n cubes per axis; with offset / n_global the box is a sub-box of a larger lattice and
    reproduces its coordinates bit for bit (multi-GPU slabs):
        x_a(i) = lo_a + (hi_a - lo_a) * ((offset_a + i) / n_global_a)."""
    lo = np.asarray(lo, dtype=np.float64)
    hi = np.asarray(hi, dtype=np.float64)
    n = np.asarray(n, dtype=np.int64)
    d = lo.size
    off = np.zeros(d, dtype=np.int64) if offset is None else np.asarray(offset, dtype=np.int64)
    ng = n if n_global is None else np.asarray(n_global, dtype=np.int64)
    axes = [lo[a] + (hi[a] - lo[a]) * ((off[a] + np.arange(n[a] + 1)) / ng[a]) for a in range(d)]
    if d == 2:
        X, Y = np.meshgrid(axes[0], axes[1], indexing="xy")
        x = np.stack([X.reshape(-1), Y.reshape(-1)], axis=1)
        I, Jx = np.meshgrid(np.arange(n[0]), np.arange(n[1]), indexing="xy")
        base = (I + (n[0] + 1) * Jx).reshape(-1)
        stride = np.array([1, n[0] + 1])
        perms = PERMS2
    else:
        Z, Y, X = np.meshgrid(axes[2], axes[1], axes[0], indexing="ij")
        x = np.stack([X.reshape(-1), Y.reshape(-1), Z.reshape(-1)], axis=1)
        Kz, Jy, I = np.meshgrid(np.arange(n[2]), np.arange(n[1]), np.arange(n[0]), indexing="ij")
        base = (I + (n[0] + 1) * (Jy + (n[1] + 1) * Kz)).reshape(-1)
        stride = np.array([1, n[0] + 1, (n[0] + 1) * (n[1] + 1)])
        perms = PERMS3
    cells = np.empty((base.size, len(perms), d + 1), dtype=np.int64)
    for t, p in enumerate(perms):
        v = base.copy()
        cells[:, t, 0] = v
        for s, a in enumerate(p):
            v = v + stride[a]
            cells[:, t, s + 1] = v
    return x, cells.reshape(-1, d + 1)
