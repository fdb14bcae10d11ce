"""Discretisation-error post-processing on the GPU.

Mirrors the error block the reference writes out in its convergence demo,
demo/interface-elasticity/main.py:327-383: interpolate the exact solution and u_h into the
Lagrange space of degree 3, integrate |e|^2 and |grad e|^2 cell by cell (the DG0-tested forms
`l2_local` / `h10_local`), and normalise the global errors by the norms of the interpolated exact
solution.  The integration runs in `phx_cell_errors`; the host only evaluates the user's
`exact_solution` callable at the reference nodes.
"""
import ctypes as C

import numpy as np

from . import _lib as L


def reference_nodes(gdim, degree=3):
    """Barycentric coordinates (n_nodes, gdim + 1) of the reference-space nodes of one cell."""
    n = C.c_int()
    L.check(L.lib.phx_reference_nodes(gdim, degree, None, C.byref(n)))
    out = np.empty((n.value, gdim + 1), dtype=np.float64)
    L.check(L.lib.phx_reference_nodes(gdim, degree, out.ctypes.data_as(C.c_void_p), C.byref(n)))
    return out


def cell_errors(mesh, u_h, exact_solution, degree=1, cells=None):
    """u_h: nodal values (nd,) or (nd, ncomp) of Lagrange degree `degree` (vertices, then edges at
    degree 2); exact_solution: callable on points x of shape (gdim, npts), as the reference's
    `exact_solution(x)`, returning (npts,) or (ncomp, npts); cells: cell indices (default: all).

    Returns a dict: `l2_local`, `h10_local` (one value per listed cell, main.py:356,377),
    `l2_relative`, `h10_relative` (main.py:360,381), and the four integrals."""
    gdim = mesh.gdim
    lam = reference_nodes(gdim)
    nb = lam.shape[0]
    cl = None if cells is None else np.ascontiguousarray(cells, dtype=np.int32)
    if cl is not None and cl.size and (cl.min() < 0 or cl.max() >= mesh.nc):
        raise ValueError("cell index out of range")
    cv = mesh.cells if cl is None else mesh.cells[cl]
    pts = np.einsum("jm,cmd->cjd", lam, mesh.x[cv]).reshape(-1, gdim)
    ue = np.asarray(exact_solution(pts.T), dtype=np.float64)
    ue = ue.reshape(1, -1) if ue.ndim == 1 else ue
    ncomp = ue.shape[0]
    u_ref = np.ascontiguousarray(ue.T.reshape(cv.shape[0], nb, ncomp))
    u_h = np.asarray(u_h, dtype=np.float64)
    u_cm = np.ascontiguousarray(u_h.reshape(u_h.shape[0], -1).T)          # component-major
    nd = mesh.nv if degree == 1 else mesh.nv + mesh.ne
    if u_cm.shape != (ncomp, nd):
        raise ValueError(f"u_h has shape {u_h.shape}, expected ({nd},) or ({nd}, {ncomp})")
    ncl = cv.shape[0]
    l2 = np.empty(ncl, dtype=np.float64)
    h10 = np.empty(ncl, dtype=np.float64)
    norms = (C.c_double * 4)()
    vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    L.check(L.lib.phx_cell_errors(mesh._h, ncomp, degree, vp(u_cm), vp(u_ref), ncl,
                                  None if cl is None else vp(cl), L.HOST, vp(l2), vp(h10), norms))
    return {"l2_local": l2, "h10_local": h10, "l2_sum": norms[0], "h10_sum": norms[1],
            "l2_norm_exact": norms[2], "h10_norm_exact": norms[3],
            "l2_relative": float(np.sqrt(norms[0] / norms[2])) if norms[2] > 0 else float("nan"),
            "h10_relative": float(np.sqrt(norms[1] / norms[3])) if norms[3] > 0 else float("nan")}
