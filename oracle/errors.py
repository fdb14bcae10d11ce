"""Cell-wise discretisation errors (oracle; test infrastructure).

Restates demo/interface-elasticity/main.py:327-383: the exact solution and u_h are interpolated
into the Lagrange space of degree 3 (primal_degree + 2), e = I(u_ex) - I(u_h), and testing
`inner(grad e, grad e)` / `inner(e, e)` with the DG0 basis gives per-cell H1-seminorm and L2
errors; the global relative errors divide by the norms of I(u_ex).

The degree-3 space is basix's default Lagrange variant (GLL-warped [3P], fenics-basix 0.9.0,
absent here): edge-interior nodes at (1 -+ 1/sqrt 5)/2, face nodes at the centroid.  Its basis
is built from the homogeneous cubic monomials in the barycentric coordinates (Vandermonde
inverse) and every integral is CLOSED FORM,
    int_K lambda^alpha = |K| d! alpha! / (d + |alpha|)!,
so nothing is shared with the quadrature of the HIP kernel.

PARITY UNPINNED against the reference (its demos print these numbers but no test or fixture
holds them).
"""
import itertools
import math

import numpy as np

from .assembly import simplex_geometry
from .assembly_quad import EDGE_VERTS, lagrange_tab

G1 = 0.5 * (1.0 - 1.0 / math.sqrt(5.0))
G2 = 0.5 * (1.0 + 1.0 / math.sqrt(5.0))


def reference_nodes(d):
    """Barycentric coordinates (nb, d+1): vertices, two nodes per edge (from the lower to the
    higher local vertex, edges in basix order), one node per face (face f opposite vertex f)."""
    n = d + 1
    rows = [np.eye(n)[i] for i in range(n)]
    ev = EDGE_VERTS["triangle" if d == 2 else "tetrahedron"]
    for a, b in ev:
        for t in (G1, G2):
            r = np.zeros(n)
            r[a], r[b] = 1.0 - t, t
            rows.append(r)
    if d == 2:
        rows.append(np.full(3, 1.0 / 3.0))
    else:
        for f in range(4):
            r = np.full(4, 1.0 / 3.0)
            r[f] = 0.0
            rows.append(r)
    return np.array(rows)


def _monomials(d):
    return [a for a in itertools.product(range(4), repeat=d + 1) if sum(a) == 3]


def _mono_integral(d, alpha):
    """(1/|K|) int_K lambda^alpha"""
    num = math.factorial(d)
    for a in alpha:
        num *= math.factorial(a)
    return num / math.factorial(d + sum(alpha))


def reference_matrices(d):
    """M (nb, nb) = (1/|K|) int N_i N_j and S (d+1, d+1, nb, nb) with
    (1/|K|) int grad N_i . grad N_j = sum_mn S[m, n, i, j] g_m . g_n."""
    lam = reference_nodes(d)
    al = _monomials(d)
    nb = len(al)
    V = np.array([[np.prod(l ** np.array(a)) for a in al] for l in lam])
    C = np.linalg.inv(V)                                   # N_i = sum_a C[a, i] lambda^alpha_a
    Mm = np.array([[_mono_integral(d, tuple(np.add(a, b))) for b in al] for a in al])
    M = C.T @ Mm @ C
    S = np.zeros((d + 1, d + 1, nb, nb))
    for m in range(d + 1):
        for n in range(d + 1):
            Sm = np.zeros((nb, nb))
            for ia, a in enumerate(al):
                if a[m] == 0:
                    continue
                for ib, b in enumerate(al):
                    if b[n] == 0:
                        continue
                    c = list(np.add(a, b))
                    c[m] -= 1
                    c[n] -= 1
                    Sm[ia, ib] = a[m] * b[n] * _mono_integral(d, tuple(c))
            S[m, n] = C.T @ Sm @ C
    return M, S


def cell_errors(topo, x, degree_h, cell_dofs_h, u_h, u_ref, cells=None):
    """u_h: (ncomp, nd) nodal values of degree `degree_h` with cell dofs `cell_dofs_h`;
    u_ref: (ncells, nb, ncomp) exact solution at the reference nodes of the listed cells.
    Returns l2_local, h10_local, norms[4] (sums, then |I u_ex|^2 and |grad I u_ex|^2)."""
    x = np.asarray(x, dtype=np.float64)
    d = x.shape[1]
    cl = np.arange(topo.nc) if cells is None else np.asarray(cells, dtype=np.int64)
    g, vol, _ = simplex_geometry(x, topo.cells)
    g, vol = g[cl], vol[cl]
    GG = np.einsum("cmd,cnd->cmn", g, g)
    M, S = reference_matrices(d)
    Nh, _, _ = lagrange_tab(topo.cell_type, degree_h, reference_nodes(d))     # (nb, nbh)
    uh_nodes = np.einsum("jb,kcb->cjk", Nh, np.asarray(u_h)[:, cell_dofs_h[cl]])   # (ncells, nb, ncomp)
    e = np.asarray(u_ref) - uh_nodes
    K = np.einsum("cmn,mnij->cij", GG, S)
    l2 = vol * np.einsum("cik,ij,cjk->c", e, M, e)
    h10 = vol * np.einsum("cik,cij,cjk->c", e, K, e)
    n2 = vol * np.einsum("cik,ij,cjk->c", u_ref, M, u_ref)
    nh = vol * np.einsum("cik,cij,cjk->c", u_ref, K, u_ref)
    return l2, h10, np.array([l2.sum(), h10.sum(), n2.sum(), nh.sum()])
