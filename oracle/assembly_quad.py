"""Quadrature-based weak-Dirichlet phi-FEM Poisson for Lagrange degree 1 or 2 (oracle; test infra).

Second, independent restatement of the forms of `demo/weak-dirichlet/flower/main.py:112-151`:
everything is integrated numerically with Stroud conical (Gauss-Jacobi) rules that are exact for
the polynomial degree at hand, the way FFCx would [3P] -- no closed-form simplex integrals.  At
degree 1 it must reproduce `oracle/assembly.py` (closed forms) to round-off, which pins the two
restatements against each other; at degree 2 it is the oracle of the P2 HIP kernels, including
the `div(grad(.))` terms of main.py:123-128,150 that vanish for P1.

PARITY UNPINNED against the reference (it holds no matrix/vector/solution golden).

DoF layout, degree 2: u at vertex v -> v, u at edge e -> nv + e, p block shifted by nd = nv + ne.
Edges are numbered by the lexicographic rank of their sorted vertex pair; local edge k follows
basix: triangle (1,2),(0,2),(0,1); tetrahedron (2,3),(1,3),(1,2),(0,3),(0,2),(0,1).
"""
import math

import numpy as np
import scipy.sparse as sp
from scipy.special import roots_jacobi

from .points import FACET_VERTS

EDGE_VERTS = {
    "triangle": np.array([[1, 2], [0, 2], [0, 1]]),
    "tetrahedron": np.array([[2, 3], [1, 3], [1, 2], [0, 3], [0, 2], [0, 1]]),
}


def build_edges(topo):
    """-> (edge_vertices (ne,2) sorted, c2e (nc, nepc))."""
    ev = EDGE_VERTS[topo.cell_type]
    alle = np.sort(topo.cells[:, ev].reshape(-1, 2), axis=1)
    uniq, inv = np.unique(alle, axis=0, return_inverse=True)
    return uniq, inv.reshape(topo.nc, ev.shape[0]).astype(np.int64)


def simplex_rule(d, degree):
    """Stroud conical product rule on the reference simplex, exact for `degree`.
    Returns barycentric points (nq, d+1) and weights summing to 1 (i.e. per unit volume)."""
    n = degree // 2 + 1
    # collapse one coordinate at a time: x_k = t_k * prod_{m<k} (1 - t_m)
    rules = []
    for k in range(d):
        x, w = roots_jacobi(n, d - 1 - k, 0)
        rules.append(((x + 1.0) / 2.0, w / 2.0 ** (d - k)))
    grids = np.meshgrid(*[r[0] for r in rules], indexing="ij")
    wgrids = np.meshgrid(*[r[1] for r in rules], indexing="ij")
    t = np.stack([g.reshape(-1) for g in grids], axis=1)
    w = np.prod(np.stack([g.reshape(-1) for g in wgrids], axis=1), axis=1)
    xs = np.zeros_like(t)
    rem = np.ones(t.shape[0])
    for k in range(d):
        xs[:, k] = t[:, k] * rem
        rem = rem * (1.0 - t[:, k])
    lam = np.concatenate([1.0 - xs.sum(axis=1, keepdims=True), xs], axis=1)
    w = w / w.sum()
    return lam, w


def lagrange_tab(cell_type, degree, lam):
    """Values N (nq, nb), barycentric gradient coefficients dN (nq, nb, d+1) with
    grad N_b = sum_m dN[q,b,m] g_m, and constant second-derivative coefficients
    H (nb, d+1, d+1): Hess N_b = sum_mn H[b,m,n] g_m g_n^T."""
    n = lam.shape[1]
    if degree == 1:
        N = lam.copy()
        dN = np.broadcast_to(np.eye(n)[None], (lam.shape[0], n, n)).copy()
        H = np.zeros((n, n, n))
        return N, dN, H
    ev = EDGE_VERTS[cell_type]
    nb = n + ev.shape[0]
    N = np.zeros((lam.shape[0], nb))
    dN = np.zeros((lam.shape[0], nb, n))
    H = np.zeros((nb, n, n))
    for i in range(n):
        N[:, i] = lam[:, i] * (2.0 * lam[:, i] - 1.0)
        dN[:, i, i] = 4.0 * lam[:, i] - 1.0
        H[i, i, i] = 4.0
    for k, (a, b) in enumerate(ev):
        N[:, n + k] = 4.0 * lam[:, a] * lam[:, b]
        dN[:, n + k, a] = 4.0 * lam[:, b]
        dN[:, n + k, b] = 4.0 * lam[:, a]
        H[n + k, a, b] = 4.0
        H[n + k, b, a] = 4.0
    return N, dN, H


class Space:
    """Scalar Lagrange space of degree 1 or 2 on a simplicial Topology."""

    def __init__(self, topo, degree):
        self.topo, self.degree = topo, degree
        if degree == 1:
            self.ndofs = topo.nv
            self.cell_dofs = topo.cells
            self.edge_vertices = None
        else:
            self.edge_vertices, c2e = build_edges(topo)
            self.ndofs = topo.nv + self.edge_vertices.shape[0]
            self.cell_dofs = np.concatenate([topo.cells, topo.nv + c2e], axis=1)

    def interpolate(self, f, x):
        """nodal interpolation of f(x) (reference numpy convention x[0], x[1], ...)."""
        pts = x if self.degree == 1 else np.concatenate(
            [x, 0.5 * (x[self.edge_vertices[:, 0]] + x[self.edge_vertices[:, 1]])], axis=0)
        return np.asarray(f(pts.T), dtype=np.float64)

    def dof_points(self, x):
        return x if self.degree == 1 else np.concatenate(
            [x, 0.5 * (x[self.edge_vertices[:, 0]] + x[self.edge_vertices[:, 1]])], axis=0)


def _geometry(x, cells):
    from .assembly import simplex_geometry
    return simplex_geometry(x, cells)


def assemble_poisson_wd_quad(topo, x, cell_tags, facet_tags, ds100, V, Vphi, phi_h, f_h, u_D,
                             pen_coef=1.0, stab_coef=1.0):
    """V: primal = auxiliary Space (degree 1 or 2); Vphi: Space of phi_h.  f_h, u_D live in V.
    Returns (A csr (2 nd x 2 nd), b, active bool)."""
    x = np.asarray(x, dtype=np.float64)
    cells = topo.cells
    d = x.shape[1]
    n = d + 1
    nd = V.ndofs
    k, kp = V.degree, Vphi.degree
    g, vol, h = _geometry(x, cells)
    rows, cols, vals = [], [], []
    b = np.zeros(2 * nd)

    def add(r, c, v):
        rows.append(np.broadcast_to(r, v.shape).reshape(-1))
        cols.append(np.broadcast_to(c, v.shape).reshape(-1))
        vals.append(v.reshape(-1))

    def phys_grad(dN, gc):
        # (nq, nb, n) x (nc, n, d) -> (nc, nq, nb, d)
        return np.einsum("qbm,cmd->cqbd", dN, gc)

    # ---- dx((1,2)): main.py:113, :143
    lam, w = simplex_rule(d, max(2 * (k - 1), 2 * k))
    N, dN, H = lagrange_tab(topo.cell_type, k, lam)
    om = np.flatnonzero((cell_tags == 1) | (cell_tags == 2))
    cd = V.cell_dofs[om]
    G = phys_grad(dN, g[om])
    K = np.einsum("q,c,cqid,cqjd->cij", w, vol[om], G, G)
    add(cd[:, :, None], cd[:, None, :], K)
    fq = np.einsum("qb,cb->cq", N, f_h[cd])
    np.add.at(b, cd, np.einsum("q,c,cq,qi->ci", w, vol[om], fq, N))

    # ---- ds(100): main.py:114   -inner(inner(grad u, n), v)
    ents = np.asarray(ds100, dtype=np.int64).reshape(-1, 2)
    if ents.size:
        flam, fw = simplex_rule(d - 1, 2 * k - 1)
        fv = FACET_VERTS[topo.cell_type]
        for lf in range(n):
            sel = ents[ents[:, 1] == lf, 0]
            if sel.size == 0:
                continue
            lamc = np.zeros((flam.shape[0], n))
            lamc[:, fv[lf]] = flam
            Nf, dNf, _ = lagrange_tab(topo.cell_type, k, lamc)
            gn = np.sqrt((g[sel, lf] ** 2).sum(axis=1))
            nrm = -g[sel, lf] / gn[:, None]
            area = d * vol[sel] * gn
            Gf = phys_grad(dNf, g[sel])
            dn = np.einsum("cqbd,cd->cqb", Gf, nrm)
            M = -np.einsum("q,c,qi,cqj->cij", fw, area, Nf, dn)
            cdf = V.cell_dofs[sel]
            add(cdf[:, :, None], cdf[:, None, :], M)

    # ---- dx(2): penalisation main.py:115-122,144-149 and div(grad) terms :123-128,150
    cut = np.flatnonzero(cell_tags == 2)
    cc = V.cell_dofs[cut]
    cphi = Vphi.cell_dofs[cut]
    lam, w = simplex_rule(d, 2 * k + 2 * kp)
    N, dN, H = lagrange_tab(topo.cell_type, k, lam)
    Np, _, _ = lagrange_tab(topo.cell_type, kp, lam)
    phq = np.einsum("qb,cb->cq", Np, phi_h[cphi])
    hc, vc = h[cut], vol[cut]
    gam = pen_coef
    uu = gam * np.einsum("q,c,qi,qj->cij", w, vc * hc ** -2, N, N)
    up = -gam * np.einsum("q,c,cq,qi,qj->cij", w, vc * hc ** -3, phq, N, N)
    pp = gam * np.einsum("q,c,cq,qi,qj->cij", w, vc * hc ** -4, phq ** 2, N, N)
    add(cc[:, :, None], cc[:, None, :], uu)
    add(cc[:, :, None], nd + cc[:, None, :], up)
    add(nd + cc[:, :, None], cc[:, None, :], up)
    add(nd + cc[:, :, None], nd + cc[:, None, :], pp)
    udq = np.einsum("qb,cb->cq", N, u_D[cc])
    np.add.at(b, cc, gam * np.einsum("q,c,cq,qi->ci", w, vc * hc ** -2, udq, N))
    np.add.at(b, nd + cc, -gam * np.einsum("q,c,cq,cq,qi->ci", w, vc * hc ** -3, udq, phq, N))
    # Laplacians are constant per cell: Lap N_b = sum_mn H[b,m,n] g_m . g_n
    lap = np.einsum("bmn,cmd,cnd->cb", H, g[cut], g[cut])
    add(cc[:, :, None], cc[:, None, :],
        stab_coef * (hc ** 2 * vc)[:, None, None] * lap[:, :, None] * lap[:, None, :])
    fq = np.einsum("qb,cb->cq", N, f_h[cc])
    np.add.at(b, cc, -stab_coef * (hc ** 2 * vc * np.einsum("q,cq->c", w, fq))[:, None] * lap)

    # ---- dS((2,3)): main.py:129-134
    fs = np.flatnonzero(((facet_tags == 2) | (facet_tags == 3)) & (topo.f2c[:, 1] >= 0))
    if fs.size:
        flam, fw = simplex_rule(d - 1, 2 * (k - 1))
        fvt = FACET_VERTS[topo.cell_type]
        cp, cm = topo.f2c[fs, 0], topo.f2c[fs, 1]
        lfp = np.argmax(topo.c2f[cp] == fs[:, None], axis=1)
        lfm = np.argmax(topo.c2f[cm] == fs[:, None], axis=1)
        gnp = np.sqrt((g[cp, lfp] ** 2).sum(axis=1))
        area = d * vol[cp] * gnp
        wgt = stab_coef * 0.5 * (h[cp] + h[cm]) * area
        # quadrature points in PHYSICAL space through the "+" cell, located in both cells by
        # their barycentric coordinates
        nb = V.cell_dofs.shape[1]
        Jall = np.zeros((fs.size, flam.shape[0], 2 * nb))
        xq = np.zeros((fs.size, flam.shape[0], d))
        for lf in range(n):
            sel = np.flatnonzero(lfp == lf)
            if sel.size:
                xq[sel] = np.einsum("qv,cvd->cqd", flam, x[cells[cp[sel]][:, fvt[lf]]])
        for side, (cs, lfs) in enumerate(((cp, lfp), (cm, lfm))):
            xc = x[cells[cs]]                                            # (nf, n, d)
            # barycentric coordinates of xq in cell cs: lam_m = 1_{m=0} + g_m . (x - x_0) pattern
            lamq = np.einsum("cmd,cqd->cqm", g[cs], xq - xc[:, None, 0, :])
            lamq[:, :, 0] += 1.0
            gn = np.sqrt((g[cs, lfs] ** 2).sum(axis=1))
            nrm = -g[cs, lfs] / gn[:, None]
            # tabulate per facet (lam differs per facet on the "-" side): vectorised over q
            for qi in range(flam.shape[0]):
                Nq, dNq, _ = lagrange_tab(topo.cell_type, k, lamq[:, qi, :])   # rows = facets
                Gq = np.einsum("cbm,cmd->cbd", dNq, g[cs])
                Jall[:, qi, side * nb:(side + 1) * nb] = np.einsum("cbd,cd->cb", Gq, nrm)
        dofs = np.concatenate([V.cell_dofs[cp], V.cell_dofs[cm]], axis=1)
        Mf = np.einsum("q,c,cqa,cqb->cab", fw, wgt, Jall, Jall)
        add(dofs[:, :, None], dofs[:, None, :], Mf)

    A = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                      shape=(2 * nd, 2 * nd)).tocsr()
    A.sum_duplicates()
    A.sort_indices()
    active = np.zeros(2 * nd, dtype=bool)
    active[V.cell_dofs[om].reshape(-1)] = True
    active[nd + cc.reshape(-1)] = True
    return A, b, active
