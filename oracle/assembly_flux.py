"""Neumann / Robin phi-FEM Poisson, mixed (u, y, p) in P1 x P1^d x DG0 with a P2 level-set
(oracle; test infrastructure).

Restates the forms of `demo/robin/square/main.py:112-168` (triangles; the Neumann demo,
`demo/neumann/square/main.py:113-158`, is the same formulation with robin_coef = 0, the facet term
on dS(3) instead of dS(2), and quadrilateral cells, which this restatement does not cover):

    a =   int_{1,2} grad u . grad v + u v                                           robin :115
        + int_{ds}  (y . n) v                                                        :116
        + gamma int_{2} [ (y + grad u).(z + grad v) + (div y + u)(div z + v)         :120-121
                          + h^-2 B(u,y,p) B(v,z,q) ]                                 :122-131
        + sigma avg(h) int_{dS(tag)} [grad u . n][grad v . n]                        :135-143
    B(u,y,p) = y . grad phi - |grad phi| kappa u + h^-1 p phi
    L =   int_{1,2} f v + gamma int_{2} [ -h^-2 g |grad phi| B(v,z,q) + f (div z + v) ]   :151-166

|grad phi_h| is not a polynomial: the integrals containing it depend on the quadrature rule.  UFL
estimates degree 10 for the Robin integrand (sqrt adds 2) and FFCx would take basix's Xiao-Gimbutas
rule of that degree [3P]; here every cell integral uses the Stroud conical rule of degree `qdeg`
(default 10), so the result matches dolfinx to quadrature accuracy, not to round-off.

DoF layout: u at vertex v -> v, y_k at vertex v -> (1 + k) nv + v, p on cell c -> (1 + d) nv + c.
PARITY UNPINNED against the reference (no matrix/vector/solution golden exists).
"""
import numpy as np
import scipy.sparse as sp

from .assembly import simplex_geometry
from .assembly_quad import lagrange_tab, simplex_rule
from .points import FACET_VERTS


def assemble_poisson_flux(topo, x, cell_tags, facet_tags, ds, Vphi, phi_h, f_h, g_h, pen_coef=1.0,
                          stab_coef=1.0, robin_coef=0.0, facet_tag=2, qdeg=10):
    """Vphi: degree-2 `assembly_quad.Space` of phi_h; f_h, g_h (u_N or u_R): P1 nodal values.
    Returns (A csr, b, active) over (1 + d) nv + nc DoFs."""
    x = np.asarray(x, dtype=np.float64)
    cells = topo.cells
    d = x.shape[1]
    n = d + 1
    nv, nc = topo.nv, topo.nc
    ntot = (1 + d) * nv + nc
    ct = topo.cell_type
    g, vol, h = simplex_geometry(x, cells)
    rows, cols, vals = [], [], []
    b = np.zeros(ntot)

    def add(r, c, v):
        rows.append(np.broadcast_to(r, v.shape).reshape(-1))
        cols.append(np.broadcast_to(c, v.shape).reshape(-1))
        vals.append(v.reshape(-1))

    # ---- dx((1,2)): robin :115, :151
    lam2, w2 = simplex_rule(d, 2)
    om = np.flatnonzero((cell_tags == 1) | (cell_tags == 2))
    cd = cells[om]
    K = np.einsum("c,cid,cjd->cij", vol[om], g[om], g[om]) + np.einsum("q,c,qi,qj->cij", w2, vol[om], lam2, lam2)
    add(cd[:, :, None], cd[:, None, :], K)
    np.add.at(b, cd, np.einsum("q,c,qj,cj,qi->ci", w2, vol[om], lam2, f_h[cd], lam2))

    # ---- ds: :116   (y . n) v
    ents = np.asarray(ds, dtype=np.int64).reshape(-1, 2)
    flam, fw = simplex_rule(d - 1, 2)
    fv = FACET_VERTS[ct]
    for lf in range(n):
        sel = ents[ents[:, 1] == lf, 0]
        if sel.size == 0:
            continue
        lamc = np.zeros((flam.shape[0], n))
        lamc[:, fv[lf]] = flam
        gn = np.sqrt((g[sel, lf] ** 2).sum(axis=1))
        nrm = -g[sel, lf] / gn[:, None]
        area = d * vol[sel] * gn
        Mf = np.einsum("q,c,qi,qj->cij", fw, area, lamc, lamc)
        cdf = cells[sel]
        for k in range(d):
            add(cdf[:, :, None], (1 + k) * nv + cdf[:, None, :], Mf * nrm[:, k, None, None])

    # ---- dx(2): :118-133, :152-165
    cut = np.flatnonzero(cell_tags == 2)
    if cut.size:
        lam, w = simplex_rule(d, qdeg)
        nq = lam.shape[0]
        cc = cells[cut]
        gc, vc, hc = g[cut], vol[cut], h[cut]
        Np, dNp, _ = lagrange_tab(ct, 2, lam)
        phn = phi_h[Vphi.cell_dofs[cut]]
        phq = np.einsum("qb,cb->cq", Np, phn)
        gphi = np.einsum("qbm,cb,cmd->cqd", dNp, phn, gc)
        ngp = np.sqrt((gphi ** 2).sum(axis=2))
        M = n * (1 + d) + 1
        ncut = cut.size
        # per local DoF: value of u, of y (vector), grad u (vector), div y, and B
        U = np.zeros((ncut, nq, M))
        Y = np.zeros((ncut, nq, M, d))
        GU = np.zeros((ncut, nq, M, d))
        DY = np.zeros((ncut, nq, M))
        B = np.zeros((ncut, nq, M))
        dofs = np.zeros((ncut, M), dtype=np.int64)
        for i in range(n):
            U[:, :, i] = lam[None, :, i]
            GU[:, :, i, :] = gc[:, None, i, :]
            B[:, :, i] = -robin_coef * ngp * lam[None, :, i]
            dofs[:, i] = cc[:, i]
            for k in range(d):
                a = n + k * n + i
                Y[:, :, a, k] = lam[None, :, i]
                DY[:, :, a] = gc[:, None, i, k]
                B[:, :, a] = lam[None, :, i] * gphi[:, :, k]
                dofs[:, a] = (1 + k) * nv + cc[:, i]
        B[:, :, M - 1] = phq / hc[:, None]
        dofs[:, M - 1] = (1 + d) * nv + cut
        T1 = Y + GU
        T2 = DY + U
        E = np.einsum("q,c,cqad,cqbd->cab", w, vc, T1, T1) + np.einsum("q,c,cqa,cqb->cab", w, vc, T2, T2) \
            + np.einsum("q,c,cqa,cqb->cab", w, vc * hc ** -2, B, B)
        add(dofs[:, :, None], dofs[:, None, :], pen_coef * E)
        gq = np.einsum("qi,ci->cq", lam, g_h[cc])
        fq = np.einsum("qi,ci->cq", lam, f_h[cc])
        r = -np.einsum("q,c,cq,cq,cqa->ca", w, vc * hc ** -2, gq, ngp, B) + np.einsum("q,c,cq,cqa->ca", w, vc, fq, T2)
        np.add.at(b, dofs, pen_coef * r)

    # ---- dS(facet_tag): :135-143
    fs = np.flatnonzero((facet_tags == facet_tag) & (topo.f2c[:, 1] >= 0))
    if fs.size:
        cp, cm = topo.f2c[fs, 0], topo.f2c[fs, 1]
        lfp = np.argmax(topo.c2f[cp] == fs[:, None], axis=1)
        lfm = np.argmax(topo.c2f[cm] == fs[:, None], axis=1)
        gnp = np.sqrt((g[cp, lfp] ** 2).sum(axis=1))
        area = d * vol[cp] * gnp
        wgt = stab_coef * 0.5 * (h[cp] + h[cm]) * area
        J = np.zeros((fs.size, 2 * n))
        for side, (cs, lfs) in enumerate(((cp, lfp), (cm, lfm))):
            gn = np.sqrt((g[cs, lfs] ** 2).sum(axis=1))
            nrm = -g[cs, lfs] / gn[:, None]
            J[:, side * n:(side + 1) * n] = np.einsum("cid,cd->ci", g[cs], nrm)
        dofs = np.concatenate([cells[cp], cells[cm]], axis=1)
        add(dofs[:, :, None], dofs[:, None, :], wgt[:, None, None] * J[:, :, None] * J[:, None, :])

    A = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                      shape=(ntot, ntot)).tocsr()
    A.sum_duplicates()
    A.sort_indices()
    active = np.zeros(ntot, dtype=bool)
    active[cd.reshape(-1)] = True
    if cut.size:
        for k in range(d):
            active[(1 + k) * nv + cells[cut].reshape(-1)] = True
        active[(1 + d) * nv + cut] = True
    return A, b, active
