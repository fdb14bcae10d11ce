"""Strong-Dirichlet ("direct") phi-FEM Poisson, u_h = phi_h w_h (oracle; test infrastructure).

Restates the forms of `demo/strong-dirichlet/flower/main.py:83-131` on simplices for a primal
(w) space of Lagrange degree 1 or 2 and a level-set space of degree 1 or 2:

    a(w, v) =   int_{cells 1,2} grad(phi w) . grad(phi v)                        main.py:104
              - int_{ds}        (grad(phi w) . n) phi v                          main.py:105
              + sigma h_T^2 int_{cells 2} div grad(phi w) div grad(phi v)        main.py:106-111
              + sigma avg(h_T) int_{facets 2,3} [grad(phi w).n][grad(phi v).n]   main.py:112-117
    L(v)    =   int_{cells 1,2} f phi v - sigma h_T^2 int_{cells 2} f div grad(phi v)   main.py:125-127

with ds = ds_bdy(100) on the background mesh (main.py:60-65) or the whole boundary of the sub-mesh
(main.py:70).  Integration is numerical (Stroud conical rules exact for the degree of each
integrand, as FFCx would generate [3P]); the "basis" psi_i = phi_h N_i is tabulated through the
product rule.  With phi_h = 1 the cell and facet terms reduce to the stiffness and ghost-penalty
blocks of `oracle/assembly.py` (closed forms), which pins this restatement against that one.

PARITY UNPINNED against the reference (no matrix/vector/solution golden exists).
"""
import numpy as np
import scipy.sparse as sp

from .assembly import simplex_geometry
from .assembly_quad import lagrange_tab, simplex_rule
from .points import FACET_VERTS


def _psi(cell_type, k, kp, lam, gc, GGc, phn, V_H=None):
    """Tabulate psi_b = phi N_b for a batch of cells at barycentric points lam (nq, n) -- or per
    cell points (nc, nq, n).  gc (nc, n, d) barycentric gradients, GGc (nc, n, n) their Gram
    matrix, phn (nc, nbp) nodal phi.  Returns val (nc,nq,nb), grad (nc,nq,nb,d), lap (nc,nq,nb)."""
    if lam.ndim == 2:
        lam = np.broadcast_to(lam[None], (gc.shape[0],) + lam.shape)
    nc, nq, n = lam.shape
    flat = lam.reshape(-1, n)
    N, dN, H = lagrange_tab(cell_type, k, flat)
    Np, dNp, Hp = lagrange_tab(cell_type, kp, flat)
    N = N.reshape(nc, nq, -1)
    dN = dN.reshape(nc, nq, -1, n)
    Np = Np.reshape(nc, nq, -1)
    dNp = dNp.reshape(nc, nq, -1, n)
    phq = np.einsum("cqb,cb->cq", Np, phn)
    cphi = np.einsum("cqbm,cb->cqm", dNp, phn)                 # grad phi = sum_m cphi_m g_m
    lphi = np.einsum("bmn,cmn,cb->c", Hp, GGc, phn)            # Laplacian of phi_h (constant)
    lapN = np.einsum("bmn,cmn->cb", H, GGc)
    cpsi = phq[:, :, None, None] * dN + N[:, :, :, None] * cphi[:, :, None, :]
    val = phq[:, :, None] * N
    grad = np.einsum("cqbm,cmd->cqbd", cpsi, gc)
    cross = np.einsum("cqm,cqbn,cmn->cqb", cphi, dN, GGc)
    lap = phq[:, :, None] * lapN[:, None, :] + 2.0 * cross + N * lphi[:, None, None]
    return val, grad, lap


def assemble_poisson_sd(topo, x, cell_tags, facet_tags, ds, V, Vphi, phi_h, f_h, stab_coef=1.0):
    """V: space of w (and of f_h); Vphi: space of phi_h (`assembly_quad.Space`).  ds: flat
    (cell, local facet) pairs.  Returns (A csr nd x nd, b, active bool[nd])."""
    x = np.asarray(x, dtype=np.float64)
    cells = topo.cells
    d = x.shape[1]
    n = d + 1
    nd = V.ndofs
    k, kp = V.degree, Vphi.degree
    ct = topo.cell_type
    g, vol, h = simplex_geometry(x, cells)
    GG = np.einsum("cmd,cnd->cmn", g, g)
    rows, cols, vals = [], [], []
    b = np.zeros(nd)

    def add(r, c, v):
        rows.append(np.broadcast_to(r, v.shape).reshape(-1))
        cols.append(np.broadcast_to(c, v.shape).reshape(-1))
        vals.append(v.reshape(-1))

    # ---- dx((1,2)): main.py:104 and the first term of :125
    lam, w = simplex_rule(d, 2 * k + kp)
    Nf, _, _ = lagrange_tab(ct, k, lam)
    om = np.flatnonzero((cell_tags == 1) | (cell_tags == 2))
    cd = V.cell_dofs[om]
    val, grad, _ = _psi(ct, k, kp, lam, g[om], GG[om], phi_h[Vphi.cell_dofs[om]])
    add(cd[:, :, None], cd[:, None, :], np.einsum("q,c,cqid,cqjd->cij", w, vol[om], grad, grad))
    fq = np.einsum("qb,cb->cq", Nf, f_h[cd])
    np.add.at(b, cd, np.einsum("q,c,cq,cqi->ci", w, vol[om], fq, val))

    # ---- dx(2): main.py:106-111 and the second term of :125-127
    cut = np.flatnonzero(cell_tags == 2)
    if cut.size:
        cc = V.cell_dofs[cut]
        val, _, lap = _psi(ct, k, kp, lam, g[cut], GG[cut], phi_h[Vphi.cell_dofs[cut]])
        sc = stab_coef * h[cut] ** 2 * vol[cut]
        add(cc[:, :, None], cc[:, None, :], np.einsum("q,c,cqi,cqj->cij", w, sc, lap, lap))
        fq = np.einsum("qb,cb->cq", Nf, f_h[cc])
        np.add.at(b, cc, -np.einsum("q,c,cq,cqi->ci", w, sc, fq, lap))

    # ---- ds: main.py:105
    ents = np.asarray(ds, dtype=np.int64).reshape(-1, 2)
    flam, fw = simplex_rule(d - 1, 2 * (k + kp) - 1)
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
        val, grad, _ = _psi(ct, k, kp, lamc, g[sel], GG[sel], phi_h[Vphi.cell_dofs[sel]])
        dn = np.einsum("cqbd,cd->cqb", grad, nrm)
        cdf = V.cell_dofs[sel]
        add(cdf[:, :, None], cdf[:, None, :], -np.einsum("q,c,cqi,cqj->cij", fw, area, val, dn))

    # ---- dS((2,3)): main.py:112-117; jump(v, n) = v+ . n+ + v- . n-
    fs = np.flatnonzero(((facet_tags == 2) | (facet_tags == 3)) & (topo.f2c[:, 1] >= 0))
    if fs.size:
        cp, cm = topo.f2c[fs, 0], topo.f2c[fs, 1]
        lfp = np.argmax(topo.c2f[cp] == fs[:, None], axis=1)
        lfm = np.argmax(topo.c2f[cm] == fs[:, None], axis=1)
        gnp = np.sqrt((g[cp, lfp] ** 2).sum(axis=1))
        area = d * vol[cp] * gnp
        wgt = stab_coef * 0.5 * (h[cp] + h[cm]) * area
        nb = V.cell_dofs.shape[1]
        xq = np.zeros((fs.size, flam.shape[0], d))
        for lf in range(n):
            sel = np.flatnonzero(lfp == lf)
            if sel.size:
                xq[sel] = np.einsum("qv,cvd->cqd", flam, x[cells[cp[sel]][:, fv[lf]]])
        J = np.zeros((fs.size, flam.shape[0], 2 * nb))
        for side, (cs, lfs) in enumerate(((cp, lfp), (cm, lfm))):
            xc = x[cells[cs]]
            lamq = np.einsum("cmd,cqd->cqm", g[cs], xq - xc[:, None, 0, :])
            lamq[:, :, 0] += 1.0
            gn = np.sqrt((g[cs, lfs] ** 2).sum(axis=1))
            nrm = -g[cs, lfs] / gn[:, None]
            _, grad, _ = _psi(ct, k, kp, lamq, g[cs], GG[cs], phi_h[Vphi.cell_dofs[cs]])
            J[:, :, side * nb:(side + 1) * nb] = np.einsum("cqbd,cd->cqb", grad, nrm)
        dofs = np.concatenate([V.cell_dofs[cp], V.cell_dofs[cm]], axis=1)
        add(dofs[:, :, None], dofs[:, None, :], np.einsum("q,c,cqa,cqb->cab", fw, wgt, J, J))

    A = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                      shape=(nd, nd)).tocsr()
    A.sum_duplicates()
    A.sort_indices()
    active = np.zeros(nd, dtype=bool)
    active[V.cell_dofs[om].reshape(-1)] = True
    return A, b, active
