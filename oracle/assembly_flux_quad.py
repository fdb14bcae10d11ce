"""Neumann / Robin phi-FEM Poisson on QUADRILATERALS, mixed (u, y, p) in Q1 x Q1^2 x DG0 with a Q2 level-set
(oracle; test infrastructure).

Restates the forms of `demo/neumann/square/main.py:113-158` (cell type `quadrilateral`, :49-50; primal / vector
degree 1, auxiliary degree 0, level-set degree 2, :36-43) -- the same formulation as `demo/robin/square/main.py:112-168`
with robin_coef = 0 and the gradient-jump term on dS(3):

    a =   int_{1,2} grad u . grad v + u v                                            neumann :114
        + int_{ds}  (y . n) v                                                         :115
        + gamma int_{2} [ (y + grad u).(z + grad v) + (div y + u)(div z + v)          :119-120
                          + h^-2 B(u,y,p) B(v,z,q) ]                                  :121-128
        + sigma avg(h) int_{dS(tag)} [grad u . n][grad v . n]                         :132-135
    B(u,y,p) = y . grad phi - |grad phi| kappa u + h^-1 p phi        (kappa = 0: Neumann)
    L =   int_{1,2} f v + gamma int_{2} [ -h^-2 g |grad phi| B(v,z,q) + f (div z + v) ]   :144-156

Cells are axis-parallel RECTANGLES in tensor-product vertex order v0 = (0,0), v1 = (1,0), v2 = (0,1), v3 = (1,1)
(what `dolfinx.mesh.create_rectangle(..., CellType.quadrilateral)` builds, :50); local facet f0 = (v0,v1),
f1 = (v0,v2), f2 = (v1,v3), f3 = (v2,v3) [3P basix]; h = cell diameter (the diagonal).  Cell integrals: tensor
Gauss rule with `nq` points per direction (the integrands with |grad phi_h| are not polynomial: agreement with
FFCx's Gauss-Jacobi rule [3P] to quadrature accuracy, not round-off); edge integrals: 3-point Gauss (exact).

Q2 level-set nodal layout: [vertex values (nv), edge-midpoint values by FACET id (nf), cell-centre values (nc)].
DoF layout: u at vertex v -> v, y_k at vertex v -> (1 + k) nv + v, p on cell c -> 3 nv + c.
PARITY UNPINNED against the reference (no matrix/vector/solution golden exists).
"""
import numpy as np
import scipy.sparse as sp

FACET_VERTS_Q = np.array([[0, 1], [0, 2], [1, 3], [2, 3]])
# outward normal of local facet lf on an axis-parallel rectangle, and (axis fixed, its reference value)
FACET_NORMAL_Q = np.array([[0.0, -1.0], [-1.0, 0.0], [1.0, 0.0], [0.0, 1.0]])
FACET_FIXED_Q = [(1, 0.0), (0, 0.0), (0, 1.0), (1, 1.0)]


def gauss01(n):
    x, w = np.polynomial.legendre.leggauss(n)
    return 0.5 * (x + 1.0), 0.5 * w


def q1_tab(xi, eta):
    """Q1 basis (npts, 4) and reference gradients (npts, 4, 2) in tensor-product vertex order."""
    N = np.stack([(1 - xi) * (1 - eta), xi * (1 - eta), (1 - xi) * eta, xi * eta], axis=1)
    dN = np.stack([np.stack([-(1 - eta), -(1 - xi)], axis=1), np.stack([(1 - eta), -xi], axis=1),
                   np.stack([-eta, (1 - xi)], axis=1), np.stack([eta, xi], axis=1)], axis=1)
    return N, dN


def _l3(t):
    """1-D quadratic Lagrange basis at nodes 0, 1/2, 1 and its derivative: (npts, 3) each."""
    L = np.stack([2 * (t - 0.5) * (t - 1), 4 * t * (1 - t), 2 * t * (t - 0.5)], axis=1)
    dL = np.stack([4 * t - 3, 4 - 8 * t, 4 * t - 1], axis=1)
    return L, dL


# Q2 local nodes: 4 vertices, 4 edge midpoints (local facet order), centre -> (ix, iy) indices into the 1-D bases
_Q2_IDX = [(0, 0), (2, 0), (0, 2), (2, 2), (1, 0), (0, 1), (2, 1), (1, 2), (1, 1)]


def q2_tab(xi, eta):
    Lx, dLx = _l3(xi)
    Ly, dLy = _l3(eta)
    N = np.stack([Lx[:, a] * Ly[:, b] for a, b in _Q2_IDX], axis=1)
    dN = np.stack([np.stack([dLx[:, a] * Ly[:, b], Lx[:, a] * dLy[:, b]], axis=1) for a, b in _Q2_IDX], axis=1)
    return N, dN


def q2_cell_dofs(topo):
    """(nc, 9) indices into the Q2 nodal array [vertices, facets, cells]."""
    nc = topo.cells.shape[0]
    return np.concatenate([topo.cells, topo.nv + topo.c2f, (topo.nv + topo.nf + np.arange(nc))[:, None]], axis=1)


def q2_dof_points(topo, x):
    """Coordinates of the Q2 nodes in the layout above."""
    fv = topo.facet_vertices
    return np.concatenate([x, 0.5 * (x[fv[:, 0]] + x[fv[:, 1]]), x[topo.cells].mean(axis=1)], axis=0)


def rect_geometry(x, cells):
    """origin, (hx, hy) of axis-parallel rectangles; raises if a cell is not one."""
    X = x[cells]
    o = X[:, 0]
    hx = X[:, 1, 0] - o[:, 0]
    hy = X[:, 2, 1] - o[:, 1]
    ok = (np.abs(X[:, 1, 1] - o[:, 1]) < 1e-12 * np.abs(hx)) & (np.abs(X[:, 2, 0] - o[:, 0]) < 1e-12 * np.abs(hy)) \
        & (np.abs(X[:, 3, 0] - X[:, 1, 0]) < 1e-12 * np.abs(hx)) & (np.abs(X[:, 3, 1] - X[:, 2, 1]) < 1e-12 * np.abs(hy)) \
        & (hx > 0) & (hy > 0)
    if not ok.all():
        raise NotImplementedError("quadrilateral assembly covers axis-parallel rectangles in tensor-product order")
    return o, hx, hy


def assemble_poisson_flux_quad(topo, x, cell_tags, facet_tags, ds, phi_h, f_h, g_h, pen_coef=1.0, stab_coef=1.0,
                               robin_coef=0.0, facet_tag=3, nq=6):
    """phi_h: Q2 nodal values [nv + nf + nc]; f_h, g_h (u_N / u_R): Q1 nodal values.
    Returns (A csr, b, active) over 3 nv + nc DoFs."""
    x = np.asarray(x, dtype=np.float64)
    cells = topo.cells
    nv, nc = topo.nv, cells.shape[0]
    ntot = 3 * nv + nc
    o, hx, hy = rect_geometry(x, cells)
    hT = np.sqrt(hx ** 2 + hy ** 2)
    rows, cols, vals = [], [], []
    b = np.zeros(ntot)

    def add(r, c, v):
        rows.append(np.broadcast_to(r, v.shape).reshape(-1))
        cols.append(np.broadcast_to(c, v.shape).reshape(-1))
        vals.append(np.ascontiguousarray(v).reshape(-1))

    g1, w1 = gauss01(nq)
    xi, eta = np.meshgrid(g1, g1, indexing="ij")
    xi, eta = xi.reshape(-1), eta.reshape(-1)
    wq = (w1[:, None] * w1[None, :]).reshape(-1)
    N, dNr = q1_tab(xi, eta)              # (q,4), (q,4,2)
    N2, dN2r = q2_tab(xi, eta)            # (q,9), (q,9,2)

    def phys(dref, hx_, hy_):
        """reference gradients (q, nb, 2) -> physical (c, q, nb, 2) on rectangles."""
        return dref[None] / np.stack([hx_, hy_], axis=1)[:, None, None, :]

    # ---- dx((1,2)): :114, :144
    om = np.flatnonzero((cell_tags == 1) | (cell_tags == 2))
    cd = cells[om]
    det = hx[om] * hy[om]
    dN = phys(dNr, hx[om], hy[om])
    K = np.einsum("q,c,cqid,cqjd->cij", wq, det, dN, dN) + np.einsum("q,c,qi,qj->cij", wq, det, N, N)
    add(cd[:, :, None], cd[:, None, :], K)
    np.add.at(b, cd, np.einsum("q,c,qj,cj,qi->ci", wq, det, N, f_h[cd], N))

    # ---- ds: :115   (y . n) v over the (cell, local facet) pairs
    ents = np.asarray(ds, dtype=np.int64).reshape(-1, 2)
    e1, ew = gauss01(3)
    for lf in range(4):
        sel = ents[ents[:, 1] == lf, 0]
        if sel.size == 0:
            continue
        ax, val = FACET_FIXED_Q[lf]
        xe = np.full(3, val) if ax == 0 else e1
        ye = np.full(3, val) if ax == 1 else e1
        Ne, _ = q1_tab(xe, ye)
        length = hy[sel] if ax == 0 else hx[sel]
        Mf = np.einsum("q,c,qi,qj->cij", ew, length, Ne, Ne)
        nrm = FACET_NORMAL_Q[lf]
        cdf = cells[sel]
        for k in range(2):
            if nrm[k] != 0.0:
                add(cdf[:, :, None], (1 + k) * nv + cdf[:, None, :], Mf * nrm[k])

    # ---- dx(2): :117-130, :145-156
    cut = np.flatnonzero(cell_tags == 2)
    if cut.size:
        cc = cells[cut]
        det = hx[cut] * hy[cut]
        hc = hT[cut]
        dN = phys(dNr, hx[cut], hy[cut])          # (c,q,4,2)
        dN2 = phys(dN2r, hx[cut], hy[cut])        # (c,q,9,2)
        phn = phi_h[q2_cell_dofs(topo)[cut]]      # (c,9)
        phq = np.einsum("qb,cb->cq", N2, phn)
        gphi = np.einsum("cqbd,cb->cqd", dN2, phn)
        ngp = np.sqrt((gphi ** 2).sum(axis=2))
        M = 13
        ncut, nqq = cut.size, wq.size
        U = np.zeros((ncut, nqq, M))
        T1 = np.zeros((ncut, nqq, M, 2))      # y + grad u
        DY = np.zeros((ncut, nqq, M))
        B = np.zeros((ncut, nqq, M))
        dofs = np.zeros((ncut, M), dtype=np.int64)
        for i in range(4):
            U[:, :, i] = N[None, :, i]
            T1[:, :, i, :] = dN[:, :, i, :]
            B[:, :, i] = -robin_coef * ngp * N[None, :, i]
            dofs[:, i] = cc[:, i]
            for k in range(2):
                a = 4 + 4 * k + i
                T1[:, :, a, k] = N[None, :, i]
                DY[:, :, a] = dN[:, :, i, k]
                B[:, :, a] = N[None, :, i] * gphi[:, :, k]
                dofs[:, a] = (1 + k) * nv + cc[:, i]
        B[:, :, M - 1] = phq / hc[:, None]
        dofs[:, M - 1] = 3 * nv + cut
        T2 = DY + U
        E = np.einsum("q,c,cqad,cqbd->cab", wq, det, T1, T1) + np.einsum("q,c,cqa,cqb->cab", wq, det, T2, T2) \
            + np.einsum("q,c,cqa,cqb->cab", wq, det * hc ** -2, B, B)
        add(dofs[:, :, None], dofs[:, None, :], pen_coef * E)
        gq = np.einsum("qi,ci->cq", N, g_h[cc])
        fq = np.einsum("qi,ci->cq", N, f_h[cc])
        r = -np.einsum("q,c,cq,cq,cqa->ca", wq, det * hc ** -2, gq, ngp, B) + np.einsum("q,c,cq,cqa->ca", wq, det, fq, T2)
        np.add.at(b, dofs, pen_coef * r)

    # ---- dS(facet_tag): :132-135   sigma avg(h) [grad u . n][grad v . n]
    fs = np.flatnonzero((facet_tags == facet_tag) & (topo.f2c[:, 1] >= 0))
    if fs.size:
        cp, cm = topo.f2c[fs, 0], topo.f2c[fs, 1]
        lfp = np.argmax(topo.c2f[cp] == fs[:, None], axis=1)
        lfm = np.argmax(topo.c2f[cm] == fs[:, None], axis=1)
        J = np.zeros((fs.size, 3, 8))            # normal derivative of the 8 local functions at the 3 edge points
        length = np.zeros(fs.size)
        for side, (cs, lfs) in enumerate(((cp, lfp), (cm, lfm))):
            for lf in range(4):
                m = np.flatnonzero(lfs == lf)
                if m.size == 0:
                    continue
                ax, val = FACET_FIXED_Q[lf]
                xe = np.full(3, val) if ax == 0 else e1
                ye = np.full(3, val) if ax == 1 else e1
                _, dNe = q1_tab(xe, ye)                       # (3,4,2)
                c_ = cs[m]
                dNp = dNe[None] / np.stack([hx[c_], hy[c_]], axis=1)[:, None, None, :]
                J[m, :, side * 4:(side + 1) * 4] = np.einsum("cqid,d->cqi", dNp, FACET_NORMAL_Q[lf])
                if side == 0:
                    length[m] = hy[c_] if ax == 0 else hx[c_]
        # both sides parametrise the shared edge in the same direction (tensor-product order on rectangles)
        wgt = stab_coef * 0.5 * (hT[cp] + hT[cm]) * length
        dofs = np.concatenate([cells[cp], cells[cm]], axis=1)
        add(dofs[:, :, None], dofs[:, None, :], np.einsum("q,c,cqa,cqb->cab", ew, wgt, J, J))

    A = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(ntot, ntot)).tocsr()
    A.sum_duplicates()
    A.sort_indices()
    active = np.zeros(ntot, dtype=bool)
    active[cd.reshape(-1)] = True
    if cut.size:
        for k in range(2):
            active[(1 + k) * nv + cells[cut].reshape(-1)] = True
        active[3 * nv + cut] = True
    return A, b, active
