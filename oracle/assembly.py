"""Weak-Dirichlet phi-FEM Poisson: element tensors, CSR, active set (oracle; test infra only).

numpy restatement of the forms of `demo/weak-dirichlet/flower/main.py`:
  bilinear form  main.py:112-135   (a9)
  linear form    main.py:142-151   (a10)
on affine simplices (triangles, tetrahedra) with P1 x P1 mixed (u, p).  All integrands are
polynomials on affine cells, so the closed-form simplex integrals used here equal any exact
quadrature FFCx would generate [3P], up to round-off.

PARITY UNPINNED: the reference holds no golden matrix/vector/solution.

DoF layout (contract shared with the HIP library): u at vertex v -> v, p at vertex v -> nv + v.
"""
import math

import numpy as np
import scipy.sparse as sp

from .points import FACET_VERTS


def simplex_geometry(x, cells):
    """grads (nc, d+1, d) of the barycentric coordinates, volume (nc,), diameter (nc,)."""
    xc = x[cells]                                  # (nc, d+1, d)
    d = xc.shape[2]
    J = np.transpose(xc[:, 1:, :] - xc[:, 0:1, :], (0, 2, 1))   # columns = edge vectors
    detJ = np.linalg.det(J)
    Jinv = np.linalg.inv(J)                        # rows = grad lambda_{1..d}
    g = np.empty((cells.shape[0], d + 1, d))
    g[:, 1:, :] = Jinv
    g[:, 0, :] = -Jinv.sum(axis=1)
    vol = np.abs(detJ) / math.factorial(d)
    h = np.zeros(cells.shape[0])
    for i in range(d + 1):
        for j in range(i + 1, d + 1):
            # ufl.CellDiameter (main.py:108): largest vertex-vertex distance
            h = np.maximum(h, np.sqrt(((xc[:, i] - xc[:, j]) ** 2).sum(axis=1)))
    return g, vol, h


def _bary_tensor(d, order):
    """T[i,j,..] = (1/|K|) int_K N_i N_j ... = d! alpha! / (d+order)!."""
    n = d + 1
    T = np.zeros((n,) * order)
    for idx in np.ndindex(*T.shape):
        alpha = np.bincount(idx, minlength=n)
        T[idx] = math.factorial(d) * np.prod([math.factorial(a) for a in alpha]) \
            / math.factorial(d + order)
    return T


class WeakDirichletSystem:
    def __init__(self, A, b, active, nv):
        self.A, self.b, self.active, self.nv = A, b, active, nv


def assemble_poisson_wd(topo, x, cell_tags, facet_tags, ds100, phi_h, f_h, u_D,
                        pen_coef=1.0, stab_coef=1.0):
    """Returns (A csr (2nv x 2nv), b (2nv), active bool (2nv)).
    cell_tags / facet_tags: dense int arrays; ds100: flat [cell, local facet, ...]."""
    x = np.asarray(x, dtype=np.float64)
    cells = topo.cells
    nv = topo.nv
    d = x.shape[1]
    n = d + 1
    g, vol, h = simplex_geometry(x, cells)
    M2, M3, M4 = _bary_tensor(d, 2), _bary_tensor(d, 3), _bary_tensor(d, 4)
    rows, cols, vals = [], [], []
    b = np.zeros(2 * nv)

    def add(r, c, v):
        rows.append(np.broadcast_to(r, v.shape).reshape(-1))
        cols.append(np.broadcast_to(c, v.shape).reshape(-1))
        vals.append(v.reshape(-1))

    # ---- main.py:113  inner(grad u, grad v) dx((1,2)) ; main.py:143 inner(f_h, v) dx((1,2))
    om = np.flatnonzero((cell_tags == 1) | (cell_tags == 2))
    cv = cells[om]
    K = vol[om, None, None] * np.einsum("cid,cjd->cij", g[om], g[om])
    add(cv[:, :, None], cv[:, None, :], K)
    np.add.at(b, cv, vol[om, None] * np.einsum("ij,cj->ci", M2, f_h[cv]))

    # ---- main.py:114  -inner(inner(grad u, n), v) ds(100)
    ents = np.asarray(ds100, dtype=np.int64).reshape(-1, 2)
    if ents.size:
        c, lf = ents[:, 0], ents[:, 1]
        # n = -g_lf/|g_lf|, |F| = d vol |g_lf|, int_F N_i = |F|/d  (i on F)
        coef = vol[c, None] * np.einsum("cjd,cd->cj", g[c], g[c, lf])     # (ne, n) over j
        for i in range(n):
            on_f = lf != i
            add(cells[c[on_f], i][:, None], cells[c[on_f]], coef[on_f])

    # ---- main.py:115-122,144-149  penalisation on cut cells
    cut = np.flatnonzero(cell_tags == 2)
    cc = cells[cut]
    hc = h[cut]
    ph = phi_h[cc]
    gam = pen_coef
    uu = gam * (hc ** -2 * vol[cut])[:, None, None] * M2[None]
    up = -gam * (hc ** -3 * vol[cut])[:, None, None] * np.einsum("ijk,ck->cij", M3, ph)
    pp = gam * (hc ** -4 * vol[cut])[:, None, None] * np.einsum("ijkl,ck,cl->cij", M4, ph, ph)
    add(cc[:, :, None], cc[:, None, :], uu)
    add(cc[:, :, None], nv + cc[:, None, :], up)
    add(nv + cc[:, :, None], cc[:, None, :], up)
    add(nv + cc[:, :, None], nv + cc[:, None, :], pp)
    ud = u_D[cc]
    np.add.at(b, cc, gam * (hc ** -2 * vol[cut])[:, None] * np.einsum("ij,cj->ci", M2, ud))
    np.add.at(b, nv + cc, -gam * (hc ** -3 * vol[cut])[:, None]
              * np.einsum("ijk,cj,ck->ci", M3, ud, ph))
    # main.py:123-128,150: div(grad(.)) of a P1 function vanishes identically.

    # ---- main.py:129-134  avg(h) jump(grad u, n) jump(grad v, n) dS((2,3))
    fs = np.flatnonzero(((facet_tags == 2) | (facet_tags == 3)) & (topo.f2c[:, 1] >= 0))
    if fs.size:
        cp, cm = topo.f2c[fs, 0], topo.f2c[fs, 1]
        lfp = np.argmax(topo.c2f[cp] == fs[:, None], axis=1)
        lfm = np.argmax(topo.c2f[cm] == fs[:, None], axis=1)
        ar = np.arange(fs.size)
        gnp = np.sqrt((g[cp, lfp] ** 2).sum(axis=1))
        gnm = np.sqrt((g[cm, lfm] ** 2).sum(axis=1))
        area = d * vol[cp] * gnp
        npl = -g[cp, lfp] / gnp[:, None]
        nmi = -g[cm, lfm] / gnm[:, None]
        Jp = np.einsum("cjd,cd->cj", g[cp], npl)
        Jm = np.einsum("cjd,cd->cj", g[cm], nmi)
        Jall = np.concatenate([Jp, Jm], axis=1)                 # (nfs, 2n)
        dofs = np.concatenate([cells[cp], cells[cm]], axis=1)
        w = stab_coef * 0.5 * (h[cp] + h[cm]) * area
        add(dofs[:, :, None], dofs[:, None, :], w[:, None, None] * Jall[:, :, None] * Jall[:, None, :])

    A = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                      shape=(2 * nv, 2 * nv)).tocsr()
    A.sum_duplicates()
    A.sort_indices()
    active = np.zeros(2 * nv, dtype=bool)
    active[cells[om].reshape(-1)] = True
    active[nv + cc.reshape(-1)] = True
    return A, b, active


def solve_direct(A, b, active):
    """main.py:162-182: MUMPS LU with null-pivot detection [3P] returns the solution with the
    null-space components at zero == solve on the active DoFs, zero elsewhere."""
    import scipy.sparse.linalg as spla
    idx = np.flatnonzero(active)
    Aa = A[idx][:, idx].tocsc()
    xa = spla.spsolve(Aa, b[idx])
    xfull = np.zeros_like(b)
    xfull[idx] = xa
    return xfull
