"""Interface linear elasticity, 5-field mixed phi-FEM (oracle; test infrastructure only).

numpy restatement of `demo/interface-elasticity/main.py:145-277` (a12): unknowns
(u_in, u_out, y_in, y_out, p), all first-order Lagrange (param1.yaml:11-15), on affine simplices:
  stiffness_in / stiffness_out            main.py:185-186, 226-227   dx((1,2)) / dx((2,3))
  penalization                            main.py:188-203, 228       dx(2)
  stabilization_facets_in / _out          main.py:205-209, 219-223, 229-230   dS(3) / dS(4)
  stabilization_cells_in / _out           main.py:211-217, 231-232   dx(2)
  boundary_in / boundary_out              main.py:182-183, 233-234   d_bdry(100) / d_bdry(101)
  right-hand side                         main.py:255-269
  Dirichlet rows of u_in on the box       main.py:158-177, 237-239, 275-277
  material law                            data.py:5-36
Every integrand is a polynomial on an affine cell, so the closed-form simplex integrals below
equal what an exact FFCx quadrature gives [3P], up to round-off.  The source `f` is taken as a
P1 nodal vector field (the demo integrates a UFL expression; the bench problems supply nodal data).

PARITY UNPINNED: the reference holds no matrix / vector / solution golden.

DoF layout: component-major blocks of nv entries:
  u_in[a] -> a,  u_out[a] -> d + a,  y_in[a][b] -> 2d + a d + b,  y_out[a][b] -> 2d + d^2 + a d + b,
  p[a] -> 2d + 2 d^2 + a;   global index = block * nv + vertex.
"""
import numpy as np
import scipy.sparse as sp

from .assembly import _bary_tensor, simplex_geometry
from .points import FACET_VERTS


def lame(E, nu):
    """data.py:5-10."""
    return E * nu / (1.0 + nu) / (1.0 - 2.0 * nu), E / 2.0 / (1.0 + nu)


class Blocks:
    def __init__(self, d):
        self.d = d
        self.uin, self.uout = 0, d
        self.yin, self.yout = 2 * d, 2 * d + d * d
        self.p = 2 * d + 2 * d * d
        self.C = 2 * d + 2 * d * d + d

    def u(self, side, a):
        return (self.uin if side == 0 else self.uout) + a

    def y(self, side, a, b):
        return (self.yin if side == 0 else self.yout) + a * self.d + b


def sigma_basis(g, lam, mu):
    """S[c, j, b] = sigma(N_j e_b) = lam g_{j,b} I + mu (e_b (x) g_j + g_j (x) e_b)  -> (nc,n,d,d,d)."""
    nc, n, d = g.shape
    S = np.zeros((nc, n, d, d, d))
    eye = np.eye(d)
    for b in range(d):
        S[:, :, b] += lam * g[:, :, b, None, None] * eye[None, None]
        S[:, :, b, b, :] += mu * g
        S[:, :, b, :, b] += mu * g
    return S


def assemble_elasticity_if(topo, x, cell_tags, facet_tags, ds100, ds101, phi_h, f_h, uD, bc_vertices,
                           E_in=1.0, nu_in=0.3, E_out=1.0e-3, nu_out=0.3, pen_coef=1.0, stab_coef=1.0):
    """f_h, uD: (nv, d) nodal vector fields.  bc_vertices: vertices carrying the Dirichlet
    condition u_in = uD.  Returns (A csr, b, active bool) over C*nv DoFs."""
    x = np.asarray(x, dtype=np.float64)
    cells = topo.cells
    nv = topo.nv
    d = x.shape[1]
    n = d + 1
    B = Blocks(d)
    g, vol, h = simplex_geometry(x, cells)
    M2, M3, M4 = _bary_tensor(d, 2), _bary_tensor(d, 3), _bary_tensor(d, 4)
    lam = [lame(E_in, nu_in)[0], lame(E_out, nu_out)[0]]
    mu = [lame(E_in, nu_in)[1], lame(E_out, nu_out)[1]]
    coef_in = (E_in / (E_in + E_out)) ** 2     # main.py:188
    coef_out = (E_out / (E_in + E_out)) ** 2   # main.py:189
    rows, cols, vals = [], [], []
    b = np.zeros(B.C * nv)

    def add(blk_r, vr, blk_c, vc, v):
        """v: (ne, nr, ncol) with rows on vertices vr (ne, nr), columns on vc (ne, ncol)."""
        r = blk_r * nv + vr[:, :, None]
        c = blk_c * nv + vc[:, None, :]
        rows.append(np.broadcast_to(r, v.shape).reshape(-1))
        cols.append(np.broadcast_to(c, v.shape).reshape(-1))
        vals.append(np.ascontiguousarray(v).reshape(-1))

    # ---- stiffness, main.py:185-186 on dx((1,2)) / dx((2,3)); rhs main.py:263-264
    for side, tags in ((0, (1, 2)), (1, (2, 3))):
        sel = np.flatnonzero(np.isin(cell_tags, tags))
        cv = cells[sel]
        S = sigma_basis(g[sel], lam[side], mu[side])
        for a in range(d):
            # eps(N_i e_a) : S[j,c] = 0.5 (S[j,c,a,:] + S[j,c,:,a]) . g_i = S[j,c,a,:] . g_i  (S symmetric)
            for c in range(d):
                K = vol[sel, None, None] * np.einsum("cjq,ciq->cij", S[:, :, c, a, :], g[sel])
                add(B.u(side, a), cv, B.u(side, c), cv, K)
            np.add.at(b, B.u(side, a) * nv + cv, vol[sel, None] * np.einsum("ij,cj->ci", M2, f_h[cv, a]))

    # ---- cut cells
    cut = np.flatnonzero(cell_tags == 2)
    cc = cells[cut]
    vc, hc, gc = vol[cut], h[cut], g[cut]
    ph = phi_h[cc]
    gphi = np.einsum("ck,ckd->cd", ph, gc)                    # grad(phi_h), constant per cell
    M = vc[:, None, None] * M2[None]
    Mphi = vc[:, None, None] * np.einsum("ijk,ck->cij", M3, ph)
    Mphi2 = vc[:, None, None] * np.einsum("ijkl,ck,cl->cij", M4, ph, ph)
    gam = pen_coef
    sgn = (1.0, -1.0)
    fbar = np.einsum("cjd->cd", f_h[cc]) / n                   # (1/|K|) int f = mean of the nodal values
    for side in (0, 1):
        S = sigma_basis(gc, lam[side], mu[side])
        W = gam * (coef_out if side == 0 else coef_in)
        for a in range(d):
            for bb in range(d):
                # main.py:191-192  (y + sigma(u)) : (z + sigma(v))
                add(B.y(side, a, bb), cc, B.y(side, a, bb), cc, W * M)                        # z-y
                for c in range(d):
                    zu = W * (vc / n)[:, None, None] * np.broadcast_to(S[:, None, :, c, a, bb], (cut.size, n, n))
                    add(B.y(side, a, bb), cc, B.u(side, c), cc, zu)                             # z-u
                    add(B.u(side, c), cc, B.y(side, a, bb), cc, np.transpose(zu, (0, 2, 1)))    # v-y
                # main.py:211-217  h^2 div(y).div(z):  div(N_j E_ce) = e_c g_{j,e}
                for e in range(d):
                    add(B.y(side, a, bb), cc, B.y(side, a, e), cc,
                        stab_coef * (hc ** 2 * vc)[:, None, None] * gc[:, :, None, bb] * gc[:, None, :, e])
                # main.py:255-260  h^2 f . div(z)
                np.add.at(b, B.y(side, a, bb) * nv + cc,
                          stab_coef * (hc ** 2 * vc * fbar[:, a])[:, None] * gc[:, :, bb])
            for c in range(d):
                vu = W * vc[:, None, None] * np.einsum("cipq,cjpq->cij", S[:, :, a], S[:, :, c])
                add(B.u(side, a), cc, B.u(side, c), cc, vu)                                     # v-u
        # main.py:193-197  h^-2 ((y_in - y_out) grad phi) . ((z_in - z_out) grad phi)
        for side2 in (0, 1):
            for a in range(d):
                for bb in range(d):
                    for e in range(d):
                        add(B.y(side, a, bb), cc, B.y(side2, a, e), cc,
                            gam * sgn[side] * sgn[side2] * (hc ** -2 * gphi[:, bb] * gphi[:, e])[:, None, None] * M)
            # main.py:198-202  h^-2 (u_in - u_out + h^-1 p phi) . (v_in - v_out + h^-1 q phi)
            for a in range(d):
                add(B.u(side, a), cc, B.u(side2, a), cc, gam * sgn[side] * sgn[side2] * (hc ** -2)[:, None, None] * M)
        for a in range(d):
            add(B.u(side, a), cc, B.p + a, cc, gam * sgn[side] * (hc ** -3)[:, None, None] * Mphi)
            add(B.p + a, cc, B.u(side, a), cc, gam * sgn[side] * (hc ** -3)[:, None, None] * Mphi)
    for a in range(d):
        add(B.p + a, cc, B.p + a, cc, gam * (hc ** -4)[:, None, None] * Mphi2)

    # ---- one-sided boundary terms, main.py:182-183 on d_bdry(100) / d_bdry(101)
    fvt = FACET_VERTS[topo.cell_type]
    for side, ents in ((0, ds100), (1, ds101)):
        ents = np.asarray(ents, dtype=np.int64).reshape(-1, 2)
        for lf in range(n):
            c = ents[ents[:, 1] == lf, 0]
            if c.size == 0:
                continue
            gn = np.sqrt((g[c, lf] ** 2).sum(axis=1))
            nrm = -g[c, lf] / gn[:, None]
            area = d * vol[c] * gn
            mF = np.zeros((n, n))
            for i in fvt[lf]:
                for j in fvt[lf]:
                    mF[i, j] = (2.0 if i == j else 1.0) / (d * (d + 1))
            for a in range(d):
                for e in range(d):
                    # (y n) . v  with  y = N_j E_ae:  N_i N_j n_e
                    add(B.u(side, a), cells[c], B.y(side, a, e), cells[c],
                        (area * nrm[:, e])[:, None, None] * mF[None])

    # ---- facet stabilisation, main.py:205-209 on dS(3) (in) and :219-223 on dS(4) (out)
    for side, ftag in ((0, 3), (1, 4)):
        fs = np.flatnonzero((facet_tags == ftag) & (topo.f2c[:, 1] >= 0))
        if fs.size == 0:
            continue
        cp, cm = topo.f2c[fs, 0], topo.f2c[fs, 1]
        J = []
        area = None
        for cs in (cp, cm):
            lfs = np.argmax(topo.c2f[cs] == fs[:, None], axis=1)
            gn = np.sqrt((g[cs, lfs] ** 2).sum(axis=1))
            nrm = -g[cs, lfs] / gn[:, None]
            if area is None:
                area = d * vol[cs] * gn
            S = sigma_basis(g[cs], lam[side], mu[side])
            J.append(np.einsum("cjbpq,cq->cjbp", S, nrm))            # sigma(N_j e_b) n
        wgt = stab_coef * 0.5 * (h[cp] + h[cm]) * area
        sides = ((cp, J[0]), (cm, J[1]))
        for (cr, Jr) in sides:
            for (ccol, Jc) in sides:
                for a in range(d):
                    for bcomp in range(d):
                        v = wgt[:, None, None] * np.einsum("cip,cjp->cij", Jr[:, :, a], Jc[:, :, bcomp])
                        add(B.u(side, a), cells[cr], B.u(side, bcomp), cells[ccol], v)

    R, Cc, Vv = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
    ndof = B.C * nv

    # ---- active set
    active = np.zeros(ndof, dtype=bool)
    v_in = np.unique(cells[np.isin(cell_tags, (1, 2))])
    v_out = np.unique(cells[np.isin(cell_tags, (2, 3))])
    v_cut = np.unique(cc)
    for a in range(d):
        active[B.u(0, a) * nv + v_in] = True
        active[B.u(1, a) * nv + v_out] = True
        active[(B.p + a) * nv + v_cut] = True
        for bb in range(d):
            active[B.y(0, a, bb) * nv + v_cut] = True
            active[B.y(1, a, bb) * nv + v_cut] = True

    # ---- Dirichlet condition on u_in, main.py:158-177,237-239,271-277: rows and columns of the
    # constrained DoFs are left out of the matrix (unit diagonal instead), their columns are
    # lifted into the right-hand side
    bc_vertices = np.asarray(bc_vertices, dtype=np.int64)
    bc_dofs = np.concatenate([B.u(0, a) * nv + bc_vertices for a in range(d)])
    bc_vals = np.concatenate([uD[bc_vertices, a] for a in range(d)])
    is_bc = np.zeros(ndof, dtype=bool)
    is_bc[bc_dofs] = True
    ubc = np.zeros(ndof)
    ubc[bc_dofs] = bc_vals
    lift = is_bc[Cc] & ~is_bc[R]
    np.subtract.at(b, R[lift], Vv[lift] * ubc[Cc[lift]])      # apply_lifting
    keep = ~is_bc[R] & ~is_bc[Cc]
    A = sp.coo_matrix((np.concatenate([Vv[keep], np.ones(bc_dofs.size)]),
                       (np.concatenate([R[keep], bc_dofs]), np.concatenate([Cc[keep], bc_dofs]))),
                      shape=(ndof, ndof)).tocsr()
    A.sum_duplicates()
    A.sort_indices()
    b[bc_dofs] = bc_vals                                       # bc.set
    active[bc_dofs] = True
    return A, b, active
