"""CPU experiment (round 3): BiCGStab iterations on the 5-field interface-elasticity system (oracle matrices) with
  J   scalar Jacobi
  L   Jacobi on every row + per-component ANISOTROPIC lattice solves on the BULK rows of the six displacement fields
      (u_in[a], u_out[a]): z = R K_a^-1 R^T r on rows whose vertex no cut-cell / facet / boundary term touches,
      K_a = sum_b kappa_ab T_b, kappa_aa = lambda + 2 mu, kappa_ab = mu (the diagonal part of -div sigma(u) for
      component a), homogeneous Dirichlet on the faces of the mesh box (u_in is prescribed there).
usage: elasticity_lattice.py [n ...]"""
import os, sys, time, warnings
import numpy as np
import scipy.sparse as sp
from scipy.fft import dstn, idstn
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import elasticity as EL, meshgen, tagging as OT
from oracle.topology import Topology
from oracle.points import FACET_VERTS
from precond_variants import bicgstab


def problem(n, E_out):
    d = 3
    x, cells = meshgen.create_box([-1.5] * d, [1.5] * d, [n] * d)
    topo = Topology("tetrahedron", cells, x.shape[0])
    phi = 1.0 - (x ** 2).sum(axis=1)
    ls = OT.NodalP1(phi)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        oc, of, _, om, _, _ = OT.compute_tags_measures("tetrahedron", x, topo, ls, 1, box_mode=True)
    cv = np.zeros(topo.nc, dtype=np.int64); cv[oc.indices] = oc.values
    fv = np.zeros(topo.nf, dtype=np.int64); fv[of.indices] = of.values
    n1 = n + 1
    v = np.arange(topo.nv)
    i, j, k = v % n1, (v // n1) % n1, v // (n1 * n1)
    bcv = np.flatnonzero((i == 0) | (i == n) | (j == 0) | (j == n) | (k == 0) | (k == n))
    f = np.stack([np.sin(x[:, 0]) + 0.2, np.cos(x[:, 1]), 0.5 * x[:, 2]], axis=1)
    uD = 0.1 * np.stack([x[:, 0] * x[:, 1], np.sin(x[:, 2]), x[:, 0] - x[:, 1]], axis=1)
    A, b, act = EL.assemble_elasticity_if(topo, x, cv, fv, om(100), om(101), phi, f, uD, bcv, E_in=1.0, E_out=E_out)
    # vertices touched by the band terms: cut cells, cells of dS(3)/dS(4) facets, cells of the ds(100)/ds(101) entities
    touched = np.zeros(topo.nv, dtype=bool)
    touched[cells[cv == 2].ravel()] = True
    for tag in (3, 4):
        fs = np.flatnonzero((fv == tag) & (topo.f2c[:, 1] >= 0))
        touched[cells[topo.f2c[fs].ravel()].ravel()] = True
    for e in (om(100), om(101)):
        touched[cells[np.asarray(e).reshape(-1, 2)[:, 0]].ravel()] = True
    return A, b, act, topo.nv, touched, bcv


def main(n, E_out=1e-3, variant="bulk"):
    t0 = time.time()
    A, b, act, nv, touched, bcv = problem(n, E_out)
    idx = np.flatnonzero(act)
    Aa = A[idx][:, idx].tocsr(); ba = b[idx]
    dg = Aa.diagonal()
    blk, vert = idx // nv, idx % nv
    isbc = np.zeros(nv, dtype=bool); isbc[bcv] = True
    h = 3.0 / n
    m = n - 1                                  # interior lattice points per axis
    k1 = np.arange(1, m + 1)
    ev = 2.0 - 2.0 * np.cos(np.pi * k1 / (m + 1))     # eigenvalues of tridiag(-1, 2, -1) with Dirichlet ends
    n1 = n + 1
    i, j, k = vert % n1, (vert // n1) % n1, vert // (n1 * n1)
    fields = []
    for side, E in ((0, 1.0), (1, E_out)):
        lam, mu = EL.lame(E, 0.3)
        for a in range(3):
            keep = ~isbc[vert]
            if variant == "bulk" or (variant == "in_all" and side == 1):
                keep &= ~touched[vert]
            rows = np.flatnonzero((blk == side * 3 + a) & keep)
            kap = [(lam + 2 * mu if bb == a else mu) * h for bb in range(3)]
            den = kap[0] * ev[:, None, None] + kap[1] * ev[None, :, None] + kap[2] * ev[None, None, :]
            fields.append((rows, i[rows] - 1, j[rows] - 1, k[rows] - 1, den))
    nbulk = sum(f[0].size for f in fields)

    def ML(r):
        z = r / dg
        for rows, ii, jj, kk, den in fields:
            if rows.size == 0:
                continue
            G = np.zeros((m, m, m))
            G[ii, jj, kk] = r[rows]
            U = idstn(dstn(G, type=1) / den, type=1)
            z[rows] = U[ii, jj, kk]
        return z

    # vertex-block Jacobi (all active DoFs of a vertex in one dense block), alone and under the bulk lattice solves
    order = np.argsort(vert, kind="stable")
    cuts = np.flatnonzero(np.diff(vert[order])) + 1
    groups = np.split(order, cuts)
    Binv = sp.block_diag([np.linalg.inv(Aa[g][:, g].toarray()) for g in groups], format="csr")
    perm = np.concatenate(groups)

    def MB(r):
        z = np.empty_like(r); z[perm] = Binv @ r[perm]; return z

    def MBL(r):
        z = MB(r)
        for rows, ii, jj, kk, den in fields:
            if rows.size == 0:
                continue
            G = np.zeros((m, m, m))
            G[ii, jj, kk] = r[rows]
            U = idstn(dstn(G, type=1) / den, type=1)
            z[rows] = U[ii, jj, kk]
        return z

    print(f"n={n} E_out={E_out} variant={variant}: {idx.size} DoFs ({nbulk} lattice rows), set-up {time.time() - t0:.0f} s", flush=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if variant == "bulk":
            t0 = time.time(); x1, it1 = bicgstab(Aa, ba, lambda r: r / dg, rtol=1e-8, maxit=20000); t1 = time.time() - t0
        else:
            x1, it1, t1 = ba * 0, -1, 0
        t0 = time.time(); x2, it2 = bicgstab(Aa, ba, ML, rtol=1e-8, maxit=20000); t2 = time.time() - t0
        x3, it3 = bicgstab(Aa, ba, MB, rtol=1e-8, maxit=20000)
        x4, it4 = bicgstab(Aa, ba, MBL, rtol=1e-8, maxit=20000)
    print(f"   vertex-block Jacobi {it3} it | vertex-block Jacobi + lattice {it4} it", flush=True)
    rr = lambda x: np.linalg.norm(Aa @ x - ba) / np.linalg.norm(ba)
    print(f"   Jacobi {it1} it (res {rr(x1):.1e}, {t1:.0f} s) | Jacobi + bulk lattice solves {it2} it (res {rr(x2):.1e}, {t2:.0f} s)", flush=True)


if __name__ == "__main__":
    for n in [int(a) for a in sys.argv[1:]] or [12, 16, 24]:
        for variant in ("bulk",):
            main(n, variant=variant)
