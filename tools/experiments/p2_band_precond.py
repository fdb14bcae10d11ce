"""CPU experiment (round 3): BiCGStab iterations on the 3-D P2 x P2 weak-Dirichlet system (quadrature oracle, sphere)
for band treatments on top of the shipped preconditioner (P1 Laplacian of the h/2 lattice on the u block, Jacobi on p):
  L      shipped: K_fine^-1 (u) | diag^-1 (p)
  L+pt   the same, then the 2 x 2 (u_i, p_i) point blocks on the DoFs that carry a p (additive, replaces both entries)
  L+cell the same + restricted additive Schwarz over the cut cells (dense 20 x 20 (u, p) block per cut cell, each DoF
         takes the average of the cells that hold it)
  L*cell multiplicative: lattice first, the cell blocks on the updated residual
usage: p2_band_precond.py [n ...]"""
import os, sys, time, warnings
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import assembly_quad as Q, meshgen, tagging as OT
from oracle.topology import Topology
from precond_variants import bicgstab


def problem(n, kphi=1):
    d = 3
    x, cells = meshgen.create_box([-1.5] * d, [1.5] * d, [n] * d)
    topo = Topology("tetrahedron", cells, x.shape[0])
    cen = np.array([0.03, -0.02, 0.01])
    phi1 = ((x - cen) ** 2).sum(axis=1) - 1.0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        oc, of, _, om, _, _ = OT.compute_tags_measures("tetrahedron", x, topo, OT.NodalP1(phi1), 1, box_mode=True)
    cv = np.zeros(topo.nc, dtype=np.int64); cv[oc.indices] = oc.values
    fv = np.zeros(topo.nf, dtype=np.int64); fv[of.indices] = of.values
    V, Vp = Q.Space(topo, 2), Q.Space(topo, kphi)
    pts = V.dof_points(x)
    phi = ((Vp.dof_points(x) - cen) ** 2).sum(axis=1) - 1.0
    uex = np.prod(np.sin(pts), axis=1)
    A, b, act = Q.assemble_poisson_wd_quad(topo, x, cv, fv, om(100), V, Vp, phi, 3 * uex, uex)
    return x, topo, cv, V, pts, A, b, act


def main(n):
    t0 = time.time()
    x, topo, cv, V, pts, A, b, act = problem(n)
    nd = V.ndofs
    idx = np.flatnonzero(act)
    Aa = A[idx][:, idx].tocsr(); ba = b[idx]
    isu = idx < nd
    nu = int(isu.sum())
    dg = Aa.diagonal()
    # fine lattice (spacing h/2) around the active u DoFs, margin 4, Dirichlet faces
    h2 = 1.5 / n
    ijk = np.rint((pts[idx[:nu]] + 1.5) / h2).astype(np.int64)
    lo = ijk.min(0) - 5; m = ijk.max(0) + 5 - lo + 1
    T = lambda k: sp.diags([-np.ones(k - 1), 2 * np.ones(k), -np.ones(k - 1)], [-1, 0, 1])
    I = [sp.identity(k) for k in m]
    K = h2 * (sp.kron(I[2], sp.kron(I[1], T(m[0]))) + sp.kron(I[2], sp.kron(T(m[1]), I[0])) + sp.kron(T(m[2]), sp.kron(I[1], I[0])))
    pos = (ijk[:, 0] - lo[0]) + m[0] * ((ijk[:, 1] - lo[1]) + m[1] * (ijk[:, 2] - lo[2]))
    Klu = spla.splu(K.tocsc())

    def ML(r):
        g = np.zeros(K.shape[0]); g[pos] = r[:nu]
        return np.concatenate([Klu.solve(g)[pos], r[nu:] / dg[nu:]])

    # point blocks
    loc = -np.ones(2 * nd, dtype=np.int64); loc[idx] = np.arange(idx.size)
    pd = idx[nu:] - nd                               # DoF ids that carry a p
    iu, ip = loc[pd], loc[pd + nd]
    ok = iu >= 0
    iu, ip = iu[ok], ip[ok]
    a11 = np.asarray(Aa[iu, iu]).ravel(); a12 = np.asarray(Aa[iu, ip]).ravel()
    a21 = np.asarray(Aa[ip, iu]).ravel(); a22 = np.asarray(Aa[ip, ip]).ravel()
    det = a11 * a22 - a12 * a21

    def MLpt(r):
        z = ML(r)
        ru, rp = r[iu], r[ip]
        z[iu] = (a22 * ru - a12 * rp) / det
        z[ip] = (-a21 * ru + a11 * rp) / det
        return z

    # cut-cell blocks
    cut = np.flatnonzero(cv == 2)
    cd = V.cell_dofs[cut]
    cnt = np.zeros(idx.size)
    ri, ci, vi = [], [], []
    Ad = Aa.tolil() if idx.size < 3000 else None
    for c in range(cut.size):
        g = np.concatenate([loc[cd[c]], loc[cd[c] + nd]])
        g = g[g >= 0]
        Bi = np.linalg.inv(Aa[g][:, g].toarray())
        ri.append(np.repeat(g, g.size)); ci.append(np.tile(g, g.size)); vi.append(Bi.ravel())
        cnt[g] += 1
    band = cnt > 0
    Z = sp.csr_matrix((np.concatenate(vi), (np.concatenate(ri), np.concatenate(ci))), shape=Aa.shape)
    w = np.where(band, 1.0 / np.maximum(cnt, 1), 0.0)

    def cellsolve(r):
        return w * (Z @ r)

    def MLcell(r):
        z = ML(r)
        zc = cellsolve(r)
        z[band] = zc[band]
        return z

    def MLcell_add(r):
        return ML(r) + cellsolve(r)

    def MLcell_mul(r):
        z = ML(r)
        return z + cellsolve(r - Aa @ z)

    def Mcell_then_L(r):
        z = cellsolve(r)
        return z + ML(r - Aa @ z)

    print(f"n={n}: {idx.size} DoFs ({nu} u, {idx.size - nu} p), {cut.size} cut cells, band DoFs {int(band.sum())}, set-up {time.time() - t0:.0f} s", flush=True)
    def MLcell_raw(r):
        return ML(r) + Z @ r

    for name, M in (("L", ML), ("L+cell (averaged sum)", MLcell_add), ("L+cell (plain sum)", MLcell_raw)):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            t0 = time.time(); xs, it = bicgstab(Aa, ba, M, rtol=1e-8, maxit=5000)
        print(f"   {name:20s} {it:5d} it  res {np.linalg.norm(Aa @ xs - ba) / np.linalg.norm(ba):.1e}  {time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__" and os.environ.get("P2_COARSE") != "1":
    for n in [int(a) for a in sys.argv[1:]] or [6, 8]:
        main(n)


def coarse_variants(n, ratios=(2, 4)):
    """Galerkin coarse corrections on top of the shipped preconditioner: trilinear functions of spacing ratio * h
    (a) on every u and p DoF, (b) on the band DoFs only (u and p of the cut cells), one set per field."""
    t0 = time.time()
    x, topo, cv, V, pts, A, b, act = problem(n)
    nd = V.ndofs
    idx = np.flatnonzero(act)
    Aa = A[idx][:, idx].tocsr(); ba = b[idx]
    nu = int((idx < nd).sum())
    dg = Aa.diagonal()
    h2 = 1.5 / n
    ijk = np.rint((pts[idx[:nu]] + 1.5) / h2).astype(np.int64)
    lo = ijk.min(0) - 5; m = ijk.max(0) + 5 - lo + 1
    T = lambda k: sp.diags([-np.ones(k - 1), 2 * np.ones(k), -np.ones(k - 1)], [-1, 0, 1])
    I = [sp.identity(k) for k in m]
    K = h2 * (sp.kron(I[2], sp.kron(I[1], T(m[0]))) + sp.kron(I[2], sp.kron(T(m[1]), I[0])) + sp.kron(T(m[2]), sp.kron(I[1], I[0])))
    pos = (ijk[:, 0] - lo[0]) + m[0] * ((ijk[:, 1] - lo[1]) + m[1] * (ijk[:, 2] - lo[2]))
    Klu = spla.splu(K.tocsc())

    def ML(r):
        g = np.zeros(K.shape[0]); g[pos] = r[:nu]
        return np.concatenate([Klu.solve(g)[pos], r[nu:] / dg[nu:]])

    loc = -np.ones(2 * nd, dtype=np.int64); loc[idx] = np.arange(idx.size)
    cut = np.flatnonzero(cv == 2)
    band = np.zeros(idx.size, dtype=bool)
    cd = V.cell_dofs[cut].ravel()
    for off in (0, nd):
        g = loc[cd + off]; band[g[g >= 0]] = True
    allp = pts[np.where(idx < nd, idx, idx - nd)]          # position of every active DoF
    fld = (idx >= nd).astype(int)
    print(f"n={n}: {idx.size} DoFs, band {int(band.sum())}, set-up {time.time() - t0:.0f} s", flush=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        xs, it = bicgstab(Aa, ba, ML, rtol=1e-8, maxit=5000)
    print(f"   L                          {it:5d} it", flush=True)
    for ratio in ratios:
        H = ratio * 3.0 / n
        mc = int(np.ceil(3.0 / H)) + 1
        for name, sel in (("all", np.ones(idx.size, dtype=bool)),):
            rows, cols, vals = [], [], []
            for f in (0, 1):
                s_ = np.flatnonzero(sel & (fld == f))
                t = (allp[s_] + 1.5) / H
                c0 = np.minimum(np.floor(t).astype(int), mc - 2)
                for d in range(8):
                    dd = np.array([d & 1, (d >> 1) & 1, d >> 2])
                    cc = c0 + dd
                    w = np.prod(1.0 - np.abs(t - cc), axis=1)
                    keep = w > 1e-14
                    rows.append(s_[keep]); cols.append(f * mc ** 3 + cc[keep, 0] + mc * (cc[keep, 1] + mc * cc[keep, 2])); vals.append(w[keep])
            R = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(idx.size, 2 * mc ** 3))
            used = np.flatnonzero(np.asarray(abs(R).sum(axis=0)).ravel() > 0)
            R = R[:, used].tocsr()
            Ac = (R.T @ Aa @ R).tocsc()
            try:
                Aclu = spla.splu(Ac)
            except RuntimeError:
                print(f"   L + C({name}, H={ratio}h): singular coarse matrix"); continue
            M = lambda r: ML(r) + R @ Aclu.solve(R.T @ r)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                xs, it = bicgstab(Aa, ba, M, rtol=1e-8, maxit=5000)
            print(f"   L + C({name:4s}, H={ratio}h, {used.size:5d}) {it:5d} it  res {np.linalg.norm(Aa @ xs - ba) / np.linalg.norm(ba):.1e}", flush=True)


if __name__ == "__main__" and os.environ.get("P2_COARSE") == "1":
    rat = tuple(int(a) for a in os.environ.get("RATIOS", "2,4").split(","))
    for n in [int(a) for a in sys.argv[1:]] or [12]:
        coarse_variants(n, rat)
