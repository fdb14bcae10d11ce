"""CPU experiment: BiCGStab iterations on the 5-field interface-elasticity system (oracle matrices, small n) with
scalar Jacobi vs vertex-block Jacobi (all active DoFs of one vertex in one dense block)."""
import os, sys, warnings
import numpy as np
import scipy.sparse as sp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import elasticity as EL, meshgen, tagging as OT
from oracle.topology import Topology
from oracle.points import FACET_VERTS
from precond_variants import bicgstab   # same loop


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
    bf = topo.boundary_facets
    c0 = topo.f2c[bf, 0]
    lf = np.array([int(np.flatnonzero(topo.c2f[c] == f)[0]) for c, f in zip(c0, bf)])
    bcv = np.unique(np.take_along_axis(cells[c0], FACET_VERTS["tetrahedron"][lf], axis=1))
    rng = np.random.default_rng(5)
    f = np.sin(x @ rng.standard_normal((d, d))) + 0.3
    uD = np.cos(x @ rng.standard_normal((d, d)))
    A, b, act = EL.assemble_elasticity_if(topo, x, cv, fv, om(100), om(101), phi, f, uD, bcv, E_in=1.0, E_out=E_out)
    return A, b, act, topo.nv


for n, E_out in [(8, 1e-3), (12, 1e-3), (12, 0.5)]:
    A, b, act, nv = problem(n, E_out)
    idx = np.flatnonzero(act)
    Aa = A[idx][:, idx].tocsr(); ba = b[idx]
    dg = Aa.diagonal()
    vert = idx % nv
    order = np.argsort(vert, kind="stable")
    vs = vert[order]
    cuts = np.flatnonzero(np.diff(vs)) + 1
    groups = np.split(order, cuts)
    Ad = Aa.tocsc()
    blocks = [np.linalg.inv(Aa[g][:, g].toarray()) for g in groups]
    perm = np.concatenate(groups)
    Binv = sp.block_diag(blocks, format="csr")
    def Mblk(r):
        z = np.empty_like(r); z[perm] = Binv @ r[perm]; return z
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        x1, it1 = bicgstab(Aa, ba, lambda r: r / dg, rtol=1e-8, maxit=20000)
        x2, it2 = bicgstab(Aa, ba, Mblk, rtol=1e-8, maxit=20000)
    print(f"n={n} E_out={E_out}: {idx.size} DoFs; scalar Jacobi {it1} it (res {np.linalg.norm(Aa@x1-ba)/np.linalg.norm(ba):.1e}); "
          f"vertex-block Jacobi {it2} it (res {np.linalg.norm(Aa@x2-ba)/np.linalg.norm(ba):.1e}); max block {max(g.size for g in groups)}", flush=True)
