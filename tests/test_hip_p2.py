"""GPU parity of the P2 x P2 path (BASELINE configs[2]: P2 elements + ghost penalty + the
div(grad) stabilisation of main.py:123-128,150) against the quadrature oracle.
Tolerances: matrix / rhs 1e-11 relative to the largest entry (different quadrature evaluation
order, FMA, atomic accumulation order); solution 1e-6 at solver rtol 1e-11."""
import warnings

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import assembly as OA
from oracle import assembly_quad as Q
from oracle import meshgen
from oracle import tagging as T
from oracle.topology import Topology

from datasets import load_mesh

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    import phifem_amd
    assert phifem_amd._lib.device_count() > 0
    return phifem_amd


def oracle_space(mesh, topo, degree):
    """Oracle space that uses the LIBRARY's edge numbering."""
    V = Q.Space.__new__(Q.Space)
    V.topo, V.degree = topo, degree
    if degree == 1:
        V.ndofs, V.cell_dofs, V.edge_vertices = topo.nv, topo.cells, None
    else:
        V.edge_vertices = mesh.edges.astype(np.int64)
        V.ndofs = topo.nv + V.edge_vertices.shape[0]
        V.cell_dofs = np.concatenate([topo.cells, topo.nv + mesh.c2e.astype(np.int64)], axis=1)
    return V


@pytest.mark.parametrize("d,n", [(2, (5, 4)), (3, (3, 4, 2))])
def test_edge_numbering_box(P, d, n):
    lo, hi = [-1.0] * d, [1.0] * d
    m = P.create_box(lo, hi, n)
    xo, co = meshgen.create_box(lo, hi, n)
    ctype = "triangle" if d == 2 else "tetrahedron"
    topo = Topology(ctype, co, xo.shape[0])
    ev, c2e = Q.build_edges(topo)
    assert m.ne == ev.shape[0]
    he, hc2e = m.edges.astype(np.int64), m.c2e.astype(np.int64)
    assert np.all(he[:, 0] < he[:, 1]) and np.unique(he, axis=0).shape[0] == m.ne
    key = {tuple(e): i for i, e in enumerate(ev)}
    to_oracle = np.array([key[tuple(e)] for e in he])
    assert np.array_equal(to_oracle[hc2e], c2e)


def test_edge_numbering_unstructured(P):
    ctype, x, cells = load_mesh("disk")
    m = P.Mesh.from_arrays(ctype, x, cells)
    topo = Topology(ctype, cells, x.shape[0])
    ev, c2e = Q.build_edges(topo)
    assert m.ne == ev.shape[0] and np.array_equal(m.edges, ev) and np.array_equal(m.c2e, c2e)
    # tetrahedra through the host sort: one Kuhn cube, shuffled
    xo, co = meshgen.create_box([0, 0, 0], [1, 1, 1], [2, 2, 2])
    m3 = P.Mesh.from_arrays("tetrahedron", xo, co)
    t3 = Topology("tetrahedron", co, xo.shape[0])
    ev3, c2e3 = Q.build_edges(t3)
    assert np.array_equal(m3.edges, ev3) and np.array_equal(m3.c2e, c2e3)


def setup(P, d, n, kphi, box=True, lin=None):
    from phifem_amd.mesh_scripts import NodalFunction
    mesh = P.create_box([-1.5] * d, [1.5] * d, [n] * d)
    cen = np.array([0.03, -0.02, 0.01][:d])
    x = mesh.x
    phi1 = ((x - cen) ** 2).sum(axis=1) - 1.0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ct, ft, sub, meas, maps = P.compute_tags_measures(mesh, NodalFunction(phi1), 1, box_mode=box,
                                                          single_layer_cut=True)
    work = mesh if box else sub
    xw = work.x
    ctype = "triangle" if d == 2 else "tetrahedron"
    topo = Topology(ctype, work.cells.astype(np.int64), work.nv)
    topo.c2f, topo.f2c, topo.nf = work.c2f.astype(np.int64), work.f2c.astype(np.int64), work.nf
    V = oracle_space(work, topo, 2)
    Vp = oracle_space(work, topo, kphi)
    pts = work.p2_dof_points()
    assert np.array_equal(pts, V.dof_points(xw))
    phi = ((Vp.dof_points(xw) - cen) ** 2).sum(axis=1) - 1.0
    if lin == "quadratic":
        def uq(p):
            return p[:, 0] ** 2 + 2 * p[:, 1] ** 2 + p[:, 0] * p[:, 1] + (p[:, 2] ** 2 if d == 3 else 0) + 0.3 * p[:, 0] + 1
        uex = uq(pts)
        f = np.full(V.ndofs, -(6.0 + (2.0 if d == 3 else 0.0)))
    else:
        uex = np.prod(np.sin(pts), axis=1)
        f = d * uex
    ds = meas(100) if box else work.boundary_facets.reshape(-1)
    A, b, act = Q.assemble_poisson_wd_quad(topo, xw, work.cell_tag_values(), work.facet_tag_values(),
                                           ds, V, Vp, phi, f, uex)
    return work, V, phi, f, uex, A, b, act


@pytest.mark.parametrize("d,n,kphi,box", [(2, 12, 1, True), (2, 10, 2, True), (3, 5, 1, True),
                                          (3, 5, 2, True), (2, 12, 1, False), (3, 5, 2, False)])
def test_p2_matrix_and_rhs_vs_oracle(P, d, n, kphi, box):
    work, V, phi, f, uex, A, b, act = setup(P, d, n, kphi, box=box)
    s = P.PhiFEMSolver(work, degree=2, levelset_degree=kphi)
    info = s.assemble(phi, f, uex)
    rowptr, col, val, rhs, dof = s.export_csr()
    H = sp.csr_matrix((val, col, rowptr), shape=(rowptr.size - 1,) * 2)
    idx = np.flatnonzero(act)
    assert info["n_active"] == idx.size and np.array_equal(dof, idx)
    Ao = A[idx][:, idx].tocsr()
    Ao.sort_indices()
    assert np.array_equal(H.indptr, Ao.indptr) and np.array_equal(H.indices, Ao.indices)
    assert np.abs(H.data - Ao.data).max() <= 1e-11 * np.abs(Ao.data).max()
    assert np.abs(rhs - b[idx]).max() <= 1e-11 * np.abs(b).max()
    xv = np.random.default_rng(1).standard_normal(idx.size)
    assert np.abs(s.spmv(xv) - Ao @ xv).max() <= 1e-11 * np.abs(Ao @ xv).max()


@pytest.mark.parametrize("d,n", [(2, 12), (3, 5)])
def test_p2_reproduces_quadratics(P, d, n):
    """P2 patch test: a quadratic u (f = -Laplace u constant, u_D = u, p = 0) satisfies the HIP
    system exactly; exercises the div(grad) terms of main.py:123-128,150.  Checked through the HIP
    SpMV (no solver involved), and in 2-D also through the solve."""
    work, V, phi, f, uex, A, b, act = setup(P, d, n, 1, lin="quadratic")
    s = P.PhiFEMSolver(work, degree=2)
    s.assemble(phi, f, uex)
    rowptr, col, val, rhs, dof = s.export_csr()
    wex = np.concatenate([uex, np.zeros(V.ndofs)])[dof]
    r = s.spmv(wex) - rhs
    assert np.abs(r).max() <= 1e-10 * np.abs(val).max()
    if d == 2:
        w = s.solve(rtol=1e-11, max_iter=50000)
        assert s.stats["relres"] <= 1e-11
        u, p = s.split(w)
        ua = act[:V.ndofs]
        assert np.abs(u[ua] - uex[ua]).max() < 1e-6
        assert np.abs(p).max() < 1e-4
        assert np.all(w[~act] == 0.0)


@pytest.mark.parametrize("d,n", [(2, 16), (3, 6)])
def test_p2_solve_vs_direct(P, d, n):
    """The iterative solve of the P2 systems against a direct solve of the oracle's matrix.  The 3-D systems
    (cond 1e7-1e8, h^-4 penalty scaling) need the breakdown-restart guard of the BiCGStab loop (scipy's
    BiCGStab diverges on them) and thousands of Jacobi iterations; the refined-lattice box preconditioner
    cuts them 3-4 x (tools/p2_3d_probe.py: 128^3, 3.1e6 DoFs: 832 instead of 3160)."""
    work, V, phi, f, uex, A, b, act = setup(P, d, n, 2)
    s = P.PhiFEMSolver(work, degree=2, levelset_degree=2)
    s.assemble(phi, f, uex)
    w = s.solve(rtol=1e-11, max_iter=100000)
    assert s.stats["relres"] <= 1e-11
    wo = OA.solve_direct(A, b, act)
    assert np.abs(w - wo).max() <= 1e-6 * np.abs(wo).max()
    inside = np.unique(V.cell_dofs[work.cell_tag_values() == 1])
    if inside.size:
        assert np.abs(w[:V.ndofs][inside] - uex[inside]).max() < 1e-1


@pytest.mark.parametrize("mesh_name,data", [("disk", "circle_in_circle"), ("square_tri", "nasty_smooth")])
@pytest.mark.parametrize("deg", [1, 2, 3])
def test_tagging_with_p2_levelset(P, mesh_name, data, deg):
    """A level-set given as a P2 function (a2: "P_k nodal values"): for a quadratic phi the P2
    interpolant IS phi, so the tags must equal those of the closed form, at every detection
    degree."""
    from datasets import MESHTAG_DATA
    from phifem_amd.mesh_scripts import NodalFunction
    ctype, x, cells = load_mesh(mesh_name)
    m = P.Mesh.from_arrays(ctype, x, cells)
    if data == "circle_in_circle":
        f = MESHTAG_DATA[data][1]
    else:
        def f(xx):
            return 0.7 * xx[0] ** 2 - 0.4 * xx[0] * xx[1] + 1.3 * xx[1] ** 2 + 0.2 * xx[0] - 0.9
    pts = m.p2_dof_points()
    nod = f(pts.T)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        c2, f2 = P.compute_tags_measures(m, NodalFunction(nod, degree=2), deg, box_mode=True)[:2]
        c1, f1 = P.compute_tags_measures(m, f, deg, box_mode=True)[:2]
    # identical up to exact-compare degeneracies of round-off (none on these data)
    assert np.array_equal(c2.values, c1.values)
    assert np.array_equal(f2.values, f1.values)


def test_p2_refined_lattice_preconditioner_2d(P):
    """P2 on a Kuhn box: the DoFs are the points of the lattice of spacing h/2; the box preconditioner on that
    lattice (P1 there is spectrally equivalent to P2 here) cuts the 2-D iteration count several times
    (CPU prototype: 2054-2611 -> 315-400) and returns the same solution."""
    from phifem_amd import _lib as L_
    res = {}
    for pc in (0, 1):
        work, V, phi, f, uex, A, b, act = setup(P, 2, 32, 2)
        L_.check(L_.lib.phx_set_option(work._h, L_.OPT_PRECOND, pc))
        s = P.PhiFEMSolver(work, degree=2, levelset_degree=2)
        s.assemble(phi, f, uex)
        res[pc] = (s.solve(rtol=1e-10, max_iter=50000), dict(s.stats))
    assert res[0][1]["precond"] == "jacobi" and res[1][1]["precond"] == "box-dst"
    assert res[1][1]["relres"] <= 1e-10 and res[0][1]["relres"] <= 1e-10
    assert res[1][1]["iterations"] < 0.5 * res[0][1]["iterations"]
    assert np.abs(res[1][0] - res[0][0]).max() <= 1e-6 * np.abs(res[0][0]).max()


def _p2_sphere(P, n, kphi=2):
    """Tagged n^3 Kuhn box around the unit sphere with P2 nodal data (no oracle assembly: sizes beyond the numpy
    oracle's reach are compared HIP against HIP, stored-everything against structured)."""
    from phifem_amd.mesh_scripts import NodalFunction
    mesh = P.create_box([-1.5] * 3, [1.5] * 3, [n] * 3)
    cen = np.array([0.03, -0.02, 0.01])
    phi1 = ((mesh.x - cen) ** 2).sum(axis=1) - 1.0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        P.compute_tags_measures(mesh, NodalFunction(phi1), 1, box_mode=True, single_layer_cut=True)
    pts = mesh.p2_dof_points()
    phi = ((pts - cen) ** 2).sum(axis=1) - 1.0
    if kphi == 1:
        phi = phi[:mesh.nv]
    uex = np.prod(np.sin(pts), axis=1)
    return mesh, phi, 3.0 * uex, uex


@pytest.mark.parametrize("n,kphi", [(28, 2), (32, 1)])
def test_structured_p2_interior_rows_match_the_stored_matrix(P, n, kphi):
    """Structured P2 systems (default on 3-D Kuhn boxes; BASELINE configs[2] needs them to fit 512^3): rows whose
    5 x 5 x 5 fine-lattice neighbourhood is untouched and tagged inside are applied from eight translation-invariant
    stencils and never assembled.  The product y = A x must equal the CSR of the stored-everything assembly (exported
    through the lazy re-assembly, which runs the generic path) to 1e-13 of |A| |x|, the right-hand sides (mass stencil
    against element quadrature) must agree to 1e-13, and both systems must give the same solution."""
    from phifem_amd import _lib as L
    mesh, phi, f, uex = _p2_sphere(P, n, kphi)
    s = P.PhiFEMSolver(mesh, degree=2, levelset_degree=kphi)
    info = s.assemble(phi, f, uex)
    assert info["stencil_rows"] > 0 and info["stencil_runs"] > 0, info
    assert info["has_csr"] == 0
    rhs_s, dof_s = s.export_rhs_dof()
    rng = np.random.default_rng(5)
    x = rng.standard_normal(info["n_active"])
    y = s.spmv(x)
    rowptr, col, val, rhs, dof = s.export_csr()          # generic path: every row assembled and stored
    M = sp.csr_matrix((val, col, rowptr), shape=(rowptr.size - 1,) * 2)
    assert np.array_equal(dof, dof_s)
    scale = np.abs(M).max() * np.abs(x).max()
    assert np.abs(y - M @ x).max() <= 1e-13 * scale * 10   # rows of ~60-200 terms
    assert np.abs(rhs_s - rhs).max() <= 1e-13 * np.abs(rhs).max() * 10
    w = s.solve(rtol=1e-10, max_iter=40000)
    assert s.stats["converged"]
    L.check(L.lib.phx_set_option(mesh._h, L.OPT_STRUCTURED, 0))
    try:
        s2 = P.PhiFEMSolver(mesh, degree=2, levelset_degree=kphi)
        info2 = s2.assemble(phi, f, uex)
        assert info2["stencil_rows"] == 0 and info2["n_active"] == info["n_active"]
        w2 = s2.solve(rtol=1e-10, max_iter=40000)
        assert s2.stats["converged"]
    finally:
        L.check(L.lib.phx_set_option(mesh._h, L.OPT_STRUCTURED, 1))
    print(f"P2 n={n}: {info['n_active']} rows, {info['stencil_rows']} from stencils in {info['stencil_runs']} runs; "
          f"iterations structured {s.stats['iterations']} / stored {s2.stats['iterations']}")
    assert np.abs(w - w2).max() <= 1e-6 * np.abs(w2).max()
    # the true residual of the structured solve through the exported (generic) matrix
    r = M @ w[dof] - rhs
    assert np.linalg.norm(r) <= 1e-8 * np.linalg.norm(rhs)


@pytest.mark.parametrize("d,n,kphi", [(2, 10, 2), (3, 5, 1), (3, 5, 2)])
def test_p2_coupling_blocks_are_bit_symmetric_when_assembled_exactly(P, d, n, kphi):
    """The (u, p) and (p, u) blocks of the cut-cell penalty are ONE table of integrals int N_r N_s phi_h (k_p2_cut computes
    each pair (r <= s) once, main.py:115-122): with the exact accumulation (deterministic=True: every slot is an exact sum,
    whatever the order) A[u_v, p_w] == A[p_w, u_v] to the last bit, and the same inside the (p, p) block.  The (u, u) block
    carries the one-sided boundary term -int (grad u . n) v and is not symmetric."""
    work, V, phi, f, uex, A, b, act = setup(P, d, n, kphi)
    s = P.PhiFEMSolver(work, degree=2, levelset_degree=kphi, deterministic=True)
    info = s.assemble(phi, f, uex)
    rowptr, col, val, rhs, dof = s.export_csr()
    H = sp.csr_matrix((val, col, rowptr), shape=(rowptr.size - 1,) * 2)
    nu = info["n_active_u"]
    Hup, Hpu, Hpp = H[:nu, nu:].tocsr(), H[nu:, :nu].T.tocsr(), H[nu:, nu:].tocsr()
    assert Hup.nnz > 0 and (Hup != Hpu).nnz == 0
    assert (Hpp != Hpp.T).nnz == 0
