"""GPU parity of the strong-Dirichlet (u_h = phi_h w_h) path, demo/strong-dirichlet/flower/main.py:
83-182, against `oracle/assembly_sd.py`.  Tolerances: matrix / rhs 1e-11 relative to the largest
entry (quadrature evaluation order, FMA, atomic accumulation order); solution 1e-6 relative at
solver rtol 1e-11 (the direct solve of the oracle matrix is the comparison)."""
import warnings

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import assembly as OA
from oracle import assembly_sd as SD
from oracle.topology import Topology

from test_hip_p2 import oracle_space

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    import phifem_amd
    assert phifem_amd._lib.device_count() > 0
    return phifem_amd


def setup(P, d, n, k, kphi, box=True):
    from phifem_amd.mesh_scripts import NodalFunction
    mesh = P.create_box([-1.5] * d, [1.5] * d, [n] * d)
    cen = np.array([0.03, -0.02, 0.01][:d])
    phi1 = ((mesh.x - cen) ** 2).sum(axis=1) - 1.0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ct, ft, sub, meas, maps = P.compute_tags_measures(mesh, NodalFunction(phi1), 1, box_mode=box)
    work = mesh if box else sub
    xw = work.x
    ctype = "triangle" if d == 2 else "tetrahedron"
    topo = Topology(ctype, work.cells.astype(np.int64), work.nv)
    topo.c2f, topo.f2c, topo.nf = work.c2f.astype(np.int64), work.f2c.astype(np.int64), work.nf
    if k == 2 or kphi == 2:
        V, Vp = oracle_space(work, topo, k), oracle_space(work, topo, kphi)
    else:
        V = Vp = oracle_space(work, topo, 1)
    phi = ((Vp.dof_points(xw) - cen) ** 2).sum(axis=1) - 1.0
    pts = V.dof_points(xw)
    g = 1.0 + 0.5 * pts[:, 0] - 0.25 * pts[:, 1]
    f = 2.0 * d * g + 4.0 * (0.5 * pts[:, 0] - 0.25 * pts[:, 1]) + np.sin(pts[:, 0])
    ds = meas(100) if box else work.boundary_facets.reshape(-1)
    A, b, act = SD.assemble_poisson_sd(topo, xw, work.cell_tag_values(), work.facet_tag_values(),
                                       ds, V, Vp, phi, f, stab_coef=0.8)
    return work, phi, f, A, b, act


@pytest.mark.parametrize("d,n,k,kphi,box", [(2, 14, 1, 1, True), (2, 14, 1, 1, False), (2, 10, 1, 2, True),
                                            (2, 10, 2, 2, True), (2, 10, 2, 1, False), (3, 6, 1, 1, True),
                                            (3, 6, 1, 1, False), (3, 5, 1, 2, True), (3, 5, 2, 2, True),
                                            (3, 5, 2, 1, False)])
def test_matrix_and_rhs_vs_oracle(P, d, n, k, kphi, box):
    work, phi, f, A, b, act = setup(P, d, n, k, kphi, box)
    s = P.StrongDirichletSolver(work, stab_coef=0.8, degree=k, levelset_degree=kphi)
    info = s.assemble(phi, f)
    rowptr, col, val, rhs, dof = s.export_csr()
    H = sp.csr_matrix((val, col, rowptr), shape=(rowptr.size - 1,) * 2)
    idx = np.flatnonzero(act)
    assert info["n_active"] == info["n_active_u"] == idx.size and np.array_equal(dof, idx)
    assert info["n_full"] == act.size
    Ao = A[idx][:, idx].tocsr()
    Ao.sort_indices()
    assert np.array_equal(H.indptr, Ao.indptr) and np.array_equal(H.indices, Ao.indices)
    assert np.abs(H.data - Ao.data).max() <= 1e-11 * np.abs(Ao.data).max()
    assert np.abs(rhs - b[idx]).max() <= 1e-11 * np.abs(b).max()


@pytest.mark.parametrize("d,n,k,box", [(2, 24, 1, True), (2, 24, 1, False), (2, 12, 2, True), (3, 8, 1, True)])
def test_solve_vs_direct(P, d, n, k, box):
    work, phi, f, A, b, act = setup(P, d, n, k, k, box)
    s = P.StrongDirichletSolver(work, stab_coef=0.8, degree=k, levelset_degree=k)
    s.assemble(phi, f)
    w = s.solve(rtol=1e-11, max_iter=100000)
    wref = OA.solve_direct(A, b, act)
    assert w.shape == wref.shape and np.all(w[~act] == 0.0)
    assert np.abs(w - wref).max() <= 1e-6 * np.abs(wref).max()
    u = s.solution(w)
    assert np.array_equal(u, w * phi)


def test_manufactured_solution_converges(P):
    """u = (1 - r^2) g on the unit disc: the nodal error of u_h = phi_h w_h falls ~4x per halving."""
    from phifem_amd.mesh_scripts import NodalFunction
    errs = []
    for n in (32, 64):
        mesh = P.create_box([-1.5, -1.5], [1.5, 1.5], [n, n])
        x = mesh.x
        phi = (x ** 2).sum(axis=1) - 1.0
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            P.compute_tags_measures(mesh, NodalFunction(phi), 1, box_mode=True)
        g = 1.0 + 0.5 * x[:, 0] + 0.25 * x[:, 1]
        uex = -phi * g
        f = 4.0 * g + 4.0 * (0.5 * x[:, 0] + 0.25 * x[:, 1])
        s = P.StrongDirichletSolver(mesh)
        s.assemble(phi, f)
        u = s.solution(s.solve(rtol=1e-11, max_iter=100000))
        inside = np.unique(mesh.cells[mesh.cell_tag_values() == 1])
        errs.append(np.sqrt(np.mean((u[inside] - uex[inside]) ** 2)))
    assert errs[0] / errs[1] > 3.0


def test_weighted_lattice_preconditioner_cuts_iterations(P):
    """A ~ S K S with S ~ |phi_h|: the sine-transform solve is applied between two nodal scalings
    (phx_precond.inc.hip); same solution, several times fewer iterations than Jacobi."""
    from phifem_amd import _lib as L
    from phifem_amd.mesh_scripts import NodalFunction
    mesh = P.create_box([-1.5] * 3, [1.5] * 3, [48] * 3)
    x = mesh.x
    phi = (x ** 2).sum(axis=1) - 1.0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        P.compute_tags_measures(mesh, NodalFunction(phi), 1, box_mode=True)
    f = 1.0 + np.sin(x[:, 0])
    out = {}
    for pre in (1, 0):
        L.check(L.lib.phx_set_option(mesh._h, L.OPT_PRECOND, pre))
        s = P.StrongDirichletSolver(mesh)
        s.assemble(phi, f)
        out[pre] = (s.solve(rtol=1e-10, max_iter=100000), s.stats["iterations"], s.stats["precond"])
    assert out[1][2] == "box-dst" and out[0][2] == "jacobi"
    assert 2 * out[1][1] < out[0][1]
    assert np.abs(out[1][0] - out[0][0]).max() <= 1e-6 * np.abs(out[0][0]).max()


@pytest.mark.parametrize("box", [True, False])
def test_flower_demo_problem(P, box):
    """The demo's own problem (demo/strong-dirichlet/flower/main.py:48-70 with the flower data, 128 x 128
    squares): background-mesh and sub-mesh modes against the oracle, matrix and solution; both modes
    give the same u_h on the shared vertices up to the solver tolerance."""
    import flower_data as F
    from phifem_amd.mesh_scripts import NodalFunction
    n = 128
    bg = P.create_rectangle([[-4.5, -4.5], [4.5, 4.5]], [n, n])
    det = F.detection_levelset(bg.x.T)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ct, ft, sub, meas, maps = P.compute_tags_measures(bg, NodalFunction(det), 1, box_mode=box)
    work = bg if box else sub
    x = work.x
    topo = Topology("triangle", work.cells.astype(np.int64), work.nv)
    topo.c2f, topo.f2c, topo.nf = work.c2f.astype(np.int64), work.f2c.astype(np.int64), work.nf
    V = oracle_space(work, topo, 1)
    phi, f = F.levelset(x.T), F.source_term(x.T)
    ds = meas(100) if box else work.boundary_facets.reshape(-1)
    A, b, act = SD.assemble_poisson_sd(topo, x, work.cell_tag_values(), work.facet_tag_values(), ds, V, V, phi, f)
    s = P.StrongDirichletSolver(work)
    s.assemble(phi, f)
    rowptr, col, val, rhs, dof = s.export_csr()
    idx = np.flatnonzero(act)
    Ao = A[idx][:, idx].tocsr()
    Ao.sort_indices()
    assert np.array_equal(dof, idx) and np.array_equal(col, Ao.indices)
    assert np.abs(val - Ao.data).max() <= 1e-11 * np.abs(Ao.data).max()
    assert np.abs(rhs - b[idx]).max() <= 1e-11 * np.abs(b).max()
    w = s.solve(rtol=1e-11, max_iter=100000)
    wo = OA.solve_direct(A, b, act)
    assert np.abs(w - wo).max() <= 1e-6 * np.abs(wo).max()
    u = s.solution(w)
    assert u.max() > 0.0          # f >= 0, u = 0 on the boundary: positive inside


def test_errors(P):
    mesh = P.create_box([-1.5, -1.5], [1.5, 1.5], [8, 8])
    with pytest.raises(NotImplementedError):
        P.StrongDirichletSolver(mesh, degree=3)
    s = P.StrongDirichletSolver(mesh)
    with pytest.raises(ValueError):          # tags not computed yet
        s.assemble(np.ones(mesh.nv), np.ones(mesh.nv))
    with pytest.raises(ValueError):          # wrong array length
        s.assemble(np.ones(mesh.nv + 1), np.ones(mesh.nv))
