"""GPU parity of the QUADRILATERAL Neumann / Robin path (`phx_assemble_poisson_flux` on a quadrilateral mesh: Q1 x
Q1^2 x DG0 with a Q2 level-set, the cell type and forms of demo/neumann/square/main.py:49-50,113-158) against
`oracle/assembly_flux_quad.py` with the same Gauss rule.  Tolerances: matrix / rhs 1e-11 relative to the largest entry
(quadrature evaluation order, FMA, atomics); solution 1e-6 relative to the direct solve of the oracle matrix."""
import warnings

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import assembly as OA
from oracle import assembly_flux_quad as FQ
from oracle.topology import Topology

from test_oracle_flux_quad import quad_mesh

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    import phifem_amd
    assert phifem_amd._lib.device_count() > 0
    return phifem_amd


def setup(P, n, kappa, ftag, box=True, nq=6):
    from phifem_amd.mesh_scripts import NodalFunction
    x0, cells0 = quad_mesh(n)
    mesh = P.Mesh.from_arrays("quadrilateral", x0, cells0.astype(np.int32))
    cen = np.array([0.03, -0.02])
    # the Q2 level-set drives the tagging too (detection degree 2: its nodes are detection points)
    phi_bg = ((mesh.q2_dof_points() - cen) ** 2).sum(axis=1) - 1.0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ct, ft, sub, meas, maps = P.compute_tags_measures(mesh, NodalFunction(phi_bg, degree=2), 2, box_mode=box)
    work = mesh if box else sub
    x = work.x
    topo = Topology("quadrilateral", work.cells.astype(np.int64), work.nv)
    topo.c2f, topo.f2c, topo.nf = work.c2f.astype(np.int64), work.f2c.astype(np.int64), work.nf
    # facet -> vertices consistent with the library's facet numbering
    fv = np.empty((work.nf, 2), dtype=np.int64)
    for lf in range(4):
        fv[topo.c2f[:, lf]] = topo.cells[:, FQ.FACET_VERTS_Q[lf]]
    topo.facet_vertices = fv
    pp = work.q2_dof_points()
    assert np.allclose(pp, FQ.q2_dof_points(topo, x))
    phi = ((pp - cen) ** 2).sum(axis=1) - 1.0
    uex = np.cos(x[:, 0]) * np.sin(x[:, 1] + 0.3)
    f = 3.0 * uex
    r = np.maximum(np.sqrt(((x - cen) ** 2).sum(axis=1)), 1e-12)
    gr = np.stack([-np.sin(x[:, 0]) * np.sin(x[:, 1] + 0.3), np.cos(x[:, 0]) * np.cos(x[:, 1] + 0.3)], axis=1)
    g = (gr * (x - cen)).sum(axis=1) / r + kappa * uex
    ds = meas(100) if box else work.boundary_facets.reshape(-1)
    A, b, act = FQ.assemble_poisson_flux_quad(topo, x, work.cell_tag_values(), work.facet_tag_values(), ds, phi, f, g,
                                              pen_coef=1.2, stab_coef=0.8, robin_coef=kappa, facet_tag=ftag, nq=nq)
    return work, phi, f, g, uex, A, b, act


@pytest.mark.parametrize("n,kappa,ftag,box", [(12, 0.0, 3, True), (12, 1.0, 2, True), (16, 0.0, 3, False), (9, 0.7, 2, False)])
def test_matrix_and_rhs_vs_oracle(P, n, kappa, ftag, box):
    work, phi, f, g, uex, A, b, act = setup(P, n, kappa, ftag, box)
    s = P.NeumannRobinSolver(work, pen_coef=1.2, stab_coef=0.8, robin_coef=kappa, facet_tag=ftag, quadrature_degree=10)
    info = s.assemble(phi, f, g)
    rowptr, col, val, rhs, dof = s.export_csr()
    H = sp.csr_matrix((val, col, rowptr), shape=(rowptr.size - 1,) * 2)
    idx = np.flatnonzero(act)
    assert info["n_active"] == idx.size and np.array_equal(dof, idx)
    assert info["n_full"] == 3 * work.nv + work.nc
    Ao = A[idx][:, idx].tocsr()
    Ao.sort_indices()
    assert np.array_equal(H.indptr, Ao.indptr) and np.array_equal(H.indices, Ao.indices)
    assert np.abs(H.data - Ao.data).max() <= 1e-11 * np.abs(Ao.data).max()
    assert np.abs(rhs - b[idx]).max() <= 1e-11 * np.abs(b).max()


@pytest.mark.parametrize("n,kappa,ftag", [(16, 0.0, 3), (16, 1.0, 2)])
def test_solve_vs_direct(P, n, kappa, ftag):
    work, phi, f, g, uex, A, b, act = setup(P, n, kappa, ftag)
    s = P.NeumannRobinSolver(work, pen_coef=1.2, stab_coef=0.8, robin_coef=kappa, facet_tag=ftag)
    s.assemble(phi, f, g)
    w = s.solve(rtol=1e-12, max_iter=400000)
    wref = OA.solve_direct(A, b, act)
    assert np.all(w[~act] == 0.0)
    assert np.abs(w - wref).max() <= 1e-6 * np.abs(wref).max()
    u, y, p = s.split(w)
    assert u.shape == (work.nv,) and y.shape == (work.nv, 2) and p.shape == (work.nc,)


def test_neumann_problem_converges_on_quadrilaterals(P):
    """du/dn = g on the unit disc, quadrilateral cells: nodal error of u_h at the inside vertices falls ~4x per halving."""
    errs = []
    for n in (16, 32):
        work, phi, f, g, uex, A, b, act = setup(P, n, 0.0, 3)
        s = P.NeumannRobinSolver(work, facet_tag=3)
        s.assemble(phi, f, g)
        u, _, _ = s.split(s.solve(rtol=1e-12, max_iter=400000))
        inside = np.unique(work.cells[work.cell_tag_values() == 1])
        errs.append(np.sqrt(np.mean((u[inside] - uex[inside]) ** 2)))
    assert errs[0] / errs[1] > 3.0, errs


def test_errors(P):
    x0, cells0 = quad_mesh(6)
    mesh = P.Mesh.from_arrays("quadrilateral", x0, cells0.astype(np.int32))
    s = P.NeumannRobinSolver(mesh, facet_tag=3)
    nq2 = mesh.nv + mesh.nf + mesh.nc
    with pytest.raises(ValueError):          # tags not computed yet
        s.assemble(np.ones(nq2), np.ones(mesh.nv), np.ones(mesh.nv))
    with pytest.raises(ValueError):          # phi_h must be Q2: vertices + facets + cells
        s.assemble(np.ones(mesh.nv), np.ones(mesh.nv), np.ones(mesh.nv))
    # a sheared mesh is not a mesh of rectangles
    from phifem_amd.mesh_scripts import NodalFunction
    xs = x0.copy()
    xs[:, 0] += 0.2 * xs[:, 1]
    sheared = P.Mesh.from_arrays("quadrilateral", xs, cells0.astype(np.int32))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        P.compute_tags_measures(sheared, NodalFunction((xs ** 2).sum(axis=1) - 1.0), 1, box_mode=True)
    s2 = P.NeumannRobinSolver(sheared, facet_tag=3)
    with pytest.raises(NotImplementedError):
        s2.assemble(np.ones(sheared.nv + sheared.nf + sheared.nc), np.ones(sheared.nv), np.ones(sheared.nv))
