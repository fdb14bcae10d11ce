"""GPU parity of the Neumann / Robin path (`phx_assemble_poisson_flux`, demo/robin/square/main.py:
98-190 on simplices) against `oracle/assembly_flux.py` with the same conical rule.  Tolerances:
matrix / rhs 1e-11 relative to the largest entry (quadrature evaluation order, FMA, atomics);
solution 1e-6 relative to the direct solve of the oracle matrix at solver rtol 1e-12."""
import warnings

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import assembly as OA
from oracle import assembly_flux as FX
from oracle.topology import Topology

from test_hip_p2 import oracle_space

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    import phifem_amd
    assert phifem_amd._lib.device_count() > 0
    return phifem_amd


def setup(P, d, n, kappa, ftag, box=True, qdeg=10):
    from phifem_amd.mesh_scripts import NodalFunction
    mesh = P.create_box([-1.5] * d, [1.5] * d, [n] * d)
    cen = np.array([0.03, -0.02, 0.01][:d])
    phi1 = ((mesh.x - cen) ** 2).sum(axis=1) - 1.0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ct, ft, sub, meas, maps = P.compute_tags_measures(mesh, NodalFunction(phi1), 1, box_mode=box)
    work = mesh if box else sub
    x = work.x
    ctype = "triangle" if d == 2 else "tetrahedron"
    topo = Topology(ctype, work.cells.astype(np.int64), work.nv)
    topo.c2f, topo.f2c, topo.nf = work.c2f.astype(np.int64), work.f2c.astype(np.int64), work.nf
    Vp = oracle_space(work, topo, 2)
    phi = ((Vp.dof_points(x) - cen) ** 2).sum(axis=1) - 1.0
    uex = np.cos(x[:, 0]) * np.sin(x[:, 1] + 0.3) * (np.cos(0.5 * x[:, 2]) if d == 3 else 1.0)
    f = (3.0 if d == 2 else 3.25) * uex
    r = np.maximum(np.sqrt(((x - cen) ** 2).sum(axis=1)), 1e-12)
    gr = np.stack([-np.sin(x[:, 0]) * np.sin(x[:, 1] + 0.3), np.cos(x[:, 0]) * np.cos(x[:, 1] + 0.3)], axis=1)
    if d == 3:
        cz = np.cos(0.5 * x[:, 2])
        gr = np.concatenate([gr * cz[:, None], (-0.5 * np.sin(0.5 * x[:, 2]) * uex / np.where(cz == 0, 1, cz))[:, None]], axis=1)
    g = (gr * (x - cen)).sum(axis=1) / r + kappa * uex
    ds = meas(100) if box else work.boundary_facets.reshape(-1)
    A, b, act = FX.assemble_poisson_flux(topo, x, work.cell_tag_values(), work.facet_tag_values(), ds, Vp,
                                         phi, f, g, pen_coef=1.2, stab_coef=0.8, robin_coef=kappa,
                                         facet_tag=ftag, qdeg=qdeg)
    return work, phi, f, g, uex, A, b, act


@pytest.mark.parametrize("d,n,kappa,ftag,box", [(2, 12, 1.0, 2, True), (2, 12, 0.0, 3, True), (2, 12, 0.7, 2, False),
                                                (3, 5, 1.0, 2, True), (3, 5, 0.0, 3, False)])
def test_matrix_and_rhs_vs_oracle(P, d, n, kappa, ftag, box):
    work, phi, f, g, uex, A, b, act = setup(P, d, n, kappa, ftag, box)
    s = P.NeumannRobinSolver(work, pen_coef=1.2, stab_coef=0.8, robin_coef=kappa, facet_tag=ftag)
    info = s.assemble(phi, f, g)
    rowptr, col, val, rhs, dof = s.export_csr()
    H = sp.csr_matrix((val, col, rowptr), shape=(rowptr.size - 1,) * 2)
    idx = np.flatnonzero(act)
    assert info["n_active"] == idx.size and np.array_equal(dof, idx)
    assert info["n_full"] == (1 + d) * work.nv + work.nc
    Ao = A[idx][:, idx].tocsr()
    Ao.sort_indices()
    assert np.array_equal(H.indptr, Ao.indptr) and np.array_equal(H.indices, Ao.indices)
    assert np.abs(H.data - Ao.data).max() <= 1e-11 * np.abs(Ao.data).max()
    assert np.abs(rhs - b[idx]).max() <= 1e-11 * np.abs(b).max()


@pytest.mark.parametrize("d,n,kappa,ftag", [(2, 16, 1.0, 2), (2, 16, 0.0, 3), (3, 6, 1.0, 2)])
def test_solve_vs_direct(P, d, n, kappa, ftag):
    work, phi, f, g, uex, A, b, act = setup(P, d, n, kappa, ftag)
    s = P.NeumannRobinSolver(work, pen_coef=1.2, stab_coef=0.8, robin_coef=kappa, facet_tag=ftag)
    s.assemble(phi, f, g)
    w = s.solve(rtol=1e-12, max_iter=400000)
    wref = OA.solve_direct(A, b, act)
    assert np.all(w[~act] == 0.0)
    assert np.abs(w - wref).max() <= 1e-6 * np.abs(wref).max()
    u, y, p = s.split(w)
    assert u.shape == (work.nv,) and y.shape == (work.nv, d) and p.shape == (work.nc,)


def test_robin_problem_converges(P):
    """du/dn + u = g on the unit disc: nodal error of u_h at the inside vertices falls ~4x per halving."""
    errs = []
    for n in (16, 32):
        work, phi, f, g, uex, A, b, act = setup(P, 2, n, 1.0, 2)
        s = P.NeumannRobinSolver(work, robin_coef=1.0)
        s.assemble(phi, f, g)
        u, _, _ = s.split(s.solve(rtol=1e-12, max_iter=400000))
        inside = np.unique(work.cells[work.cell_tag_values() == 1])
        errs.append(np.sqrt(np.mean((u[inside] - uex[inside]) ** 2)))
    assert errs[0] / errs[1] > 3.0


def test_errors(P):
    mesh = P.create_box([-1.5, -1.5], [1.5, 1.5], [8, 8])
    s = P.NeumannRobinSolver(mesh)
    with pytest.raises(ValueError):          # tags not computed yet
        s.assemble(np.ones(mesh.nv + mesh.ne), np.ones(mesh.nv), np.ones(mesh.nv))
    with pytest.raises(ValueError):          # phi_h must be P2
        s.assemble(np.ones(mesh.nv), np.ones(mesh.nv), np.ones(mesh.nv))
    with pytest.raises(ValueError):
        P.NeumannRobinSolver(mesh, facet_tag=9).assemble(np.ones(mesh.nv + mesh.ne), np.ones(mesh.nv), np.ones(mesh.nv))
