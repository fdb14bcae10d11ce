"""GPU parity of the interface-elasticity path (a12, BASELINE configs[3] in 2-D and 3-D form)
against the numpy oracle.  Tolerance: 1e-11 relative to the largest matrix / vector entry
(atomic accumulation order, FMA)."""
import warnings

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import elasticity as EL
from oracle import tagging as T
from oracle.topology import Topology

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    import phifem_amd
    assert phifem_amd._lib.device_count() > 0
    return phifem_amd


def setup(P, d, n, E_out, centre=None, f_scale=1.0):
    from phifem_amd.mesh_scripts import NodalFunction
    mesh = P.create_box([-1.5] * d, [1.5] * d, [n] * d)
    x = mesh.x
    cen = np.zeros(d) if centre is None else np.asarray(centre)[:d]
    phi = 1.0 - ((x - cen) ** 2).sum(axis=1)            # data.py:39-40
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        P.compute_tags_measures(mesh, NodalFunction(phi), 1, box_mode=True)   # main.py:115-117
    ctype = "triangle" if d == 2 else "tetrahedron"
    topo = Topology(ctype, mesh.cells.astype(np.int64), mesh.nv)
    topo.c2f, topo.f2c, topo.nf = mesh.c2f.astype(np.int64), mesh.f2c.astype(np.int64), mesh.nf
    bf = mesh.boundary_facets
    from oracle.points import FACET_VERTS
    bcv = np.unique(np.take_along_axis(mesh.cells[bf[:, 0]], FACET_VERTS[ctype][bf[:, 1]], axis=1))
    return mesh, topo, x, phi, bcv


@pytest.mark.parametrize("d,n,E_out", [(2, 12, 1.0e-3), (2, 16, 1.0), (3, 5, 1.0e-3), (3, 6, 0.5)])
def test_elasticity_matrix_and_rhs_vs_oracle(P, d, n, E_out):
    mesh, topo, x, phi, bcv = setup(P, d, n, E_out, centre=[0.04, -0.03, 0.02])
    rng = np.random.default_rng(5)
    f = np.sin(x @ rng.standard_normal((d, d))) + 0.3
    uD = np.cos(x @ rng.standard_normal((d, d)))
    from phifem_amd.mesh_scripts import BoundaryMeasure
    meas = BoundaryMeasure(mesh, True)
    A, b, act = EL.assemble_elasticity_if(topo, x, mesh.cell_tag_values(), mesh.facet_tag_values(),
                                          meas(100), meas(101), phi, f, uD, bcv, E_in=1.0, E_out=E_out)
    s = P.InterfaceElasticitySolver(mesh, E_in=1.0, E_out=E_out)
    info = s.assemble(phi, f, uD, bcv)
    rowptr, col, val, rhs, dof = s.export_csr()
    idx = np.flatnonzero(act)
    assert info["n_active"] == idx.size and np.array_equal(dof, idx)
    H = sp.csr_matrix((val, col, rowptr), shape=(idx.size, idx.size))
    Ao = A[idx][:, idx].tocsr()
    scale = np.abs(Ao.data).max()
    assert abs(H - Ao).max() <= 1e-11 * scale
    assert np.abs(rhs - b[idx]).max() <= 1e-11 * max(np.abs(b).max(), 1e-300)
    xv = rng.standard_normal(idx.size)
    y = s.spmv(xv)
    assert np.abs(y - Ao @ xv).max() <= 1e-11 * np.abs(Ao @ xv).max()


@pytest.mark.parametrize("d,n", [(2, 16), (3, 6)])
def test_elasticity_patch_test_through_hip(P, d, n):
    """Same material, linear displacement: the exact nodal vector satisfies the HIP system."""
    mesh, topo, x, phi, bcv = setup(P, d, n, 1.0)
    G = np.array([[0.3, -0.2, 0.1], [0.15, 0.25, -0.05], [0.05, 0.1, -0.3]])[:d, :d]
    ulin = x @ G.T + 0.1
    lam, mu = EL.lame(1.0, 0.3)
    sig = lam * np.trace(G) * np.eye(d) + mu * (G + G.T)
    s = P.InterfaceElasticitySolver(mesh, E_in=1.0, E_out=1.0)
    s.assemble(phi, np.zeros((mesh.nv, d)), ulin, bcv)
    rowptr, col, val, rhs, dof = s.export_csr()
    B, nv = EL.Blocks(d), mesh.nv
    w = np.zeros(B.C * nv)
    for a in range(d):
        for side in (0, 1):
            w[B.u(side, a) * nv:(B.u(side, a) + 1) * nv] = ulin[:, a]
            for bb in range(d):
                w[B.y(side, a, bb) * nv:(B.y(side, a, bb) + 1) * nv] = -sig[a, bb]
    r = s.spmv(w[dof]) - rhs
    assert np.abs(r).max() <= 1e-10 * np.abs(val).max()
    blocks = s.blocks(w)
    assert blocks["u_in"].shape == (nv, d) and blocks["y_out"].shape == (nv, d, d)


def test_elasticity_solve_2d_same_material(P):
    """With equal materials (cond ~5e4) the Jacobi-BiCGStab solve reproduces the linear field."""
    d, n = 2, 16
    mesh, topo, x, phi, bcv = setup(P, d, n, 1.0)
    G = np.array([[0.3, -0.2], [0.15, 0.25]])
    ulin = x @ G.T + 0.1
    s = P.InterfaceElasticitySolver(mesh, E_in=1.0, E_out=1.0)
    s.assemble(phi, np.zeros((mesh.nv, d)), ulin, bcv)
    w = s.solve(rtol=1e-12, max_iter=50000)
    assert s.stats["relres"] <= 1e-12
    b = s.blocks(w)
    vin = np.unique(mesh.cells[mesh.cell_tag_values() != 3])
    assert np.abs(b["u_in"][vin] - ulin[vin]).max() < 1e-7


def test_elasticity_demo_problem_2d(P):
    """The demo's interface problem (E_in = 1, E_out = 1e-3, nu = 0.3, data.py:14-22, exact
    solution data.py:43-49, f = -div(sigma_in(cos_vec))/E_in, main.py:150) solved on the GPU:
    the error against the exact solution falls with h."""
    import sympy as sy
    E_in, E_out, nu = 1.0, 1e-3, 0.3
    X, Y = sy.symbols("x y")
    r = sy.sqrt(X ** 2 + Y ** 2)
    u = sy.Matrix([sy.cos(r), sy.cos(r)])
    lam, mu = EL.lame(E_in, nu)
    grad = u.jacobian([X, Y])
    sig = lam * (grad[0, 0] + grad[1, 1]) * sy.eye(2) + mu * (grad + grad.T)
    f_sym = -sy.Matrix([sy.diff(sig[0, 0], X) + sy.diff(sig[0, 1], Y),
                        sy.diff(sig[1, 0], X) + sy.diff(sig[1, 1], Y)]) / E_in
    ffun = sy.lambdify((X, Y), f_sym, "numpy")
    errs, its = [], []
    for n in (15, 30):
        mesh, topo, x, phi, bcv = setup(P, 2, n, E_out)
        rr = np.sqrt((x ** 2).sum(axis=1))
        val = np.cos(rr) - np.cos(1.0) / E_in
        val = np.where(rr < 1.0, val * (E_in / E_out), val)
        ue = np.stack([val, val], axis=1)
        xs = np.where(np.abs(x) < 1e-12, 1e-9, x)
        fh = np.array(ffun(xs[:, 0], xs[:, 1])).reshape(2, -1).T
        s = P.InterfaceElasticitySolver(mesh, E_in=E_in, E_out=E_out, nu_in=nu, nu_out=nu)
        s.assemble(phi, fh, ue, bcv)
        w = s.solve(rtol=1e-10, max_iter=200000)
        assert s.stats["relres"] <= 1e-10
        its.append(s.stats["iterations"])
        vin = np.unique(mesh.cells[mesh.cell_tag_values() == 1])
        errs.append(np.abs(s.blocks(w)["u_in"][vin] - ue[vin]).max() / np.abs(ue[vin]).max())
    print("elasticity 2-D demo: iterations", its, "errors", errs)
    assert errs[0] < 2e-2 and errs[0] / errs[1] > 2.5


@pytest.mark.parametrize("E_out", [1.0, 1.0e-3])
def test_elasticity_solve_3d_vs_direct(P, E_out):
    """3-D interface problem (configs[3] in miniature): GPU Jacobi-BiCGStab against a direct solve
    of the oracle's system."""
    import scipy.sparse.linalg as spla
    d, n = 3, 8
    mesh, topo, x, phi, bcv = setup(P, d, n, E_out)
    f = np.stack([np.sin(x[:, 0]) + 0.2, np.cos(x[:, 1]), 0.5 * x[:, 2]], axis=1)
    uD = 0.1 * np.stack([x[:, 0] * x[:, 1], np.sin(x[:, 2]), x[:, 0] - x[:, 1]], axis=1)
    from phifem_amd.mesh_scripts import BoundaryMeasure
    meas = BoundaryMeasure(mesh, True)
    A, b, act = EL.assemble_elasticity_if(topo, x, mesh.cell_tag_values(), mesh.facet_tag_values(),
                                          meas(100), meas(101), phi, f, uD, bcv, E_in=1.0, E_out=E_out)
    s = P.InterfaceElasticitySolver(mesh, E_in=1.0, E_out=E_out)
    s.assemble(phi, f, uD, bcv)
    w = s.solve(rtol=1e-11, max_iter=200000)
    print("elasticity 3-D n=8 E_out", E_out, s.stats)
    assert s.stats["relres"] <= 1e-11
    idx = np.flatnonzero(act)
    wo = np.zeros_like(w)
    wo[idx] = spla.spsolve(A[idx][:, idx].tocsc(), b[idx])
    assert np.abs(w - wo).max() <= 1e-6 * np.abs(wo).max()
    assert np.all(w[~act] == 0.0)


@pytest.mark.parametrize("d,n,ratio", [(3, 12, 6), (2, 40, 8), (3, 11, 5)])
def test_elasticity_coarse_correction_vs_direct(P, d, n, ratio):
    """Two-level preconditioner (PHX_OPT_EL_COARSE: vertex blocks + Galerkin coarse problem on trilinear functions of
    spacing ratio * h, probed with the solver's own operator and inverted densely): the solution still equals a direct
    solve of the oracle's system and the option switches it (boxes this small have no smooth bulk modes to remove: about
    the same iteration count).  (3, 11, 5): a last coarse cell that is cut short by the box."""
    import scipy.sparse.linalg as spla
    E_out = 1.0e-3
    mesh, topo, x, phi, bcv = setup(P, d, n, E_out, centre=[0.02, -0.01, 0.03])
    f = np.stack([np.sin(x[:, 0]) + 0.2, np.cos(x[:, 1]), 0.5 * x[:, -1]][:d], axis=1)
    uD = 0.1 * np.stack([x[:, 0] * x[:, 1], np.sin(x[:, -1]), x[:, 0] - x[:, 1]][:d], axis=1)
    from phifem_amd.mesh_scripts import BoundaryMeasure
    meas = BoundaryMeasure(mesh, True)
    A, b, act = EL.assemble_elasticity_if(topo, x, mesh.cell_tag_values(), mesh.facet_tag_values(),
                                          meas(100), meas(101), phi, f, uD, bcv, E_in=1.0, E_out=E_out)
    idx = np.flatnonzero(act)
    wo = np.zeros(A.shape[0])
    wo[idx] = spla.spsolve(A[idx][:, idx].tocsc(), b[idx])
    its = {}
    for coarse in (0, ratio):
        s = P.InterfaceElasticitySolver(mesh, E_in=1.0, E_out=E_out, deterministic=True, coarse=coarse)
        s.assemble(phi, f, uD, bcv)
        w = s.solve(rtol=1e-11, max_iter=200000)
        assert s.stats["relres"] <= 1e-11 and s.stats["converged"]
        assert s.stats["precond"] == ("vertex-block-jacobi+coarse" if coarse else "vertex-block-jacobi"), s.stats
        assert np.abs(w - wo).max() <= 1e-6 * np.abs(wo).max()
        assert np.all(w[~act] == 0.0)
        its[coarse] = s.stats["iterations"]
    print(f"elasticity d={d} n={n}: {its[0]} iterations with the vertex blocks, {its[ratio]} with the coarse correction (H = {ratio} h)")
    assert its[ratio] <= 1.3 * its[0]
    with pytest.raises(ValueError):
        P._lib.check(P._lib.lib.phx_set_option(mesh._h, P._lib.OPT_EL_COARSE, 3))


def test_elasticity_coarse_correction_cuts_the_iterations(P):
    """32^3 box (too large for a direct solve here): the two preconditioners reach the same solution, the coarse
    correction in clearly fewer iterations (CPU prototype: 129 -> 89 at rtol 1e-8)."""
    d, n = 3, 32
    mesh, topo, x, phi, bcv = setup(P, d, n, 1.0e-3)
    f = np.stack([np.sin(x[:, 0]) + 0.2, np.cos(x[:, 1]), 0.5 * x[:, 2]], axis=1)
    uD = 0.1 * np.stack([x[:, 0] * x[:, 1], np.sin(x[:, 2]), x[:, 0] - x[:, 1]], axis=1)
    sol, its = {}, {}
    for coarse in (0, 8):
        s = P.InterfaceElasticitySolver(mesh, E_in=1.0, E_out=1.0e-3, deterministic=True, coarse=coarse)
        s.assemble(phi, f, uD, bcv)
        sol[coarse] = s.solve(rtol=1e-10, max_iter=200000)
        assert s.stats["converged"] and s.stats["relres"] <= 1e-10
        its[coarse] = s.stats["iterations"]
        s._free()
    print(f"elasticity 32^3: {its[0]} iterations with the vertex blocks, {its[8]} with the coarse correction")
    assert np.abs(sol[0] - sol[8]).max() <= 1e-6 * np.abs(sol[0]).max()
    assert its[8] <= 0.8 * its[0]
