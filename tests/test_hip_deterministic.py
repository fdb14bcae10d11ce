"""PHX_OPT_DETERMINISTIC: the scattering assemblies (P2 weak Dirichlet, interface elasticity) sum f64
atomics in arrival order, so two identical runs produced matrices that differ in their last bits and BiCGStab -- 700 to
900 iterations on these systems -- turned that into a +-13 % spread of the iteration count.  With the option the element
kernels accumulate exactly (two passes: per-slot exponent, then two accumulators whose partial sums are exact in f64)
and the dot products are summed in a fixed order: same bits, same iteration count, same solution on every run."""
import warnings

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    import phifem_amd
    assert phifem_amd._lib.device_count() > 0
    return phifem_amd


def _p2_problem(P, n):
    from phifem_amd.mesh_scripts import NodalFunction
    mesh = P.create_box([-1.5] * 3, [1.5] * 3, [n] * 3)
    cen = np.array([0.03, -0.02, 0.01])
    phi1 = ((mesh.x - cen) ** 2).sum(axis=1) - 1.0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        P.compute_tags_measures(mesh, NodalFunction(phi1), 1, box_mode=True, single_layer_cut=True)
    pts = mesh.p2_dof_points()
    phi = ((pts - cen) ** 2).sum(axis=1) - 1.0
    uex = np.prod(np.sin(pts), axis=1)
    return mesh, phi, 3.0 * uex, uex


@pytest.mark.parametrize("n", [12, 28])     # 28: structured system (stencil rows), 12: every row stored
def test_p2_runs_are_bit_identical(P, n):
    mesh, phi, f, uex = _p2_problem(P, n)
    runs = []
    for rep in range(3):
        s = P.PhiFEMSolver(mesh, degree=2, levelset_degree=2, deterministic=rep < 2)
        info = s.assemble(phi, f, uex)
        rhs, dof = s.export_rhs_dof()
        rowptr, col, val, _, _ = s.export_csr()
        w = s.solve(rtol=1e-9, max_iter=40000)
        assert s.stats["converged"]
        runs.append((rhs, val, w, s.stats["iterations"], col, rowptr))
    a, b, c = runs
    assert np.array_equal(a[0], b[0]), "right-hand sides differ between two deterministic runs"
    assert np.array_equal(a[1], b[1]), "matrix values differ between two deterministic runs"
    assert a[3] == b[3] and np.array_equal(a[2], b[2]), (a[3], b[3])
    # against the plain atomics: the same numbers to round-off
    assert np.array_equal(a[4], c[4]) and np.array_equal(a[5], c[5])
    assert np.abs(a[1] - c[1]).max() <= 1e-13 * np.abs(c[1]).max()
    assert np.abs(a[0] - c[0]).max() <= 1e-13 * np.abs(c[0]).max()
    # (two solves to rtol 1e-9 of a system of condition 1e6-1e7: the solutions agree to ~cond * rtol)
    assert np.abs(a[2] - c[2]).max() <= 1e-4 * np.abs(c[2]).max()
    print(f"P2 n={n}: iterations deterministic {a[3]} / {b[3]}, plain atomics {c[3]}")


def test_elasticity_runs_are_bit_identical(P):
    from phifem_amd.mesh_scripts import NodalFunction
    n = 10
    mesh = P.create_box([-1.5] * 3, [1.5] * 3, [n] * 3)
    x = mesh.x
    phi = 1.0 - (x ** 2).sum(axis=1)
    f = np.stack([np.sin(x[:, 0]) + 0.2, np.cos(x[:, 1]), 0.5 * x[:, 2]], axis=1)
    uD = 0.1 * np.stack([x[:, 0] * x[:, 1], np.sin(x[:, 2]), x[:, 0] - x[:, 1]], axis=1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        P.compute_tags_measures(mesh, NodalFunction(phi), 1, box_mode=True)
    n1 = n + 1
    v = np.arange(mesh.nv)
    i, j, k = v % n1, (v // n1) % n1, v // (n1 * n1)
    bcv = np.flatnonzero((i == 0) | (i == n) | (j == 0) | (j == n) | (k == 0) | (k == n))
    runs = []
    for rep in range(3):
        s = P.InterfaceElasticitySolver(mesh, deterministic=rep < 2)
        s.assemble(phi, f, uD, bcv)
        rowptr, col, val, rhs, dof = s.export_csr()
        w = s.solve(rtol=1e-9, max_iter=200000)
        assert s.stats["converged"]
        runs.append((rhs, val, w, s.stats["iterations"]))
    a, b, c = runs
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert a[3] == b[3] and np.array_equal(a[2], b[2]), (a[3], b[3])
    assert np.abs(a[1] - c[1]).max() <= 1e-13 * np.abs(c[1]).max()
    assert np.abs(a[2] - c[2]).max() <= 1e-4 * np.abs(c[2]).max()
    print(f"elasticity n={n}: iterations deterministic {a[3]} / {b[3]}, plain atomics {c[3]}")
