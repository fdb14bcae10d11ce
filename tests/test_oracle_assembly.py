"""CPU checks of the assembly oracles (PARITY UNPINNED against the reference): the
closed-form P1 restatement and the quadrature restatement must agree with each other, pass the
polynomial patch tests, and converge at the expected rate."""
import warnings

import numpy as np
import pytest

from oracle import assembly as OA
from oracle import assembly_quad as Q
from oracle import meshgen
from oracle import tagging as T
from oracle.topology import Topology


def problem(d, n, centre=None):
    x, cells = meshgen.create_box([-1.5] * d, [1.5] * d, [n] * d)
    ctype = "triangle" if d == 2 else "tetrahedron"
    topo = Topology(ctype, cells, x.shape[0])
    cen = np.zeros(d) if centre is None else np.asarray(centre)[:d]
    phi = ((x - cen) ** 2).sum(axis=1) - 1.0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ct, ft, _, meas, _, _ = T.compute_tags_measures(ctype, x, topo, T.NodalP1(phi), 1,
                                                        box_mode=True, single_layer_cut=True)
    cv = np.zeros(topo.nc, dtype=np.int64)
    cv[ct.indices] = ct.values
    return x, topo, cv, ft.values, meas(100), phi


@pytest.mark.parametrize("d,n", [(2, 16), (3, 6)])
def test_closed_form_equals_quadrature_p1(d, n):
    x, topo, cv, fv, ds, phi = problem(d, n, [0.03, -0.02, 0.01])
    uex = np.prod(np.sin(x), axis=1)
    A1, b1, a1 = OA.assemble_poisson_wd(topo, x, cv, fv, ds, phi, d * uex, uex)
    V = Q.Space(topo, 1)
    A2, b2, a2 = Q.assemble_poisson_wd_quad(topo, x, cv, fv, ds, V, V, phi, d * uex, uex)
    assert np.array_equal(a1, a2)
    assert abs(A1 - A2).max() <= 1e-13 * abs(A1).max()
    assert np.abs(b1 - b2).max() <= 1e-13 * np.abs(b1).max()


@pytest.mark.parametrize("d,n", [(2, 24), (3, 8)])
def test_p1_patch_test(d, n):
    """f = 0, u_D = u linear: residual of the exact nodal vector vanishes, solve returns it."""
    x, topo, cv, fv, ds, phi = problem(d, n)
    ulin = x @ np.arange(1, d + 1) + 0.5
    A, b, act = OA.assemble_poisson_wd(topo, x, cv, fv, ds, phi, np.zeros(topo.nv), ulin)
    w = np.concatenate([ulin, np.zeros(topo.nv)])
    r = A @ w - b
    assert np.abs(r[act]).max() < 1e-12 and np.all(r[~act] == 0.0)
    ws = OA.solve_direct(A, b, act)
    ua = act[:topo.nv]
    assert np.abs(ws[:topo.nv][ua] - ulin[ua]).max() < 1e-10


@pytest.mark.parametrize("d,n,kphi", [(2, 12, 1), (2, 12, 2), (3, 5, 1), (3, 5, 2)])
def test_p2_patch_test(d, n, kphi):
    """P2 reproduces quadratics (div(grad) terms of main.py:123-128,150 included)."""
    x, topo, cv, fv, ds, phi1 = problem(d, n, [0.03, -0.02, 0.01])
    V2, Vp = Q.Space(topo, 2), Q.Space(topo, kphi)
    cen = np.array([0.03, -0.02, 0.01][:d])
    phi = ((Vp.dof_points(x) - cen) ** 2).sum(axis=1) - 1.0
    pts = V2.dof_points(x)
    u2 = pts[:, 0] ** 2 + 2 * pts[:, 1] ** 2 + pts[:, 0] * pts[:, 1] + (pts[:, 2] ** 2 if d == 3 else 0) + 1.0
    f2 = np.full(V2.ndofs, -(6.0 + (2.0 if d == 3 else 0.0)))
    A, b, act = Q.assemble_poisson_wd_quad(topo, x, cv, fv, ds, V2, Vp, phi, f2, u2)
    w = np.concatenate([u2, np.zeros(V2.ndofs)])
    assert np.abs((A @ w - b)[act]).max() < 1e-10
    ws = OA.solve_direct(A, b, act)
    ua = act[:V2.ndofs]
    assert np.abs(ws[:V2.ndofs][ua] - u2[ua]).max() < 1e-7


def test_p1_convergence_rate_2d():
    """Error at the inside vertices falls ~4x per halving of h (mirrors the slope check of
    demo/interface-elasticity/main.py:392-400)."""
    errs = []
    for n in (32, 64):
        x, topo, cv, fv, ds, phi = problem(2, n)
        uex = np.prod(np.sin(x), axis=1)
        A, b, act = OA.assemble_poisson_wd(topo, x, cv, fv, ds, phi, 2 * uex, uex)
        w = OA.solve_direct(A, b, act)
        inside = np.unique(topo.cells[cv == 1])
        errs.append(np.sqrt(np.mean((w[:topo.nv][inside] - uex[inside]) ** 2)))
    assert errs[0] / errs[1] > 3.0


def test_matrix_is_not_symmetric_and_cg_is_not_applicable():
    """main.py:114 has no transposed partner (hard part 1)."""
    x, topo, cv, fv, ds, phi = problem(2, 24)
    uex = np.prod(np.sin(x), axis=1)
    A, b, act = OA.assemble_poisson_wd(topo, x, cv, fv, ds, phi, 2 * uex, uex)
    idx = np.flatnonzero(act)
    Aa = A[idx][:, idx]
    assert abs(Aa - Aa.T).max() > 1e-3 * abs(Aa).max()
    assert np.all(Aa.diagonal() > 0)
