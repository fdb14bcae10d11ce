"""CPU checks of the quadrilateral Neumann / Robin oracle (`oracle/assembly_flux_quad.py`, the formulation of
demo/neumann/square/main.py:113-158 on its own cell type; PARITY UNPINNED): structure, symmetry without the one-sided
term, quadrature sensitivity confined to the non-polynomial terms, and second-order convergence of the manufactured
Neumann and Robin problems on the unit disc."""
import warnings

import numpy as np
import pytest

from oracle import assembly as OA
from oracle import assembly_flux_quad as FQ
from oracle import tagging as T
from oracle.topology import Topology


def quad_mesh(n, lo=-1.5, hi=1.5):
    t = np.linspace(lo, hi, n + 1)
    X, Y = np.meshgrid(t, t, indexing="xy")
    x = np.stack([X.reshape(-1), Y.reshape(-1)], axis=1)
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="xy")
    v0 = (j * (n + 1) + i).reshape(-1)
    cells = np.stack([v0, v0 + 1, v0 + n + 1, v0 + n + 2], axis=1)     # tensor-product order
    return x, cells


def disc_problem(n, kappa):
    x, cells = quad_mesh(n)
    topo = Topology("quadrilateral", cells, x.shape[0])
    ls = lambda p: p[0] ** 2 + p[1] ** 2 - 1.0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ct, ft, _, meas, _, _ = T.compute_tags_measures("quadrilateral", x, topo, ls, 1, box_mode=True)
    cv = np.zeros(topo.nc, dtype=np.int64)
    cv[ct.indices] = ct.values
    pp = FQ.q2_dof_points(topo, x)
    phi = (pp ** 2).sum(axis=1) - 1.0
    uex = np.cos(x[:, 0]) * np.sin(x[:, 1] + 0.3)
    gux = -np.sin(x[:, 0]) * np.sin(x[:, 1] + 0.3)
    guy = np.cos(x[:, 0]) * np.cos(x[:, 1] + 0.3)
    r = np.maximum(np.sqrt((x ** 2).sum(axis=1)), 1e-12)
    g = (gux * x[:, 0] + guy * x[:, 1]) / r + kappa * uex
    f = 3.0 * uex
    return x, topo, cv, ft.values, meas(100), phi, f, g, uex


@pytest.mark.parametrize("kappa,ftag", [(0.0, 3), (1.0, 2)])
def test_second_order_convergence(kappa, ftag):
    errs = []
    for n in (16, 32):
        x, topo, cv, fv, ds, phi, f, g, uex = disc_problem(n, kappa)
        A, b, act = FQ.assemble_poisson_flux_quad(topo, x, cv, fv, ds, phi, f, g, robin_coef=kappa, facet_tag=ftag)
        w = OA.solve_direct(A, b, act)
        inside = np.unique(topo.cells[cv == 1])
        errs.append(np.sqrt(np.mean((w[inside] - uex[inside]) ** 2)))
    assert errs[0] / errs[1] > 3.0, errs


def test_structure_and_symmetry():
    x, topo, cv, fv, ds, phi, f, g, uex = disc_problem(12, 0.5)
    A, b, act = FQ.assemble_poisson_flux_quad(topo, x, cv, fv, ds, phi, f, g, robin_coef=0.5)
    nv, nc = topo.nv, topo.nc
    assert A.shape == (3 * nv + nc,) * 2
    assert np.array_equal(np.flatnonzero(act[3 * nv:]), np.flatnonzero(cv == 2))       # p on cut cells only
    cutv = np.unique(topo.cells[cv == 2])
    assert np.array_equal(np.flatnonzero(act[nv:2 * nv]), cutv)                         # y on their vertices
    A0, _, _ = FQ.assemble_poisson_flux_quad(topo, x, cv, fv, np.zeros(0, dtype=np.int64), phi, f, g, robin_coef=0.5)
    assert abs(A0 - A0.T).max() <= 1e-12 * abs(A0).max()
    assert abs(A - A.T).max() > 0
    # the bulk block (Q1 stiffness + mass on rectangles) has the closed-form diagonal 2/3 (hx/hy + hy/hx) + hx hy / 9
    inside = np.flatnonzero(cv == 1)
    h = 3.0 / 12
    K = A[:nv, :nv]
    v_int = np.flatnonzero((x ** 2).sum(axis=1) < 0.25)      # far from Gamma_h: no stabilisation term reaches them
    assert len(v_int) > 0 and np.all(np.isin(v_int, np.unique(topo.cells[inside])))
    assert np.allclose(K.diagonal()[v_int], 4 * (2.0 / 3.0 + h * h / 9.0), rtol=1e-12)


def test_quadrature_only_moves_the_nonpolynomial_terms():
    x, topo, cv, fv, ds, phi, f, g, uex = disc_problem(12, 1.0)
    A6, b6, _ = FQ.assemble_poisson_flux_quad(topo, x, cv, fv, ds, phi, f, g, robin_coef=1.0, nq=6)
    A8, b8, _ = FQ.assemble_poisson_flux_quad(topo, x, cv, fv, ds, phi, f, g, robin_coef=1.0, nq=8)
    assert 0 < abs(A6 - A8).max() < 1e-6 * abs(A6).max()
    N6, _, _ = FQ.assemble_poisson_flux_quad(topo, x, cv, fv, ds, phi, f, g, robin_coef=0.0, nq=6)
    N8, _, _ = FQ.assemble_poisson_flux_quad(topo, x, cv, fv, ds, phi, f, g, robin_coef=0.0, nq=8)
    assert abs(N6 - N8).max() < 1e-12 * abs(N6).max()
    assert np.abs(b6 - b8).max() < 1e-6 * np.abs(b6).max()


def test_rejects_non_rectangles():
    x, cells = quad_mesh(4)
    x = x.copy()
    x[7] += [0.05, 0.02]
    topo = Topology("quadrilateral", cells, x.shape[0])
    with pytest.raises(NotImplementedError):
        FQ.rect_geometry(x, topo.cells)


def test_neumann_demo_data_matches_reference_fixture():
    """demo/neumann/square/data.py of this repo against values of the reference's data module at 400 seeded points
    (tests/golden/neumann_data.npz, made by tests/golden/make_neumann_data.py)."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("neumann_demo_data", os.path.join(root, "demo", "neumann", "square", "data.py"))
    mine = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mine)
    gold = np.load(os.path.join(root, "tests", "golden", "neumann_data.npz"))
    for name in ("detection_levelset", "levelset", "exact_solution", "source_term", "neumann_data"):
        got = getattr(mine, name)(gold["x"].copy())
        assert np.abs(got - gold[name]).max() <= 1e-12 * max(1.0, np.abs(gold[name]).max()), name
    assert np.array_equal(np.sign(mine.detection_levelset(gold["x"])), np.sign(gold["detection_levelset"]))
