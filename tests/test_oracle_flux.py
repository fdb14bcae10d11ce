"""CPU checks of the Neumann / Robin oracle (`oracle/assembly_flux.py`; PARITY UNPINNED): the
discrete problem is consistent for a solution in the discrete space, and the manufactured
Neumann and Robin problems on the unit disc converge at second order."""
import numpy as np
import pytest

from oracle import assembly as OA
from oracle import assembly_flux as FX
from oracle import assembly_quad as Q

from test_oracle_assembly import problem


def disc_problem(n, kappa):
    x, topo, cv, fv, ds, _ = problem(2, n)
    Vp = Q.Space(topo, 2)
    pp = Vp.dof_points(x)
    phi = (pp ** 2).sum(axis=1) - 1.0
    uex = np.cos(x[:, 0]) * np.sin(x[:, 1] + 0.3)
    gux = -np.sin(x[:, 0]) * np.sin(x[:, 1] + 0.3)
    guy = np.cos(x[:, 0]) * np.cos(x[:, 1] + 0.3)
    r = np.maximum(np.sqrt((x ** 2).sum(axis=1)), 1e-12)
    g = (gux * x[:, 0] + guy * x[:, 1]) / r + kappa * uex      # du/dn + kappa u, extended radially
    f = 3.0 * uex                                               # -lap u + u
    return x, topo, cv, fv, ds, Vp, phi, f, g, uex


@pytest.mark.parametrize("kappa,ftag", [(0.0, 3), (1.0, 2)])
def test_second_order_convergence(kappa, ftag):
    errs = []
    for n in (16, 32):
        x, topo, cv, fv, ds, Vp, phi, f, g, uex = disc_problem(n, kappa)
        A, b, act = FX.assemble_poisson_flux(topo, x, cv, fv, ds, Vp, phi, f, g, robin_coef=kappa,
                                             facet_tag=ftag)
        w = OA.solve_direct(A, b, act)
        inside = np.unique(topo.cells[cv == 1])
        errs.append(np.sqrt(np.mean((w[inside] - uex[inside]) ** 2)))
    assert errs[0] / errs[1] > 3.0


def test_structure_and_symmetry():
    x, topo, cv, fv, ds, Vp, phi, f, g, uex = disc_problem(12, 0.5)
    A, b, act = FX.assemble_poisson_flux(topo, x, cv, fv, ds, Vp, phi, f, g, robin_coef=0.5)
    nv, nc = topo.nv, topo.nc
    assert A.shape == (3 * nv + nc,) * 2
    # p lives on cut cells only, y on their vertices
    assert np.array_equal(np.flatnonzero(act[3 * nv:]), np.flatnonzero(cv == 2))
    cutv = np.unique(topo.cells[cv == 2])
    assert np.array_equal(np.flatnonzero(act[nv:2 * nv]), cutv)
    # without the one-sided ds term the form is symmetric
    A0, _, _ = FX.assemble_poisson_flux(topo, x, cv, fv, np.zeros(0, dtype=np.int64), Vp, phi, f, g, robin_coef=0.5)
    assert abs(A0 - A0.T).max() <= 1e-12 * abs(A0).max()
    assert abs(A - A.T).max() > 0


def test_quadrature_degree_only_moves_the_nonpolynomial_terms():
    """|grad phi_h| is the only non-polynomial factor: rules of degree 10 and 14 agree to
    quadrature accuracy; with robin_coef = 0 the matrix is polynomial and they agree to round-off."""
    x, topo, cv, fv, ds, Vp, phi, f, g, uex = disc_problem(12, 1.0)
    A10, b10, _ = FX.assemble_poisson_flux(topo, x, cv, fv, ds, Vp, phi, f, g, robin_coef=1.0, qdeg=10)
    A14, b14, _ = FX.assemble_poisson_flux(topo, x, cv, fv, ds, Vp, phi, f, g, robin_coef=1.0, qdeg=14)
    assert 0 < abs(A10 - A14).max() < 1e-6 * abs(A10).max()
    N10, _, _ = FX.assemble_poisson_flux(topo, x, cv, fv, ds, Vp, phi, f, g, robin_coef=0.0, qdeg=10)
    N14, _, _ = FX.assemble_poisson_flux(topo, x, cv, fv, ds, Vp, phi, f, g, robin_coef=0.0, qdeg=14)
    assert abs(N10 - N14).max() < 1e-12 * abs(N10).max()
    assert np.abs(b10 - b14).max() < 1e-6 * np.abs(b10).max()


def test_robin_demo_data_matches_reference_fixture():
    """demo/robin/square/data.py of this repo against values of the reference's data module at 400
    seeded points (tests/golden/robin_data.npz, made by tests/golden/make_robin_data.py)."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("robin_demo_data", os.path.join(root, "demo", "robin", "square", "data.py"))
    mine = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mine)
    gold = np.load(os.path.join(root, "tests", "golden", "robin_data.npz"))
    assert float(gold["robin_coef"]) == mine.ROBIN_COEF
    for name in ("detection_levelset", "levelset", "exact_solution", "source_term", "robin_data"):
        got = getattr(mine, name)(gold["x"].copy())
        assert np.abs(got - gold[name]).max() <= 1e-12 * max(1.0, np.abs(gold[name]).max()), name
    # the sign pattern (what the tagging sees) is identical
    assert np.array_equal(np.sign(mine.detection_levelset(gold["x"])), np.sign(gold["detection_levelset"]))
