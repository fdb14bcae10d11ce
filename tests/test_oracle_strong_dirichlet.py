"""CPU checks of the strong-Dirichlet oracle (`oracle/assembly_sd.py`; PARITY UNPINNED against the
reference): it is pinned against the closed-form weak-Dirichlet restatement where the
two coincide (phi_h = 1), it is consistent (a polynomial u = phi_h w_h in the discrete space has a
vanishing residual), and it converges at second order."""
import numpy as np
import pytest

from oracle import assembly as OA
from oracle import assembly_quad as Q
from oracle import assembly_sd as SD

from test_oracle_assembly import problem


@pytest.mark.parametrize("d,n", [(2, 16), (3, 6)])
def test_unit_levelset_reduces_to_weak_dirichlet_uu_block(d, n):
    """phi_h = 1: grad(phi w) = grad w, so main.py:104,105,112-117 of the strong demo are
    main.py:113,114,129-134 of the weak one (closed-form oracle with the penalisation off)."""
    x, topo, cv, fv, ds, phi = problem(d, n, [0.03, -0.02, 0.01])
    f = np.prod(np.cos(x), axis=1)
    V = Q.Space(topo, 1)
    A, b, act = SD.assemble_poisson_sd(topo, x, cv, fv, ds, V, V, np.ones(topo.nv), f, stab_coef=0.7)
    Aw, bw, aw = OA.assemble_poisson_wd(topo, x, cv, fv, ds, phi, f, np.zeros(topo.nv),
                                        pen_coef=0.0, stab_coef=0.7)
    nv = topo.nv
    assert np.array_equal(act, aw[:nv])
    assert abs(A - Aw[:nv, :nv]).max() <= 1e-13 * abs(A).max()
    assert np.abs(b - bw[:nv]).max() <= 1e-13 * np.abs(b).max()


@pytest.mark.parametrize("d,n,k", [(2, 12, 1), (2, 10, 2), (3, 5, 1), (3, 5, 2)])
def test_consistency_for_polynomial_solution(d, n, k):
    """phi_h = r^2 - 1 exactly in P2, w a polynomial of degree k: u = phi w is smooth, f = -lap u
    lies in the w space, every jump vanishes and the residual A w - b is round-off."""
    x, topo, cv, fv, ds, _ = problem(d, n)
    V, Vp = Q.Space(topo, k), Q.Space(topo, 2)
    pp, pw = Vp.dof_points(x), V.dof_points(x)
    phi = (pp ** 2).sum(axis=1) - 1.0
    gw = np.array([0.5, -0.25, 0.3][:d])
    w = 1.0 + pw @ gw
    xgw = pw @ gw
    lapw = 0.0
    if k == 2:
        w = w + 0.2 * pw[:, 0] ** 2 - 0.1 * pw[:, 0] * pw[:, 1]
        xgw = xgw + 0.4 * pw[:, 0] ** 2 - 0.2 * pw[:, 0] * pw[:, 1]
        lapw = 0.4
    f = -(2.0 * d * w + 4.0 * xgw + ((pw ** 2).sum(axis=1) - 1.0) * lapw)
    A, b, act = SD.assemble_poisson_sd(topo, x, cv, fv, ds, V, Vp, phi, f)
    r = A @ w - b
    assert np.abs(r[act]).max() <= 1e-11 * np.abs(b).max()
    assert np.all(r[~act] == 0.0)


def test_second_order_convergence_2d():
    errs = []
    for n in (24, 48):
        x, topo, cv, fv, ds, phi = problem(2, n)
        V = Q.Space(topo, 1)
        g = 1.0 + 0.5 * x[:, 0] + 0.25 * x[:, 1]
        uex = (1.0 - (x ** 2).sum(axis=1)) * g
        f = 4.0 * g + 4.0 * (0.5 * x[:, 0] + 0.25 * x[:, 1])
        A, b, act = SD.assemble_poisson_sd(topo, x, cv, fv, ds, V, V, phi, f)
        w = OA.solve_direct(A, b, act)
        inside = np.unique(topo.cells[cv == 1])
        errs.append(np.sqrt(np.mean((w[inside] * phi[inside] - uex[inside]) ** 2)))
    assert errs[0] / errs[1] > 3.0


def test_bilinear_form_is_symmetric_up_to_the_boundary_term():
    x, topo, cv, fv, ds, phi = problem(2, 16)
    V = Q.Space(topo, 1)
    A, _, _ = SD.assemble_poisson_sd(topo, x, cv, fv, ds, V, V, phi, np.ones(topo.nv))
    A0, _, _ = SD.assemble_poisson_sd(topo, x, cv, fv, np.zeros(0, dtype=np.int64), V, V, phi, np.ones(topo.nv))
    assert abs(A0 - A0.T).max() <= 1e-13 * abs(A0).max()
    assert abs(A - A.T).max() > 0.0
