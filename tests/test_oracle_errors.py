"""CPU checks of the error-evaluation oracle (`oracle/errors.py`, PARITY UNPINNED): the degree-3
reference space reproduces cubics, its closed-form mass / stiffness integrals agree with an
independent quadrature, and the per-cell errors of a known field have their analytic values."""
import numpy as np
import pytest

from oracle import assembly_quad as Q
from oracle import errors as E
from oracle import meshgen
from oracle.topology import Topology


@pytest.mark.parametrize("d", [2, 3])
def test_reference_space_is_nodal_and_complete(d):
    lam = E.reference_nodes(d)
    assert lam.shape == ((10, 3) if d == 2 else (20, 4))
    assert np.allclose(lam.sum(axis=1), 1.0)
    M, S = E.reference_matrices(d)
    # partition of unity: sum_ij M_ij = 1 (per unit volume); S is symmetric under (m, i) <-> (n, j)
    assert abs(M.sum() - 1.0) < 1e-12
    assert np.abs(S - S.transpose(1, 0, 3, 2)).max() < 1e-11
    # mass matrix against the conical quadrature of the assembly oracle (degree 6)
    q, w = Q.simplex_rule(d, 6)
    al = E._monomials(d)
    V = np.array([[np.prod(l ** np.array(a)) for a in al] for l in lam])
    C = np.linalg.inv(V)
    Nq = np.array([[np.prod(l ** np.array(a)) for a in al] for l in q]) @ C
    assert np.abs(np.einsum("q,qi,qj->ij", w, Nq, Nq) - M).max() < 1e-13


@pytest.mark.parametrize("d,n", [(2, 3), (3, 2)])
def test_cubic_is_reproduced_and_linear_error_is_analytic(d, n):
    x, cells = meshgen.create_box([0.0] * d, [1.0] * d, [n] * d)
    topo = Topology("triangle" if d == 2 else "tetrahedron", cells, x.shape[0])
    lam = E.reference_nodes(d)
    pts = np.einsum("jm,cmd->cjd", lam, x[cells])

    def cubic(p):
        return p[..., 0] ** 3 - 2.0 * p[..., 0] * p[..., 1] ** 2 + p[..., -1] + 0.5

    # u_h = 0: the "error" is the interpolant itself; a cubic is interpolated exactly, so the
    # global integrals are those of the cubic over the unit box
    l2, h10, norms = E.cell_errors(topo, x, 1, cells, np.zeros((1, topo.nv)), cubic(pts)[..., None])
    ref = Q.Space(topo, 1)
    lamq, wq = Q.simplex_rule(d, 6)
    g, vol, _ = E.simplex_geometry(x, cells)
    xq = np.einsum("qm,cmd->cqd", lamq, x[cells])
    exact_l2 = np.einsum("q,c,cq->", wq, vol, cubic(xq) ** 2)
    assert abs(norms[0] - exact_l2) < 1e-12 and abs(norms[2] - exact_l2) < 1e-12
    gx = 3 * xq[..., 0] ** 2 - 2 * xq[..., 1] ** 2
    gy = -4.0 * xq[..., 0] * xq[..., 1] + (1.0 if d == 2 else 0.0)
    gsq = gx ** 2 + gy ** 2 + (1.0 if d == 3 else 0.0)
    assert abs(norms[1] - np.einsum("q,c,cq->", wq, vol, gsq)) < 1e-11
    # u_h = the linear part: the error is exactly the rest
    uh = (x[:, -1] + 0.5)[None]
    l2b, h10b, nb = E.cell_errors(topo, x, 1, cells, uh, cubic(pts)[..., None])
    rest = xq[..., 0] ** 3 - 2.0 * xq[..., 0] * xq[..., 1] ** 2
    assert np.abs(l2b - np.einsum("q,c,cq->c", wq, vol, rest ** 2)).max() < 1e-13
    assert abs(nb[2] - exact_l2) < 1e-12
    assert ref.ndofs == topo.nv


def test_subset_of_cells_and_vector_fields():
    x, cells = meshgen.create_box([0.0, 0.0], [1.0, 2.0], [3, 4])
    topo = Topology("triangle", cells, x.shape[0])
    lam = E.reference_nodes(2)
    sel = np.array([5, 0, 17])
    pts = np.einsum("jm,cmd->cjd", lam, x[cells[sel]])
    uref = np.stack([np.sin(pts[..., 0]), pts[..., 1] ** 2], axis=-1)
    uh = np.stack([np.sin(x[:, 0]), x[:, 1] ** 2])
    l2, h10, norms = E.cell_errors(topo, x, 1, cells, uh, uref, cells=sel)
    assert l2.shape == (3,) and np.all(l2 > 0) and np.all(h10 > 0)
    a, _, _ = E.cell_errors(topo, x, 1, cells, uh[:1], uref[..., :1], cells=sel)
    b, _, _ = E.cell_errors(topo, x, 1, cells, uh[1:], uref[..., 1:], cells=sel)
    assert np.allclose(l2, a + b, rtol=1e-13)
