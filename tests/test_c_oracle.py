"""The C/OpenMP restatement (oracle/phifem_oracle.c, the CPU baseline of bench.py) against the
numpy oracle: identical tags, same matrix size, same solution."""
import warnings

import numpy as np
import pytest

from oracle import assembly as OA
from oracle import c_oracle, meshgen
from oracle import tagging as T
from oracle.topology import Topology


@pytest.mark.parametrize("n", [10, 16])
def test_c_oracle_matches_numpy_oracle(n):
    r = c_oracle.poisson_sphere(n, threads=2, rtol=1e-11, want_fields=True)
    assert r["bad_facets"] == 0 and r["relres"] <= 1e-11
    x, cells = meshgen.create_box([-1.5] * 3, [1.5] * 3, [n] * 3)
    topo = Topology("tetrahedron", cells, x.shape[0])
    phi = (x ** 2).sum(axis=1) - 1.0
    uex = np.prod(np.sin(x), axis=1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ct, ft, _, meas, _, _ = T.compute_tags_measures("tetrahedron", x, topo, T.NodalP1(phi), 1,
                                                        box_mode=True, single_layer_cut=True)
    assert np.array_equal(r["cell_tags"], ct.values)
    # facet numbering of the C oracle is the closed-form one: compare through vertex tuples
    assert np.array_equal(np.bincount(r["facet_tags"], minlength=7), np.bincount(ft.values, minlength=7))
    cv = np.zeros(topo.nc, dtype=np.int64)
    cv[ct.indices] = ct.values
    A, b, act = OA.assemble_poisson_wd(topo, x, cv, ft.values, meas(100), phi, 3.0 * uex, uex)
    assert int(r["n_active"]) == int(act.sum())
    idx = np.flatnonzero(act)
    assert int(r["nnz"]) == A[idx][:, idx].nnz
    w = OA.solve_direct(A, b, act)
    assert np.abs(r["u_full"] - w).max() <= 1e-7 * np.abs(w).max()


def test_c_oracle_box_preconditioner_same_solution_fewer_iterations():
    """The sine-transform box preconditioner of the CPU baseline (the algorithm the GPU path runs):
    same solution as the direct solve, far fewer iterations than Jacobi."""
    n = 24
    r = c_oracle.poisson_sphere(n, threads=2, rtol=1e-11, want_fields=True, precond=1)
    assert r["pc_built"] == 1 and r["relres_pc"] <= 1e-11
    assert 0 < r["iterations_pc"] < 0.6 * r["iterations"]
    rj = c_oracle.poisson_sphere(n, threads=2, rtol=1e-11, want_fields=True, precond=0)
    scale = np.abs(rj["u_full"]).max()
    assert np.abs(r["u_full"] - rj["u_full"]).max() <= 1e-8 * scale
