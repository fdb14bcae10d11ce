"""The library's own dense inverse (blocked Gauss-Jordan with partial pivoting, phifem_amd/csrc/phx_dense.inc.hip) that the
elasticity coarse correction applies to its Galerkin matrix (no vendor LAPACK on the product path).
Floating point: compared with numpy.linalg.inv, tolerance 1e-10 * cond-independent scale on well-conditioned random
matrices (||A^-1 A - I||_max)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def hip_inverse(A):
    import phifem_amd  # noqa: F401
    from phifem_amd import _lib as L
    a = np.ascontiguousarray(A, dtype=np.float64).copy()
    sing = C.c_int(0)
    L.check(L.lib.phx_dense_inverse(0, a.shape[0], a.ctypes.data_as(C.c_void_p), C.byref(sing)))
    return a, sing.value


@pytest.mark.parametrize("n", [1, 2, 5, 31, 32, 33, 64, 100, 257, 1000])
def test_dense_inverse_matches_numpy(n):
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n)) + 0.1 * np.eye(n)
    Ai, sing = hip_inverse(A)
    assert sing == 0
    assert np.abs(Ai @ A - np.eye(n)).max() <= 1e-9 * max(1.0, np.linalg.cond(A) / 1e3)
    assert np.abs(Ai - np.linalg.inv(A)).max() <= 1e-9 * np.abs(np.linalg.inv(A)).max() * max(1.0, np.linalg.cond(A) / 1e3)


def test_dense_inverse_needs_its_pivoting():
    """Zero diagonal, permutation-like structure, and rows whose pivots sit far below the diagonal block."""
    n = 96
    rng = np.random.default_rng(7)
    P = np.eye(n)[rng.permutation(n)]
    A = P + 1e-3 * rng.standard_normal((n, n))
    np.fill_diagonal(A, 0.0)
    Ai, sing = hip_inverse(A)
    assert sing == 0 and np.abs(Ai @ A - np.eye(n)).max() <= 1e-10
    # a non-symmetric matrix: the result is the inverse, not its transpose
    B = np.triu(rng.standard_normal((40, 40))) + 5.0 * np.eye(40)
    Bi, sing = hip_inverse(B)
    assert sing == 0 and np.abs(Bi @ B - np.eye(40)).max() <= 1e-12 and np.abs(np.tril(Bi, -1)).max() <= 1e-14


def test_dense_inverse_reports_a_singular_matrix():
    A = np.ones((50, 50))
    _, sing = hip_inverse(A)
    assert sing == 1
    Z = np.zeros((10, 10))
    _, sing = hip_inverse(Z)
    assert sing == 1


def test_dense_inverse_at_the_size_of_the_coarse_matrix():
    """n = 6000 (the Galerkin matrix of the 256^3 elasticity box has ~10^4 rows): residual and time."""
    import time
    n = 6000
    rng = np.random.default_rng(1)
    A = rng.standard_normal((n, n)) / np.sqrt(n) + np.eye(n)
    t0 = time.perf_counter()
    Ai, sing = hip_inverse(A)
    dt = time.perf_counter() - t0
    assert sing == 0
    x = rng.standard_normal(n)
    assert np.abs(Ai @ (A @ x) - x).max() <= 1e-10 * np.abs(x).max()
    print(f"dense inverse n = {n}: {dt:.2f} s including the two host copies of {8 * n * n / 1e6:.0f} MB")
