"""GPU checks of the fictitious-domain (box sine-transform) preconditioner.

Floating point: the lattice Poisson solve is compared with scipy's DST-I solve of the same operator,
relative tolerance 1e-12 (f64 FFTs of length <= 2048)."""
import ctypes as C

import numpy as np
import pytest
import scipy.fft as sf

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    import phifem_amd
    assert phifem_amd._lib.device_count() > 0
    return phifem_amd


def scipy_box_solve(f, L, h):
    c = [h[1] * h[2] / h[0], h[0] * h[2] / h[1], h[0] * h[1] / h[2]]
    lam = [c[a] * (2.0 - 2.0 * np.cos(np.pi * np.arange(1, L[a]) / L[a])) for a in range(3)]
    lam3 = lam[2][:, None, None] + lam[1][None, :, None] + lam[0][None, None, :]
    F = sf.dstn(f, type=1)
    return sf.idstn(F / lam3, type=1)


@pytest.mark.parametrize("L", [(64, 64, 64), (96, 64, 128), (192, 128, 64), (256, 96, 64), (64, 384, 96),
                               (512, 64, 64), (64, 64, 768), (1024, 64, 64)])
@pytest.mark.parametrize("f32", [0, 1])
def test_box_poisson_solve_matches_scipy(P, L, f32):
    from phifem_amd import _lib as L_
    rng = np.random.default_rng(11)
    h = (0.011, 0.017, 0.013)
    f = rng.standard_normal((L[2] - 1, L[1] - 1, L[0] - 1))
    ref = scipy_box_solve(f, L, h)
    u = np.ascontiguousarray(f.copy())
    Lc = (C.c_int * 3)(*L)
    hc = (C.c_double * 3)(*h)
    L_.check(L_.lib.phx_box_poisson_solve(0, Lc, hc, f32, u.ctypes.data_as(C.c_void_p)))
    tol = 2e-5 if f32 else 1e-12   # f32 transforms of length <= 1024 in three axes
    assert np.abs(u - ref).max() <= tol * np.abs(ref).max()
    if f32:
        return
    # and it really inverts the 7-point operator
    c = [h[1] * h[2] / h[0], h[0] * h[2] / h[1], h[0] * h[1] / h[2]]
    up = np.pad(u, 1)
    Ku = (c[0] * (2 * up[1:-1, 1:-1, 1:-1] - up[1:-1, 1:-1, :-2] - up[1:-1, 1:-1, 2:])
          + c[1] * (2 * up[1:-1, 1:-1, 1:-1] - up[1:-1, :-2, 1:-1] - up[1:-1, 2:, 1:-1])
          + c[2] * (2 * up[1:-1, 1:-1, 1:-1] - up[:-2, 1:-1, 1:-1] - up[2:, 1:-1, 1:-1]))
    assert np.abs(Ku - f).max() <= 1e-10 * np.abs(f).max()
