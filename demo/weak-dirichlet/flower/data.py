"""Problem data of the weak-Dirichlet "flower" demo, for this demo (the same module backs the parity tests as tests/flower_data.py).

Follows demo/weak-dirichlet/flower/data.py: smooth flower level-set via a graded smooth-min
(:10-53), non-smooth detection level-set = min over the 9 circles (:57-82), piecewise-constant
source (:85-99), zero Dirichlet data (:103-104).  Written independently on (2, npts) arrays.
"""
import numpy as np

_R8 = np.cos(np.pi / 8.0) + np.sin(np.pi / 8.0)
_PETAL_CENTRE_RADIUS = 2.0 * _R8
_PETAL_RADIUS = np.sqrt(2.0) * 2.0 * _R8 * np.sin(np.pi / 8.0)


def _circles(x):
    """phi_0 (radius 2 around the origin) and the 8 petal circles, data.py:28-49 / :58-79."""
    out = [x[0] ** 2 + x[1] ** 2 - 2.0 ** 2]
    for i in range(1, 9):
        cx = _PETAL_CENTRE_RADIUS * np.cos(i * np.pi / 4.0)
        cy = _PETAL_CENTRE_RADIUS * np.sin(i * np.pi / 4.0)
        out.append((x[0] - cx) ** 2 + (x[1] - cy) ** 2 - _PETAL_RADIUS ** 2)
    return out


def detection_levelset(x):
    """data.py:57-82."""
    cs = _circles(x)
    val = cs[0]
    for c in cs[1:]:
        val = np.minimum(val, c)
    return val


def _graded_smin(x, a, b, kmin=0.0, kmax=1.0):
    """data.py:18-22: smooth-min whose blending width k decays away from r = 2 (atan ramp)."""
    r = np.sqrt(x[0] ** 2 + x[1] ** 2)
    k = kmax * ((np.pi / 2.0 - np.arctan(50.0 * (r - 2.0))) / np.pi / 2.0) + kmin
    pa = np.maximum(k - a, 0.0)
    pb = np.maximum(k - b, 0.0)
    return np.maximum(k, np.minimum(a, b)) - np.sqrt(pa ** 2 + pb ** 2)


def levelset(x):
    """data.py:27-53."""
    cs = _circles(x)
    val = cs[0]
    for c in cs[1:]:
        val = _graded_smin(x, val, c)
    return val


def source_term(x):
    """data.py:85-99: 10 inside the disc of radius r1/sqrt(2) around the first petal centre."""
    d2 = (x[0] - _PETAL_CENTRE_RADIUS) ** 2 + x[1] ** 2
    return np.where(d2 <= _PETAL_RADIUS ** 2 / 2.0, 10.0, 0.0)


def dirichlet_data(x):
    """data.py:103-104."""
    return np.zeros_like(x[0])
