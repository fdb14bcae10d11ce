"""Problem data of the Robin demo: a unit square tilted by pi/6, -lap u + u = f inside,
du/dn + u = g on its boundary, u = cos(2 pi X) cos(2 pi Y) in the square's own axes (X, Y).

Restates demo/robin/square/data.py of the reference (same functions of x with shape (gdim, npts))
so that the demo solves the same problem; written against the rotated coordinates directly.
"""
import numpy as np

TILT = np.pi / 6.0
ROBIN_COEF = 1.0


def _rot(angle, x):
    """Coordinates of x in axes turned by `angle` (2-D part; a third row passes through)."""
    c, s = np.cos(angle), np.sin(angle)
    out = np.array(x, dtype=np.float64, copy=True)
    out[0] = c * x[0] + s * x[1]
    out[1] = -s * x[0] + c * x[1]
    return out


def detection_levelset(x):
    """|.|_1 distance to the tilted square: negative inside (data.py:17-19)."""
    r = _rot(TILT - np.pi / 4.0, x)
    return np.abs(r[0]) + np.abs(r[1]) - np.sqrt(2.0) / 2.0


def levelset(x):
    """Smooth level-set vanishing on the square's sides (data.py:21-25)."""
    half = np.full_like(np.asarray(x, dtype=np.float64), 0.5)
    r = _rot(TILT, x - _rot(-TILT, half))
    return -np.sin(np.pi * r[0]) * np.sin(np.pi * r[1])


def exact_solution(x):
    r = _rot(TILT, x)
    return np.cos(2.0 * np.pi * r[0]) * np.cos(2.0 * np.pi * r[1])


def source_term(x):
    return (8.0 * np.pi ** 2 + 1.0) * exact_solution(x)


def robin_data(x):
    """du/dn + kappa u with the outward normal of the nearest side, extended to the plane by the
    sectors |Y| < X, |X| < Y, |Y| < -X and the rest (data.py:36-55)."""
    r = _rot(TILT, x)
    X, Y = r[0], r[1]
    dX = -2.0 * np.pi * np.sin(2.0 * np.pi * X) * np.cos(2.0 * np.pi * Y)
    dY = -2.0 * np.pi * np.cos(2.0 * np.pi * X) * np.sin(2.0 * np.pi * Y)
    flux = -dY                                   # bottom side: n = (0, -1)
    flux = np.where(np.abs(Y) < X, dX, flux)     # right
    flux = np.where(np.abs(X) < Y, dY, flux)     # top
    flux = np.where(np.abs(Y) < -X, -dX, flux)   # left
    return flux + ROBIN_COEF * exact_solution(x)
