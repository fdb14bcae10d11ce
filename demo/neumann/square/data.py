"""Problem data of the Neumann demo: a unit square tilted by pi/6, -lap u + u = f inside, du/dn = g on its boundary,
u = cos(2 pi X) cos(2 pi Y) in the square's own axes (X, Y).

Restates demo/neumann/square/data.py of the reference (same functions of x with shape (gdim, npts)) so that the demo
solves the same problem; written against the rotated coordinates directly.  Pinned to values generated from the
reference's module: tests/golden/neumann_data.npz (tests/golden/make_neumann_data.py).
"""
import numpy as np

TILT = np.pi / 6.0


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


def neumann_data(x):
    """The derivative of u along the axis of the nearest side, extended to the plane by the sectors |Y| < X,
    |X| < Y, |Y| < -X and the rest (data.py:36-55; the reference takes dY in the bottom and dX in the left sector
    without the sign of the outward normal -- both vanish on the square's sides, where the datum is used)."""
    r = _rot(TILT, x)
    X, Y = r[0], r[1]
    dX = -2.0 * np.pi * np.sin(2.0 * np.pi * X) * np.cos(2.0 * np.pi * Y)
    dY = -2.0 * np.pi * np.cos(2.0 * np.pi * X) * np.sin(2.0 * np.pi * Y)
    g = dY.copy()
    g = np.where(np.abs(Y) < X, dX, g)
    g = np.where(np.abs(X) < Y, dY, g)
    g = np.where(np.abs(Y) < -X, dX, g)
    return g
