"""Level-set inputs of the reference's tests, restated as data (test vectors).

tests/test_compute_meshtags.py:18-104 and tests/test_one_sided_integral.py:15-96 define the
level-sets by a handful of numbers; those numbers are the test vectors.
Each entry: name -> (mesh name, callable in the reference's numpy convention x[0], x[1]).
"""
import numpy as np


def quadric(x0, a, x1, b, c):
    def f(x):
        return (a * x[0] - x0) ** 2 + (b * x[1] - x1) ** 2 + c
    f.quadric = (x0, a, x1, b, c)
    return f


def square_ls(r):
    def f(x):
        return np.maximum(np.abs(x[0]), np.abs(x[1])) - r
    return f


def nasty(x):
    th = np.abs(np.arctan2(x[1], x[0]))
    return np.sqrt(x[0] ** 2 + x[1] ** 2) * (th * np.sin(1.0 / th)) - 0.25


def nasty_interpolated(x):
    """`nasty` as the reference's discretize=True leg sees it (tests/test_compute_meshtags.py:153-158).
    On the half line y = 0, x > 0 the formula is NaN (sin(1/0)); dolfinx interpolates at pushed-forward
    reference nodes whose y is a round-off away from 0 [3P basix], where r*th*sin(1/th) - 0.25 evaluates
    to its limit -0.25.  OBSERVED: with that value the 8 goldens of degrees 1 and 3 are reproduced
    exactly; degree 2 keeps NaN on some edge midpoints and stays an expected failure."""
    with np.errstate(all="ignore"):
        v = np.asarray(nasty(x), dtype=np.float64)
    return np.where(np.isnan(v), -0.25, v)


def line(x):
    return x[0] + 0.35


MESHTAG_DATA = {
    "circle_in_circle": ("disk", quadric(0.0, 1.0, 0.0, 1.0, -0.125)),
    "boundary_crossing_circle": ("disk", quadric(0.0, 1.0, -0.5, 1.0, -0.125)),
    "circle_in_square": ("square_quad", quadric(0.0, 1.0, 0.0, 1.0, -0.125)),
    "square_in_square": ("square_tri", square_ls(1.0)),
    "ellipse_in_square": ("square_quad", quadric(0.0, 1.0, 0.1, 0.3, -0.65)),
    "circle_near_boundary": ("coarse_square", quadric(0.5, 1.0, 0.5, 1.0, -0.2)),
    "nasty_levelset": ("square_tri", nasty),
}
# exact float compares on level-sets that vanish exactly on mesh nodes / are NaN
# depend on FFCx/basix round-off [3P]; reported, never gated.
FP_FRAGILE = {"square_in_square"}
FP_FRAGILE_DISCRETIZED = {"square_in_square", "nasty_levelset"}
# Round 3: tests/golden/meshes.npz holds the EXACT coordinates of the reference's mesh files (rounds 1-2: h5dump's
# 6-digit text).  With them the ellipse x^2 + (0.3 y - 0.1)^2 = 0.65 passes EXACTLY (in exact arithmetic) through the
# mesh vertices (+-0.8, 0) and (+-0.8, 2/3) of the 30 x 30 square: the sign of phi there is round-off, and at detection
# degree 3 one cell hung on whether a 1e-16 sample is absorbed by the running sum (67 / 68 cut cells).  Round 4: the
# detection sums scale their terms by |det J| as FFCx does (oracle/tagging.py:_ratio) and all 8 cases are reproduced;
# tools/r04/oracle_orders.py lists the evaluation orders tried.
FP_FRAGILE_DEGREE = set()


def is_fragile(name, deg, disc):
    """Golden case decided by floating-point round-off of FFCx / basix [3P]: reported (xfail), never gated."""
    return name in (FP_FRAGILE_DISCRETIZED if disc else FP_FRAGILE) or (name, deg) in FP_FRAGILE_DEGREE

ONE_SIDED_DATA = {
    "line_in_square_quad": ("square_quad", line, lambda n: n[:, 0] + n[:, 1]),
    "square_in_square_quad": ("square_quad", square_ls(0.35),
                              lambda n: np.abs(n[:, 0]) + np.abs(n[:, 1])),
    "square_in_square_tri": ("square_tri", square_ls(0.325),
                             lambda n: np.abs(n[:, 0]) + np.abs(n[:, 1])),
}


def load_mesh(name):
    import os
    m = np.load(os.path.join(os.path.dirname(__file__), "golden", "meshes.npz"))
    x = m[name + "_x"]
    cells = m[name + "_cells"]
    ctype = str(m[name + "_type"])
    if ctype == "quadrilateral":
        cells = cells[:, [0, 1, 3, 2]]  # XDMF cyclic -> tensor-product order
    return ctype, x, cells
