"""the finite set of plausible FFCx evaluation orders for x_q and phi(x_q), tried on the
floating-point-degenerate golden cases (development record; CPU only).  For every variant the cell-tag histogram of
the NON-discretised leg (UFL expression of SpatialCoordinate) is compared with the golden histogram.
usage: python tools/r04/oracle_orders.py"""
import itertools
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from datasets import MESHTAG_DATA, load_mesh  # noqa: E402
from oracle import points as P  # noqa: E402
from oracle.topology import Topology  # noqa: E402

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "tags_golden.npz"))


def hist(v, hi):
    return np.bincount(np.asarray(v, dtype=np.int64), minlength=hi + 1)[1:hi + 1]


def xq_variants(ctype, pts, xc):
    """physical points of the detection points: dict name -> (gdim, nc, npts).  xc: (nc, nvpc, gdim)."""
    out = {}
    N = P.shape_functions(ctype, pts)                      # closed form, left to right
    nv = N.shape[1]
    for perm in ([tuple(range(nv))] + ([tuple(reversed(range(nv)))] if nv <= 4 else [])):
        acc = None
        for i in perm:
            term = N[:, i][None, :, None] * xc[:, i, None, :]
            acc = term if acc is None else acc + term
        out["sumN" + "".join(map(str, perm))] = np.moveaxis(acc, 2, 0)
    # affine map x0 + J X (columns of J = edge vectors), X components in order
    if ctype == "triangle":
        J0, J1 = xc[:, 1] - xc[:, 0], xc[:, 2] - xc[:, 0]
    else:
        J0, J1 = xc[:, 1] - xc[:, 0], xc[:, 2] - xc[:, 0]
    X, Y = pts[:, 0], pts[:, 1]
    a = xc[:, 0, None, :] + J0[:, None, :] * X[None, :, None] + J1[:, None, :] * Y[None, :, None]
    out["x0+JX"] = np.moveaxis(a, 2, 0)
    b = (J0[:, None, :] * X[None, :, None] + J1[:, None, :] * Y[None, :, None]) + xc[:, 0, None, :]
    out["JX+x0"] = np.moveaxis(b, 2, 0)
    if ctype == "quadrilateral":
        # Q1 tabulated as products of 1-D tables with the table entries of exact nodes snapped (FFCx clamps table values
        # within 1e-9 of -1, 0, 1) -- identical to the closed form at these lattice points; and the bilinear map written
        # as nested 1-D interpolation (x along the bottom and the top edge, then between them)
        bot = xc[:, 0, None, :] * (1.0 - X)[None, :, None] + xc[:, 1, None, :] * X[None, :, None]
        top = xc[:, 2, None, :] * (1.0 - X)[None, :, None] + xc[:, 3, None, :] * X[None, :, None]
        c = bot * (1.0 - Y)[None, :, None] + top * Y[None, :, None]
        out["nested1d"] = np.moveaxis(c, 2, 0)
    return out


def phi_variants(name):
    mesh, f = MESHTAG_DATA[name]
    out = {}
    if hasattr(f, "quadric"):
        x0, a, x1, b, c = f.quadric
        out["(ax-x0)^2+(by-x1)^2+c"] = lambda x: (a * x[0] - x0) ** 2 + (b * x[1] - x1) ** 2 + c
        out["c+((..)^2+(..)^2)"] = lambda x: c + ((a * x[0] - x0) ** 2 + (b * x[1] - x1) ** 2)
        out["(..)^2+((..)^2+c)"] = lambda x: (a * x[0] - x0) ** 2 + ((b * x[1] - x1) ** 2 + c)
        out["pow()"] = lambda x: np.power(a * x[0] - x0, 2.0) + np.power(b * x[1] - x1, 2.0) + c
        out["expanded"] = lambda x: (a * a) * x[0] * x[0] - 2 * a * x0 * x[0] + x0 * x0 + (b * b) * x[1] * x[1] - 2 * b * x1 * x[1] + x1 * x1 + c
    else:
        out["as written"] = f
    return out


def ratio_variants(phi_q, scale):
    """d = sum / sum|.| with the per-term scaling FFCx applies (|detJ|, weight 1.0), left to right."""
    out = {}
    for nm, s in (("plain", None), ("phi*detJ", scale)):
        num = np.zeros(phi_q.shape[0])
        den = np.zeros(phi_q.shape[0])
        for q in range(phi_q.shape[1]):
            t = phi_q[:, q] if s is None else phi_q[:, q] * s
            num = num + t
            den = den + np.abs(t)
        d = np.full_like(num, 0.5)
        ok = den > 0
        d[ok] = num[ok] / den[ok]
        out[nm] = d
    return out


def main():
    warnings.simplefilter("ignore")
    rows = []
    for name, degs in (("ellipse_in_square", (3,)), ("square_in_square", (1, 2, 3)), ("circle_in_square", (1, 2, 3)),
                       ("circle_near_boundary", (1, 2, 3)), ("circle_in_circle", (1, 2, 3)), ("boundary_crossing_circle", (1, 2, 3))):
        mesh, f = MESHTAG_DATA[name]
        ctype, x, cells = load_mesh(mesh)
        topo = Topology(ctype, cells, x.shape[0])
        xc = x[topo.cells]                                  # (nc, nvpc, 2)
        if ctype == "triangle":
            detj = np.abs((xc[:, 1, 0] - xc[:, 0, 0]) * (xc[:, 2, 1] - xc[:, 0, 1]) - (xc[:, 1, 1] - xc[:, 0, 1]) * (xc[:, 2, 0] - xc[:, 0, 0]))
        else:
            detj = np.abs((xc[:, 1, 0] - xc[:, 0, 0]) * (xc[:, 2, 1] - xc[:, 0, 1]) - (xc[:, 1, 1] - xc[:, 0, 1]) * (xc[:, 2, 0] - xc[:, 0, 0]))
        for deg in degs:
            pts = P.cell_detection_points(ctype, deg)
            key = f"{name}_{deg}_cells_tags:v"
            gh = hist(GOLD[key], 3)
            for (xn, xq), (pn, pf) in itertools.product(xq_variants(ctype, pts, xc).items(), phi_variants(name).items()):
                with np.errstate(all="ignore"):
                    ph = np.asarray(pf(xq.reshape(2, -1)), dtype=np.float64).reshape(xq.shape[1], xq.shape[2])
                for rn, d in ratio_variants(ph, detj).items():
                    tags = np.zeros(topo.nc, dtype=np.int8)
                    tags[(d > -1.0) & (d < 1.0)] = 2
                    tags[d == 1.0] = 3
                    tags[d == -1.0] = 1
                    h = hist(tags, 3)
                    rows.append((name, deg, xn, pn, rn, tuple(h), tuple(gh), bool(np.array_equal(h, gh))))
    cur = None
    for r in rows:
        if (r[0], r[1]) != cur:
            cur = (r[0], r[1])
            print(f"== {r[0]} degree {r[1]}: golden (inside, cut, outside) = {r[6]}")
        print(f"   x_q {r[2]:10s} phi {r[3]:24s} sum {r[4]:9s} -> {r[5]} {'OK' if r[7] else 'differs'}")
    # summary: variants that reproduce EVERY case listed
    from collections import defaultdict
    score = defaultdict(list)
    for r in rows:
        score[(r[2], r[3] if r[3] != "as written" else "*", r[4])].append((r[0], r[1], r[7]))
    print("\nvariants (x_q, phi, sum) and the cases they reproduce:")
    for k, v in score.items():
        bad = [f"{n}_{d}" for n, d, ok in v if not ok]
        print(f"   {k}: {sum(ok for _, _, ok in v)} / {len(v)}  fails: {bad}")


if __name__ == "__main__":
    main()
