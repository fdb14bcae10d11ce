#!/usr/bin/env python3
"""Two-material linear elasticity with a circular interface (5-field mixed phi-FEM) on the MI355X
path: the counterpart of the reference's demo/interface-elasticity/main.py with its param1.yaml
(E_in = 1, E_out = 1e-3, nu = 0.3, phi = 1 - r^2, all degrees 1, box [-1.5,1.5]^2, h-refinement
loop with relative error slopes).

    python main.py [--iterations 4] [--mesh-size 0.2]

The reference refines with dolfinx.mesh.refine; the structured background mesh is simply
regenerated with twice the cells.  Errors as in the reference: u_h and the exact solution in the
degree-3 Lagrange space, cell-wise H10 / L2 integrals (`phifem_amd.postprocess.cell_errors`).
"""
import argparse
import os
import sys
import warnings

import numpy as np
import sympy as sy

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))

import phifem_amd as P  # noqa: E402
from phifem_amd.mesh_scripts import NodalFunction  # noqa: E402
from phifem_amd.postprocess import cell_errors  # noqa: E402

E_in, nu_in, E_out, nu_out = 1.0, 0.3, 1.0e-3, 0.3


def lame(E, nu):
    return E * nu / (1.0 + nu) / (1.0 - 2.0 * nu), E / 2.0 / (1.0 + nu)


def exact_solution(x):
    r = np.sqrt(x[:, 0] ** 2 + x[:, 1] ** 2)
    val = np.cos(r) - np.cos(1.0) / E_in
    val = np.where(r < 1.0, val * (E_in / E_out), val)
    return np.stack([val, val], axis=1)


def source():
    """f = -div(sigma_in((cos r, cos r))) / E_in as a numpy function."""
    X, Y = sy.symbols("x y")
    r = sy.sqrt(X ** 2 + Y ** 2)
    u = sy.Matrix([sy.cos(r), sy.cos(r)])
    lam, mu = lame(E_in, nu_in)
    g = u.jacobian([X, Y])
    s = lam * (g[0, 0] + g[1, 1]) * sy.eye(2) + mu * (g + g.T)
    f = -sy.Matrix([sy.diff(s[0, 0], X) + sy.diff(s[0, 1], Y),
                    sy.diff(s[1, 0], X) + sy.diff(s[1, 1], Y)]) / E_in
    fn = sy.lambdify((X, Y), f, "numpy")

    def call(x):
        xs = np.where(np.abs(x) < 1e-12, 1e-9, x)
        return np.array(fn(xs[:, 0], xs[:, 1])).reshape(2, -1).T
    return call


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=4)
    ap.add_argument("--mesh-size", type=float, default=0.2)
    args = ap.parse_args()
    f = source()
    n = int(round(3.0 / args.mesh_size))
    dofs, h10s, l2s = [], [], []
    for it in range(args.iterations):
        mesh = P.create_rectangle([[-1.5, -1.5], [1.5, 1.5]], [n, n])
        x = mesh.x
        phi = 1.0 - (x ** 2).sum(axis=1)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            P.compute_tags_measures(mesh, NodalFunction(phi), 1, box_mode=True)
        i, j = np.arange(mesh.nv) % (n + 1), np.arange(mesh.nv) // (n + 1)
        bc = np.flatnonzero((i == 0) | (i == n) | (j == 0) | (j == n))
        ue = exact_solution(x)
        solver = P.InterfaceElasticitySolver(mesh, E_in, nu_in, E_out, nu_out, 1.0, 1.0)
        info = solver.assemble(phi, f(x), ue, bc)
        w = solver.blocks(solver.solve(rtol=1e-10, max_iter=500000))
        # main.py:296-325 of the reference: u_h = u_in inside, u_out outside, their mean on the cut cells
        tags = mesh.cell_tag_values()
        on = lambda t: np.isin(np.arange(mesh.nv), mesh.cells[tags == t])  # noqa: E731
        v_in, v_cut, v_out = on(1), on(2), on(3)
        u_in = np.where((v_in | v_cut)[:, None], w["u_in"], 0.0)
        u_out = np.where((v_out | v_cut)[:, None], w["u_out"], 0.0)
        u_in[v_cut] *= 0.5
        u_out[v_cut] *= 0.5
        u = u_in + u_out
        # main.py:327-383: cell-wise H10 / L2 errors in the degree-3 space, on the GPU
        e = cell_errors(mesh, u, lambda p: exact_solution(p.T).T, degree=1)
        dofs.append(2 * mesh.nv)
        h10s.append(e["h10_relative"])
        l2s.append(e["l2_relative"])
        print(f"n = {n:4d}: {info['n_active']:8d} active DoFs, {solver.stats['iterations']:6d} iterations, "
              f"H10 relative error {e['h10_relative']:.3e}, L2 relative error {e['l2_relative']:.3e}")
        n *= 2
    if len(dofs) > 1:
        print("H10 relative error slope:", np.polyfit(np.log(dofs), np.log(h10s), 1)[0])
        print("L2 relative error slope:", np.polyfit(np.log(dofs), np.log(l2s), 1)[0])


if __name__ == "__main__":
    main()
