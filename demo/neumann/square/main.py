#!/usr/bin/env python3
"""Neumann phi-FEM on a tilted square with QUADRILATERAL cells, on the MI355X path.

The counterpart of the reference's demo/neumann/square/main.py: mixed (u, y, p) in Q1 x Q1^2 x DG0, Q2 level-set,
gamma = sigma = 1, 200 x 200 background quadrilaterals on [-1,1]^2 (main.py:36-50), detection degree 1, the
gradient-jump term on dS(3), and the same two modes --

    python main.py bg     solve on the background mesh (one-sided ds_bdy(100))
    python main.py sub    solve on the sub-mesh of the cells tagged 1/2

-- with phifem_amd in place of dolfinx / PETSc / MUMPS.  Prints the relative l2 error of u_h at the vertices of the
inside cells and writes <mode>_output/solution.npz.
"""
import argparse
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..", "..")))

import phifem_amd as P  # noqa: E402
from phifem_amd.mesh_scripts import NodalFunction  # noqa: E402

from data import detection_levelset, exact_solution, levelset, neumann_data, source_term  # noqa: E402


def rectangle_of_quadrilaterals(bbox, n):
    """dolfinx.mesh.create_rectangle(..., CellType.quadrilateral) as arrays: tensor-product vertex order."""
    (x0, y0), (x1, y1) = bbox
    tx, ty = np.linspace(x0, x1, n[0] + 1), np.linspace(y0, y1, n[1] + 1)
    X, Y = np.meshgrid(tx, ty, indexing="xy")
    x = np.stack([X.reshape(-1), Y.reshape(-1)], axis=1)
    i, j = np.meshgrid(np.arange(n[0]), np.arange(n[1]), indexing="xy")
    v0 = (j * (n[0] + 1) + i).reshape(-1)
    cells = np.stack([v0, v0 + 1, v0 + n[0] + 1, v0 + n[0] + 2], axis=1).astype(np.int32)
    return x, cells


def main():
    ap = argparse.ArgumentParser(prog="main.py", description="Run neumann phiFEM demo.")
    ap.add_argument("mesh_type", choices=["bg", "sub"],
                    help="solve on the background mesh (bg) or on a submesh (sub)")
    ap.add_argument("--cells", type=int, default=200, help="background quadrilaterals per direction")
    args = ap.parse_args()
    out_dir = os.path.join(HERE, args.mesh_type + "_output")
    os.makedirs(out_dir, exist_ok=True)

    x, cells = rectangle_of_quadrilaterals([[-1.0, -1.0], [1.0, 1.0]], [args.cells, args.cells])
    bg_mesh = P.Mesh.from_arrays("quadrilateral", x, cells)
    # the Q2 interpolant of the detection level-set, sampled with detection degree 1 (main.py:55-63)
    detection_h = NodalFunction(detection_levelset(bg_mesh.q2_dof_points().T), degree=2)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        if args.mesh_type == "bg":
            cells_tags, facets_tags, _, ds_bdy, _ = P.compute_tags_measures(bg_mesh, detection_h, 1, box_mode=True)
            mesh = bg_mesh
        else:
            cells_tags, facets_tags, mesh, _, _ = P.compute_tags_measures(bg_mesh, detection_h, 1, box_mode=False)

    solver = P.NeumannRobinSolver(mesh, pen_coef=1.0, stab_coef=1.0, robin_coef=0.0, facet_tag=3)
    phi_h = levelset(mesh.q2_dof_points().T)                                # levelset_degree = 2
    info = solver.assemble(phi_h, source_term(mesh.x.T), neumann_data(mesh.x.T))
    w = solver.solve(rtol=1e-10, max_iter=500000)
    u_h, y_h, p_h = solver.split(w)
    inside = np.unique(mesh.cells[mesh.cell_tag_values() == 1])
    uex = exact_solution(mesh.x.T)
    err = np.sqrt(np.sum((u_h[inside] - uex[inside]) ** 2) / np.sum(uex[inside] ** 2))
    print(f"{args.mesh_type}: {mesh.nc} quadrilaterals, {info['n_active']} active DoFs, {info['nnz']} non-zeros, "
          f"{solver.stats['iterations']} BiCGStab iterations, residual {solver.stats['relres']:.1e}, "
          f"relative nodal l2 error {err:.3e}")
    np.savez(os.path.join(out_dir, "solution.npz"), x=mesh.x, cells=mesh.cells, u=u_h, y=y_h, p=p_h,
             cell_tags=mesh.cell_tag_values())
    # solution.xdmf: of.write_mesh(mesh); of.write_function(u) of the reference (heavy data as raw binary, no HDF5)
    P.io.write_solution(os.path.join(out_dir, "solution.xdmf"), mesh, u=u_h, y=y_h, p=p_h, cell_tags=mesh.cell_tag_values())


if __name__ == "__main__":
    main()
