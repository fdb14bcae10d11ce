#!/usr/bin/env python3
"""Robin phi-FEM on a tilted square, on the MI355X path.

The counterpart of the reference's demo/robin/square/main.py: mixed (u, y, p) in P1 x P1^2 x DG0, P2
level-set, gamma = sigma = 1, 200 x 200 background squares on [-1,1]^2 split into triangles, detection
degree 1, and the same two modes --

    python main.py bg     solve on the background mesh (one-sided ds_bdy(100))
    python main.py sub    solve on the sub-mesh of the cells tagged 1/2

-- with phifem_amd in place of dolfinx / PETSc / MUMPS.  Prints the relative H1 error over the cells
tagged 1/2 (reference space of degree 3, main.py:229-262) and writes <mode>_output/solution.npz.
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
from phifem_amd.postprocess import cell_errors  # noqa: E402

from data import ROBIN_COEF, detection_levelset, exact_solution, levelset, robin_data, source_term  # noqa: E402


def main():
    ap = argparse.ArgumentParser(prog="main.py", description="Run Robin phiFEM demo.")
    ap.add_argument("mesh_type", choices=["bg", "sub"],
                    help="solve on the background mesh (bg) or on a submesh (sub)")
    ap.add_argument("--cells", type=int, default=200, help="background squares per direction")
    args = ap.parse_args()
    out_dir = os.path.join(HERE, args.mesh_type + "_output")
    os.makedirs(out_dir, exist_ok=True)

    bg_mesh = P.create_rectangle([[-1.0, -1.0], [1.0, 1.0]], [args.cells, args.cells])
    # the P2 interpolant of the detection level-set is sampled at the vertices only (detection degree 1)
    detection_h = NodalFunction(detection_levelset(bg_mesh.x.T))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        if args.mesh_type == "bg":
            cells_tags, facets_tags, _, ds_bdy, _ = P.compute_tags_measures(bg_mesh, detection_h, 1, box_mode=True)
            mesh = bg_mesh
        else:
            cells_tags, facets_tags, mesh, _, _ = P.compute_tags_measures(bg_mesh, detection_h, 1, box_mode=False)

    solver = P.NeumannRobinSolver(mesh, pen_coef=1.0, stab_coef=1.0, robin_coef=ROBIN_COEF, facet_tag=2)
    phi_h = levelset(mesh.p2_dof_points().T)                              # levelset_degree = 2
    info = solver.assemble(phi_h, source_term(mesh.x.T), robin_data(mesh.x.T))
    w = solver.solve(rtol=1e-10, max_iter=500000)
    u_h, y_h, p_h = solver.split(w)
    omega_h = np.flatnonzero(np.isin(mesh.cell_tag_values(), (1, 2)))
    e = cell_errors(mesh, u_h, exact_solution, degree=1, cells=omega_h)
    h1 = np.sqrt((e["h10_sum"] + e["l2_sum"]) / (e["h10_norm_exact"] + e["l2_norm_exact"]))
    print(f"{args.mesh_type}: {mesh.nc} cells, {info['n_active']} active DoFs, {info['nnz']} non-zeros, "
          f"{solver.stats['iterations']} BiCGStab iterations, residual {solver.stats['relres']:.1e}, "
          f"relative H1 error {h1:.3e}")
    np.savez(os.path.join(out_dir, "solution.npz"), x=mesh.x, cells=mesh.cells, u=u_h, y=y_h, p=p_h,
             cell_tags=mesh.cell_tag_values(), h1_local=e["h10_local"] + e["l2_local"], h1_cells=omega_h)
    # solution.xdmf: of.write_mesh(mesh); of.write_function(u) of the reference (heavy data as raw binary, no HDF5)
    P.io.write_solution(os.path.join(out_dir, "solution.xdmf"), mesh, u=u_h, y=y_h, p=p_h, cell_tags=mesh.cell_tag_values())


if __name__ == "__main__":
    main()
