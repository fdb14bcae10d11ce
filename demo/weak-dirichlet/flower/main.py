#!/usr/bin/env python3
"""Weak-Dirichlet phi-FEM Poisson on the "flower" domain, on the MI355X path.

The counterpart of the reference's demo/weak-dirichlet/flower/main.py: same problem (level-sets,
source, gamma = sigma = 1, 200 x 200 background squares on [-4.5,4.5]^2, detection degree 1,
single-layer cut), same two modes --

    python main.py bg     solve on the background mesh (box mode, one-sided ds(100))
    python main.py sub    solve on the sub-mesh of the cells tagged 1/2

-- with phifem_amd in place of dolfinx / PETSc / MUMPS.  Writes <mode>_output/solution.npz
(vertex coordinates, cells, u_h, p_h, cell tags) instead of XDMF.
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

from data import detection_levelset, dirichlet_data, levelset, source_term  # noqa: E402


def main():
    ap = argparse.ArgumentParser(prog="main.py", description="Run weak dirichlet phiFEM demo.")
    ap.add_argument("mesh_type", choices=["bg", "sub"],
                    help="solve on the background mesh (bg) or on a submesh (sub)")
    ap.add_argument("--cells", type=int, default=200, help="background squares per direction")
    ap.add_argument("--degree", type=int, default=1, choices=[1, 2], help="primal degree")
    args = ap.parse_args()
    out_dir = os.path.join(HERE, args.mesh_type + "_output")
    os.makedirs(out_dir, exist_ok=True)

    pen_coef, stab_coef = 1.0, 1.0
    bg_mesh = P.create_rectangle([[-4.5, -4.5], [4.5, 4.5]], [args.cells, args.cells])
    detection_h = NodalFunction(detection_levelset(bg_mesh.x.T))          # P1 interpolant

    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        if args.mesh_type == "bg":
            cells_tags, facets_tags, _, ds_bdy, _ = P.compute_tags_measures(
                bg_mesh, detection_h, 1, box_mode=True, single_layer_cut=True)
            mesh = bg_mesh
        else:
            cells_tags, facets_tags, mesh, _, _ = P.compute_tags_measures(
                bg_mesh, detection_h, 1, box_mode=False, single_layer_cut=True)

    solver = P.PhiFEMSolver(mesh, pen_coef=pen_coef, stab_coef=stab_coef, degree=args.degree)
    pts = mesh.x if args.degree == 1 else mesh.p2_dof_points()
    phi_h = levelset(mesh.x.T)                                            # levelset_degree = 1
    info = solver.assemble(phi_h, source_term(pts.T), dirichlet_data(pts.T))
    w = solver.solve(rtol=1e-10, max_iter=100000)
    u_h, p_h = solver.split(w)
    print(f"{args.mesh_type}: {mesh.nc} cells, {info['n_active']} active DoFs, {info['nnz']} non-zeros, "
          f"{solver.stats['iterations']} BiCGStab iterations, residual {solver.stats['relres']:.1e}, "
          f"max u_h = {u_h.max():.6f}")
    np.savez(os.path.join(out_dir, "solution.npz"), x=mesh.x, cells=mesh.cells, u=u_h, p=p_h,
             cell_tags=mesh.cell_tag_values())
    # solution.xdmf: of.write_mesh(mesh); of.write_function(u) of the reference (heavy data as raw binary, no HDF5)
    P.io.write_solution(os.path.join(out_dir, "solution.xdmf"), mesh, u=u_h, cell_tags=mesh.cell_tag_values())


if __name__ == "__main__":
    main()
