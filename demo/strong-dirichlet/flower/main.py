#!/usr/bin/env python3
"""Strong-Dirichlet ("direct") phi-FEM Poisson on the "flower" domain, on the MI355X path.

The counterpart of the reference's demo/strong-dirichlet/flower/main.py: u_h = phi_h w_h with one
scalar unknown, same level-sets and source, sigma = 1, 200 x 200 background squares on
[-4.5,4.5]^2, detection degree 1, and the same two modes --

    python main.py bg     solve on the background mesh (one-sided ds_bdy(100))
    python main.py sub    solve on the sub-mesh of the cells tagged 1/2 (ds = its whole boundary)

-- with phifem_amd in place of dolfinx / PETSc / MUMPS.  Writes <mode>_output/solution.npz
(vertex coordinates, cells, u_h, w_h, cell tags) instead of XDMF.
"""
import argparse
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..", "..")))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..", "weak-dirichlet", "flower")))  # same data

import phifem_amd as P  # noqa: E402
from phifem_amd.mesh_scripts import NodalFunction  # noqa: E402

from data import detection_levelset, levelset, source_term  # noqa: E402


def main():
    ap = argparse.ArgumentParser(prog="main.py", description="Run strong dirichlet phiFEM demo.")
    ap.add_argument("mesh_type", choices=["bg", "sub"],
                    help="solve on the background mesh (bg) or on a submesh (sub)")
    ap.add_argument("--cells", type=int, default=200, help="background squares per direction")
    args = ap.parse_args()
    out_dir = os.path.join(HERE, args.mesh_type + "_output")
    os.makedirs(out_dir, exist_ok=True)

    stab_coef = 1.0
    bg_mesh = P.create_rectangle([[-4.5, -4.5], [4.5, 4.5]], [args.cells, args.cells])
    detection_h = NodalFunction(detection_levelset(bg_mesh.x.T))          # P1 interpolant
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        if args.mesh_type == "bg":
            cells_tags, facets_tags, _, ds_bdy, _ = P.compute_tags_measures(bg_mesh, detection_h, 1, box_mode=True)
            mesh = bg_mesh
        else:
            cells_tags, facets_tags, mesh, _, _ = P.compute_tags_measures(bg_mesh, detection_h, 1, box_mode=False)

    solver = P.StrongDirichletSolver(mesh, stab_coef=stab_coef, degree=1, levelset_degree=1)
    phi_h = levelset(mesh.x.T)
    info = solver.assemble(phi_h, source_term(mesh.x.T))
    w_h = solver.solve(rtol=1e-10, max_iter=100000)
    u_h = solver.solution(w_h)                                            # solution_degree = 1
    print(f"{args.mesh_type}: {mesh.nc} cells, {info['n_active']} active DoFs, {info['nnz']} non-zeros, "
          f"{solver.stats['iterations']} BiCGStab iterations ({solver.stats['precond']}), "
          f"residual {solver.stats['relres']:.1e}, max u_h = {u_h.max():.6f}")
    np.savez(os.path.join(out_dir, "solution.npz"), x=mesh.x, cells=mesh.cells, u=u_h, w=w_h,
             cell_tags=mesh.cell_tag_values())
    # solution.xdmf: of.write_mesh(mesh); of.write_function(u) of the reference (heavy data as raw binary, no HDF5)
    P.io.write_solution(os.path.join(out_dir, "solution.xdmf"), mesh, u=u_h, cell_tags=mesh.cell_tag_values())


if __name__ == "__main__":
    main()
