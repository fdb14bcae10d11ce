"""Strong-Dirichlet (u = phi w) Poisson on the unit sphere / disc: assembly time, Krylov iterations
with the lattice preconditioner and with Jacobi, nodal error.  usage: sd_probe.py d n [degree]"""
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import phifem_amd as P  # noqa: E402
from phifem_amd import _lib as L  # noqa: E402
from phifem_amd.mesh_scripts import NodalFunction  # noqa: E402

d = int(sys.argv[1])
n = int(sys.argv[2])
k = int(sys.argv[3]) if len(sys.argv) > 3 else 1
warnings.simplefilter("ignore")
mesh = P.create_box([-1.5] * d, [1.5] * d, [n] * d)
x = mesh.x
phi1 = (x ** 2).sum(axis=1) - 1.0
t0 = time.perf_counter()
P.compute_tags_measures(mesh, NodalFunction(phi1), 1, box_mode=True)
t_tag = time.perf_counter() - t0
pts = x if k == 1 else mesh.p2_dof_points()
phi = (pts ** 2).sum(axis=1) - 1.0
g = np.prod(np.cos(pts), axis=1)
uex = -phi * g
# -lap(phi' g), phi' = 1 - r^2:  2 d g + 4 x.grad g + d phi' g  (lap g = -d g)
xg = sum(-pts[:, i] * np.tan(pts[:, i]) for i in range(d)) * g
f = 2 * d * g + 4 * xg - d * phi * g
for pre in (1, 0):
    L.check(L.lib.phx_set_option(mesh._h, L.OPT_PRECOND, pre))
    s = P.StrongDirichletSolver(mesh, degree=k, levelset_degree=k)
    t0 = time.perf_counter()
    info = s.assemble(phi, f)
    t_asm = time.perf_counter() - t0
    t0 = time.perf_counter()
    try:
        w = s.solve(rtol=1e-8, max_iter=50000)
    except Exception as e:  # noqa: BLE001
        print("precond", pre, "failed:", e)
        continue
    t_sol = time.perf_counter() - t0
    u = s.solution(w)
    ins = np.unique(mesh.cells[mesh.cell_tag_values() == 1])
    err = np.abs(u[ins] - uex[ins]).max()
    print(f"precond={pre} n_active={info['n_active']} nnz={info['nnz']} tag={t_tag*1e3:.1f}ms asm={t_asm*1e3:.1f}ms "
          f"solve={t_sol*1e3:.1f}ms its={s.stats['iterations']} relres={s.stats['relres']:.2e} "
          f"{s.stats['precond']} err_inf={err:.3e}", flush=True)
