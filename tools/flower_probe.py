"""Repeated solves of the 2-D flower problem (BASELINE configs[0]) per preconditioner and tolerance
(development aid)."""
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np  # noqa: E402
import flower_data as F  # noqa: E402
import phifem_amd as P  # noqa: E402
from phifem_amd import _lib as L  # noqa: E402
from phifem_amd.mesh_scripts import NodalFunction  # noqa: E402

warnings.simplefilter("ignore")
n = 128
mesh = P.create_rectangle([[-4.5, -4.5], [4.5, 4.5]], [n, n])
x = mesh.x
det, phi, f, ud = F.detection_levelset(x.T), F.levelset(x.T), F.source_term(x.T), F.dirichlet_data(x.T)
P.compute_tags_measures(mesh, NodalFunction(det), 1, box_mode=True, single_layer_cut=True)
for pc in (1, 2, 0):
    L.check(L.lib.phx_set_option(mesh._h, L.OPT_PRECOND, pc))
    for rtol in (1e-8, 1e-10, 1e-11, 1e-12):
        for rep in range(3):
            s = P.PhiFEMSolver(mesh)
            s.assemble(phi, f, ud)
            try:
                s.solve(rtol=rtol, max_iter=20000)
                print(pc, rtol, s.stats["iterations"], "%.2e" % s.stats["relres"], flush=True)
            except Exception as e:
                print(pc, rtol, "FAILED", str(e)[:90], flush=True)
