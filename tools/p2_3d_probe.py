"""3-D P2 solves with the refined-lattice box preconditioner vs Jacobi (development aid)."""
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import phifem_amd as P  # noqa: E402
from phifem_amd import _lib as L  # noqa: E402
from phifem_amd.mesh_scripts import NodalFunction  # noqa: E402

warnings.simplefilter("ignore")
for n in [int(a) for a in sys.argv[1:]]:
    for pc in ([int(os.environ.get("PC", "1"))] if n > 128 else [1, 0]):
        mesh = P.create_box([-1.5] * 3, [1.5] * 3, [n] * 3)
        L.check(L.lib.phx_set_option(mesh._h, L.OPT_PRECOND, pc))
        x = mesh.x
        cen = np.array([0.03, -0.02, 0.01])
        P.compute_tags_measures(mesh, NodalFunction(((x - cen) ** 2).sum(axis=1) - 1.0), 1, box_mode=True,
                                single_layer_cut=True)
        pts = mesh.p2_dof_points()
        phi = ((pts - cen) ** 2).sum(axis=1) - 1.0
        uex = np.prod(np.sin(pts), axis=1)
        s = P.PhiFEMSolver(mesh, degree=2, levelset_degree=2)
        info = s.assemble(phi, 3.0 * uex, uex)
        try:
            w = s.solve(rtol=1e-8, max_iter=30000)
            inside = phi < -0.2
            err = np.abs(w[:pts.shape[0]][inside] - uex[inside]).max()
            print(n, "precond" if pc else "jacobi ", info["n_active"], s.stats["iterations"], "%.1e" % s.stats["relres"],
                  "max err inside %.2e" % err, "solve %.3f s" % s.stats["seconds"],
                  "assemble %.3f s" % mesh.timings()["assemble"], "nnz %d" % info["nnz"], flush=True)
        except Exception as e:
            print(n, "precond" if pc else "jacobi ", info["n_active"], "FAILED", str(e)[:80], flush=True)
