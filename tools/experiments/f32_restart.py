"""2-D flower problem (BASELINE configs[0]) and 3-D level-sets: iterations per preconditioner precision (argv[1] =
PHX_OPT_PRECOND value) at several tolerances; the restart threshold comes from PHX_RESTART_DROP (development aid)."""
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import flower_data as F  # noqa: E402
import phifem_amd as P  # noqa: E402
from phifem_amd import _lib as L  # noqa: E402
from phifem_amd.mesh_scripts import NodalFunction  # noqa: E402

warnings.simplefilter("ignore")
pc = int(sys.argv[1])
tag = "pc=%d drop=%s" % (pc, os.environ.get("PHX_RESTART_DROP", "default"))
for n in (128, 400):
    mesh = P.create_rectangle([[-4.5, -4.5], [4.5, 4.5]], [n, n])
    x = mesh.x
    det, phi, f, ud = F.detection_levelset(x.T), F.levelset(x.T), F.source_term(x.T), F.dirichlet_data(x.T)
    P.compute_tags_measures(mesh, NodalFunction(det), 1, box_mode=True, single_layer_cut=True)
    L.check(L.lib.phx_set_option(mesh._h, L.OPT_PRECOND, pc))
    for rtol in (1e-8, 1e-11):
        its = []
        for rep in range(3):
            s = P.PhiFEMSolver(mesh)
            s.assemble(phi, f, ud)
            try:
                s.solve(rtol=rtol, max_iter=3000)
                its.append("%d(%d)" % (s.stats["iterations"], s.stats.get("restarts", -1)))
            except Exception as e:
                its.append("FAIL")
        print(tag, "flower n=%d rtol=%g:" % (n, rtol), " ".join(its), flush=True)
for name, fn in (("torus", lambda x: (np.sqrt(x[:, 0] ** 2 + x[:, 1] ** 2) - 0.8) ** 2 + x[:, 2] ** 2 - 0.16),
                 ("ellipsoid", lambda x: ((x[:, 0] - 0.13) / 1.2) ** 2 + ((x[:, 1] + 0.07) / 0.6) ** 2 + (x[:, 2] / 0.9) ** 2 - 1.0)):
    mesh = P.create_box([-1.5] * 3, [1.5] * 3, [96] * 3)
    L.check(L.lib.phx_set_option(mesh._h, L.OPT_PRECOND, pc))
    x = mesh.x
    phi = fn(x)
    P.compute_tags_measures(mesh, NodalFunction(phi), 1, box_mode=True, single_layer_cut=True)
    uex = np.prod(np.sin(x), axis=1)
    for rtol in (1e-8, 1e-11):
        s = P.PhiFEMSolver(mesh)
        s.assemble(phi, 3.0 * uex, uex)
        try:
            s.solve(rtol=rtol, max_iter=3000)
            print(tag, name, "96^3 rtol=%g: %d(%d)" % (rtol, s.stats["iterations"], s.stats.get("restarts", -1)), flush=True)
        except Exception as e:
            print(tag, name, "FAIL", str(e)[:80], flush=True)
