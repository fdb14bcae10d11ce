"""Stage timings of the 3-D interface-elasticity path on growing boxes (development aid)."""
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import phifem_amd as P  # noqa: E402
from phifem_amd.mesh_scripts import NodalFunction  # noqa: E402

warnings.simplefilter("ignore")
for n in [int(a) for a in sys.argv[1:]]:
    mesh = P.create_box([-1.5] * 3, [1.5] * 3, [n] * 3)
    x = mesh.x
    phi = 1.0 - (x ** 2).sum(axis=1)
    P.compute_tags_measures(mesh, NodalFunction(phi), 1, box_mode=True)
    bf = mesh.boundary_facets
    fv = np.array([[1, 2, 3], [0, 2, 3], [0, 1, 3], [0, 1, 2]])
    bcv = np.unique(np.take_along_axis(mesh.cells[bf[:, 0]], fv[bf[:, 1]], axis=1))
    f = np.stack([np.sin(x[:, 0]) + 0.2, np.cos(x[:, 1]), 0.5 * x[:, 2]], axis=1)
    uD = 0.1 * np.stack([x[:, 0] * x[:, 1], np.sin(x[:, 2]), x[:, 0] - x[:, 1]], axis=1)
    s = P.InterfaceElasticitySolver(mesh)
    t0 = time.time()
    info = s.assemble(phi, f, uD, bcv)
    t1 = time.time()
    w = s.solve(rtol=1e-8, max_iter=100000)
    t2 = time.time()
    print(n, "assemble %.3f s solve %.3f s" % (t1 - t0, t2 - t1), info, s.stats, mesh.timings())
    print("   DoF/s", info["n_active"] / (t2 - t0), "spmv", s.spmv_bench(20))
