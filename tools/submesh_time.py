"""Time of compute_tags_measures(box_mode=False) (tag, sub-mesh, tag transfer) on an n^3 box with the unit sphere."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import phifem_amd as P
from phifem_amd.mesh_scripts import Quadric
warnings.simplefilter("ignore")
for n in [int(a) for a in sys.argv[1:]] or [128]:
    m = P.create_box([-1.5] * 3, [1.5] * 3, [n] * 3)
    ls = Quadric([0.0, 0.0, 0.0], [1.0, 1.0, 1.0], -1.0)
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ct, ft, sub, meas, maps = P.compute_tags_measures(m, ls, 1, box_mode=False, single_layer_cut=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        print(f"n={n} rep {rep}: {t1 - t0:.3f} s  sub-mesh: {sub.nc} cells, {sub.nv} vertices, {sub.nf} facets", flush=True)
        del sub
