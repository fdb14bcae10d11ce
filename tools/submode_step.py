"""One step of the reference's DEFAULT flow (box_mode=False: tags on the sub-mesh of Omega_h, forms assembled and
solved there) on an n^3 box with the unit sphere -- timing aid."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import phifem_amd as P
from phifem_amd.mesh_scripts import Quadric
warnings.simplefilter("ignore")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
m = P.create_box([-1.5] * 3, [1.5] * 3, [n] * 3)
ls = Quadric([0.0, 0.0, 0.0], [1.0, 1.0, 1.0], -1.0)
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ct, ft, sub, meas, maps = P.compute_tags_measures(m, ls, 1, box_mode=False, single_layer_cut=True)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    xs = torch.from_numpy(sub.x).to("cuda:0")
    phi = (xs ** 2).sum(dim=1) - 1.0
    uex = torch.sin(xs[:, 0]) * torch.sin(xs[:, 1]) * torch.sin(xs[:, 2])
    s = P.PhiFEMSolver(sub)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    info = s.assemble(phi, 3.0 * uex, uex)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    out = torch.empty(2 * sub.nv, dtype=torch.float64, device="cuda:0")
    s.solve(rtol=1e-8, max_iter=5000, out=out)
    torch.cuda.synchronize(); t4 = time.perf_counter()
    print(f"n={n} rep {rep}: tag+submesh {1e3*(t1-t0):.1f} ms, assemble {1e3*(t3-t2):.1f} ms, solve {1e3*(t4-t3):.1f} ms "
          f"({s.stats['iterations']} it, {s.stats['precond']}), active {info['n_active']}", flush=True)
    del s, sub
