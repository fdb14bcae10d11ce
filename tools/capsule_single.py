"""The weak-scaling capsule of `bench.py --gpus N` (N slabs of n^3) on ONE mesh: the reference iteration count the
slab-partitioned runs are compared with.  usage: capsule_single.py n N"""
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import phifem_amd as P  # noqa: E402
from phifem_amd.mesh_scripts import NodalFunction  # noqa: E402

n, w = int(sys.argv[1]), int(sys.argv[2])
mesh = P.create_box([-1.5, -1.5, -1.5 * w], [1.5, 1.5, 1.5 * w], [n, n, n * w])
x = torch.from_numpy(mesh.x).cuda()
dz = torch.clamp(torch.abs(x[:, 2]) - 1.5 * (w - 1), min=0.0)
phi = (x[:, 0] ** 2 + x[:, 1] ** 2 + dz ** 2 - 1.0).contiguous()
uex = (torch.sin(x[:, 0]) * torch.sin(x[:, 1]) * torch.sin(x[:, 2])).contiguous()
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    P.compute_tags_measures(mesh, NodalFunction(phi), 1, box_mode=True, single_layer_cut=True)
s = P.PhiFEMSolver(mesh)
info = s.assemble(phi, (3.0 * uex).contiguous(), uex)
out = torch.empty(info["n_full"], dtype=torch.float64, device="cuda")
s.solve(rtol=1e-8, out=out)
print(f"single mesh {n}x{n}x{n * w}: dofs {info['n_active']} iterations {s.stats['iterations']} relres {s.stats['relres']:.2e} "
      f"precond {s.stats['precond']} L {s.stats['precond_L']}", flush=True)
