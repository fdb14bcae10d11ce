"""Step time (tag + assemble + solve) of a Kuhn box that arrives as ARRAYS in shuffled vertex / cell / local order
(Mesh.from_arrays: served by the generated box behind it) against the generated box itself.  usage: [n]"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import phifem_amd as P
from phifem_amd.mesh_scripts import NodalFunction
warnings.simplefilter("ignore")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
box = P.create_box([-1.5] * 3, [1.5] * 3, [n] * 3)
x, cells = box.x, box.cells
rng = np.random.default_rng(3)
perm = rng.permutation(x.shape[0]); inv = np.empty_like(perm); inv[perm] = np.arange(perm.size)
xs = x[perm]
cs = inv[cells][rng.permutation(cells.shape[0])].astype(np.int32)
t0 = time.perf_counter(); arr = P.Mesh.from_arrays("tetrahedron", xs, cs); t_create = time.perf_counter() - t0
def step(m, xx, reps=3):
    dev = torch.device("cuda", 0)
    phi = torch.from_numpy((xx ** 2).sum(axis=1) - 1.0).to(dev); uex = torch.from_numpy(np.prod(np.sin(xx), axis=1)).to(dev)
    f = 3.0 * uex; out = torch.empty(2 * xx.shape[0], dtype=torch.float64, device=dev)
    s = P.PhiFEMSolver(m); best = 1e9
    for _ in range(reps + 1):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        P.compute_tags_measures(m, NodalFunction(phi), 1, box_mode=True, single_layer_cut=True)
        info = s.assemble(phi, f, uex); s.solve(rtol=1e-8, out=out)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best, info, s.stats
tb, ib, sb = step(box, x); ta, ia, sa = step(arr, xs)
print(f"n={n}: from_arrays create {t_create:.2f} s | step generated box {1e3*tb:.2f} ms ({sb['iterations']} it, stencil rows {ib['stencil_rows']}) | "
      f"caller-supplied arrays {1e3*ta:.2f} ms ({sa['iterations']} it, stencil rows {ia['stencil_rows']}) | ratio {ta/tb:.3f}")
