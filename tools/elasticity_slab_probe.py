"""Tag + assemble one interior slab of BASELINE configs[3] (256^3 interface elasticity on 8 GPUs: rank 3 of 8,
256 x 256 x (32 + 8 ghost) cubes) on one GPU, without the solve (which needs the neighbours): memory and
stage times of the per-GPU share (development aid).  usage: elasticity_slab_probe.py [nxy] [nz_per_rank]"""
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import phifem_amd  # noqa: E402,F401
from phifem_amd.distributed import ElasticitySlabProblem  # noqa: E402
from phifem_amd.mesh_scripts import NodalFunction, _tag_cells, _tag_facets  # noqa: E402

nxy = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nz = int(sys.argv[2]) if len(sys.argv) > 2 else 32
warnings.simplefilter("ignore")
p = ElasticitySlabProblem(nxy, nz, rank=3, world=8, device=0)
p.setup()
torch.cuda.synchronize()
for rep in range(5):
    t0 = time.perf_counter()
    staged = _tag_cells(p.mesh, NodalFunction(p.phi), 1, single_layer_cut=False)
    _tag_facets(p.mesh, staged, 1)
    info = p.solver.assemble(p.phi, p.f, p.u_D, p.bc_vertices)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    free, total = torch.cuda.mem_get_info()
    print(f"pass {rep}: tag+assemble {1e3 * (t1 - t0):.0f} ms; {info}; device memory in use {(total - free) / 2**30:.1f} GiB",
          flush=True)
print(p.mesh.timings())
print("spmv", p.solver.spmv_bench(5))
