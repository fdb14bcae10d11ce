"""Host-side cost of the distributed driver around the native RCCL loop, measured with a one-rank
communicator on one GPU (development aid).  usage: dist_overhead.py [cubes]"""
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import phifem_amd  # noqa: E402,F401
from phifem_amd.dist_solver import DistributedKrylov  # noqa: E402
from phifem_amd.distributed import SlabProblem  # noqa: E402
from phifem_amd.mesh_scripts import NodalFunction, _tag_cells, _tag_facets  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
warnings.simplefilter("ignore")
p = SlabProblem(n)
p.setup()
dk = DistributedKrylov(p)
for it in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    staged = _tag_cells(p.mesh, NodalFunction(p.phi), 1, single_layer_cut=True)
    dk.agree_on_exterior()
    _tag_facets(p.mesh, staged, 1)
    info = p.solver.assemble(p.phi, p.f, p.u_ex)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    st = dk.solve(p.out)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    p.solver.solve(rtol=p.rtol, max_iter=p.max_iter, out=p.out)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print(f"pass {it}: tag+assemble {1e3 * (t1 - t0):.1f} ms | distributed solve call {1e3 * (t2 - t1):.1f} ms "
          f"(loop {1e3 * st['seconds']:.1f} ms, {st['iterations']} it, path {dk.path}) | "
          f"plain solve {1e3 * (t3 - t2):.1f} ms ({p.solver.stats['iterations']} it)", flush=True)
dist.destroy_process_group()
