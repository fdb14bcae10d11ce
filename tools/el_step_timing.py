"""Where an elasticity step spends its host time (development aid): el_step_timing.py n"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phifem_amd.distributed import ElasticitySlabProblem
from phifem_amd.mesh_scripts import NodalFunction, _tag_cells, _tag_facets
warnings.simplefilter("ignore")
n = int(sys.argv[1])
p = ElasticitySlabProblem(n, n, rtol=1e-8)
p.setup()
def tick(label, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    print(f"  {label:20s} {1e3 * (time.perf_counter() - t0):9.1f} ms", flush=True); return r
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
    print("step", rep, flush=True)
    st = tick("tag_cells", lambda: _tag_cells(p.mesh, NodalFunction(p._tag_levelset()), 1, single_layer_cut=p.single_layer_cut))
    tick("tag_facets", lambda: _tag_facets(p.mesh, st, 1))
    tick("solver._free", lambda: p.solver._free())
    tick("assemble", lambda: p._assemble())
    tick("solve", lambda: p.solver.solve(rtol=1e-8, max_iter=200000, out=p.out))
    free, tot = torch.cuda.mem_get_info()
    print(f"  in use {(tot - free) / 2**30:.1f} GB, iterations {p.solver.stats['iterations']}", flush=True)
