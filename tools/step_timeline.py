"""Wall time of the host-level calls of one bench step (development aid): where the time between kernels goes."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import phifem_amd  # noqa
from phifem_amd.distributed import SlabProblem
from phifem_amd.mesh_scripts import NodalFunction, _tag_cells, _tag_facets
warnings.simplefilter("ignore")
p = SlabProblem(int(sys.argv[1]) if len(sys.argv) > 1 else 256)
p.setup()
for _ in range(2):
    p.step()
def tick(label, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    print(f"{label:28s} {1e3 * (time.perf_counter() - t0):8.3f} ms", flush=True); return r
for rep in range(2):
    print("--- step", rep)
    st = tick("_tag_cells", lambda: _tag_cells(p.mesh, NodalFunction(p.phi), 1, single_layer_cut=True))
    tick("_tag_facets", lambda: _tag_facets(p.mesh, st, 1))
    tick("solver._free", lambda: p.solver._free())
    tick("solver.assemble", lambda: p.solver.assemble(p.phi, p.f, p.u_ex))
    tick("solver.solve", lambda: p.solver.solve(rtol=1e-8, out=p.out))
    tick("mesh.timings+precond_info", lambda: (p.mesh.timings(), p.solver.precond_info()))
    print(p.mesh.timings())
