"""Tag + assemble of the structured P2 system alone, twice (development aid: wrap in rocprofv3 --kernel-trace --stats to see
where the assembly of BASELINE configs[2] spends its time).  usage: p2_assemble_only.py [cubes]"""
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from phifem_amd.distributed import P2Problem, _tag_cells, _tag_facets  # noqa: E402
from phifem_amd.mesh_scripts import NodalFunction  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
warnings.simplefilter("ignore")
p = P2Problem(n)
p.setup()
for it in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    staged = _tag_cells(p.mesh, NodalFunction(p._tag_levelset()), 1, single_layer_cut=p.single_layer_cut)
    _tag_facets(p.mesh, staged, 1)
    info = p._assemble()
    torch.cuda.synchronize()
    t = p.mesh.timings()
    print(f"pass {it}: wall {time.perf_counter() - t0:.3f} s, assemble {t['assemble']:.3f} s, n_active {info['n_active']}", flush=True)
