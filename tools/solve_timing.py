"""Wall-clock of repeated solves of the bench system with and without the box preconditioner
(development aid).  usage: solve_timing.py [cubes]"""
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import phifem_amd  # noqa: E402,F401
from phifem_amd import _lib as L  # noqa: E402
from phifem_amd.distributed import SlabProblem  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
warnings.simplefilter("ignore")
p = SlabProblem(n)
p.setup()
for pc in (1, 0, 1):
    L.check(L.lib.phx_set_option(p.mesh._h, L.OPT_PRECOND, pc))
    p.step()
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        p.solver.solve(rtol=p.rtol, max_iter=p.max_iter, out=p.out)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        print(f"precond={pc} solve wall {1e3 * (t1 - t0):.2f} ms, library timer {1e3 * p.solver.stats['seconds']:.2f} ms, "
              f"{p.solver.stats['iterations']} it, relres {p.solver.stats['relres']:.2e}", flush=True)
