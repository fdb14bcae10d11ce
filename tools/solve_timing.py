"""Wall-clock of repeated solves of the bench system per preconditioner (0 Jacobi, 1 box sine transforms
in f64, 2 in f32) -- development aid.  usage: solve_timing.py [cubes] [precond ...]"""
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
modes = [int(a) for a in sys.argv[2:]] or [1, 2, 0]
warnings.simplefilter("ignore")
p = SlabProblem(n)
p.setup()
for pc in modes:
    L.check(L.lib.phx_set_option(p.mesh._h, L.OPT_PRECOND, pc))
    p.step()
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        p.solver.solve(rtol=p.rtol, max_iter=p.max_iter, out=p.out)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        print(f"precond={pc} solve wall {1e3 * (t1 - t0):.2f} ms, library timer {1e3 * p.solver.stats['seconds']:.2f} ms, "
              f"{p.solver.stats['iterations']} it, relres {p.solver.stats['relres']:.2e} {p.solver.stats['precond_L']}", flush=True)
