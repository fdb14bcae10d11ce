"""Set up the bench problem once and launch the plain SpMV kernel a few times (for rocprofv3
--pmc passes, which must not be combined with tracing of other domains)."""
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import phifem_amd  # noqa: E402,F401
from phifem_amd.distributed import SlabProblem  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
warnings.simplefilter("ignore")
p = SlabProblem(n)
p.setup()
res = p.step()
print(res)
print(p.solver.info())
print(p.solver.spmv_bench(reps))
