"""Set up the bench problem once and launch the plain SpMV kernel a few times (for rocprofv3
--pmc passes, which must not be combined with tracing of other domains).
usage: spmv_only.py [cubes] [reps] [xcd_group ...]  (each group size is timed in turn)"""
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import phifem_amd  # noqa: E402,F401
from phifem_amd import _lib as L  # noqa: E402
from phifem_amd.distributed import SlabProblem  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
groups = [int(a) for a in sys.argv[3:]] or [0]
warnings.simplefilter("ignore")
p = SlabProblem(n)
p.setup()
res = p.step()
print(res)
print(p.solver.info())
for g in groups:
    L.check(L.lib.phx_set_option(p.mesh._h, L.OPT_SPMV_XCD_GROUP, g))
    print("xcd_group", g, p.solver.spmv_bench(reps), flush=True)
if os.environ.get("PHX_COMPARE_RAW"):
    # the same system with raw (not value-indexed) slices
    L.check(L.lib.phx_set_option(p.mesh._h, L.OPT_SPMV_XCD_GROUP, 0))
    L.check(L.lib.phx_set_option(p.mesh._h, L.OPT_SPMV_VALUE_INDEX, 0))
    print(p.step())
    print(p.solver.info())
    print("raw slices", p.solver.spmv_bench(reps), flush=True)
