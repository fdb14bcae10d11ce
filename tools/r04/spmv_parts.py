"""SpMV of the bench problem alone (development aid; rocprofv3 --pmc passes wrap this).  PHX_SPMV_PART=1 / 2 in the
environment restricts the launch to the SELL-16 blocks / the stencil blocks.
usage: spmv_parts.py [256 | config5 | p2:256] [reps]   (PHX_SELL_EXP=<variant>: the stored rows alone, phx_spmv_exp.inc.hip)"""
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import phifem_amd  # noqa: E402,F401
from phifem_amd.distributed import SlabProblem, P2Problem  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "256"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
warnings.simplefilter("ignore")
p = SlabProblem(128, nxy=1024) if what == "config5" else P2Problem(int(what[3:]), rtol=1e-2) if what.startswith("p2:") else SlabProblem(int(what))
p.setup()
if os.environ.get("PLANE_ROWS"):
    from phifem_amd import _lib as L
    L.check(L.lib.phx_set_option(p.mesh._h, L.OPT_STENCIL_PLANE_ROWS, int(os.environ["PLANE_ROWS"])))
res = p.step()
info = p.solver.info()
sb = p.solver.spmv_bench(reps)
sec = 1e-3 * sb["ms"]
if os.environ.get("PHX_SELL_EXP"):
    print(f"{what} PHX_SELL_EXP={os.environ['PHX_SELL_EXP']}: stored rows alone {1e6 * sec:.1f} us, padded entries {info['sell_padded_nnz']}, "
          f"{12.0 * info['sell_padded_nnz'] / sec / 1e12:.2f} TB/s of the stored stream; raw {sb}", flush=True)
    sys.exit(0)
req = info["spmv_matrix_bytes"] + 16.0 * info["n_active"]
print(f"{what} part={os.environ.get('PHX_SPMV_PART', '0')}: n={info['n_active']} stencil_rows={info['stencil_rows']} "
      f"sell_nnz={info['sell_nnz']} padded={info['sell_padded_nnz']} stream={info['spmv_matrix_bytes'] / 1e6:.1f} MB "
      f"required={req / 1e6:.1f} MB  spmv {1e6 * sec:.1f} us  ({req / sec / 1e12:.2f} TB/s of required bytes)", flush=True)
