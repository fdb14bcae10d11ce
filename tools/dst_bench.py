"""Times the sine-transform passes of the preconditioner lattice (development aid).
usage: dst_bench.py L0 L1 L2 [f32=1] [reps=50]; tile sizes via PHX_DST_LDS_KB"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import phifem_amd  # noqa: E402,F401
from phifem_amd import _lib as L  # noqa: E402

Ls = [int(a) for a in sys.argv[1:4]]
f32 = int(sys.argv[4]) if len(sys.argv) > 4 else 1
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 50
out = (C.c_double * 3)()
L.check(L.lib.phx_box_dst_bench(0, (C.c_int * 3)(*Ls), f32, reps, out))
pts = (Ls[0] - 1) * (Ls[1] - 1) * (Ls[2] - 1)
b = 2 * (4 if f32 else 8) * pts
print(f"L={Ls} f32={f32} LDS_KB={os.environ.get('PHX_DST_LDS_KB', 'default')}: "
      f"x {out[0]:.1f} us ({b / out[0] / 1e6:.2f} TB/s)  y {out[1]:.1f} us ({b / out[1] / 1e6:.2f} TB/s)  "
      f"z-solve {out[2]:.1f} us ({b / out[2] / 1e6:.2f} TB/s)", flush=True)
