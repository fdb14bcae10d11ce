#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 600 python tools/submesh_time.py 128 256 2>&1 | grep "n="
PHX_TOPOLOGY_HOST=1 timeout -k 10 600 python - <<'PY' 2>&1 | grep "from_arrays"
import sys, time, warnings, numpy as np
sys.path.insert(0, '.')
import phifem_amd as P
from oracle import meshgen
warnings.simplefilter("ignore")
x, cells = meshgen.create_box([-1.5]*3, [1.5]*3, [64]*3)
t0 = time.perf_counter(); m = P.Mesh.from_arrays("tetrahedron", x, cells.astype(np.int32)); t1 = time.perf_counter()
print("from_arrays 64^3 host topology:", round(t1 - t0, 3), "s")
PY
timeout -k 10 600 python - <<'PY' 2>&1 | grep "from_arrays"
import sys, time, warnings, numpy as np
sys.path.insert(0, '.')
import phifem_amd as P
from oracle import meshgen
warnings.simplefilter("ignore")
x, cells = meshgen.create_box([-1.5]*3, [1.5]*3, [64]*3)
for r in range(2):
    t0 = time.perf_counter(); m = P.Mesh.from_arrays("tetrahedron", x, cells.astype(np.int32)); t1 = time.perf_counter()
    print("from_arrays 64^3 device topology:", round(t1 - t0, 3), "s")
PY
