#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 600 python - <<'PY' > $O/r2_p2flaky.log 2>&1
import sys, numpy as np, warnings
sys.path.insert(0, 'tests')
import phifem_amd as P
import test_hip_p2 as T
from oracle import assembly as OA
warnings.simplefilter("ignore")
for rep in range(6):
    work, V, phi, f, uex, A, b, act = T.setup(P, 3, 6, 2)
    s = P.PhiFEMSolver(work, degree=2, levelset_degree=2)
    s.assemble(phi, f, uex)
    w = s.solve(rtol=1e-11, max_iter=100000)
    wo = OA.solve_direct(A, b, act)
    print(rep, s.stats["iterations"], s.stats["relres"], np.abs(w - wo).max() / np.abs(wo).max(), flush=True)
PY
echo "rc=$?"; tail -8 $O/r2_p2flaky.log
