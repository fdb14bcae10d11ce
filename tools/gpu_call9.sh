#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
export PHIFEM_DIST_BACKEND=gloo
timeout -k 10 300 python tools/capsule_single.py 64 5 2>&1 | grep "single mesh"
for ex in 1 0; do
  PHIFEM_PRECOND_EXACT=$ex timeout -k 10 500 python bench.py --gpus 5 --cubes 64 --steps 1 --warmup 1 --no-cpu-baseline > $O/r2_n5_$ex.json 2> $O/r2_n5_$ex.err; echo "rc=$?"
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r2_n5_$ex.json') if l.startswith('{')][-1])
c=d['config']; print('exact=$ex', 'N', d['n_gpus'], 'ranks_seen', c['ranks_seen'], 'dofs', c['active_dofs'], 'iterations', c['iterations'], 'relres', c['relres'], 'loop', c['dist_loop'], 'ms', round(d['ms_per_step'],1))
PY
done
