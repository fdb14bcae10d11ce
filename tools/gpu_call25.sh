#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
LIB=$(ls $R/tests/fake_rccl/*.so | head -1)
for n in 5; do
  PHIFEM_DIST_BACKEND=gloo PHIFEM_NATIVE_LOOP=1 PHX_RCCL_LIB=$LIB timeout -k 10 500 python bench.py --gpus $n --cubes 48 --steps 2 --warmup 1 --no-cpu-baseline > $O/r2_ranks$n.json 2> $O/r2_ranks$n.err; echo "ranks $n rc=$?"
  python - $n <<'PY'
import json,sys
d=json.loads([l for l in open('gpurun_out/r2_ranks%s.json'%sys.argv[1]) if l.startswith('{')][-1])
c=d['config']
print(d['n_gpus'], c.get('ranks_seen'), c.get('dist_loop'), c.get('dist_backend'), c.get('iterations'), c.get('converged'), c.get('precond_exact', None), round(d['ms_per_step'],1), d.get('valid', True))
PY
done
