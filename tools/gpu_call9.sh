#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 300 python tools/experiments/f32_restart.py 1 2>&1 | grep "pc=" 
for d in 0 1e-3 1e-4 1e-5; do
  PHX_RESTART_DROP=$d timeout -k 10 300 python tools/experiments/f32_restart.py 2 2>&1 | grep "pc="
done
for cfg in "1 0" "2 0" "2 1e-3" "2 1e-4" "2 1e-5"; do set -- $cfg
  PHX_PRECOND=$1 PHX_RESTART_DROP=$2 timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/r2_b9.json 2> $O/r2_b9.err; echo "bench rc=$? precond=$1 drop=$2"
  python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2_b9.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['config']['iterations'], d['config']['relres'], d['config']['stage_ms'])
PY
done
