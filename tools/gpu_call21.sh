#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
for sp in 1 0 1 0; do
  PHX_SYNC_SPIN=$sp timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/r2_b21.json 2> $O/r2_b21.err; echo "bench rc=$? spin=$sp"
  python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2_b21.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['config']['iterations'], d['config']['stage_ms'])
PY
done
