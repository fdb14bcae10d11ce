#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
for sp in 1 64 4096; do
  PHX_FACET_SPREAD=$sp timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/r2_b19.json 2> $O/r2_b19.err; echo "bench rc=$? spread=$sp"
  python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2_b19.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['config']['iterations'], d['config']['relres'], d['config']['stage_ms'])
PY
done
