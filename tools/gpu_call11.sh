#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_hip_assembly.py tests/test_hip_precond.py -x -q > $O/r2_t11.log 2>&1; echo "pytest rc=$?"; tail -5 $O/r2_t11.log
for cs in 0 1; do
  PHX_FACET_SCATTER=$cs timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/r2_b11.json 2> $O/r2_b11.err; echo "bench rc=$? scatter=$cs"
  python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2_b11.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['config']['iterations'], d['config']['relres'], d['config']['stage_ms'])
PY
done
