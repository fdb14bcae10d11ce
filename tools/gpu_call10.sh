#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_hip_precond.py -x -q > $O/r2_t10.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r2_t10.log
for pv in 1 0; do
  PHX_DST_PERSIST=$pv timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/r2_b10.json 2> $O/r2_b10.err; echo "bench rc=$? persist=$pv"
  python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2_b10.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['config']['iterations'], d['config']['stage_ms'], d['roofline']['avg_launch_us'], d['roofline_other']['avg_launch_us'])
PY
done
