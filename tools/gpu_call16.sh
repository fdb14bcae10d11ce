#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/r2_b16.json 2> $O/r2_b16.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2_b16.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['config']['iterations'], d['config']['stage_ms'], d['roofline']['avg_launch_us'], d['roofline_other']['avg_launch_us'])
PY
