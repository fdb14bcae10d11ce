#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
for pr in 4 8 16; do
  PHX_DST_PAIRS=$pr PHX_DST_LDS_KB=150 timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/r2_b31.json 2> $O/r2_b31.err; echo "bench rc=$? pairs=$pr"
  python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2_b31.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['config']['iterations'], d['config']['stage_ms']['solve'], d['roofline_other']['avg_launch_us'])
PY
done
