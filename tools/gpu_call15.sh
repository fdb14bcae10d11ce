#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_hip_tagging.py tests/test_hip_assembly.py tests/test_hip_elasticity.py tests/test_hip_fullsize.py -x -q -m gpu > $O/r2_t15.log 2>&1; echo "pytest rc=$?"; tail -5 $O/r2_t15.log
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/r2_b15.json 2> $O/r2_b15.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2_b15.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['config']['iterations'], d['config']['relres'], d['config']['stage_ms'])
PY
