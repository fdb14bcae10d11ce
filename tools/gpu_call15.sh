#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_hip_assembly.py tests/test_hip_precond.py tests/test_hip_p2.py tests/test_hip_strong_dirichlet.py -x -q > $O/r2_t15.log 2>&1; echo "pytest rc=$?"; tail -4 $O/r2_t15.log
timeout -k 10 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/r2_bench15.json 2> $O/r2_bench15.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2_bench15.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['config']['iterations'], d['config']['stage_ms'], d['roofline']['avg_launch_us'], d['roofline_other']['avg_launch_us'])
PY
