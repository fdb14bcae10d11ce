#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_hip_assembly.py -x -q > $O/r2_t5.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r2_t5.log
for cfg in "8 1" "8 0" "1 0"; do set -- $cfg; echo "chunks=$1 xcd=$2"; PHX_SELL_CHUNKS=$1 PHX_SELL_XCD=$2 timeout -k 10 300 python tools/spmv_only.py 256 50 2>&1 | grep xcd_group; done
for cfg in "8 1" "1 0"; do set -- $cfg; PHX_SELL_CHUNKS=$1 PHX_SELL_XCD=$2 timeout -k 10 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/r2_bench5.json 2> $O/r2_bench5.err; echo "bench rc=$? chunks=$1 xcd=$2"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2_bench5.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['config']['iterations'], d['config']['stage_ms'], d['roofline']['avg_launch_us'], d['roofline_other']['avg_launch_us'])
PY
done
