#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s2
cd $GRAFT_REPO_ROOT
S=gpurun_out/s2/summary3.txt
timeout -k 10 900 python -m pytest tests/test_hip_tagging.py tests/test_hip_assembly.py tests/test_hip_precond.py tests/test_capi_host.py tests/test_hip_fullsize.py -x -q -m gpu > gpurun_out/s2/t3.log 2>&1; echo "tests rc=$?" | tee -a $S
tail -4 gpurun_out/s2/t3.log | tee -a $S
for v in new facetgen; do
  if [ $v = facetgen ]; then export PHX_FACET_GENERIC=1; fi
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 > gpurun_out/s2/bench3_$v.json 2> gpurun_out/s2/bench3_$v.err; echo "bench $v rc=$?" | tee -a $S
  python - $v <<'PY' | tee -a $S
import json, sys
d = json.load(open(f"gpurun_out/s2/bench3_{sys.argv[1]}.json"))
print(sys.argv[1], d["ms_per_step"], d["config"]["stage_ms"], d["config"]["iterations"], d["config"]["relres"], "spmv us", d["roofline"]["avg_launch_us"], "y us", d["roofline_other"]["avg_launch_us"])
PY
done
