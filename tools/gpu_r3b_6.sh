#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s2; mkdir -p $O
cd $R
S=$O/summary6.txt
timeout -k 10 900 python -m pytest tests/test_hip_assembly.py tests/test_hip_precond.py tests/test_hip_fullsize.py tests/test_hip_multirank.py -x -q -m gpu > $O/t6.log 2>&1; echo "tests rc=$?" | tee -a $S
tail -4 $O/t6.log | tee -a $S
for v in new rocprim; do
  if [ $v = rocprim ]; then export PHX_SORT_ROCPRIM=1; fi
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench6_$v.json 2> $O/bench6_$v.err; echo "bench $v rc=$?" | tee -a $S
  python - $v <<'PY' | tee -a $S
import json, sys
d = json.load(open(f"gpurun_out/s2/bench6_{sys.argv[1]}.json"))
print(sys.argv[1], d["ms_per_step"], d["config"]["stage_ms"], d["config"]["iterations"], d["config"]["relres"], d["config"]["system"])
PY
done
