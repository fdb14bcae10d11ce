#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_hip_precond.py -x -q > $O/r2_t12.log 2>&1; echo "pytest rc=$?"; grep -E "assert|Error" $O/r2_t12.log | head -5; tail -4 $O/r2_t12.log
for g in 0 1; do
  if [ $g = 1 ]; then export PHX_DST_GENERIC=1; fi
  timeout -k 10 120 python tools/dst_bench.py 192 192 182 0 50
  timeout -k 10 120 python tools/dst_bench.py 256 256 256 0 50
done
