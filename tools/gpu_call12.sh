#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_hip_precond.py -x -q > $O/r2_t12.log 2>&1; echo "pytest rc=$?"; grep -E "assert|Error" $O/r2_t12.log | head -5; tail -4 $O/r2_t12.log
PHX_Z_TRIDIAG=0 timeout -k 10 300 python -m pytest tests/test_hip_precond.py -x -q -k "box_poisson and not 180 and not 38 and not 1025 and not (64-128-2) and not (64-64-3) and not 40 and not 20 and not 30" 2>&1 | tail -3
for g in 0 1; do
  if [ $g = 1 ]; then export PHX_DST_GENERIC=1; fi
  timeout -k 10 120 python tools/dst_bench.py 192 192 182 0 50
  timeout -k 10 120 python tools/dst_bench.py 384 384 128 0 30
  timeout -k 10 120 python tools/dst_bench.py 768 768 192 0 20
done
