#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 1100 python -m pytest tests/test_hip_multirank.py tests/test_hip_assembly.py -x -q -s > $O/r2_t8.log 2>&1; echo "pytest rc=$?"; grep -E "iterations, single|passed|failed|Error|error" $O/r2_t8.log | tail -20; tail -5 $O/r2_t8.log
