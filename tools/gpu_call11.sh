#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 1100 python -m pytest tests/test_hip_precond.py tests/test_hip_fullsize.py -x -q > $O/r2_t11.log 2>&1; echo "pytest rc=$?"; tail -15 $O/r2_t11.log
