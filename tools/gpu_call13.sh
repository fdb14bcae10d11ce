#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 1150 python -m pytest tests -m gpu -x -q > $O/r2_gpu_tests_full.log 2>&1; echo "pytest rc=$?"; tail -6 $O/r2_gpu_tests_full.log
