#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_hip_tagging.py -x -q -m gpu -k "meshtags_keep or topology or overwrite" > $O/r2_t26.log 2>&1; echo "pytest rc=$?"; tail -12 $O/r2_t26.log
