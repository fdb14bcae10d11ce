#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_hip_flux_quad.py tests/test_hip_flux.py -x -q > $O/r2_t17.log 2>&1; echo "pytest rc=$?"; tail -25 $O/r2_t17.log
cd demo/neumann/square && timeout -k 10 300 python main.py bg --cells 100 && timeout -k 10 300 python main.py bg && timeout -k 10 300 python main.py sub
