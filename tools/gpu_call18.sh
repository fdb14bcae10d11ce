#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_hip_tagging.py tests/test_hip_flux.py tests/test_hip_flux_quad.py tests/test_hip_p2.py tests/test_hip_strong_dirichlet.py -x -q -m gpu > $O/r2_t18.log 2>&1; echo "pytest rc=$?"; tail -25 $O/r2_t18.log
