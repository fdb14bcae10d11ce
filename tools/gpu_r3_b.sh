#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_hip_p2.py -x -q -m gpu -s > gpurun_out/b_p2.log 2>&1
echo "p2 rc=$?"; tail -30 gpurun_out/b_p2.log
