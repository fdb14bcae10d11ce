#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu --deselect tests/test_hip_fullsize.py --deselect tests/test_hip_multirank.py > gpurun_out/h_gpu.log 2>&1
echo "gpu rc=$?"; tail -25 gpurun_out/h_gpu.log
