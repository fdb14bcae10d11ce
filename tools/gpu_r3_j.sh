#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
( while true; do echo "alive $(date +%s)" >> gpurun_out/j_alive.log; sleep 60; done ) &
MON=$!
timeout -k 10 1100 python -m pytest tests -x -q -m gpu --durations=12 > gpurun_out/j_all.log 2>&1
echo "all rc=$?"; kill $MON
tail -30 gpurun_out/j_all.log
