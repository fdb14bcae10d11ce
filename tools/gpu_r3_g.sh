#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_precond.py tests/test_hip_assembly.py tests/test_hip_strong_dirichlet.py -x -q -m gpu > gpurun_out/g_tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/g_tests.log
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/g_bench.json 2> gpurun_out/g_bench.err
echo "bench rc=$?"; python - <<PY
import json
d=json.load(open("gpurun_out/g_bench.json"))
c=d["config"]; print(d["value"], d["ms_per_step"], c["iterations"], c["stage_ms"], d["roofline"]["avg_launch_us"], d["roofline_other"]["avg_launch_us"])
PY
