#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_hip_p2.py -x -q -m gpu > gpurun_out/e_p2.log 2>&1 || { tail -20 gpurun_out/e_p2.log; exit 1; }
tail -2 gpurun_out/e_p2.log
for n in 256 512; do
  timeout -k 10 900 python bench.py --config3 --cubes $n --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/e_p2_$n.json 2> gpurun_out/e_p2_$n.err
  echo "cubes $n rc=$?"; python - <<PY
import json
d=json.load(open("gpurun_out/e_p2_$n.json"))
c=d["config"]; print(d["value"], d["ms_per_step"], c["active_dofs"], c["iterations"], c["converged"], c["stage_ms"], d["roofline"]["avg_launch_us"], c["system"])
PY
done
