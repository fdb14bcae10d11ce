#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_hip_elasticity.py tests/test_hip_deterministic.py tests/test_hip_multirank.py -x -q -m gpu -s > gpurun_out/k_el.log 2>&1
echo "el rc=$?"; grep -E "iterations|passed|failed|Error|assert" gpurun_out/k_el.log | tail -12
for n in 64 96; do
timeout -k 10 600 python bench.py --config4 --cubes $n --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/k_el_$n.json 2> gpurun_out/k_el_$n.err
python - <<PY
import json
d=json.load(open("gpurun_out/k_el_$n.json")); c=d["config"]; print("config4 cubes $n:", d["value"], d["ms_per_step"], c["active_dofs"], c["iterations"], c["converged"], c["stage_ms"])
PY
done
