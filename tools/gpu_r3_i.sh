#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_hip_deterministic.py tests/test_hip_p2.py tests/test_hip_elasticity.py -x -q -m gpu -s > gpurun_out/i_det.log 2>&1
echo "det rc=$?"; grep -E "iterations|passed|failed|Error|assert" gpurun_out/i_det.log | tail -20
for rep in 1 2; do
timeout -k 10 600 python bench.py --config3 --cubes 128 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/i_p2_128_$rep.json 2> gpurun_out/i_p2_128_$rep.err
python - <<PY
import json
d=json.load(open("gpurun_out/i_p2_128_$rep.json")); c=d["config"]; print("cubes 128 run $rep:", d["ms_per_step"], c["iterations"], c["relres"], c["stage_ms"])
PY
done
timeout -k 10 300 python tools/caller_lattice_step.py 128 2>&1 | tail -2
