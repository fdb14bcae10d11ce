#!/bin/bash
# bench.py --config3 under PHX_P2S_TILE = ...: SpMV pair time and step time (the iteration count moves with the last bits)
set -e
n=${1:-256}
mkdir -p gpurun_out/r04
for t in ${TILES_AB:-16 32}; do
  PHX_P2S_TILE=$t timeout -k 10 400 python bench.py --config3 --cubes $n --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r04/b_p2_tile$t.log 2>&1
  python3 - gpurun_out/r04/b_p2_tile$t.log $t <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("tile", sys.argv[2], "ms/step", round(d["ms_per_step"], 1), "its", d["config"]["iterations"], d["config"]["stage_ms"],
      "spmv pair us", round(d["roofline"]["avg_launch_us"], 1), "padded", d["config"]["system"]["sell_padded_nnz"])
PY
done
