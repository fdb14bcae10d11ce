#!/bin/bash
# A/B of PHX_SELL_TILE (order of the stored rows of the structured P1 system) on the default workload / the configs[4] slab
set -e
mkdir -p gpurun_out/r04
extra=""; [ "$1" = "config5" ] && extra="--config5"
for v in ${TILES_AB:-0 8 16}; do
  PHX_SELL_TILE=$v timeout -k 10 400 python bench.py $extra --steps 5 --warmup 2 --no-cpu-baseline --no-configs4-extra > gpurun_out/r04/b_p1_tile$v.log 2>&1
  python3 - gpurun_out/r04/b_p1_tile$v.log $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"] if "spmv" in d["roofline"]["kernel"] else d.get("roofline_other", {})
print("tile", sys.argv[2], "ms/step", round(d["ms_per_step"], 2), "its", d["config"]["iterations"], "spmv us", round(r.get("avg_launch_us", 0), 1),
      "padded", d["config"]["system"]["sell_padded_nnz"], d["config"]["stage_ms"])
PY
done
