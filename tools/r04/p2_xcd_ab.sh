#!/bin/bash
# A/B of the SELL block -> XCD map on the structured P2 system (BASELINE configs[2], 256^3): PHX_SPMV_XCD_GROUP = 0 round robin,
# G: XCD k takes G consecutive blocks of every run of 8 G (PHX_SELL_XCD=1, contiguous eighths, measured: 1680 us against 1277).  Same box, back to back.  Usage: gpurun -- bash tools/r04/p2_xcd_ab.sh [cubes]
set -e
n=${1:-256}
mkdir -p gpurun_out/r04
for v in ${GROUPS_AB:-0 32 64 128}; do
  PHX_SPMV_XCD_GROUP=$v timeout -k 10 400 python bench.py --config3 --cubes $n --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r04/b_p2_xcd$v.log 2>&1
  python3 - gpurun_out/r04/b_p2_xcd$v.log <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "ms/step", round(d["ms_per_step"], 1), "its", d["config"]["iterations"], d["config"]["stage_ms"],
      "spmv pair us", round(d["roofline"]["avg_launch_us"], 1))
PY
done
