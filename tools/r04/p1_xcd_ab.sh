#!/bin/bash
# A/B of PHX_SPMV_XCD_GROUP on the default workload (P1, 256^3) and, with "config5", on the configs[4] slab.  Same box.
set -e
mkdir -p gpurun_out/r04
extra=""; [ "$1" = "config5" ] && extra="--config5"
for v in ${GROUPS_AB:-0 16 64}; do
  PHX_SPMV_XCD_GROUP=$v timeout -k 10 400 python bench.py $extra --steps 5 --warmup 2 --no-cpu-baseline --no-configs4-extra > gpurun_out/r04/b_p1_xcd$v.log 2>&1
  python3 - gpurun_out/r04/b_p1_xcd$v.log <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"] if "spmv" in d["roofline"]["kernel"] else d.get("roofline_other", {})
print(sys.argv[1], "ms/step", round(d["ms_per_step"], 2), "its", d["config"]["iterations"], "spmv us", round(r.get("avg_launch_us", 0), 1))
PY
done
