#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/ks4 -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --config4 --cubes 96 --steps 1 --warmup 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/l_el.json 2> $GRAFT_REPO_ROOT/gpurun_out/l_el.err
echo rc=$?
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('/tmp/ks4/p_kernel_stats.csv')))
for r in rows[:16]:
    print(f"{r['Name'][:90]:90s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:10.3f} ms  avg {float(r['AverageNs'])/1e3:10.1f} us {float(r['Percentage']):6.2f}%")
PY
