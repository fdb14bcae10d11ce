#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/prof_p2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_p2 -o p2_256 -- python3 $GRAFT_REPO_ROOT/bench.py --config3 --cubes 256 --steps 1 --warmup 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/f_p2_256.json 2> $GRAFT_REPO_ROOT/gpurun_out/f_p2_256.err
echo rc=$?
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_p2 -name "*kernel_stats*" | head
f=$(find gpurun_out/prof_p2 -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:18]:
    print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>7s} avg_us {float(r["AverageNs"])/1e3:10.1f} tot_ms {float(r["TotalDurationNs"])/1e6:9.1f} {r["Percentage"]}%')
PY
find gpurun_out/prof_p2 -name "*kernel_trace.csv" -delete
