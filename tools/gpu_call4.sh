#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_hip_assembly.py tests/test_hip_precond.py -x -q > $O/r2_t4.log 2>&1; echo "pytest rc=$?"; tail -25 $O/r2_t4.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/r2_prof4 -o p4 --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/r2_prof4.log 2>&1; echo "prof rc=$?"
python3 - <<PY
import csv,sys
rows=list(csv.DictReader(open('$O/r2_prof4/p4_kernel_stats.csv')))
for r in rows[:45]:
    print(f"{r['Name'][:70]:70s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:9.3f} ms avg {float(r['AverageNs'])/1e3:9.1f} us {r['Percentage']}")
PY
