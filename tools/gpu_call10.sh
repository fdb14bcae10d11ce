#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_hip_fullsize.py -x -q > $O/r2_t10.log 2>&1; echo "pytest rc=$?"; tail -15 $O/r2_t10.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/r2_prof10 -o p --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/r2_prof10.log 2>&1; echo "prof rc=$?"
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$O/r2_prof10/p_kernel_stats.csv')))
with open('$O/r2_kernel_stats10.txt','w') as f:
    for r in rows[:60]:
        line=f"{r['Name'][:90]:90s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:9.3f} ms avg {float(r['AverageNs'])/1e3:9.1f} us {r['Percentage']}"
        f.write(line+"\n")
        print(line)
PY
rm -f $O/r2_prof10/p_kernel_trace.csv
