#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --stats -d /tmp/sm -o p --output-format csv -- python3 $R/tools/submode_step.py 256 2 > $O/r2_sm28.log 2>&1; echo "rc=$?"
grep "n=" $O/r2_sm28.log
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('/tmp/sm/p_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)/1e6
print("kernel total ms", round(tot,1))
for r in rows[:14]: print(r['Name'][:70], r['Calls'], round(float(r['TotalDurationNs'])/1e6,2))
try:
    rows=list(csv.DictReader(open('/tmp/sm/p_memory_copy_stats.csv')))
    for r in rows: print('COPY', r['Name'][:40], r['Calls'], round(float(r['TotalDurationNs'])/1e6,2))
except Exception as e: print(e)
PY
