#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/ks -o p --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/r2_ks12.log 2>&1; echo "stats rc=$?"
python3 - <<PY
import csv
rows=list(csv.DictReader(open('/tmp/ks/p_kernel_stats.csv')))
skip=('k_dst','k_spmv','k_tri_z','k_update','k_copy_list')
with open('$O/r2_ks12.txt','w') as f:
    for r in rows:
        if any(k in r['Name'] for k in skip): continue
        f.write(f"{r['Name'][:90]:90s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/3e6:10.3f} {float(r['AverageNs'])/1e3:10.1f}\n")
PY
python3 $R/tools/gaps.py /tmp/ks/p_kernel_trace.csv 8 > $O/r2_gaps12.txt 2>&1
head -45 $O/r2_ks12.txt
