#!/bin/bash
# rocprofv3 kernel table of one python tool: kstats.sh <out name> <script> [args...]  -> gpurun_out/r04/<name>_stats.txt
R=$GRAFT_REPO_ROOT; name=$1; shift
mkdir -p $R/gpurun_out/r04
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks_$name && timeout -k 10 800 rocprofv3 --kernel-trace --stats -d /tmp/ks_$name -o p --output-format csv -- python3 "$@" > $R/gpurun_out/r04/$name.log 2>&1 )
grep -v "rocprofv3\|output_stream\|tool.cpp" $R/gpurun_out/r04/$name.log | tail -5
python3 - /tmp/ks_$name/p_kernel_stats.csv $R/gpurun_out/r04/${name}_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
with open(sys.argv[2], "w") as f:
    for r in rows[:60]:
        f.write(f"{r['Name'][:100]:100s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:10.1f} {float(r['Percentage']):7.3f}\n")
PY
head -${LINES_OUT:-32} $R/gpurun_out/r04/${name}_stats.txt | cut -c1-140
if [ -n "$GAPS_US" ]; then python3 $R/tools/gaps.py /tmp/ks_$name/p_kernel_trace.csv $GAPS_US > $R/gpurun_out/r04/${name}_gaps.txt 2>&1; head -60 $R/gpurun_out/r04/${name}_gaps.txt | cut -c1-200; fi
