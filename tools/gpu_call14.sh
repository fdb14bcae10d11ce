#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/tr -o t --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/r2_tr14.log 2>&1; echo "rc=$?"
python3 $R/tools/gaps.py /tmp/tr/t_kernel_trace.csv 8 > $O/r2_gaps14.txt; head -70 $O/r2_gaps14.txt; tail -3 $O/r2_gaps14.txt
