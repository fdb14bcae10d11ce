#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
cp phifem_amd/libphifem_hip.so /tmp/libphifem_orig.so
for v in 1 3 4; do
  cp phifem_amd/libphifem_pad$v.so phifem_amd/libphifem_hip.so
  timeout -k 10 300 python -m pytest tests/test_hip_precond.py -x -q -k "lattice or box_poisson" > $O/r2_t24_$v.log 2>&1; echo "pad $v pytest rc=$? $(tail -1 $O/r2_t24_$v.log)"
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/kp$v -o p --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/r2_ks24_$v.log 2>&1; echo "pad $v rc=$?"
  python3 - <<PY
import csv
for r in csv.DictReader(open('/tmp/kp$v/p_kernel_stats.csv')):
    if 'k_dst' in r['Name'] or 'k_tri' in r['Name']: print(r['Name'][:60], round(float(r['AverageNs'])/1e3,2))
PY
  cd $R
done
cp /tmp/libphifem_orig.so phifem_amd/libphifem_hip.so
