#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
cp phifem_amd/libphifem_hip.so /tmp/libphifem_orig.so
for v in 1 2; do
  cp phifem_amd/libphifem_exp$v.so phifem_amd/libphifem_hip.so
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/ks$v -o p --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/r2_ks20_$v.log 2>&1; echo "variant $v rc=$?"
  python3 - <<PY
import csv
for r in csv.DictReader(open('/tmp/ks$v/p_kernel_stats.csv')):
    if 'k_assemble_facets' in r['Name'] or 'k_assemble_ds' in r['Name']: print(r['Name'][:50], float(r['AverageNs'])/1e3)
PY
  cd $R
done
cp /tmp/libphifem_orig.so phifem_amd/libphifem_hip.so
