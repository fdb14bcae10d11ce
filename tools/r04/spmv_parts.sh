#!/bin/bash
# SpMV alone, whole launch / SELL-16 blocks / stencil blocks, with FETCH_SIZE / WRITE_SIZE and the raw request counters
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
WHAT=${1:-256}
for part in 0 1 2; do
  PHX_SPMV_PART=$part timeout -k 10 200 python tools/r04/spmv_parts.py $WHAT 20 2>&1 | grep -v amdgpu.ids | tail -1
done | tee $O/spmv_parts_$WHAT.txt
cd /tmp && export TMPDIR=/tmp
for part in 0 1 2; do
  for ctr in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $ctr | tr ' ' '+')
    rm -rf /tmp/pm
    PHX_SPMV_PART=$part timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-include-regex 'k_spmv_sell' -d /tmp/pm -o c --output-format csv -- python3 $R/tools/r04/spmv_parts.py $WHAT 3 > /tmp/pm.log 2>&1 || { echo "pmc $ctr failed"; tail -3 /tmp/pm.log; continue; }
    echo "part $part $tag: $(python3 $R/tools/pmc_summary.py /tmp/pm k_spmv_sell | grep -v dispatches | tr -s ' ' | tr '\n' ';')"
  done
done | tee $O/spmv_parts_pmc_$WHAT.txt
