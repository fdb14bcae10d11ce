#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for pr in 0 32768; do
  rm -rf /tmp/pm
  PLANE_ROWS=$pr PHX_SPMV_PART=2 timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex 'k_spmv_sell' -d /tmp/pm -o c --output-format csv -- python3 $R/tools/r04/spmv_parts.py config5 3 > /tmp/pm.log 2>&1 || { echo "pmc failed"; tail -3 /tmp/pm.log; }
  grep "config5 part" /tmp/pm.log
  echo "plane_rows $pr: $(python3 $R/tools/pmc_summary.py /tmp/pm k_spmv_sell | grep -v dispatches | tr -s ' ' | tr '\n' ';')"
done | tee $O/stencil_pmc.txt
