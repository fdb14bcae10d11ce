#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
for v in 0 1 2 3 4 5 7; do
  PHX_SELL_EXP=$v timeout -k 10 200 python tools/spmv_only.py 256 50 2>&1 | grep xcd_group | sed "s/^/var $v: /"
done
