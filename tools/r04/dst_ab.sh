#!/bin/bash
# A/B of the long sine-transform passes: one wavefront per pair (phx_dst_pair.inc.hip) against the block-synchronous
# kernels (PHX_DST_LONG_OLD=1).  Run on the GPU box.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_hip_precond.py tests/test_hip_multirank.py -x -q -m gpu 2>&1 | tail -5 || exit 1
for shape in "768 768 194" "384 384 354" "1024 1024 60" "192 192 182" "256 256 250" "512 512 400"; do
  timeout -k 10 120 python tools/dst_bench.py $shape 0 20 || exit 1

done 2>&1 | tee $O/dst_ab.txt
