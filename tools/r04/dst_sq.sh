#!/bin/bash
# SQ counters of the sine-transform / tridiagonal passes on a given lattice (run on the GPU box).
# usage: dst_sq.sh "L0 L1 L2" tag
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
SHAPE=${1:-"768 768 194"}; TAG=${2:-768}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sq1 /tmp/sq2
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --kernel-include-regex 'k_dst|k_tri' -d /tmp/sq1 -o s --output-format csv -- python3 $R/tools/dst_bench.py $SHAPE 0 3 > $O/pmc_sq1.log 2>&1 || { tail -5 $O/pmc_sq1.log; exit 1; }
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-include-regex 'k_dst|k_tri' -d /tmp/sq2 -o s --output-format csv -- python3 $R/tools/dst_bench.py $SHAPE 0 3 > $O/pmc_sq2.log 2>&1 || { tail -5 $O/pmc_sq2.log; exit 1; }
(python3 $R/tools/pmc_summary.py /tmp/sq1; python3 $R/tools/pmc_summary.py /tmp/sq2) > $O/pmc_dst_sq_$TAG.txt 2>&1
cat $O/pmc_dst_sq_$TAG.txt
