#!/bin/bash
# round-2 GPU call 1: full GPU suite, default bench, SQ counters of the sine-transform passes
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r2_gpu_tests.log 2>&1
echo "pytest rc=$?" | tee -a $O/r2_gpu_tests.log
tail -5 $O/r2_gpu_tests.log
timeout -k 10 400 python bench.py --steps 5 --warmup 1 > $O/r2_bench1.json 2> $O/r2_bench1.err
echo "bench rc=$?"; tail -c 3000 $O/r2_bench1.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU \
  --kernel-include-regex 'k_dst' -d $O/r2_pmc_sq1 -o sq1 --output-format csv -- python3 $R/tools/dst_bench.py 192 192 192 0 5 > $O/r2_pmc_sq1.log 2>&1
echo "pmc1 rc=$?"
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
  --kernel-include-regex 'k_dst' -d $O/r2_pmc_sq2 -o sq2 --output-format csv -- python3 $R/tools/dst_bench.py 192 192 192 0 5 > $O/r2_pmc_sq2.log 2>&1
echo "pmc2 rc=$?"
python3 $R/tools/pmc_summary.py $O/r2_pmc_sq1 > $O/r2_pmc_sq1.txt 2>&1
python3 $R/tools/pmc_summary.py $O/r2_pmc_sq2 > $O/r2_pmc_sq2.txt 2>&1
cat $O/r2_pmc_sq1.txt $O/r2_pmc_sq2.txt
