#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for p in 1 2; do
  export PHX_SPMV_PART=$p
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU --kernel-include-regex 'k_spmv' -d $O/r2_pmc6a_$p -o a --output-format csv -- python3 $R/tools/spmv_only.py 256 10 > $O/r2_pmc6a_$p.log 2>&1; echo "a$p rc=$?"
  timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-include-regex 'k_spmv' -d $O/r2_pmc6b_$p -o b --output-format csv -- python3 $R/tools/spmv_only.py 256 10 > $O/r2_pmc6b_$p.log 2>&1; echo "b$p rc=$?"
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex 'k_spmv' -d $O/r2_pmc6c_$p -o c --output-format csv -- python3 $R/tools/spmv_only.py 256 10 > $O/r2_pmc6c_$p.log 2>&1; echo "c$p rc=$?"
  timeout -k 10 200 rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum --kernel-include-regex 'k_spmv' -d $O/r2_pmc6d_$p -o d --output-format csv -- python3 $R/tools/spmv_only.py 256 10 > $O/r2_pmc6d_$p.log 2>&1; echo "d$p rc=$?"
  for q in a b c d; do echo "== part $p pass $q"; python3 $R/tools/pmc_summary.py $O/r2_pmc6${q}_$p k_spmv; done
done
