#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  rm -rf /tmp/pc
  timeout -k 10 120 rocprofv3 --pmc $ctr -d /tmp/pc -o c --output-format csv -- $R/tools/r04/pmc_calib > /tmp/pc.log 2>&1 || { echo "pmc $ctr failed"; tail -3 /tmp/pc.log; continue; }
  grep "bytes per launch" /tmp/pc.log
  python3 $R/tools/pmc_summary.py /tmp/pc calib
done 2>&1 | tee $O/pmc_calibration.txt
