#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/experiments/rccl_same_gpu.py > $O/r2_rccl17.log 2>&1; echo "rc=$?"; tail -15 $O/r2_rccl17.log
