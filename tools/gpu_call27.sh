#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 600 python tools/submode_step.py 256 6 2>&1 | grep -E "n=|Error|error" | head -12
