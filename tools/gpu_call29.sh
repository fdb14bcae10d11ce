#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
for cmd in "demo/weak-dirichlet/flower/main.py bg" "demo/weak-dirichlet/flower/main.py sub" "demo/weak-dirichlet/flower/main.py sub --degree 2" "demo/strong-dirichlet/flower/main.py sub" "demo/robin/square/main.py bg" "demo/neumann/square/main.py sub"; do
  echo "== $cmd"; timeout -k 10 200 python $cmd 2>&1 | tail -3 | cut -c1-200
done
