#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
for t in 0 4 8 16; do
  echo "== tile $t"
  PHX_SELL_TILE=$t timeout -k 10 200 python tools/spmv_only.py 256 50 > $O/r2_tile$t.log 2>&1 || { echo "spmv_only failed"; tail -5 $O/r2_tile$t.log; exit 1; }
  grep -E "xcd_group" $O/r2_tile$t.log; grep -oE "'sell_padded_nnz': [0-9]+|'sell_nnz': [0-9]+|'iterations': [0-9]+" $O/r2_tile$t.log | tr '\n' ' '; echo
  PHX_SELL_TILE=$t PHX_SPMV_PART=1 timeout -k 10 200 python tools/spmv_only.py 256 50 2>&1 | grep xcd_group
done
for t in 0 8; do
  PHX_SELL_TILE=$t timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/r2_bench7_$t.json 2> $O/r2_bench7_$t.err; echo "bench rc=$? tile=$t"
  python - $t <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r2_bench7_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print(d['ms_per_step'], d['config']['iterations'], d['config']['stage_ms'], d['roofline']['avg_launch_us'], d['roofline_other']['avg_launch_us'])
PY
done
