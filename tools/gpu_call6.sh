#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 300 python bench.py --config5 --steps 2 --warmup 1 --no-cpu-baseline > $O/r2_c5.json 2> $O/r2_c5.err; echo "config5 rc=$?"; tail -1 $O/r2_c5.json | cut -c1-600
timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/r2_def.json 2> $O/r2_def.err; echo "default rc=$?"; tail -1 $O/r2_def.json | cut -c1-400
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > $O/r2_full.log 2>&1; echo "pytest rc=$?"; tail -5 $O/r2_full.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r2_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/r2_smoke.log
