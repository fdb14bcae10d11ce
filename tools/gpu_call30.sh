#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 600 python bench.py --config3 --cubes 256 --steps 1 --warmup 1 --no-cpu-baseline > $O/r2_c3.json 2> $O/r2_c3.err; echo "c3 rc=$?"
timeout -k 10 600 python bench.py --config4 --cubes 96 --steps 1 --warmup 1 --no-cpu-baseline > $O/r2_c4.json 2> $O/r2_c4.err; echo "c4 rc=$?"
python - <<'PY'
import json
for f in ('r2_c3','r2_c4'):
    d=json.loads([l for l in open('gpurun_out/%s.json'%f) if l.startswith('{')][-1])
    print(f, round(d['ms_per_step'],1), d['config'].get('iterations'), d['config'].get('relres'), d['config'].get('converged'), d['config'].get('stage_ms'))
PY
