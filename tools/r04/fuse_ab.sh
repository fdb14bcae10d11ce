#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_hip_assembly.py tests/test_hip_precond.py tests/test_hip_deterministic.py -x -q -m gpu > $O/t_fuse.log 2>&1; tail -4 $O/t_fuse.log
for fz in 1 0; do
  PHX_FUSE_VEC=$fz timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-configs4-extra > $O/bf_d_$fz.json 2>/dev/null
  PHX_FUSE_VEC=$fz timeout -k 10 300 python bench.py --config5 --steps 3 --warmup 1 --no-cpu-baseline > $O/bf_c5_$fz.json 2>/dev/null
done
python3 - <<PY
import json
for f in ["bf_d_1","bf_d_0","bf_c5_1","bf_c5_0"]:
    try:
        d=json.loads([l for l in open(f"$O/{f}.json") if l.startswith("{")][-1]); c=d["config"]
        print(f, round(d["value"]), round(d["ms_per_step"],2), c["iterations"], c["converged"], c["relres"], {k:round(v,2) for k,v in c["stage_ms"].items()})
    except Exception as e: print(f, "ERR", e)
PY
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kc && timeout -k 10 120 rocprofv3 --kernel-trace --stats -d /tmp/kc -o p --output-format csv -- $R/tools/r04/pmc_calib > /tmp/kc.log 2>&1
python3 - <<PY
import csv
for r in csv.DictReader(open('/tmp/kc/p_kernel_stats.csv')):
    print(r['Name'][:40], r['Calls'], float(r['AverageNs'])/1e3, 'us')
PY
