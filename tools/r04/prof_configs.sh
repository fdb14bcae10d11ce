#!/bin/bash
# rocprofv3 kernel tables of the non-default workloads (run on the GPU box; outputs under gpurun_out/r04/).
# usage: prof_configs.sh [tag]
set -o pipefail
R=$GRAFT_REPO_ROOT; TAG=${1:-base}; O=$R/gpurun_out/r04/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
stats() {  # name, bench args...
  local name=$1; shift
  rm -rf /tmp/ks_$name
  timeout -k 10 420 rocprofv3 --kernel-trace --stats -d /tmp/ks_$name -o p --output-format csv -- python3 $R/bench.py "$@" --no-cpu-baseline > $O/stats_$name.log 2>&1
  echo "stats $name rc=$?"
  python3 - "$name" "$O" "$*" <<'PY'
import csv, sys
name, O, args = sys.argv[1:4]
try:
    rows = list(csv.DictReader(open(f'/tmp/ks_{name}/p_kernel_stats.csv')))
except OSError as e:
    print('no stats', e); sys.exit(0)
with open(f'{O}/kernel_stats_{name}.txt', 'w') as f:
    f.write(f"# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py {args} --no-cpu-baseline  (MI355X)\n")
    f.write(f"{'Name':100s} {'Calls':>6s} {'TotalMs':>10s} {'AvgUs':>10s} {'Pct':>7s}\n")
    for r in rows:
        f.write(f"{r['Name'][:100]:100s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:10.1f} {float(r['Percentage']):7.3f}\n")
PY
  grep '^{' $O/stats_$name.log > $O/bench_$name.json
}
stats config5 --config5 --steps 2 --warmup 1 && \
stats config4_96 --config4 --cubes 96 --steps 2 --warmup 1 && \
stats config3_256 --config3 --cubes 256 --steps 1 --warmup 1
ls -la $O
