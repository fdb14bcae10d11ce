#!/bin/bash
# Regenerates the evidence committed under profiles/r04/ (run on the GPU box; outputs under gpurun_out/r04p/).
# usage: make_profiles_r04.sh [part ...]   parts: default config5 config3 config4 big   (no argument: all but `big`)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04p; mkdir -p $O; cd $R
PARTS=${@:-default config5 config3 config4}
has() { [[ " $PARTS " == *" $1 "* ]]; }
fmt_stats() {  # name, args
  python3 - "$1" "$O" "$2" <<'PY'
import csv, sys
name, O, args = sys.argv[1:4]
rows = list(csv.DictReader(open(f'/tmp/ks_{name}/p_kernel_stats.csv')))
with open(f'{O}/kernel_stats_{name}.txt', 'w') as f:
    f.write(f"# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py {args}  (MI355X, round-4 build)\n")
    f.write(f"{'Name':100s} {'Calls':>6s} {'TotalMs':>10s} {'AvgUs':>10s} {'Pct':>7s}\n")
    for r in rows:
        f.write(f"{r['Name'][:100]:100s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:10.1f} {float(r['Percentage']):7.3f}\n")
PY
}
stats() {  # name, bench args...
  local name=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks_$name && timeout -k 10 420 rocprofv3 --kernel-trace --stats -d /tmp/ks_$name -o p --output-format csv -- python3 $R/bench.py "$@" > /tmp/ks_$name.log 2>&1 ); echo "stats $name rc=$?"
  fmt_stats $name "$*"
}
pmc() {  # name, regex, bench args...: two passes (FETCH_SIZE, WRITE_SIZE), never combined with tracing
  local name=$1 rx=$2; shift 2
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmc_${name}_$c && timeout -k 10 600 rocprofv3 --pmc $c --kernel-include-regex "$rx" -d /tmp/pmc_${name}_$c -o c --output-format csv -- python3 $R/bench.py "$@" > /tmp/pmc_${name}_$c.log 2>&1 ); echo "pmc $name $c rc=$?"
  done
}
rec() {  # name, kernel substring(s), key, json with the bench line, which roofline ("spmv" | "dst"), out, note
  local alg=$(python3 -c "import json,sys;d=json.loads([l for l in open('$4') if l.startswith('{')][-1]);r=d['roofline'] if '$5' in d['roofline']['kernel'] else d['roofline_other'];print(r['bytes_per_launch'])")
  python3 $R/tools/pmc_record.py /tmp/pmc_$1_FETCH_SIZE /tmp/pmc_$1_WRITE_SIZE "$2" "$3" $alg $O/$6 "$7" > /dev/null && echo "record $6 ok"
}
if has default; then
  timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $O/bench_default_n1.json 2> $O/bench_default_n1.err; echo "bench rc=$?"
  stats bench_256_steps2 --steps 2 --warmup 1 --no-cpu-baseline --no-configs4-extra
  cp /tmp/ks_bench_256_steps2/p_kernel_stats.csv $O/kernel_stats_bench_256_steps2.csv
  python3 $R/tools/gaps.py /tmp/ks_bench_256_steps2/p_kernel_trace.csv 8 > $O/step_gaps.txt 2>&1
  pmc d256 'k_spmv_sell|k_dst_yw|k_tri_z|k_dst_xw' --steps 1 --warmup 1 --no-cpu-baseline --no-configs4-extra
  rec d256 "k_spmv_sell" k_spmv_sell $O/bench_default_n1.json spmv pmc_spmv_256.json "structured SpMV (stencil blocks + SELL-16 blocks in one launch), 256^3 default workload"
  rec d256 "k_dst_yw<192>" k_dst_yw $O/bench_default_n1.json dst pmc_dst_256.json "sine transform along y of the 192 x 192 x 182 preconditioner lattice, f64, wave-mode kernel"
  rec d256 "k_tri_z" k_tri_z $O/bench_default_n1.json dst pmc_tri_256.json "tridiagonal z pass of the preconditioner lattice, f64"
fi
if has config5; then
  timeout -k 10 500 python bench.py --config5 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_config5_n1.json 2> $O/bench_config5_n1.err; echo "c5 rc=$?"
  stats config5 --config5 --steps 2 --warmup 1 --no-cpu-baseline
  pmc c5 'k_spmv_sell|k_dst_yp|k_dst_xp|k_tri_z' --config5 --steps 1 --warmup 0 --no-cpu-baseline
  rec c5 "k_spmv_sell" k_spmv_sell $O/bench_config5_n1.json spmv pmc_spmv_config5.json "structured SpMV, 1024 x 1024 x 128 slab (BASELINE configs[4])"
  rec c5 "k_dst_yp<768" k_dst_yp $O/bench_config5_n1.json dst pmc_dst_config5.json "sine transform along y of the 768 x 768 x 194 lattice, one wavefront per pair of lines"
  rec c5 "k_dst_xp<768" k_dst_xp $O/bench_config5_n1.json dst pmc_dstx_config5.json "sine transform along x (gathering / scattering passes), 768 x 768 x 194 lattice"
  rec c5 "k_tri_z" k_tri_z $O/bench_config5_n1.json dst pmc_tri_config5.json "tridiagonal z pass, 768 x 768 x 194 lattice"
fi
if has config3; then
  timeout -k 10 600 python bench.py --config3 --cubes 256 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_config3_p2_256_n1.json 2> $O/bench_config3_256.err; echo "c3-256 rc=$?"
  stats config3_256 --config3 --cubes 256 --steps 1 --warmup 0 --no-cpu-baseline
  pmc c3 'k_spmv_sell|k_spmv_p2s|k_dst_yp|k_tri_z' --config3 --cubes 256 --steps 1 --warmup 0 --no-cpu-baseline
  rec c3 "k_spmv_sell+k_spmv_p2s" "k_spmv_sell+k_spmv_p2s" $O/bench_config3_p2_256_n1.json spmv pmc_spmv_config3_256.json "SpMV of the structured P2 system (SELL-16 band rows + class stencils), 256^3"
  rec c3 "k_dst_yp<384" k_dst_yp $O/bench_config3_p2_256_n1.json dst pmc_dst_config3_256.json "sine transform along y of the 384 x 384 x 354 fine lattice"
  # the line again, now quoting the traffic just measured
  cp $O/pmc_spmv_config3_256.json $O/pmc_dst_config3_256.json $R/profiles/r04/
  timeout -k 10 600 python bench.py --config3 --cubes 256 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_config3_p2_256_n1.json 2> $O/bench_config3_256.err; echo "c3-256 (final) rc=$?"
fi
if has config4; then
  timeout -k 10 600 python bench.py --config4 --cubes 96 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_config4_el_96_n1.json 2> $O/bench_config4.err; echo "c4 rc=$?"
  stats config4_96 --config4 --cubes 96 --steps 2 --warmup 1 --no-cpu-baseline
  pmc c4 'k_spmv_sell|k_bj_apply' --config4 --cubes 96 --steps 1 --warmup 0 --no-cpu-baseline
  rec c4 "k_spmv_sell" k_spmv_sell $O/bench_config4_el_96_n1.json spmv pmc_spmv_config4_96.json "SpMV of the interface-elasticity system (SELL-64, value-indexed slices), 96^3"
  cp $O/pmc_spmv_config4_96.json $R/profiles/r04/
  timeout -k 10 600 python bench.py --config4 --cubes 96 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_config4_el_96_n1.json 2> $O/bench_config4.err; echo "c4 (final) rc=$?"
fi
if has big; then
  timeout -k 10 1000 python bench.py --config3 --cubes 512 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_config3_p2_512_n1.json 2> $O/bench_config3_512.err; echo "c3-512 rc=$?"
  timeout -k 10 600 python bench.py --config4 --cubes 256 --steps 2 --warmup 2 --no-cpu-baseline > $O/bench_config4_el_256_n1.json 2> $O/bench_config4_256.err; echo "c4-256 rc=$?"
fi
rm -f $O/*.err
ls -la $O; python3 - <<PY
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); c=d['config']
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],2), c['iterations'], c['converged'], {k:round(v,2) for k,v in c['stage_ms'].items()}, d['roofline'].get('traffic'), d.get('configs4_slab',{}).get('ms_per_step'))
    except Exception as e: print(f, 'ERR', e)
PY
