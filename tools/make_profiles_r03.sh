#!/bin/bash
# Regenerates the evidence committed under profiles/r03/ (run on the GPU box; outputs under gpurun_out/r03/).
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O; cd $R
timeout -k 10 500 python bench.py --steps 20 --warmup 2 > $O/bench_default_n1.json 2> $O/bench_default_n1.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/ks -o p --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/kernel_stats.log 2>&1; echo "stats rc=$?"
python3 - <<PY
import csv
rows=list(csv.DictReader(open('/tmp/ks/p_kernel_stats.csv')))
with open('$O/kernel_stats_bench_256_steps2.txt','w') as f:
    f.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline\n")
    f.write("# (MI355X, 256^3 box, 3 pipeline passes incl. warm-up; round-3 build)\n")
    f.write(f"{'Name':100s} {'Calls':>6s} {'TotalMs':>10s} {'AvgUs':>10s} {'Pct':>7s}\n")
    for r in rows:
        f.write(f"{r['Name'][:100]:100s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:10.1f} {float(r['Percentage']):7.3f}\n")
PY
cp /tmp/ks/p_kernel_stats.csv $O/kernel_stats_bench_256_steps2.csv
python3 $R/tools/gaps.py /tmp/ks/p_kernel_trace.csv 8 > $O/step_gaps.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-include-regex 'k_spmv_sell|k_dst_yw|k_tri_z|k_dst_xw' -d /tmp/pmc_$c -o c --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_$c.log 2>&1; echo "pmc $c rc=$?"
done
ALG=$(python3 -c "import json;d=json.loads([l for l in open('$O/bench_default_n1.json') if l.startswith('{')][-1]);r=d['roofline'] if 'spmv' in d['roofline']['kernel'] else d['roofline_other'];print(r['bytes_per_launch'])")
ALGD=$(python3 -c "import json;d=json.loads([l for l in open('$O/bench_default_n1.json') if l.startswith('{')][-1]);r=d['roofline'] if 'dst' in d['roofline']['kernel'] else d['roofline_other'];print(r['bytes_per_launch'])")
python3 $R/tools/pmc_record.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE "k_spmv_sell" k_spmv_sell $ALG $O/pmc_spmv_256.json "structured SpMV (stencil blocks + SELL-16 blocks in one launch), 256^3 default workload"
python3 $R/tools/pmc_record.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE "k_dst_yw<192>" k_dst_yw $ALGD $O/pmc_dst_256.json "sine transform along y of the 192 x 192 x 182 preconditioner lattice, f64, wave-mode kernel (rows outside the active interval of a plane skipped)"
python3 $R/tools/pmc_record.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE "k_tri_z" k_tri_z $ALGD $O/pmc_tri_256.json "tridiagonal z pass of the preconditioner lattice, f64"
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --kernel-include-regex 'k_dst|k_tri' -d /tmp/sq1 -o s --output-format csv -- python3 $R/tools/dst_bench.py 192 192 182 0 5 > $O/pmc_sq1.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-include-regex 'k_dst|k_tri' -d /tmp/sq2 -o s --output-format csv -- python3 $R/tools/dst_bench.py 192 192 182 0 5 > $O/pmc_sq2.log 2>&1
(python3 $R/tools/pmc_summary.py /tmp/sq1; python3 $R/tools/pmc_summary.py /tmp/sq2) > $O/pmc_dst_sq.txt 2>&1
cd $R
timeout -k 10 500 python bench.py --config5 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_config5_n1.json 2> $O/bench_config5_n1.err; echo "c5 rc=$?"
timeout -k 10 600 python bench.py --config3 --cubes 256 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_config3_p2_256_n1.json 2> $O/bench_config3_256.err; echo "c3-256 rc=$?"
timeout -k 10 900 python bench.py --config3 --cubes 512 --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_config3_p2_512_n1.json 2> $O/bench_config3_512.err; echo "c3-512 rc=$?"
timeout -k 10 600 python bench.py --config4 --cubes 96 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_config4_el_96_n1.json 2> $O/bench_config4.err; echo "c4 rc=$?"
timeout -k 10 600 python bench.py --config4 --cubes 256 --steps 2 --warmup 2 --no-cpu-baseline > $O/bench_config4_el_256_n1.json 2> $O/bench_config4_256.err; echo "c4-256 rc=$?"
timeout -k 10 300 python tools/caller_lattice_step.py 128 > $O/caller_lattice_step_128.txt 2>&1; tail -1 $O/caller_lattice_step_128.txt
rm -f $O/*.log
ls -la $O; python3 - <<PY
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); c=d['config']
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],2), c['iterations'], c['converged'], {k:round(v,2) for k,v in c['stage_ms'].items()})
    except Exception as e: print(f, 'ERR', e)
PY
