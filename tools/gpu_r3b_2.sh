#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s2
cd $GRAFT_REPO_ROOT
S=gpurun_out/s2/summary2.txt
timeout -k 10 400 python -m pytest tests/test_hip_precond.py -x -q -m gpu > gpurun_out/s2/t_precond2.log 2>&1; echo "precond tests rc=$?" | tee -a $S
tail -3 gpurun_out/s2/t_precond2.log | tee -a $S
for L in "192 192 183" "256 256 100" "512 512 40"; do
  timeout -k 10 120 python tools/dst_bench.py $L 0 50 2>&1 | tail -1 | tee -a $S
done
timeout -k 10 300 python bench.py --steps 10 --warmup 3 > gpurun_out/s2/bench_new2.json 2> gpurun_out/s2/bench_new2.err; echo "bench rc=$?" | tee -a $S
python - <<'PY' | tee -a $S
import json
for f in ("bench_new2",):
    d = json.load(open(f"gpurun_out/s2/{f}.json"))
    print(f, d["ms_per_step"], d["config"]["stage_ms"], d["config"]["iterations"], "spmv us", d["roofline"]["avg_launch_us"], "y us", d["roofline_other"]["avg_launch_us"])
PY
PHX_SELL_XCD=1 timeout -k 10 300 python bench.py --steps 10 --warmup 3 > gpurun_out/s2/bench_xcd.json 2> gpurun_out/s2/bench_xcd.err; echo "bench xcd rc=$?" | tee -a $S
python - <<'PY' | tee -a $S
import json
for f in ("bench_xcd",):
    d = json.load(open(f"gpurun_out/s2/{f}.json"))
    print(f, d["ms_per_step"], d["config"]["stage_ms"], d["config"]["iterations"], "spmv us", d["roofline"]["avg_launch_us"], "y us", d["roofline_other"]["avg_launch_us"])
PY
for part in 0 1 2; do
PHX_SPMV_PART=$part timeout -k 10 200 python tools/spmv_only.py 256 50 0 2>&1 | grep xcd_group | sed "s/^/part=$part /" | tee -a $S
PHX_SELL_XCD=1 PHX_SPMV_PART=$part timeout -k 10 200 python tools/spmv_only.py 256 50 0 2>&1 | grep xcd_group | sed "s/^/SELL_XCD part=$part /" | tee -a $S
done
