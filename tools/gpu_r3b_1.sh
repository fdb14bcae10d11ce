#!/bin/bash
# round 3, session 2, call 1: new wave-mode DST kernels -- correctness, A/B timing, bench
set -o pipefail
mkdir -p gpurun_out/s2
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_hip_precond.py -x -q -m gpu > gpurun_out/s2/t_precond.log 2>&1; echo "precond tests rc=$?" | tee -a gpurun_out/s2/summary.txt
tail -3 gpurun_out/s2/t_precond.log | tee -a gpurun_out/s2/summary.txt
for L in "192 192 183" "256 256 100" "512 512 40"; do
  timeout -k 10 120 python tools/dst_bench.py $L 0 50 2>&1 | tail -1 | tee -a gpurun_out/s2/summary.txt
  PHX_DST_OLD=1 timeout -k 10 120 python tools/dst_bench.py $L 0 50 2>&1 | tail -1 | sed 's/^/OLD /' | tee -a gpurun_out/s2/summary.txt
done
timeout -k 10 300 python bench.py --steps 10 --warmup 3 > gpurun_out/s2/bench_new.json 2> gpurun_out/s2/bench_new.err; echo "bench rc=$?" | tee -a gpurun_out/s2/summary.txt
cat gpurun_out/s2/bench_new.json | tee -a gpurun_out/s2/summary.txt
PHX_DST_OLD=1 timeout -k 10 300 python bench.py --steps 10 --warmup 3 > gpurun_out/s2/bench_old.json 2> gpurun_out/s2/bench_old.err; echo "bench old rc=$?" | tee -a gpurun_out/s2/summary.txt
cat gpurun_out/s2/bench_old.json | tee -a gpurun_out/s2/summary.txt
