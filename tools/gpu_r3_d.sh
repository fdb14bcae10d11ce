#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
( while true; do rocm-smi --showmeminfo vram 2>/dev/null | grep "Used" >> gpurun_out/d_mem.log; sleep 5; done ) &
MON=$!
timeout -k 10 1000 python bench.py --config3 --cubes 512 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/d_p2_512.json 2> gpurun_out/d_p2_512.err
echo "cubes 512 rc=$?"
kill $MON
tail -5 gpurun_out/d_p2_512.err
head -c 2500 gpurun_out/d_p2_512.json
sort -t: -k3 -n gpurun_out/d_mem.log | tail -1
