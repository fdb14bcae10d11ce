#!/bin/bash
# round 3, call A: multi-rank tests with the async stand-in + overlap, then the whole GPU suite, then a bench line
set -o pipefail
mkdir -p gpurun_out
python -c 'import __graft_entry__ as g; g.build()' > gpurun_out/a_build.log 2>&1 || { tail -20 gpurun_out/a_build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_hip_multirank.py -x -q -m gpu -s > gpurun_out/a_multirank.log 2>&1
echo "multirank rc=$?"; tail -15 gpurun_out/a_multirank.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu --deselect tests/test_hip_multirank.py > gpurun_out/a_gpu.log 2>&1
echo "gpu rc=$?"; tail -5 gpurun_out/a_gpu.log
timeout -k 10 300 python bench.py --steps 5 --warmup 2 > gpurun_out/a_bench.json 2> gpurun_out/a_bench.err
echo "bench rc=$?"; cat gpurun_out/a_bench.json | head -c 3000
