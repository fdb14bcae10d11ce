#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of k_spmv_p2s and of k_spmv_sell SEPARATELY on the structured P2 system (256^3): one pair of
# counter passes, records by tools/pmc_record.py.  Output: gpurun_out/r04/pmc_p2_{p2s,sell}.json
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r04
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pp_$c && timeout -k 10 400 rocprofv3 --pmc $c --kernel-include-regex "k_spmv_p2s|k_spmv_sell" -d /tmp/pp_$c -o c --output-format csv -- python3 $R/bench.py --config3 --cubes 256 --steps 1 --warmup 0 --no-cpu-baseline > /tmp/pp_$c.log 2>&1 ); echo "pmc $c rc=$?"
done
python3 $R/tools/pmc_record.py /tmp/pp_FETCH_SIZE /tmp/pp_WRITE_SIZE k_spmv_p2s k_spmv_p2s 0 $R/gpurun_out/r04/pmc_p2_p2s.json "class-stencil rows of the structured P2 system alone, 256^3"
python3 $R/tools/pmc_record.py /tmp/pp_FETCH_SIZE /tmp/pp_WRITE_SIZE k_spmv_sell k_spmv_sell 0 $R/gpurun_out/r04/pmc_p2_sell.json "stored rows (SELL-16, tile order) of the structured P2 system alone, 256^3"
grep -h "traffic_bytes_per_launch\|FETCH_SIZE_KB\|WRITE_SIZE_KB\|kernel_key" $R/gpurun_out/r04/pmc_p2_p2s.json $R/gpurun_out/r04/pmc_p2_sell.json
