#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s2; mkdir -p $O
cd $R
S=$O/summary5.txt
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/t5.log 2>&1; echo "tests rc=$?" | tee -a $S
tail -4 $O/t5.log | tee -a $S
timeout -k 10 300 python bench.py --steps 10 --warmup 3 > $O/bench5.json 2> $O/bench5.err; echo "bench rc=$?" | tee -a $S
python - <<'PY' | tee -a $S
import json
d = json.load(open("gpurun_out/s2/bench5.json"))
print(d["ms_per_step"], d["config"]["stage_ms"], d["config"]["iterations"], d["config"]["relres"], "spmv us", d["roofline"]["avg_launch_us"], "y us", d["roofline_other"]["avg_launch_us"])
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/ks -o p --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/ks.log 2>&1; echo "stats rc=$?"
python3 $R/tools/gaps.py /tmp/ks/p_kernel_trace.csv 8 > $O/step_gaps5.txt 2>&1
python3 - <<PY
import csv
rows = list(csv.DictReader(open('/tmp/ks/p_kernel_trace.csv')))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("void k_tag_cells")]
seg = rows[starts[-1]:]
t0 = int(seg[0]["Start_Timestamp"])
with open('$O/timeline5.txt', 'w') as f:
    prev = t0
    for r in seg:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if (s - t0) / 1e6 > 9.0: break
        f.write(f"{(s - t0) / 1e3:9.1f} us  gap {(s - prev) / 1e3:6.1f}  dur {(e - s) / 1e3:7.1f}  {r['Kernel_Name'][:70]}\n")
        prev = e
PY
tail -3 $O/step_gaps5.txt
