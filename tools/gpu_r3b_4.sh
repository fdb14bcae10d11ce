#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/ks -o p --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/ks.log 2>&1; echo "stats rc=$?"
python3 $R/tools/gaps.py /tmp/ks/p_kernel_trace.csv 8 > $O/step_gaps4.txt 2>&1
python3 - <<PY
import csv
rows = list(csv.DictReader(open('/tmp/ks/p_kernel_trace.csv')))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("void k_tag_cells")]
seg = rows[starts[-1]:]
t0 = int(seg[0]["Start_Timestamp"])
with open('$O/timeline4.txt', 'w') as f:
    prev = t0
    for r in seg:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if (s - t0) / 1e6 > 9.5: break
        f.write(f"{(s - t0) / 1e3:9.1f} us  gap {(s - prev) / 1e3:6.1f}  dur {(e - s) / 1e3:7.1f}  {r['Kernel_Name'][:70]}\n")
        prev = e
PY
tail -3 $O/step_gaps4.txt
