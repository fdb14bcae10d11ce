"""Idle gaps between consecutive kernels of a rocprofv3 --kernel-trace CSV (development aid).
usage: gaps.py trace.csv [min_gap_us]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last step only: find the last k_tag_cells launch
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("void k_tag_cells")]
i0 = starts[-1]
seg = rows[i0:]
t0 = int(seg[0]["Start_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
span = int(seg[-1]["End_Timestamp"]) - t0
print(f"last step: {len(seg)} kernels, span {span / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, idle {(span - busy) / 1e6:.3f} ms")
tot = 0.0
for a, b in zip(seg[:-1], seg[1:]):
    gap = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
    if gap > thr:
        tot += gap
        print(f"  +{(int(a['End_Timestamp']) - t0) / 1e6:8.3f} ms gap {gap:7.1f} us after {a['Kernel_Name'][:50]:50s} before {b['Kernel_Name'][:40]}")
print(f"gaps > {thr} us: {tot / 1e3:.3f} ms")
small = sum(max(0, (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3) for a, b in zip(seg[:-1], seg[1:]) if (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 <= thr)
print(f"gaps <= {thr} us: {small / 1e3:.3f} ms")
