"""Sum rocprofv3 --pmc counter_collection CSVs per kernel name (development aid).
usage: pmc_summary.py dir [name-substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(int)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        if sub and sub not in k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            cnt[k] += 1
for k in acc:
    print(k, "dispatches", cnt[k])
    for c, v in sorted(acc[k].items()):
        print(f"   {c:28s} {v / cnt[k]:16.1f} per dispatch")
