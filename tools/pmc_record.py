"""Builds a committed PMC traffic record (profiles/rNN/pmc_<kernel>.json) from two rocprofv3 --pmc passes
(FETCH_SIZE and WRITE_SIZE, collected separately as MI355X_MICROARCH.md prescribes).
usage: pmc_record.py fetch_dir write_dir kernel_substring kernel_key algorithmic_bytes out.json [note]"""
import csv
import glob
import json
import os
import sys

fetch_dir, write_dir, sub, key, alg, out = sys.argv[1:7]
note = sys.argv[7] if len(sys.argv) > 7 else ""


def mean_counter(d, counter):
    tot, n = 0.0, 0
    per = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"] and r["Counter_Name"] == counter:
                per.setdefault(r["Dispatch_Id"], 0.0)
                per[r["Dispatch_Id"]] += float(r["Counter_Value"])
    for v in per.values():
        tot += v
        n += 1
    return (tot / n if n else None), n


fetch_kb, nf = mean_counter(fetch_dir, "FETCH_SIZE")
write_kb, nw = mean_counter(write_dir, "WRITE_SIZE")
rec = {
    "kernel_key": key, "kernel_match": sub, "dispatches": {"FETCH_SIZE": nf, "WRITE_SIZE": nw},
    "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
    "correction": "gfx950: FETCH_SIZE tallies 128-B requests at 64 B -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact "
                  "for wide streaming stores; units KB",
    "algorithmic_bytes_per_launch": float(alg),
    "traffic_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0 if fetch_kb is not None and write_kb is not None else None,
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) -- python3 bench.py "
              "--steps 1 --warmup 1 --no-cpu-baseline", "note": note,
}
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec))
