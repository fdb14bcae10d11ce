"""Builds a committed PMC traffic record (profiles/rNN/pmc_<kernel>.json) from two rocprofv3 --pmc passes
(FETCH_SIZE and WRITE_SIZE, collected separately as MI355X_MICROARCH.md prescribes).
usage: pmc_record.py fetch_dir write_dir kernel_substring[+substring...] kernel_key algorithmic_bytes out.json [note]
Several substrings joined by '+': one logical launch is one dispatch of EACH kernel (the SpMV of a structured P2 system =
k_spmv_sell + k_spmv_p2s); their per-dispatch means are added.
Environment PMC_FETCH_FACTOR (default 2.0): bytes moved per FETCH_SIZE byte.  Calibration (tools/r04/pmc_calib.hip,
profiles/r04/pmc_calibration.txt; 2 GiB buffer): 8- and 16-byte-per-lane streams and 128-byte tile rows issue ONE
TCC_EA0_RDREQ per 128 bytes, which FETCH_SIZE tallies at 64 (TCC_EA0_RDREQ_32B = 0 everywhere) -> x2, as
MI355X_MICROARCH.md says for wide streaming reads; a gather of one 8-byte entry per 128-byte line also issues one request
per line and runs at the streaming rate only if each request moves the whole line (360 us for 1.68e7 lines: 5.96 TB/s at
128 bytes, 2.98 TB/s at 64) -> x2 as well: the L2 fills whole 128-byte lines."""
import csv
import glob
import json
import os
import sys

fetch_dir, write_dir, subs, key, alg, out = sys.argv[1:7]
note = sys.argv[7] if len(sys.argv) > 7 else ""
factor = float(os.environ.get("PMC_FETCH_FACTOR", "2.0"))


def mean_counter_one(d, counter, sub):
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


def mean_counter(d, counter):
    tot, cnt = 0.0, []
    for sub in subs.split("+"):
        m, n = mean_counter_one(d, counter, sub)
        if m is None:
            return None, 0
        tot += m
        cnt.append(n)
    return tot, min(cnt)


sub = subs
fetch_kb, nf = mean_counter(fetch_dir, "FETCH_SIZE")
write_kb, nw = mean_counter(write_dir, "WRITE_SIZE")
rec = {
    "kernel_key": key, "kernel_match": sub, "dispatches": {"FETCH_SIZE": nf, "WRITE_SIZE": nw},
    "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
    "correction": f"FETCH_SIZE x {factor:g}: every TCC_EA0_RDREQ moves a 128-byte line and is tallied at 64 bytes -- calibrated "
                  "for 8- / 16-byte-per-lane streams, 128-byte tile rows and one-entry-per-line gathers "
                  "(profiles/r04/pmc_calibration.txt; MI355X_MICROARCH.md, HBM); WRITE_SIZE exact; units KB",
    "algorithmic_bytes_per_launch": float(alg),
    "traffic_bytes_per_launch": (factor * fetch_kb + write_kb) * 1024.0 if fetch_kb is not None and write_kb is not None else None,
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) -- python3 bench.py "
              "--steps 1 --warmup 1 --no-cpu-baseline", "note": note,
}
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec))
