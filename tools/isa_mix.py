"""Static instruction mix of one kernel of a device assembly listing (development aid; no GPU needed).
usage: isa_mix.py listing.s mangled-name-substring [--dump]
Make the listing with   hipcc <flags of the Makefile> --cuda-device-only -S phx_solve.hip -o listing.s"""
import collections
import re
import sys

path, key = sys.argv[1], sys.argv[2]
dump = "--dump" in sys.argv
lines = open(path).read().split("\n")
start = None
for i, l in enumerate(lines):
    if re.match(r"^[A-Za-z_][\w$.]*:", l) and key in l.split(":")[0]:
        start = i
        break
if start is None:
    sys.exit("kernel not found")
body = []
for l in lines[start + 1:]:
    if l.startswith("\t.end_amdhsa_kernel") or l.startswith("\t.section") or "s_endpgm" in l and False:
        break
    if l.strip().startswith(".Lfunc_end"):
        break
    body.append(l)
ops = collections.Counter()
cls = collections.Counter()
for l in body:
    s = l.strip()
    if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
        continue
    op = s.split()[0]
    ops[op] += 1
    if op.startswith("v_") and ("f64" in op or "f32" in op and "cvt" not in op):
        c = "valu_fp"
    elif op.startswith("v_cvt"):
        c = "valu_cvt"
    elif op.startswith("v_"):
        c = "valu_int/other"
    elif op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier"):
        c = "wait"
    elif op.startswith("s_"):
        c = "salu"
    elif op.startswith("ds_"):
        c = "lds"
    elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"):
        c = "vmem"
    else:
        c = "other"
    cls[c] += 1
tot = sum(cls.values())
print(lines[start])
print("total", tot, dict(cls))
for op, n in ops.most_common(45):
    print(f"  {n:5d} {op}")
for l in lines[start:start + 4000]:
    m = re.search(r"\.(vgpr_count|sgpr_count|lds_size|scratch)", l)
if dump:
    print("\n".join(body))
