"""How redundant are the matrix values?  (development aid: sizing of a value-indexed SELL variant)
usage: value_stats.py [cubes]"""
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import phifem_amd  # noqa: E402,F401
from phifem_amd.distributed import SlabProblem  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
warnings.simplefilter("ignore")
p = SlabProblem(n)
p.setup()
p.step()
rowptr, col, val, rhs, dof = p.solver.export_csr()
nr = rowptr.size - 1
rows = np.repeat(np.arange(nr), np.diff(rowptr))
diag = np.zeros(nr)
d = col == rows
diag[rows[d]] = val[d]
sv = val / diag[col]                      # A D^-1, as stored in the SELL copy
nzm = sv != 0.0
print("rows", nr, "nnz", val.size, "nonzero", int(nzm.sum()))
print("distinct raw values", np.unique(val[nzm]).size, "distinct scaled values", np.unique(sv[nzm]).size)
lens = np.bincount(rows[nzm], minlength=nr)
print("row length histogram", np.bincount(lens)[:45])
# order rows as the SELL copy does: stable by decreasing length
order = np.argsort(-lens, kind="stable")
rank = np.empty(nr, dtype=np.int64)
rank[order] = np.arange(nr)
r2 = rank[rows[nzm]]
v2 = sv[nzm]
c2 = rank[col[nzm]]
for blk in (64, 256, 1024, 4096, 16384, 65536):
    b = r2 // blk
    o = np.lexsort((v2, b))
    bs, vs = b[o], v2[o]
    new = np.ones(vs.size, dtype=bool)
    new[1:] = (bs[1:] != bs[:-1]) | (vs[1:] != vs[:-1])
    per = np.bincount(bs[new])
    tot = np.bincount(bs)
    print(f"block {blk:6d} rows: distinct/block  median {np.median(per):8.0f}  p90 {np.percentile(per, 90):8.0f}  max {per.max():8d}"
          f"   blocks with <=256: {np.mean(per <= 256):.3f}  <=65536: {np.mean(per <= 65536):.3f}"
          f"   nnz share in <=256 blocks {tot[per <= 256].sum() / tot.sum():.3f}")
delta = c2 - r2
print("column delta |max|", np.abs(delta).max(), " share within int16", np.mean(np.abs(delta) < 32768),
      " within +-2^23", np.mean(np.abs(delta) < (1 << 23)))
for blk in (64,):
    b = r2 // blk
    far = np.abs(delta) >= 32768
    bad = np.zeros(b.max() + 1, dtype=bool)
    bad[b[far]] = True
    print("slices with every delta in int16:", 1.0 - bad.mean())

# column compressibility per slice: col = base[k] + lane + delta, delta in int8 / int16
lens2 = np.bincount(r2, minlength=nr)
o = np.lexsort((c2, r2))            # entries by (row, col); SELL keeps CSR (column-sorted) order per row
r3, c3 = r2[o], c2[o]
start = np.zeros(nr + 1, dtype=np.int64)
np.cumsum(lens2, out=start[1:])
kpos = np.arange(r3.size) - start[r3]
sl = r3 // 64
lane = r3 % 64
key = sl * 64 + kpos                # (slice, k) group (k < 64 in 3-D P1)
assert kpos.max() < 64
d = c3 - lane
gmin = np.full(key.max() + 1, np.iinfo(np.int64).max)
gmax = np.full(key.max() + 1, np.iinfo(np.int64).min)
np.minimum.at(gmin, key, d)
np.maximum.at(gmax, key, d)
rng = (gmax - gmin)[key]            # range of the group each entry belongs to
nsl = sl.max() + 1
worst = np.zeros(nsl, dtype=np.int64)
np.maximum.at(worst, sl, rng)
ent = np.bincount(sl, minlength=nsl)
for name, lim in (("int8", 255), ("int16", 65535)):
    okm = worst <= lim
    print(f"slices whose every (slice,k) column group spans <= {lim}: {okm.mean():.3f} of slices, "
          f"{ent[okm].sum() / ent.sum():.3f} of entries")
