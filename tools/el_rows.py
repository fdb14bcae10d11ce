import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import phifem_amd as P
from phifem_amd.mesh_scripts import NodalFunction
warnings.simplefilter("ignore")
n = int(sys.argv[1])
mesh = P.create_box([-1.5] * 3, [1.5] * 3, [n] * 3)
x = mesh.x
phi = 1.0 - (x ** 2).sum(axis=1)
P.compute_tags_measures(mesh, NodalFunction(phi), 1, box_mode=True)
bf = mesh.boundary_facets
fv = np.array([[1, 2, 3], [0, 2, 3], [0, 1, 3], [0, 1, 2]])
bcv = np.unique(np.take_along_axis(mesh.cells[bf[:, 0]], fv[bf[:, 1]], axis=1))
f = np.stack([np.sin(x[:, 0]) + 0.2, np.cos(x[:, 1]), 0.5 * x[:, 2]], axis=1)
uD = 0.1 * np.stack([x[:, 0] * x[:, 1], np.sin(x[:, 2]), x[:, 0] - x[:, 1]], axis=1)
s = P.InterfaceElasticitySolver(mesh)
info = s.assemble(phi, f, uD, bcv)
rowptr, col, val, rhs, dof = s.export_csr()
lens = np.diff(rowptr)
nv = mesh.nv
blk = dof // nv
vert = dof % nv
ct = mesh.cell_tag_values()
cutv = np.zeros(nv, bool); cutv[mesh.cells[ct == 2].reshape(-1)] = True
# vertices of cells adjacent (by facet) to cut cells
f2c = mesh.f2c; c2f = mesh.c2f
cutc = ct == 2
adj = np.zeros(mesh.nc, bool)
fc = c2f[cutc].reshape(-1)
nb = f2c[fc].reshape(-1); nb = nb[nb >= 0]
adj[nb] = True
nearv = np.zeros(nv, bool); nearv[mesh.cells[adj].reshape(-1)] = True
print(info)
print("blocks present", np.unique(blk))
for name, mask in (("untouched", ~nearv[vert]), ("near-not-cut", nearv[vert] & ~cutv[vert]), ("cut", cutv[vert])):
    l = lens[mask]
    if l.size: print(name, "rows", l.size, "len max", l.max(), "p99", np.percentile(l, 99), "mean", l.mean(), "blocks", np.unique(blk[mask]))
print("rows > 64:", (lens > 64).sum(), " > 128:", (lens > 128).sum(), "> 256:", (lens > 256).sum(), "max", lens.max(), "of", lens.size)
