"""Ad-hoc stage timing of tag -> assemble -> solve on growing boxes (development aid)."""
import sys, time, warnings
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import phifem_amd as P
from phifem_amd import _lib as L
from phifem_amd.mesh_scripts import NodalFunction
import ctypes as C
warnings.simplefilter('ignore')
for n in [int(a) for a in sys.argv[1:]]:
    t0=time.time()
    mesh=P.create_box([-1.5]*3,[1.5]*3,[n]*3)
    mesh.synchronize(); t_mesh=time.time()-t0
    dev=torch.device('cuda:0')
    x=torch.empty((mesh.nv,3),dtype=torch.float64,device=dev)
    L.check(L.lib.phx_mesh_get_array(mesh._h, L.ARR_COORDS, C.c_void_p(x.data_ptr()), L.DEVICE))
    phi=(x**2).sum(1)-1.0
    uex=torch.sin(x[:,0])*torch.sin(x[:,1])*torch.sin(x[:,2])
    f=3*uex
    torch.cuda.synchronize()
    for rep in range(2):
        t0=time.time()
        ct=P.mesh_scripts._tag_cells(mesh, NodalFunction(phi), 1, True)
        P.mesh_scripts._tag_facets(mesh, ct, 1)
        t1=time.time()
        s=P.PhiFEMSolver(mesh)
        info=s.assemble(phi,f,uex)
        t2=time.time()
        out=torch.empty(2*mesh.nv,dtype=torch.float64,device=dev)
        s.solve(rtol=1e-8,max_iter=20000,out=out)
        torch.cuda.synchronize()
        t3=time.time()
        print(n,'mesh %.3f tag %.4f asm %.4f solve %.4f'%(t_mesh,t1-t0,t2-t1,t3-t2), info, s.stats, mesh.timings())
        print('   DoF/s', info['n_active']/(t3-t0), 'spmv', s.spmv_bench(50))
    del s, mesh
