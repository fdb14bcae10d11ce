"""Two and three ranks sharing the one GPU of the test box (gloo, point-to-point staged through
the host -- RCCL refuses several ranks per device): the slab-partitioned HIP path must reproduce
the single-mesh HIP solution of the same global problem."""
import os
import socket
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, n, port, outdir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from phifem_amd.distributed import SlabProblem
        prob = SlabProblem(n, rank=rank, world=world, device=0, rtol=1e-11)
        prob.setup()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            res = prob.step()
        lay = prob.lay
        plane = (n + 1) * (n + 1)
        nv = prob.mesh.nv
        w = prob.out.cpu().numpy()
        vplane = np.arange(nv) // plane + lay["k0"]
        owned = (vplane >= lay["P0"]) & (vplane < lay["P1"])
        gid = np.arange(nv) + lay["k0"] * plane
        np.savez(os.path.join(outdir, f"r{rank}.npz"), gid=gid[owned], u=w[:nv][owned],
                 p=w[nv:][owned], it=res["iterations"], relres=res["relres"],
                 n_owned=res["n_active_owned"])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_slabs_on_one_gpu_match_single_mesh(world, tmp_path):
    import torch
    import torch.multiprocessing as mp
    import phifem_amd as P
    from phifem_amd import _lib as L
    from phifem_amd.mesh_scripts import NodalFunction
    n = 20
    mp.spawn(_worker, args=(world, n, _free_port(), str(tmp_path)), nprocs=world, join=True)
    # the same global problem on one mesh
    mesh = P.create_box([-1.5, -1.5, -1.5 * world], [1.5, 1.5, 1.5 * world], [n, n, n * world])
    x = mesh.x
    zs = x[:, 2] / float(world)
    phi = x[:, 0] ** 2 + x[:, 1] ** 2 + zs ** 2 - 1.0
    uex = np.sin(x[:, 0]) * np.sin(x[:, 1]) * np.sin(zs)
    f = (2.0 + 1.0 / float(world * world)) * uex
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        P.compute_tags_measures(mesh, NodalFunction(phi), 1, box_mode=True, single_layer_cut=True)
    s = P.PhiFEMSolver(mesh)
    info = s.assemble(phi, f, uex)
    wref = s.solve(rtol=1e-11)
    nvg = mesh.nv
    u = np.full(nvg, np.nan)
    p = np.full(nvg, np.nan)
    n_owned = 0
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), f"r{r}.npz"))
        assert np.all(np.isnan(u[d["gid"]]))
        u[d["gid"]] = d["u"]
        p[d["gid"]] = d["p"]
        n_owned += int(d["n_owned"])
        assert d["relres"] <= 1e-11
    assert not np.any(np.isnan(u))
    assert n_owned == info["n_active"]
    scale = np.abs(wref).max()
    assert np.abs(u - wref[:nvg]).max() <= 1e-7 * scale
    assert np.abs(p - wref[nvg:]).max() <= 1e-7 * scale
