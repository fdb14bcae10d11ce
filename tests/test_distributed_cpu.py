"""World-size-2 and -3 runs of the slab-partitioned solver on CPU (gloo): the multi-GPU host logic
(slab layout, ghost layers, ownership, halo lists, halo exchange, batched scalar all-reduce) with
the numpy stand-in of the phase kernels.  The distributed solution must equal the single-mesh
oracle solution."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cpu_backend import CpuBackend, assemble_local
from oracle import assembly as OA
from phifem_amd.dist_solver import DistributedSolver
from phifem_amd.distributed import GHOST_LAYERS, slab_layout


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def worker(rank, world, n, port, outdir, sphere=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lay = slab_layout(n, rank, world)
        x, topo, cv, A, b, act = assemble_local(n, world, lay["k0"], lay["k1"], has_exterior=True, sphere=sphere)
        be = CpuBackend(A, b, act, topo.nv)
        plane = (n + 1) * (n + 1)
        ds = DistributedSolver(be, dist, torch, rank, world, plane, lay["k0"], lay["P0"], lay["P1"],
                               lay["k1"] - lay["k0"] + 1, rtol=1e-11, max_iter=4000, check_every=4)
        out = torch.zeros(2 * topo.nv, dtype=torch.float64)
        st = ds.solve(out)
        # owned entries in global vertex numbering
        w = out.numpy()
        nv = topo.nv
        vplane = np.arange(nv) // plane + lay["k0"]
        owned = (vplane >= lay["P0"]) & (vplane < lay["P1"])
        gid = np.arange(nv) + lay["k0"] * plane
        np.savez(os.path.join(outdir, f"r{rank}.npz"), gid=gid[owned], u=w[:nv][owned],
                 p=w[nv:][owned], it=st["iterations"], relres=st["relres"], n_owned=st["n_owned"])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_slab_solver_matches_single_mesh(world, tmp_path):
    """world = 8: the rank count of the driver's scaling run -- eight slabs of four cube layers, so
    that the ghost layers of a slab reach exactly one neighbour."""
    n = 8 if world < 8 else 4
    port = free_port()
    mp.spawn(worker, args=(world, n, port, str(tmp_path)), nprocs=world, join=True)
    # single-mesh reference
    x, topo, cv, A, b, act = assemble_local(n, world, 0, n * world)
    wref = OA.solve_direct(A, b, act)
    nvg = topo.nv
    u = np.full(nvg, np.nan)
    p = np.full(nvg, np.nan)
    n_owned = 0
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), f"r{r}.npz"))
        assert np.all(np.isnan(u[d["gid"]])), "a vertex is owned by two ranks"
        u[d["gid"]] = d["u"]
        p[d["gid"]] = d["p"]
        n_owned += int(d["n_owned"])
        assert d["relres"] <= 1e-11 and d["it"] > 0
    assert not np.any(np.isnan(u)), "a vertex is owned by no rank"
    assert n_owned == int(act.sum()), "owned active DoFs do not add up to the global system"
    scale = np.abs(wref).max()
    assert np.abs(u - wref[:nvg]).max() <= 1e-7 * scale
    assert np.abs(p - wref[nvg:]).max() <= 1e-7 * scale


def test_empty_end_slabs_join_the_collectives(tmp_path):
    """BASELINE configs[4] in miniature: the unit sphere in a box four slabs tall.  Ranks 0
    and 3 own no active DoF (their slab + ghost layers do not touch the sphere); they must run the same
    collectives with zero contributions and empty halos, and the two middle ranks must reproduce the
    single-mesh solution."""
    n, world = 6, 4
    port = free_port()
    mp.spawn(worker, args=(world, n, port, str(tmp_path), True), nprocs=world, join=True)
    x, topo, cv, A, b, act = assemble_local(n, world, 0, n * world, sphere=True)
    wref = OA.solve_direct(A, b, act)
    nvg = topo.nv
    u = np.zeros(nvg)
    owned = []
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), f"r{r}.npz"))
        u[d["gid"]] = d["u"]
        owned.append(int(d["n_owned"]))
        assert d["relres"] <= 1e-11 and d["it"] > 0    # every rank saw the same (all-reduced) residual
    assert owned[0] == 0 and owned[3] == 0 and owned[1] > 0 and owned[2] > 0
    assert sum(owned) == int(act.sum())
    assert np.abs(u - wref[:nvg]).max() <= 1e-7 * np.abs(wref).max()


def test_eight_slabs_with_empty_ends(tmp_path):
    """The layout of `bench.py --config5 --gpus 8` in miniature: unit sphere, eight slabs, the end slabs (here ranks 0-1
    and 6-7) own nothing and still take part in every collective."""  # noqa
    n, world = 6, 8
    port = free_port()
    mp.spawn(worker, args=(world, n, port, str(tmp_path), True), nprocs=world, join=True)
    x, topo, cv, A, b, act = assemble_local(n, world, 0, n * world, sphere=True)
    wref = OA.solve_direct(A, b, act)
    nvg = topo.nv
    u = np.zeros(nvg)
    owned = []
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), f"r{r}.npz"))
        u[d["gid"]] = d["u"]
        owned.append(int(d["n_owned"]))
        assert d["relres"] <= 1e-11 and d["it"] > 0
    assert owned[0] == 0 and owned[7] == 0 and max(owned) > 0
    assert sum(owned) == int(act.sum())
    assert np.abs(u - wref[:nvg]).max() <= 1e-7 * np.abs(wref).max()


def test_slab_layout_covers_the_box():
    for world in (1, 2, 3, 8):
        n = 16
        planes = []
        for r in range(world):
            lay = slab_layout(n, r, world)
            assert lay["k0"] == max(0, r * n - GHOST_LAYERS)
            assert lay["k1"] == min(n * world, (r + 1) * n + GHOST_LAYERS)
            planes += list(range(lay["P0"], lay["P1"]))
        assert planes == list(range(n * world + 1))


def test_lattice_of_a_caller_supplied_box_is_recognised_in_any_vertex_order():
    """Host half of ArraySlabProblem (the partitioned solve itself: tests/test_hip_multirank.py): the tensor lattice behind
    shuffled vertices, the lattice index of every vertex, and what is refused."""
    from oracle import meshgen
    from phifem_amd.distributed import detect_kuhn_lattice
    lo, hi, n = [-1.4, -1.6, -1.5], [1.6, 1.5, 1.7], [5, 7, 6]
    x, _ = meshgen.create_box(lo, hi, n)
    perm = np.random.default_rng(3).permutation(x.shape[0])
    lo2, hi2, n2, lat = detect_kuhn_lattice(x[perm])
    assert np.allclose(lo2, lo, atol=0, rtol=1e-15) and np.allclose(hi2, hi, atol=0, rtol=1e-15) and list(n2) == n
    assert np.array_equal(lat, perm)                       # generated order IS lattice order: x[perm][v] is point perm[v]
    bad = x[perm].copy()
    bad[4, 2] += 1e-9
    with pytest.raises(ValueError, match="uniform lattice"):
        detect_kuhn_lattice(bad)
    with pytest.raises(ValueError, match="fill a tensor lattice"):
        detect_kuhn_lattice(x[perm][1:])
    with pytest.raises(ValueError, match="z-slabs"):
        detect_kuhn_lattice(x[:, :2])
    # every slab's lattice points map to distinct caller vertices and the slabs cover the box
    world, nz = 3, n[2]
    lat2v = np.empty(lat.size, dtype=np.int64)
    lat2v[lat] = np.arange(lat.size)
    plane = (n[0] + 1) * (n[1] + 1)
    owned = []
    for r in range(world):
        lay = slab_layout(nz // world, r, world)
        owned.append(lat2v[lay["P0"] * plane:lay["P1"] * plane])
    allv = np.concatenate(owned)
    assert allv.size == x.shape[0] and np.unique(allv).size == x.shape[0]
