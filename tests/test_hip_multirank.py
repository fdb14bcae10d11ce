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
    dz = np.maximum(np.abs(x[:, 2]) - 1.5 * (world - 1), 0.0)   # the capsule of SlabProblem
    phi = x[:, 0] ** 2 + x[:, 1] ** 2 + dz ** 2 - 1.0
    uex = np.sin(x[:, 0]) * np.sin(x[:, 1]) * np.sin(x[:, 2])
    f = 3.0 * uex
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


def _worker_el(rank, world, nxy, nzr, port, outdir, native=None, coarse=-1):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if native is not None:
        os.environ["PHIFEM_NATIVE_LOOP"] = "1" if native else "0"
        os.environ["PHX_RCCL_LIB"] = _FAKE
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from phifem_amd.distributed import ElasticitySlabProblem
        prob = ElasticitySlabProblem(nxy, nzr, rank=rank, world=world, device=0, rtol=1e-11, coarse=coarse)
        prob.setup()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            res = prob.step()
        lay = prob.lay
        plane = (nxy + 1) * (nxy + 1)
        nv = prob.mesh.nv
        w = prob.out.cpu().numpy().reshape(27, nv)
        vplane = np.arange(nv) // plane + lay["k0"]
        owned = (vplane >= lay["P0"]) & (vplane < lay["P1"])
        gid = np.arange(nv) + lay["k0"] * plane
        np.savez(os.path.join(outdir, f"e{rank}.npz"), gid=gid[owned], w=w[:, owned],
                 relres=res["relres"], n_owned=res["n_active_owned"], it=res["iterations"], precond=res["precond"],
                 path=prob.dk.path)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_elasticity_slabs_match_single_mesh(world, tmp_path):
    """configs[3] in miniature: the slab-partitioned 5-field elasticity system (27 component blocks
    per vertex, Dirichlet rows on the global box faces only) against the single-mesh solve."""
    import torch.multiprocessing as mp
    import phifem_amd as P
    from phifem_amd.mesh_scripts import NodalFunction
    nxy, nzr = 12, 12 // world
    mp.spawn(_worker_el, args=(world, nxy, nzr, _free_port(), str(tmp_path)), nprocs=world, join=True)
    mesh = P.create_box([-1.5] * 3, [1.5] * 3, [nxy, nxy, nzr * world])
    x = mesh.x
    phi = 1.0 - (x ** 2).sum(axis=1)
    f = np.stack([np.sin(x[:, 0]) + 0.2, np.cos(x[:, 1]), 0.5 * x[:, 2]], axis=1)
    uD = 0.1 * np.stack([x[:, 0] * x[:, 1], np.sin(x[:, 2]), x[:, 0] - x[:, 1]], axis=1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        P.compute_tags_measures(mesh, NodalFunction(phi), 1, box_mode=True)
    n1 = nxy + 1
    v = np.arange(mesh.nv)
    i, j, k = v % n1, (v // n1) % n1, v // (n1 * n1)
    bcv = np.flatnonzero((i == 0) | (i == nxy) | (j == 0) | (j == nxy) | (k == 0) | (k == nzr * world))
    s = P.InterfaceElasticitySolver(mesh)
    info = s.assemble(phi, f, uD, bcv)
    wref = s.solve(rtol=1e-11, max_iter=200000).reshape(27, mesh.nv)
    got = np.full((27, mesh.nv), np.nan)
    n_owned = 0
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), f"e{r}.npz"))
        got[:, d["gid"]] = d["w"]
        n_owned += int(d["n_owned"])
        assert d["relres"] <= 1e-11
    assert not np.any(np.isnan(got)) and n_owned == info["n_active"]
    assert np.abs(got - wref).max() <= 1e-6 * np.abs(wref).max()


# ---------------------------------------------------------------------------------------------------
# The NATIVE loop (phx_solve_distributed: pack kernel -> ncclSend/ncclRecv group -> unpack kernel,
# ncclAllReduce of the batched dot products, collective preconditioner vote) with more than one rank.
# RCCL refuses two ranks per device and the test box has one GPU, so the nine RCCL entry points the library
# binds with dlopen are served by the host-staged stand-in of tests/fake_rccl (PHX_RCCL_LIB).
# ---------------------------------------------------------------------------------------------------
_FAKE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fake_rccl", "libfake_rccl.so")


def _worker_native(rank, world, n, nxy, port, outdir, native, exact=True):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["PHIFEM_NATIVE_LOOP"] = "1" if native else "0"
    os.environ["PHIFEM_PRECOND_EXACT"] = "1" if exact else "0"
    os.environ["PHX_RCCL_LIB"] = _FAKE
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from phifem_amd.distributed import SlabProblem
        prob = SlabProblem(n, rank=rank, world=world, device=0, rtol=1e-11, nxy=nxy)
        prob.setup()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            res = prob.step()
        lay = prob.lay
        nx = prob.nxy
        plane = (nx + 1) * (nx + 1)
        nv = prob.mesh.nv
        w = prob.out.cpu().numpy()
        vplane = np.arange(nv) // plane + lay["k0"]
        owned = (vplane >= lay["P0"]) & (vplane < lay["P1"])
        gid = np.arange(nv) + lay["k0"] * plane
        np.savez(os.path.join(outdir, f"r{rank}.npz"), gid=gid[owned], u=w[:nv][owned],
                 p=w[nv:][owned], it=res["iterations"], relres=res["relres"],
                 n_owned=res["n_active_owned"], path=prob.dk.path, converged=res["converged"],
                 precond=res["precond"], exact=bool(res.get("precond_exact", False)), overlap=bool(prob.dk.overlap),
                 library=str(prob.dk.library))
    finally:
        dist.destroy_process_group()


def _single_mesh(n, world, nxy=None):
    import phifem_amd as P
    from phifem_amd.mesh_scripts import NodalFunction
    nx = nxy or n
    zext = 1.5 * world * n / nx if nxy else 1.5 * world
    mesh = P.create_box([-1.5, -1.5, -zext], [1.5, 1.5, zext], [nx, nx, n * world])
    x = mesh.x
    cyl = 0.0 if nxy else 1.5 * (world - 1)
    dz = np.maximum(np.abs(x[:, 2]) - cyl, 0.0)
    phi = x[:, 0] ** 2 + x[:, 1] ** 2 + dz ** 2 - 1.0
    uex = np.sin(x[:, 0]) * np.sin(x[:, 1]) * np.sin(x[:, 2])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        P.compute_tags_measures(mesh, NodalFunction(phi), 1, box_mode=True, single_layer_cut=True)
    s = P.PhiFEMSolver(mesh)
    info = s.assemble(phi, 3.0 * uex, uex)
    return mesh, info, s.solve(rtol=1e-11), s.stats


def _collect(tmp_path, world, nvg):
    u = np.full(nvg, np.nan)
    p = np.full(nvg, np.nan)
    rows = []
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), f"r{r}.npz"))
        assert np.all(np.isnan(u[d["gid"]]))
        u[d["gid"]] = d["u"]
        p[d["gid"]] = d["p"]
        rows.append(d)
    assert not np.any(np.isnan(u))
    return u, p, rows


@pytest.mark.skipif(not os.path.exists(_FAKE), reason="tests/fake_rccl/libfake_rccl.so not built (build())")
# five ranks: a test box lets six processes share its GPU, and this process is one of them (the driver's scaling run has
# eight ranks, one per GPU; eight slabs run on the CPU backend in tests/test_distributed_cpu.py)
@pytest.mark.parametrize("world,native,exact", [(2, True, True), (3, True, True), (2, False, True), (4, False, True),
                                                (2, True, False), (5, True, True)])
def test_native_loop_multi_rank_matches_single_mesh(world, native, exact, tmp_path):
    """exact = True: the slab-exact preconditioner (x / y sine transforms rank-local, tridiagonal z solves continued
    across the ranks through one all-gather): the SAME operator as the single-mesh preconditioner, so the
    iteration count must be the single-mesh count (+- the rounding of a different summation order).
    exact = False: rank-local block Jacobi over the slabs (more iterations)."""
    import torch.multiprocessing as mp
    n = 20
    mp.spawn(_worker_native, args=(world, n, None, _free_port(), str(tmp_path), native, exact), nprocs=world, join=True)
    mesh, info, wref, st = _single_mesh(n, world)
    u, p, rows = _collect(tmp_path, world, mesh.nv)
    assert all(str(d["path"]) == ("native" if native else "python") for d in rows), "the wrong loop ran"
    if native:
        # the overlapped exchange passed the self-test of every communicator, so the loop ran with the overlap on
        assert all(bool(d["overlap"]) for d in rows), "the overlapped halo exchange did not pass its self-test"
        # phx_comm_library names the file the collective entry points really come from (here: the stand-in)
        assert all(str(d["library"]).endswith("libfake_rccl.so") for d in rows), [str(d["library"]) for d in rows]
    assert all(bool(d["converged"]) and d["relres"] <= 1e-11 for d in rows)
    assert len({int(d["it"]) for d in rows}) == 1, "ranks stopped at different iterations"
    assert all(str(d["precond"]) == "box-dst" for d in rows)
    assert all(bool(d["exact"]) == exact for d in rows)
    its = int(rows[0]["it"])
    print(f"world {world} native {native} exact {exact}: {its} iterations, single mesh {st['iterations']}")
    if exact:
        # the same operator, assembled twice with atomics in different orders: at rtol 1e-11 BiCGStab's count moves by
        # about a tenth between two such solves of ONE mesh (84 / 94 seen for this case); a lost preconditioner doubles it
        assert abs(its - st["iterations"]) <= max(5, st["iterations"] // 5), (its, st["iterations"])
    assert sum(int(d["n_owned"]) for d in rows) == info["n_active"]
    scale = np.abs(wref).max()
    assert np.abs(u - wref[:mesh.nv]).max() <= 1e-7 * scale
    assert np.abs(p - wref[mesh.nv:]).max() <= 1e-7 * scale


@pytest.mark.skipif(not os.path.exists(_FAKE), reason="tests/fake_rccl/libfake_rccl.so not built (build())")
@pytest.mark.parametrize("world", [2, 3, 4])
def test_elasticity_coarse_correction_across_slabs(world, tmp_path):
    """The coarse correction of the elasticity solve on a partitioned box (native loop): the coarse lattice is the one
    of the global box, the coarse matrix and every coarse right-hand side are summed over the ranks -- the SAME
    preconditioner as on the single mesh, so the same solution in (about) the same number of iterations."""
    import torch.multiprocessing as mp
    import phifem_amd as P
    from phifem_amd.mesh_scripts import NodalFunction
    nxy, nzr, ratio = 24, 24 // world, 6
    mp.spawn(_worker_el, args=(world, nxy, nzr, _free_port(), str(tmp_path), True, ratio), nprocs=world, join=True)
    mesh = P.create_box([-1.5] * 3, [1.5] * 3, [nxy, nxy, nzr * world])
    x = mesh.x
    phi = 1.0 - (x ** 2).sum(axis=1)
    f = np.stack([np.sin(x[:, 0]) + 0.2, np.cos(x[:, 1]), 0.5 * x[:, 2]], axis=1)
    uD = 0.1 * np.stack([x[:, 0] * x[:, 1], np.sin(x[:, 2]), x[:, 0] - x[:, 1]], axis=1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        P.compute_tags_measures(mesh, NodalFunction(phi), 1, box_mode=True)
    n1 = nxy + 1
    v = np.arange(mesh.nv)
    i, j, k = v % n1, (v // n1) % n1, v // (n1 * n1)
    bcv = np.flatnonzero((i == 0) | (i == nxy) | (j == 0) | (j == nxy) | (k == 0) | (k == nzr * world))
    s = P.InterfaceElasticitySolver(mesh, deterministic=True, coarse=ratio)
    info = s.assemble(phi, f, uD, bcv)
    wref = s.solve(rtol=1e-11, max_iter=200000).reshape(27, mesh.nv)
    assert s.stats["precond"] == "vertex-block-jacobi+coarse"
    got = np.full((27, mesh.nv), np.nan)
    rows = []
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), f"e{r}.npz"))
        got[:, d["gid"]] = d["w"]
        rows.append(d)
        assert d["relres"] <= 1e-11 and str(d["path"]) == "native" and str(d["precond"]) == "vertex-block-jacobi+coarse", (d["path"], d["precond"])
    assert not np.any(np.isnan(got)) and sum(int(d["n_owned"]) for d in rows) == info["n_active"]
    assert np.abs(got - wref).max() <= 1e-6 * np.abs(wref).max()
    its = {int(d["it"]) for d in rows}
    print(f"world {world}: {its} iterations across the slabs, {s.stats['iterations']} on the single mesh")
    # (the vertex blocks alone need 2.5 x the iterations: a quarter is the noise of the assembly's summation order)
    assert len(its) == 1 and abs(its.pop() - s.stats["iterations"]) <= max(8, s.stats["iterations"] // 4)


@pytest.mark.skipif(not os.path.exists(_FAKE), reason="tests/fake_rccl/libfake_rccl.so not built (build())")
@pytest.mark.parametrize("native,world", [(False, 4), (True, 4), (True, 5)])
def test_empty_end_slabs_on_the_gpu(native, world, tmp_path):
    """BASELINE configs[4] in miniature: unit sphere, four (five) slabs, the end ranks do not touch
    the domain -> empty systems (PHX_OPT_ALLOW_EMPTY) that still join every collective, in the Python-driven
    and in the native loop."""
    import torch.multiprocessing as mp
    n, nxy = (10, 16) if world == 4 else (8, 16)
    mp.spawn(_worker_native, args=(world, n, nxy, _free_port(), str(tmp_path), native, True), nprocs=world, join=True)
    mesh, info, wref, st = _single_mesh(n, world, nxy)
    u, p, rows = _collect(tmp_path, world, mesh.nv)
    owned = [int(d["n_owned"]) for d in rows]
    assert owned[0] == 0 and owned[world - 1] == 0 and owned[world // 2 - 1] > 0 and owned[world // 2] > 0
    assert sum(owned) == info["n_active"]
    assert all(str(d["path"]) == ("native" if native else "python") for d in rows)
    assert all(bool(d["converged"]) for d in rows) and len({int(d["it"]) for d in rows}) == 1
    scale = np.abs(wref).max()
    assert np.abs(u - wref[:mesh.nv]).max() <= 1e-7 * scale
    assert np.abs(p - wref[mesh.nv:]).max() <= 1e-7 * scale


def _run_bench(extra_env, *args):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), *args], env=env,
                          capture_output=True, text=True, timeout=900)


def test_bench_gpus_2_launches_two_ranks():
    """`python bench.py --gpus 2` without a launcher starts two ranks itself (gloo rehearsal on the one GPU) and
    prints ONE line that says so."""
    import json
    r = _run_bench({"PHIFEM_DIST_BACKEND": "gloo"}, "--gpus", "2", "--cubes", "32", "--steps", "1",
                   "--warmup", "0", "--no-cpu-baseline")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["ranks_seen"] == 2 and d["config"]["converged"]
    assert d["config"]["parallelism"] == "slab2" and d["scaling"] == "weak" and d["value"] > 0


def test_bench_rccl_on_one_device_fails_loudly():
    """Two RCCL ranks need two GPUs: on the one-GPU box the run must fail, not report n_gpus = 1."""
    r = _run_bench({"PHIFEM_DIST_BACKEND": "nccl"}, "--gpus", "2", "--cubes", "32", "--steps", "1",
                   "--warmup", "0", "--no-cpu-baseline")
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "need 2 GPUs" in r.stderr


# ---------------------------------------------------------------------------------------------------
# A CALLER-SUPPLIED Kuhn box, partitioned (ArraySlabProblem): every rank gets the whole vertex array in the caller's
# (here: shuffled) numbering, recognises the lattice, works on its own slab and hands its owned part back in the caller's
# numbering.  Cross-section 10 x 14 (not square), box not centred.
# ---------------------------------------------------------------------------------------------------
def _caller_box():
    from oracle import meshgen
    lo, hi, n = [-1.4, -1.6, -1.5], [1.6, 1.5, 1.7], [10, 14, 12]
    x, _ = meshgen.create_box(lo, hi, n)
    perm = np.random.default_rng(11).permutation(x.shape[0])
    xc = np.ascontiguousarray(x[perm])                      # caller numbering: a shuffle of the lattice order
    phi = (xc[:, 0] / 1.1) ** 2 + (xc[:, 1] / 0.9) ** 2 + (xc[:, 2] / 1.2) ** 2 - 1.0
    uD = np.sin(xc[:, 0]) * np.cos(xc[:, 1]) + 0.3 * xc[:, 2]
    f = 2.0 * np.sin(xc[:, 0]) * np.cos(xc[:, 1])
    return lo, hi, n, perm, xc, phi, f, uD


def _worker_arrays(rank, world, port, outdir, native):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["PHIFEM_NATIVE_LOOP"] = "1" if native else "0"
    os.environ["PHX_RCCL_LIB"] = _FAKE
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from phifem_amd.distributed import ArraySlabProblem
        _, _, _, _, xc, phi, f, uD = _caller_box()
        prob = ArraySlabProblem(xc, phi, f, uD, rank=rank, world=world, device=0, rtol=1e-11)
        prob.setup()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            res = prob.step()
        v, u, p = prob.solution()
        np.savez(os.path.join(outdir, f"a{rank}.npz"), v=v, u=u, p=p, relres=res["relres"],
                 n_owned=res["n_active_owned"], path=prob.dk.path)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,native", [(2, True), (3, False)])
def test_caller_supplied_box_partitioned_matches_single_mesh(world, native, tmp_path):
    import torch.multiprocessing as mp
    import phifem_amd as P
    from phifem_amd.mesh_scripts import NodalFunction
    lo, hi, n, perm, xc, phi, f, uD = _caller_box()
    mp.spawn(_worker_arrays, args=(world, _free_port(), str(tmp_path), native), nprocs=world, join=True)
    # the same problem on one generated mesh (lattice order); lattice point g is caller vertex inv[g]
    mesh = P.create_box(lo, hi, n)
    inv = np.argsort(perm)                                  # xc[inv] = x: lattice point g is caller vertex inv[g]
    lat_phi, lat_f, lat_uD = phi[inv], f[inv], uD[inv]
    assert np.abs(mesh.x - xc[inv]).max() <= 1e-14
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        P.compute_tags_measures(mesh, NodalFunction(lat_phi), 1, box_mode=True, single_layer_cut=True)
    s = P.PhiFEMSolver(mesh)
    info = s.assemble(lat_phi, lat_f, lat_uD)
    wref = s.solve(rtol=1e-11)
    nv = mesh.nv
    u = np.full(nv, np.nan)
    p = np.full(nv, np.nan)
    n_owned = 0
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), f"a{r}.npz"))
        assert np.all(np.isnan(u[d["v"]]))                  # every caller vertex owned by exactly one rank
        u[d["v"]], p[d["v"]] = d["u"], d["p"]
        n_owned += int(d["n_owned"])
        assert d["relres"] <= 1e-11
        assert str(d["path"]) == ("native" if native else "python")
    assert not np.any(np.isnan(u)) and n_owned == info["n_active"]
    scale = np.abs(wref).max()
    assert np.abs(u[inv] - wref[:nv]).max() <= 1e-7 * scale
    assert np.abs(p[inv] - wref[nv:]).max() <= 1e-7 * scale


def test_caller_supplied_mesh_that_is_no_lattice_is_refused():
    from phifem_amd.distributed import detect_kuhn_lattice
    _, _, _, _, xc, _, _, _ = _caller_box()
    lo, hi, n, lat = detect_kuhn_lattice(xc)
    assert list(n) == [10, 14, 12] and np.unique(lat).size == xc.shape[0]
    bad = xc.copy()
    bad[5, 1] += 1e-9
    with pytest.raises(ValueError, match="uniform lattice"):
        detect_kuhn_lattice(bad)
    with pytest.raises(ValueError, match="fill a tensor lattice"):
        detect_kuhn_lattice(xc[:-1])
