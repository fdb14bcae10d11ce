"""GPU parity of assembly (a9, a10) and solve (a11) through the C ABI against the numpy oracle.

Floating point: the HIP path sums element contributions with f64 atomics in arbitrary order and
uses FMA, the oracle sums in COO order without FMA.  Tolerances (relative to the largest matrix
/ vector entry): matrix and rhs 1e-12; solution 1e-6 at solver rtol 1e-10."""
import warnings

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import assembly as OA
from oracle import tagging as T
from oracle.topology import Topology

pytestmark = pytest.mark.gpu
MAT_TOL = 1e-12
SOL_TOL = 1e-6


@pytest.fixture(scope="module")
def P():
    import phifem_amd
    assert phifem_amd._lib.device_count() > 0
    return phifem_amd


def setup_problem(P, d, n, box_mode=True, centre=None, lin=False):
    from phifem_amd.mesh_scripts import NodalFunction
    lo, hi = [-1.5] * d, [1.5] * d
    mesh = P.create_box(lo, hi, [n] * d)
    x = mesh.x
    centre = np.zeros(d) if centre is None else np.asarray(centre)
    phi = ((x - centre) ** 2).sum(axis=1) - 1.0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ct, ft, sub, meas, maps = P.compute_tags_measures(
            mesh, NodalFunction(phi), 1, box_mode=box_mode, single_layer_cut=True)
    work = mesh if box_mode else sub
    xw = work.x
    phiw = ((xw - centre) ** 2).sum(axis=1) - 1.0
    if lin:
        uex = xw @ np.arange(1, d + 1) + 0.5
        f = np.zeros(work.nv)
    else:
        uex = np.prod(np.sin(xw), axis=1)
        f = d * uex
    ctype = "triangle" if d == 2 else "tetrahedron"
    topo = Topology(ctype, work.cells.astype(np.int64), work.nv)
    topo.c2f = work.c2f.astype(np.int64)
    topo.f2c = work.f2c.astype(np.int64)
    topo.nf = work.nf
    ds = meas(100) if box_mode else work.boundary_facets.reshape(-1)
    A, b, act = OA.assemble_poisson_wd(topo, xw, work.cell_tag_values(), work.facet_tag_values(),
                                       ds, phiw, f, uex)
    return work, phiw, f, uex, A, b, act


def hip_matrix(solver):
    rowptr, col, val, rhs, dof = solver.export_csr()
    n = rowptr.size - 1
    return sp.csr_matrix((val, col, rowptr), shape=(n, n)), rhs, dof


@pytest.mark.parametrize("d,n,box", [(2, 24, True), (2, 40, True), (3, 8, True), (3, 12, True),
                                     (2, 24, False), (3, 8, False)])
def test_matrix_and_rhs_vs_oracle(P, d, n, box):
    work, phi, f, uex, A, b, act = setup_problem(P, d, n, box_mode=box, centre=[0.03, -0.02, 0.01][:d])
    s = P.PhiFEMSolver(work)
    info = s.assemble(phi, f, uex)
    H, rhs, dof = hip_matrix(s)
    idx = np.flatnonzero(act)
    assert info["n_active"] == idx.size
    assert np.array_equal(dof, idx), "active DoF numbering differs"
    Ao = A[idx][:, idx].tocsr()
    Ao.sort_indices()
    # same structural pattern (sorted columns), same values
    assert np.array_equal(H.indptr, Ao.indptr)
    assert np.array_equal(H.indices, Ao.indices)
    scale = np.abs(Ao.data).max()
    assert np.abs(H.data - Ao.data).max() <= MAT_TOL * scale
    assert np.abs(rhs - b[idx]).max() <= MAT_TOL * max(np.abs(b).max(), 1e-300)
    # SpMV kernel (SELL copy without the explicit zeros) against the CSR product
    rng = np.random.default_rng(0)
    xv = rng.standard_normal(idx.size)
    y = s.spmv(xv)
    yo = Ao @ xv
    assert np.abs(y - yo).max() <= 1e-12 * np.abs(yo).max()
    # sell_nnz: non-zeros one SpMV applies; sell_padded_nnz: SELL entries stored (the rows applied from the
    # stencil of a structured system store nothing)
    per_row = 5 if d == 2 else 7
    assert info["sell_nnz"] <= info["nnz"]
    assert info["sell_padded_nnz"] + per_row * info["stencil_rows"] >= info["sell_nnz"]


@pytest.mark.parametrize("d,n", [(2, 32), (3, 10)])
def test_patch_test_linear_solution(P, d, n):
    """f = 0, u_D = u linear: P1 reproduces u exactly and p = 0 (consistency of main.py:112-151)."""
    work, phi, f, uex, A, b, act = setup_problem(P, d, n, lin=True)
    s = P.PhiFEMSolver(work)
    s.assemble(phi, f, uex)
    w = s.solve(rtol=1e-13, max_iter=5000)
    u, p = s.split(w)
    ua = act[:work.nv]
    assert np.abs(u[ua] - uex[ua]).max() < 1e-8
    assert np.abs(p).max() < 1e-6
    assert np.all(w[~act] == 0.0)  # MUMPS ICNTL(24)=1 semantics: null-space components at zero


@pytest.mark.parametrize("d,n,box", [(2, 48, True), (3, 12, True), (3, 16, True), (2, 32, False)])
def test_solve_vs_direct(P, d, n, box):
    work, phi, f, uex, A, b, act = setup_problem(P, d, n, box_mode=box)
    s = P.PhiFEMSolver(work)
    s.assemble(phi, f, uex)
    w = s.solve(rtol=1e-10)
    assert s.stats["relres"] <= 1e-10 and s.stats["iterations"] > 0
    wo = OA.solve_direct(A, b, act)
    assert np.abs(w - wo).max() <= SOL_TOL * np.abs(wo).max()
    assert np.all(w[~act] == 0.0)
    # true residual of the returned vector on the oracle's matrix
    r = A @ w - b
    assert np.linalg.norm(r[act]) <= 1e-8 * np.linalg.norm(b)


def test_device_resident_inputs_and_outputs(P):
    import torch
    work, phi, f, uex, A, b, act = setup_problem(P, 3, 10)
    s = P.PhiFEMSolver(work)
    dev = torch.device("cuda:0")
    tphi, tf, tu = (torch.from_numpy(a).to(dev) for a in (phi, f, uex))
    s.assemble(tphi, tf, tu)
    out = torch.empty(2 * work.nv, dtype=torch.float64, device=dev)
    s.solve(rtol=1e-10, out=out)
    torch.cuda.synchronize()
    wo = OA.solve_direct(A, b, act)
    assert np.abs(out.cpu().numpy() - wo).max() <= SOL_TOL * np.abs(wo).max()


def test_convergence_rate_2d(P):
    """L2-type error at the inside vertices falls ~4x per halving of h (P1, smooth solution),
    mirroring the slope check of demo/interface-elasticity/main.py:392-400."""
    errs = []
    for n in (32, 64):
        work, phi, f, uex, A, b, act = setup_problem(P, 2, n)
        s = P.PhiFEMSolver(work)
        s.assemble(phi, f, uex)
        w = s.solve(rtol=1e-11)
        inside = np.unique(work.cells[work.cell_tag_values() == 1])
        errs.append(np.sqrt(np.mean((w[:work.nv][inside] - uex[inside]) ** 2)))
    assert errs[0] / errs[1] > 3.0


def test_phase_api_equals_native_solve(P):
    """The externally driven iteration (multi-GPU driver, world = 1: no halo, no all-reduce) must
    reproduce the native solver loop on the same system."""
    import torch
    from phifem_amd.dist_solver import DistributedSolver, HipBackend
    from phifem_amd.distributed import SlabProblem
    prob = SlabProblem(16, rank=0, world=1, device=0, rtol=1e-10)
    prob.setup()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = prob.step()
    native = prob.out.clone()
    dev = torch.device("cuda", 0)
    be = HipBackend(prob.solver, dev)
    be.use_current_stream()
    lay = prob.lay
    ds = DistributedSolver(be, None, torch, 0, 1, (prob.n + 1) ** 2, lay["k0"], lay["P0"], lay["P1"],
                           lay["k1"] - lay["k0"] + 1, rtol=1e-10)
    assert ds.n_owned == prob.solver.info()["n_active"] and not ds.halos
    out = torch.zeros_like(native)
    st = ds.solve(out, profile_spmv=False)
    # atomic accumulation order moves the convergence point by a check interval at most
    assert abs(st["iterations"] - res["iterations"]) <= 16
    assert torch.allclose(out, native, rtol=0, atol=1e-9 * float(native.abs().max()))


def test_config0_flower_2d(P):
    """BASELINE configs[0]: 2-D weak-Dirichlet Poisson, 'flower' level-set, P1, 128x128 background
    mesh on [-4.5,4.5]^2, detection degree 1, box mode, single-layer cut, gamma = sigma = 1
    (demo/weak-dirichlet/flower/main.py:42-62): HIP vs the CPU oracle, stage by stage."""
    import flower_data as F
    from oracle import meshgen
    from phifem_amd.mesh_scripts import NodalFunction
    n = 128
    mesh = P.create_rectangle([[-4.5, -4.5], [4.5, 4.5]], [n, n])
    x = mesh.x
    xo, co = meshgen.create_box([-4.5, -4.5], [4.5, 4.5], [n, n])
    assert np.array_equal(x, xo) and np.array_equal(mesh.cells, co)
    assert (mesh.nc, mesh.nv, mesh.nf) == (32768, 16641, 49408)   # SURVEY 8(a) sizes
    det = F.detection_levelset(x.T)
    phi = F.levelset(x.T)
    f = F.source_term(x.T)
    ud = F.dirichlet_data(x.T)
    topo = Topology("triangle", co, x.shape[0])
    topo.c2f, topo.f2c, topo.nf = mesh.c2f.astype(np.int64), mesh.f2c.astype(np.int64), mesh.nf
    topo.boundary_facets = np.flatnonzero(topo.f2c[:, 1] < 0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        hc, hf, _, hmeas, _ = P.compute_tags_measures(mesh, NodalFunction(det), 1, box_mode=True,
                                                      single_layer_cut=True)
        oc, of, _, omeas, _, _ = T.compute_tags_measures("triangle", x, topo, T.NodalP1(det), 1,
                                                         box_mode=True, single_layer_cut=True)
    assert np.array_equal(hc.values, oc.values) and np.array_equal(hf.values, of.values)
    assert np.array_equal(hmeas(100), omeas(100)) and np.array_equal(hmeas(101), omeas(101))
    assert set(np.unique(hc.values)) == {1, 2, 3}
    A, b, act = OA.assemble_poisson_wd(topo, x, mesh.cell_tag_values(), mesh.facet_tag_values(),
                                       hmeas(100), phi, f, ud)
    s = P.PhiFEMSolver(mesh, pen_coef=1.0, stab_coef=1.0)
    s.assemble(phi, f, ud)
    H, rhs, dof = hip_matrix(s)
    idx = np.flatnonzero(act)
    Ao = A[idx][:, idx].tocsr()
    Ao.sort_indices()
    assert np.array_equal(dof, idx) and np.array_equal(H.indices, Ao.indices)
    assert np.abs(H.data - Ao.data).max() <= MAT_TOL * np.abs(Ao.data).max()
    assert np.abs(rhs - b[idx]).max() <= MAT_TOL * np.abs(b).max()
    w = s.solve(rtol=1e-11)
    wo = OA.solve_direct(A, b, act)
    assert np.abs(w - wo).max() <= SOL_TOL * np.abs(wo).max()
    assert w[:mesh.nv].max() > 0.0 and np.all(w[~act] == 0.0)


def test_native_rccl_loop_world1(P):
    """phx_solve_distributed with a one-rank RCCL communicator (no halo; the all-reduces are
    skipped for one rank) reproduces the native single-GPU solve: binds librccl at run time."""
    import ctypes as C
    import torch
    from phifem_amd import _lib as L
    from phifem_amd.dist_solver import DistributedSolver, HipBackend
    from phifem_amd.distributed import SlabProblem
    prob = SlabProblem(16, rank=0, world=1, device=0, rtol=1e-10)
    prob.setup()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = prob.step()
    native = prob.out.clone()
    uid = (C.c_ubyte * 128)()
    L.check(L.lib.phx_comm_unique_id(uid))
    comm = C.c_void_p()
    L.check(L.lib.phx_comm_create(1, 0, uid, 0, C.byref(comm)))
    dev = torch.device("cuda", 0)
    be = HipBackend(prob.solver, dev)
    be.use_current_stream()
    lay = prob.lay
    ds = DistributedSolver(be, None, torch, 0, 1, (prob.n + 1) ** 2, lay["k0"], lay["P0"], lay["P1"],
                           lay["k1"] - lay["k0"] + 1, rtol=1e-10)
    out = torch.zeros_like(native)
    st = (C.c_double * 8)()
    peers, counts, idx = (C.c_int * 1)(), (C.c_int64 * 1)(), (C.c_void_p * 1)()
    L.check(L.lib.phx_solve_distributed(be.sys, comm, 0, peers, counts, idx, 1e-10, 20000,
                                        C.c_void_p(out.data_ptr()), L.DEVICE, st))
    torch.cuda.synchronize()
    assert abs(int(st[0]) - res["iterations"]) <= 16
    assert torch.allclose(out, native, rtol=0, atol=1e-9 * float(native.abs().max()))
    # halo code path (pack kernel -> ncclSend/ncclRecv -> unpack kernel) with the rank as its own
    # neighbour: entries at `recv` positions must come back as the entries at `send` positions
    n = be.n
    vec = torch.arange(n, dtype=torch.float64, device=dev)
    k = n // 4
    assert 3 * k + 2 < n
    send = torch.arange(0, k, dtype=torch.int64, device=dev) * 3 + 1
    recv = torch.arange(0, k, dtype=torch.int64, device=dev) * 3 + 2
    peers[0] = 0
    counts2 = (C.c_int64 * 2)(k, k)
    idx2 = (C.c_void_p * 2)(send.data_ptr(), recv.data_ptr())
    L.check(L.lib.phx_halo_selftest(be.sys, comm, 1, peers, counts2, idx2, C.c_void_p(vec.data_ptr())))
    torch.cuda.synchronize()
    assert torch.equal(vec[recv], send.to(torch.float64))
    assert torch.equal(vec[send], send.to(torch.float64))
    L.check(L.lib.phx_comm_destroy(comm))


@pytest.mark.parametrize("d,n", [(2, 64), (3, 32), (3, 20), (3, 27), (2, 100), (2, 96)])
def test_value_indexed_slices_are_bit_identical(P, d, n):
    """SELL slices stored as dictionary + byte codes (PHX_OPT_SPMV_VALUE_INDEX) must give the very same
    products as raw doubles.  Two assemblies differ in the last bits on rows that receive atomic
    scatter (cut cells, facets), so bit equality is asserted on the rows whose CSR values are
    bit-identical in both -- the gathered interior rows, which are the ones that get indexed."""
    from phifem_amd import _lib as L
    work, phi, f, uex, A, b, act = setup_problem(P, d, n)
    # every row stored: the interior rows this test is about are not stored at all by structured systems
    L.check(L.lib.phx_set_option(work._h, L.OPT_STRUCTURED, 0))
    rng = np.random.default_rng(5)
    ys, sols, infos, mats = [], [], [], []
    for flag in (1, 0):
        L.check(L.lib.phx_set_option(work._h, L.OPT_SPMV_VALUE_INDEX, flag))
        s = P.PhiFEMSolver(work)
        info = s.assemble(phi, f, uex)
        infos.append(info)
        if flag:
            x = rng.standard_normal(info["n_active"])
        ys.append(s.spmv(x))
        sols.append(s.solve(rtol=1e-10))
        mats.append(hip_matrix(s)[0])
    L.check(L.lib.phx_set_option(work._h, L.OPT_SPMV_VALUE_INDEX, 1))
    print(d, n, {k: infos[0][k] for k in ("n_slices", "indexed_slices", "indexed_slices_lds")})
    assert infos[1]["indexed_slices"] == 0
    # box rows are built from the exact lattice spacing: interior rows repeat bit for bit for any n
    assert infos[0]["indexed_slices"] > 0
    assert infos[0]["spmv_matrix_bytes"] < infos[1]["spmv_matrix_bytes"]
    M0, M1 = mats
    assert np.array_equal(M0.indptr, M1.indptr) and np.array_equal(M0.indices, M1.indices)
    differs = np.add.reduceat((M0.data != M1.data).astype(np.int64), M0.indptr[:-1]) > 0
    same = ~differs
    assert same.mean() > 0.2
    assert np.array_equal(ys[0][same], ys[1][same])
    scale = np.abs(M0).max() * np.abs(x).max()
    assert np.abs(ys[0] - ys[1]).max() <= 1e-13 * scale
    assert np.abs(ys[0] - M0 @ x).max() <= 1e-13 * scale
    assert np.abs(sols[0] - sols[1]).max() <= 1e-8 * np.abs(sols[1]).max()


@pytest.mark.parametrize("d,n", [(3, 20), (3, 27), (2, 64), (2, 100), (3, 12)])
def test_structured_interior_rows_match_the_stored_matrix(P, d, n):
    """Structured systems (default on Kuhn boxes): the translation-invariant interior rows are applied from a
    stencil over runs of consecutive rows and never stored.  The product y = A x of that operator must equal
    the CSR matrix (exported from a re-assembly with PHX_OPT_EXPORT_CSR) applied to the same x, the stored-everything
    system (PHX_OPT_STRUCTURED = 0) must give the same product and solution, and the CSR itself still matches the
    oracle.  Floating-point tolerance: 1e-13 of |A| |x| (different summation order within a row)."""
    from phifem_amd import _lib as L
    work, phi, f, uex, A, b, act = setup_problem(P, d, n)
    rng = np.random.default_rng(17)
    s = P.PhiFEMSolver(work)
    info = s.assemble(phi, f, uex)
    assert info["stencil_rows"] > 0 and info["stencil_runs"] > 0, info
    assert info["stencil_rows"] + info["n_slices"] * 16 >= info["n_active"]   # stored rows: SELL slices of 16
    x = rng.standard_normal(info["n_active"])
    y = s.spmv(x)
    M, rhs, dof = hip_matrix(s)          # lazy CSR: re-assembled with the export option
    assert M.nnz == info["nnz"]          # the structural count is known without the CSR
    scale = np.abs(M).max() * np.abs(x).max()
    assert np.abs(y - M @ x).max() <= 1e-13 * scale
    w = s.solve(rtol=1e-11)
    assert s.stats["converged"] and s.stats["precond"] == "box-dst"
    # the same problem with every row stored
    L.check(L.lib.phx_set_option(work._h, L.OPT_STRUCTURED, 0))
    try:
        s2 = P.PhiFEMSolver(work)
        info2 = s2.assemble(phi, f, uex)
        assert info2["stencil_rows"] == 0 and info2["nnz"] == info["nnz"]
        y2 = s2.spmv(x)
        w2 = s2.solve(rtol=1e-11)
    finally:
        L.check(L.lib.phx_set_option(work._h, L.OPT_STRUCTURED, 1))
    assert np.abs(y - y2).max() <= 1e-13 * scale
    assert np.abs(w - w2).max() <= 1e-8 * np.abs(w2).max()
    # and without the box preconditioner the unscaled u columns get their Jacobi scaling as a preconditioner
    L.check(L.lib.phx_set_option(work._h, L.OPT_PRECOND, 0))
    try:
        s3 = P.PhiFEMSolver(work)
        s3.assemble(phi, f, uex)
        w3 = s3.solve(rtol=1e-11, max_iter=50000)
        assert s3.stats["converged"] and s3.stats["precond"] == "jacobi"
    finally:
        L.check(L.lib.phx_set_option(work._h, L.OPT_PRECOND, 1))
    assert np.abs(w - w3).max() <= 1e-7 * np.abs(w).max()


def test_stencil_blocks_by_plane_eighths_give_the_same_product(P):
    """PHX_OPT_STENCIL_PLANE_ROWS: on large lattice planes XCD k walks the k-th eighth of every plane (an L2 placement of
    the stencil blocks).  Forced here on a small box (threshold 1 row per plane): every row is still applied exactly
    once -- the same product as with the default placement -- and the solve takes the same iterations."""
    from phifem_amd import _lib as L
    work, phi, f, uex, A, b, act = setup_problem(P, 3, 20)
    rng = np.random.default_rng(5)
    s = P.PhiFEMSolver(work)
    info = s.assemble(phi, f, uex)
    x = rng.standard_normal(info["n_active"])
    y = s.spmv(x)
    w = s.solve(rtol=1e-11)
    it = s.stats["iterations"]
    L.check(L.lib.phx_set_option(work._h, L.OPT_STENCIL_PLANE_ROWS, 1))
    try:
        s2 = P.PhiFEMSolver(work)
        info2 = s2.assemble(phi, f, uex)
        assert info2["stencil_rows"] == info["stencil_rows"] > 0
        y2 = s2.spmv(x)
        w2 = s2.solve(rtol=1e-11)
    finally:
        L.check(L.lib.phx_set_option(work._h, L.OPT_STENCIL_PLANE_ROWS, 32768))
    # (two assemblies differ in their last bits: the ghost penalty is summed by f64 atomics in arrival order)
    assert np.abs(y - y2).max() <= 1e-13 * np.abs(y).max()
    # (BiCGStab to rtol 1e-11 turns the last-bit differences of two assemblies into a few iterations more or less)
    assert abs(s2.stats["iterations"] - it) <= max(4, it // 5) and np.abs(w - w2).max() <= 1e-9 * np.abs(w).max()


def test_export_after_retagging_warns(P):
    """the lazy CSR export re-assembles from the CURRENT tags; when the mesh was tagged again after
    assemble() it must say so instead of silently exporting a different system, and `has_csr` tells the two cases
    apart (a system assembled with PHX_OPT_EXPORT_CSR exports its own copy, no warning)."""
    from phifem_amd import _lib as L
    from phifem_amd.mesh_scripts import NodalFunction, _tag_cells, _tag_facets
    work, phi, f, uex, A, b, act = setup_problem(P, 3, 12)
    s = P.PhiFEMSolver(work)
    info = s.assemble(phi, f, uex)
    assert info["has_csr"] == 0
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)
        s.export_csr()                      # tags unchanged since assemble(): no warning
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        st = _tag_cells(work, NodalFunction(phi), 1, single_layer_cut=True)
        _tag_facets(work, st, 1)
    with pytest.warns(RuntimeWarning, match="tagged again"):
        s.export_csr()
    L.check(L.lib.phx_set_option(work._h, L.OPT_EXPORT_CSR, 1))
    try:
        s2 = P.PhiFEMSolver(work)
        assert s2.assemble(phi, f, uex)["has_csr"] == 1
    finally:
        L.check(L.lib.phx_set_option(work._h, L.OPT_EXPORT_CSR, 0))
