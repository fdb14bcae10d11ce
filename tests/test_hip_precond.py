"""GPU checks of the fictitious-domain (box sine-transform) preconditioner.

Floating point: the lattice Poisson solve is compared with scipy's DST-I solve of the same operator,
relative tolerance 1e-12 (f64 FFTs of length <= 2048)."""
import ctypes as C

import numpy as np
import pytest
import scipy.fft as sf

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    import phifem_amd
    assert phifem_amd._lib.device_count() > 0
    return phifem_amd


def scipy_box_solve(f, L, h):
    c = [h[1] * h[2] / h[0], h[0] * h[2] / h[1], h[0] * h[1] / h[2]]
    lam = [c[a] * (2.0 - 2.0 * np.cos(np.pi * np.arange(1, L[a]) / L[a])) for a in range(3)]
    lam3 = lam[2][:, None, None] + lam[1][None, :, None] + lam[0][None, None, :]
    F = sf.dstn(f, type=1)
    return sf.idstn(F / lam3, type=1)


@pytest.mark.parametrize("L", [(64, 64, 64), (128, 64, 192), (192, 128, 64), (256, 128, 64), (64, 384, 128),
                               (512, 64, 64), (64, 64, 768), (1024, 64, 64), (768, 64, 40), (64, 768, 20), (384, 192, 30),
                               # z is solved as a tridiagonal system: any column length, no transform
                               (64, 64, 180), (128, 64, 38), (64, 128, 2), (64, 64, 3), (64, 64, 1025),
                               # chunk shapes of the z pass: 8 / 16 wavefronts of 64 columns, 16 wavefronts of 2 x 32 columns
                               (64, 64, 195), (64, 64, 258), (64, 64, 355), (64, 64, 500), (64, 64, 701),
                               # lengths served by the wave-mode kernels (phx_dst_wave.inc.hip) in x AND in y, odd plane counts
                               (192, 192, 21), (256, 512, 4), (512, 256, 7), (192, 256, 2)])
@pytest.mark.parametrize("f32", [0, 1])
def test_box_poisson_solve_matches_scipy(P, L, f32):
    from phifem_amd import _lib as L_
    rng = np.random.default_rng(11)
    h = (0.011, 0.017, 0.013)
    f = rng.standard_normal((L[2] - 1, L[1] - 1, L[0] - 1))
    ref = scipy_box_solve(f, L, h)
    u = np.ascontiguousarray(f.copy())
    Lc = (C.c_int * 3)(*L)
    hc = (C.c_double * 3)(*h)
    L_.check(L_.lib.phx_box_poisson_solve(0, Lc, hc, f32, u.ctypes.data_as(C.c_void_p)))
    tol = 2e-5 if f32 else 1e-12   # f32 transforms of length <= 1024 in three axes
    assert np.abs(u - ref).max() <= tol * np.abs(ref).max()
    if f32:
        return
    # and it really inverts the 7-point operator
    c = [h[1] * h[2] / h[0], h[0] * h[2] / h[1], h[0] * h[1] / h[2]]
    up = np.pad(u, 1)
    Ku = (c[0] * (2 * up[1:-1, 1:-1, 1:-1] - up[1:-1, 1:-1, :-2] - up[1:-1, 1:-1, 2:])
          + c[1] * (2 * up[1:-1, 1:-1, 1:-1] - up[1:-1, :-2, 1:-1] - up[1:-1, 2:, 1:-1])
          + c[2] * (2 * up[1:-1, 1:-1, 1:-1] - up[:-2, 1:-1, 1:-1] - up[2:, 1:-1, 1:-1]))
    assert np.abs(Ku - f).max() <= 1e-10 * np.abs(f).max()


def _solve_case(P, n, phi_fn, precond):
    """tag -> assemble -> solve of the weak-Dirichlet problem for a nodal level-set on an n^3 box."""
    import warnings
    from phifem_amd import _lib as L_
    from phifem_amd.mesh_scripts import NodalFunction
    mesh = P.create_box([-1.5] * 3, [1.5] * 3, [n] * 3)
    L_.check(L_.lib.phx_set_option(mesh._h, L_.OPT_PRECOND, precond))
    x = mesh.x
    phi = phi_fn(x)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        P.compute_tags_measures(mesh, NodalFunction(phi), 1, box_mode=True, single_layer_cut=True)
    uex = np.prod(np.sin(x), axis=1)
    s = P.PhiFEMSolver(mesh)
    s.assemble(phi, 3.0 * uex, uex)
    w = s.solve(rtol=1e-9, max_iter=5000)
    return w, s.stats, phi, uex


LEVELSETS = {
    "sphere": lambda x: (x ** 2).sum(axis=1) - 1.0,
    "offset_ellipsoid": lambda x: ((x[:, 0] - 0.13) / 1.2) ** 2 + ((x[:, 1] + 0.07) / 0.6) ** 2 + (x[:, 2] / 0.9) ** 2 - 1.0,
    "two_balls": lambda x: np.minimum(((x - [0.65, 0.1, 0.0]) ** 2).sum(axis=1) - 0.36,
                                      ((x + [0.65, 0.0, 0.2]) ** 2).sum(axis=1) - 0.30),
    "torus": lambda x: (np.sqrt(x[:, 0] ** 2 + x[:, 1] ** 2) - 0.8) ** 2 + x[:, 2] ** 2 - 0.16,
    # crosses the x = 1.5 face of the background box: one-sided boundary terms, lattice box past the mesh
    "boundary_crossing": lambda x: ((x - [0.9, 0.2, 0.0]) ** 2).sum(axis=1) - 1.0,
}


@pytest.mark.parametrize("name", list(LEVELSETS))
@pytest.mark.parametrize("precond", [1, 2])
def test_preconditioned_solve_equals_jacobi_solve(P, name, precond):
    """Same solution (to the solver tolerance) with the box preconditioner in f32 / f64 and with Jacobi,
    in far fewer iterations, on level-sets that are not a centred sphere."""
    n = 48
    w_j, st_j, phi, uex = _solve_case(P, n, LEVELSETS[name], 0)
    w_p, st_p, _, _ = _solve_case(P, n, LEVELSETS[name], precond)
    assert st_j["precond"] == "jacobi" and st_p["precond"] == "box-dst"
    assert st_p["relres"] <= 1e-9 and st_j["relres"] <= 1e-9
    assert st_p["iterations"] < 0.6 * st_j["iterations"]
    assert np.abs(w_p - w_j).max() <= 1e-6 * np.abs(w_j).max()
    inside = phi < -0.1
    assert inside.sum() > 100
    assert np.abs(w_p[:uex.size][inside] - uex[inside]).max() < 5e-2   # sanity: discretisation error at n = 48 (0.03 on the sphere)


def test_preconditioner_in_2d(P):
    """2-D boxes: the lattice gets a dummy third axis with coefficient 0; same solution as with Jacobi."""
    import warnings
    from phifem_amd import _lib as L_
    from phifem_amd.mesh_scripts import NodalFunction
    res = {}
    for pc in (0, 1):
        mesh = P.create_box([-1.5, -1.5], [1.5, 1.5], [160, 160])
        L_.check(L_.lib.phx_set_option(mesh._h, L_.OPT_PRECOND, pc))
        x = mesh.x
        phi = (x ** 2).sum(axis=1) - 1.0
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            P.compute_tags_measures(mesh, NodalFunction(phi), 1, box_mode=True, single_layer_cut=True)
        uex = np.prod(np.sin(x), axis=1)
        s = P.PhiFEMSolver(mesh)
        s.assemble(phi, 2.0 * uex, uex)
        res[pc] = (s.solve(rtol=1e-9, max_iter=20000), dict(s.stats))
    assert res[0][1]["precond"] == "jacobi" and res[1][1]["precond"] == "box-dst"
    assert res[1][1]["relres"] <= 1e-9
    assert res[1][1]["iterations"] < 0.6 * res[0][1]["iterations"]
    assert np.abs(res[1][0] - res[0][0]).max() <= 1e-6 * np.abs(res[0][0]).max()


@pytest.mark.parametrize("d,n", [(2, 128), (3, 40)])
def test_preconditioner_on_submesh_of_a_box(P, d, n):
    """box_mode=False: the sub-mesh of Omega_h keeps the parent's lattice, so the box preconditioner applies
    (demo/weak-dirichlet/flower/main.py `sub`)."""
    import warnings
    from phifem_amd import _lib as L_
    from phifem_amd.mesh_scripts import NodalFunction
    res = {}
    for pc in (0, 1):
        mesh = P.create_box([-1.5] * d, [1.5] * d, [n] * d)
        x = mesh.x
        phi = (x ** 2).sum(axis=1) - 1.0
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            _, _, sub, _, _ = P.compute_tags_measures(mesh, NodalFunction(phi), 1, box_mode=False, single_layer_cut=True)
        L_.check(L_.lib.phx_set_option(sub._h, L_.OPT_PRECOND, pc))
        xs = sub.x
        phis = (xs ** 2).sum(axis=1) - 1.0
        uex = np.prod(np.sin(xs), axis=1)
        s = P.PhiFEMSolver(sub)
        s.assemble(phis, float(d) * uex, uex)
        res[pc] = (s.solve(rtol=1e-9, max_iter=20000), dict(s.stats))
    assert res[0][1]["precond"] == "jacobi" and res[1][1]["precond"] == "box-dst"
    assert res[1][1]["relres"] <= 1e-9
    assert res[1][1]["iterations"] < 0.6 * res[0][1]["iterations"]
    assert np.abs(res[1][0] - res[0][0]).max() <= 1e-6 * np.abs(res[0][0]).max()


@pytest.mark.parametrize("d,n", [(3, 16), (2, 48)])
def test_caller_supplied_lattice_mesh_gets_the_box_preconditioner(P, d, n):
    """a mesh that arrives as arrays (what a dolfinx caller hands over after create_box /
    create_rectangle) with its vertices on a tensor lattice -- in ANY vertex order -- is recognised
    by phx_mesh_create and gets the fictitious-domain preconditioner: the iteration count of the generated box
    (within 15 %: BiCGStab counts move by a few iterations with the summation order), not the ~4x larger Jacobi
    count; same solution."""
    import warnings
    from phifem_amd.mesh_scripts import NodalFunction
    box = P.create_box([-1.5] * d, [1.5] * d, [n] * d)
    x, cells = box.x, box.cells
    rng = np.random.default_rng(3)
    perm = rng.permutation(x.shape[0])          # new vertex id -> old vertex id
    inv = np.empty_like(perm)
    inv[perm] = np.arange(perm.size)
    xs = x[perm]
    cs = inv[cells][rng.permutation(cells.shape[0])]
    cs = np.take_along_axis(cs, rng.permuted(np.tile(np.arange(d + 1), (cs.shape[0], 1)), axis=1), axis=1).astype(np.int32)
    mesh = P.Mesh.from_arrays("tetrahedron" if d == 3 else "triangle", xs, cs)   # any vertex, cell and local order

    def run(m, xx):
        phi = (xx ** 2).sum(axis=1) - 1.0
        uex = np.prod(np.sin(xx), axis=1)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            P.compute_tags_measures(m, NodalFunction(phi), 1, box_mode=True, single_layer_cut=True)
        s = P.PhiFEMSolver(m)
        info = s.assemble(phi, float(d) * uex, uex)
        w = s.solve(rtol=1e-10, max_iter=20000)
        # the exported system (lazy CSR) in the caller's numbering is satisfied by the returned solution
        rowptr, col, val, rhs, dof = s.export_csr()
        import scipy.sparse as sp
        A = sp.csr_matrix((val, col, rowptr), shape=(rowptr.size - 1,) * 2)
        assert np.linalg.norm(A @ w[dof] - rhs) <= 1e-8 * np.linalg.norm(rhs)
        return w, dict(s.stats, **info)
    w_box, st_box = run(box, x)
    w_arr, st_arr = run(mesh, xs)
    assert st_box["precond"] == "box-dst" and st_arr["precond"] == "box-dst"
    # the caller's mesh is served by the generated box behind it -- closed-form rows, stencil operator
    assert st_arr["stencil_rows"] == st_box["stencil_rows"] > 0 and st_arr["n_active"] == st_box["n_active"]
    assert abs(st_arr["iterations"] - st_box["iterations"]) <= max(3, 0.15 * st_box["iterations"]), \
        (st_arr["iterations"], st_box["iterations"])
    nv = x.shape[0]
    # vertex perm[i] of the box is vertex i of the shuffled mesh
    assert np.abs(w_arr[:nv] - w_box[:nv][perm]).max() <= 1e-7 * np.abs(w_box).max()
    assert np.abs(w_arr[nv:] - w_box[nv:][perm]).max() <= 1e-7 * np.abs(w_box).max()
    # a mesh whose vertices are NOT on the lattice (one vertex moved by 1e-7 h: far above round-off, far below
    # the old 1e-6 h snap) is not replaced by the uniform box -- it keeps the generic path (every row stored) and its own
    # geometry; the solution moves by about that much
    xp = xs.copy()
    xp[xs.shape[0] // 2, 0] += 1e-7 * (3.0 / n)
    mesh_p = P.Mesh.from_arrays("tetrahedron" if d == 3 else "triangle", xp, cs)
    w_p, st_p = run(mesh_p, xp)
    assert st_p["stencil_rows"] == 0
    assert np.abs(w_p - w_arr).max() <= 1e-5 * np.abs(w_arr).max()
