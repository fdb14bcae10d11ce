"""PhiFEMSolver: the assemble -> solve sequence of the weak-Dirichlet demo over the C ABI.

The reference writes this sequence out in each demo (demo/weak-dirichlet/flower/main.py:102-186:
UFL forms, dolfinx assemble_matrix/assemble_vector, PETSc KSP + MUMPS); ROADMAP.md:15 plans a
class-based interface for it.  This class is that interface for the P1 x P1 weak-Dirichlet
Poisson problem, with the element integration, scatter and Krylov solve in HIP.
"""
import ctypes as C
import warnings

import numpy as np

from . import _lib as L


class ConvergenceWarning(RuntimeWarning):
    """The Krylov solve stopped at max_iter with a relative residual above rtol.  The reference solves
    directly (MUMPS LU, demo/weak-dirichlet/flower/main.py:162-182), so its callers assume an accurate
    solution: an unconverged iterate must not pass silently."""


def check_converged(stats, rtol, strict=False):
    """Warn (or raise with strict=True) when `stats` describes an unconverged solve."""
    if stats.get("converged", True):
        return
    msg = (f"BiCGStab did not converge: relative residual {stats['relres']:.3e} > rtol {rtol:g} after "
           f"{stats['iterations']} iterations")
    if strict:
        raise ArithmeticError(msg)
    warnings.warn(msg, ConvergenceWarning, stacklevel=3)


class PhiFEMSolver:
    def __init__(self, mesh, pen_coef=1.0, stab_coef=1.0, degree=1, levelset_degree=1, deterministic=False):
        """deterministic=True (PHX_OPT_DETERMINISTIC): bit-reproducible assembly (degree 2: the element kernels run
        twice and accumulate exactly) and Krylov dot products -- the same matrix bits, iteration count and solution
        on every run; costs one more pass of the element kernels.

        mesh: a tagged `phifem_amd.Mesh` (box mode) or the sub-mesh returned by
        `compute_tags_measures(..., box_mode=False)`; coefficients as main.py:42-43;
        degree = primal_degree = auxiliary degree (main.py:38, 76-78), levelset_degree as main.py:40.
        Degree-2 nodal arrays list the vertex values first, then the edge-midpoint values
        (`mesh.p2_dof_points()`)."""
        if degree not in (1, 2) or levelset_degree not in (1, 2):
            raise NotImplementedError("Lagrange degrees 1 and 2 are implemented")
        if degree == 1 and levelset_degree != 1:
            raise NotImplementedError("a P2 level-set needs degree = 2")
        self.degree, self.levelset_degree = degree, levelset_degree
        self.mesh = mesh
        self.pen_coef = float(pen_coef)
        self.stab_coef = float(stab_coef)
        self._sys = None
        self.stats = {}
        self.deterministic = bool(deterministic)

    def _apply_options(self):
        L.check(L.lib.phx_set_option(self.mesh._h, L.OPT_DETERMINISTIC, int(getattr(self, "deterministic", False))))
        if hasattr(self, "coarse"):
            L.check(L.lib.phx_set_option(self.mesh._h, L.OPT_EL_COARSE, self.coarse))

    def __del__(self):
        self._free()

    def _free(self):
        try:
            if self._sys is not None:
                L.lib.phx_system_destroy(self._sys)
                self._sys = None
        except Exception:
            pass

    def _arr(self, a, n):
        """Nodal array of the C ABI: numpy (converted) or a torch tensor, which is handed to the kernels as a
        raw `double*` and therefore has to BE one: float64, contiguous, n values, and -- on the device -- on
        the mesh's GPU."""
        if hasattr(a, "data_ptr"):
            import torch
            if a.dtype != torch.float64:
                raise ValueError(f"nodal tensor has dtype {a.dtype}, the C ABI takes float64")
            if not a.is_contiguous():
                raise ValueError("nodal tensor is not contiguous")
            if a.numel() != n:
                raise ValueError(f"nodal tensor has {a.numel()} values, the space has {n} DoFs")
            if a.is_cuda and a.device.index != self.mesh.device:
                raise ValueError(f"nodal tensor lives on cuda:{a.device.index}, the mesh on cuda:{self.mesh.device}")
            return a
        a = np.ascontiguousarray(a, dtype=np.float64)
        if a.shape[0] != n:
            raise ValueError(f"nodal array has {a.shape[0]} values, the space has {n} DoFs")
        return a

    def assemble(self, phi_h, f_h, u_D):
        """Bilinear + linear form of main.py:112-154 (nodal P1 data, numpy or device tensors)."""
        self._free()
        nd = self.ndofs
        nphi = self.mesh.nv if self.levelset_degree == 1 else nd
        phi_h = self._arr(phi_h, nphi)
        f_h, u_D = (self._arr(a, nd) for a in (f_h, u_D))
        locs = {L.ptr(a)[1] for a in (phi_h, f_h, u_D)}
        if len(locs) != 1:
            raise ValueError("phi_h, f_h and u_D must all live on the host or all on the device")
        self._keep = (phi_h, f_h, u_D)   # inputs stay alive as long as the system they were read for
        self._tag_generation = self.mesh._tag_generation
        self._sys = self._assemble_raw()
        return self.info()

    def _assemble_raw(self):
        phi_h, f_h, u_D = self._keep
        loc = L.ptr(phi_h)[1]
        h = C.c_void_p()
        self._apply_options()
        if self.degree == 1:
            L.check(L.lib.phx_assemble_poisson_wd(
                self.mesh._h, self.pen_coef, self.stab_coef, L.ptr(phi_h)[0], L.ptr(f_h)[0],
                L.ptr(u_D)[0], loc, C.byref(h)))
        else:
            L.check(L.lib.phx_assemble_poisson_wd_p2(
                self.mesh._h, self.pen_coef, self.stab_coef, L.ptr(phi_h)[0], self.levelset_degree,
                L.ptr(f_h)[0], L.ptr(u_D)[0], loc, C.byref(h)))
        return h

    @property
    def ndofs(self):
        """DoFs per field in the full numbering (vertices, plus edges at degree 2)."""
        return self.mesh.nv if self.degree == 1 else self.mesh.nv + self.mesh.ne

    def info(self):
        i = (C.c_int64 * 14)()
        L.check(L.lib.phx_system_info(self._sys, i))
        keys = ("n_active", "n_active_u", "nnz", "n_full", "sell_padded_nnz", "slot_capacity",
                "sell_nnz", "n_slices", "indexed_slices", "spmv_matrix_bytes", "indexed_slices_lds",
                "stencil_rows", "stencil_runs", "has_csr")
        return dict(zip(keys, (int(v) for v in i)))

    def export_csr(self):
        """(scipy-style rowptr, col, val, rhs, dof) of the active system, for inspection/tests.

        The CSR copy is LAZY: P1 systems on Kuhn boxes are assembled straight into the solver formats
        (stencil-coded interior rows + SELL, PHX_OPT_STRUCTURED) and never form it; the first export
        re-assembles the same inputs once with PHX_OPT_EXPORT_CSR set and reads the CSR of that system."""
        i = self.info()

        def buffers(n, nnz):
            arrs = (np.empty(n + 1, dtype=np.int64), np.empty(nnz, dtype=np.int32), np.empty(nnz, dtype=np.float64),
                    np.empty(n, dtype=np.float64), np.empty(n, dtype=np.int64))
            return arrs, [a.ctypes.data_as(C.c_void_p) for a in arrs]

        if i["has_csr"]:
            (rowptr, col, val, rhs, dof), args = buffers(i["n_active"], i["nnz"])
            L.check(L.lib.phx_system_export(self._sys, *args))
        elif getattr(self, "_keep", None) is not None and getattr(self, "_tag_generation", None) is not None:
            if self._tag_generation != self.mesh._tag_generation:
                warnings.warn("the mesh was tagged again after assemble(): the exported CSR is rebuilt from the CURRENT "
                              "tags and equals the solved system only if they are unchanged (set PHX_OPT_EXPORT_CSR "
                              "before assembling to keep the solved system's own copy)", RuntimeWarning, stacklevel=2)
            L.check(L.lib.phx_set_option(self.mesh._h, L.OPT_EXPORT_CSR, 1))
            try:
                h = self._assemble_raw()
            finally:
                L.check(L.lib.phx_set_option(self.mesh._h, L.OPT_EXPORT_CSR, 0))
            try:
                # sizes of the RE-ASSEMBLED system: a structured system only estimates its structural count
                j = (C.c_int64 * 14)()
                L.check(L.lib.phx_system_info(h, j))
                (rowptr, col, val, rhs, dof), args = buffers(int(j[0]), int(j[2]))
                L.check(L.lib.phx_system_export(h, *args))
            finally:
                L.lib.phx_system_destroy(h)
        else:
            (rowptr, col, val, rhs, dof), args = buffers(i["n_active"], i["nnz"])
            L.check(L.lib.phx_system_export(self._sys, *args))   # reports "assembled without its CSR copy"
        return rowptr, col, val, rhs, dof

    def export_rhs_dof(self):
        """(rhs, dof) of the active system without the matrix: right-hand side in active numbering and the map
        active row -> full DoF index (works at sizes where the CSR copy would not fit the host)."""
        n = self.info()["n_active"]
        rhs = np.empty(n, dtype=np.float64)
        dof = np.empty(n, dtype=np.int64)
        L.check(L.lib.phx_system_export(self._sys, None, None, None, rhs.ctypes.data_as(C.c_void_p),
                                        dof.ctypes.data_as(C.c_void_p)))
        return rhs, dof

    def solve(self, rtol=1e-8, max_iter=20000, out=None, profile_spmv=False, strict=False):
        """Replaces the KSP/MUMPS block of main.py:162-182.  Returns the mixed solution in the
        full numbering [u (nv), p (nv)] with inactive DoFs at zero; `out` may be a device
        tensor of 2*nv doubles.  `stats["converged"]` says whether rtol was reached; if not, a
        `ConvergenceWarning` is issued (strict=True: ArithmeticError) -- the reference's direct solve
        never returns an inaccurate x silently."""
        nfull = self.info()["n_full"]
        if out is None:
            out = np.empty(nfull, dtype=np.float64)
        elif hasattr(out, "data_ptr"):
            import torch
            if out.dtype != torch.float64 or not out.is_contiguous() or out.numel() != nfull:
                raise ValueError(f"`out` must be a contiguous float64 tensor of {nfull} values")
        p, loc = L.ptr(out)
        st = (C.c_double * 8)()
        self._apply_options()
        L.check(L.lib.phx_set_option(self.mesh._h, L.OPT_PROFILE_SPMV, int(profile_spmv)))
        L.check(L.lib.phx_solve(self._sys, 0, float(rtol), int(max_iter), p, loc, st))
        self.stats = {"iterations": int(st[0]), "relres": st[1], "seconds": st[2],
                      "spmv": int(st[3]), "spmv_avg_s": st[4], "spmv_timed": int(st[5]),
                      "converged": bool(st[6]), "restarts": int(st[7])}
        self.stats.update(self.precond_info())
        check_converged(self.stats, rtol, strict)
        return out

    def split(self, w):
        """solution_wh.split() (main.py:185): (u, p) views of the mixed vector."""
        nd = self.ndofs
        return w[:nd], w[nd:]

    def precond_info(self):
        """State of the fictitious-domain preconditioner after a solve (phx_precond_info)."""
        o = (C.c_double * 8)()
        L.check(L.lib.phx_precond_info(self._sys, o))
        return {"precond": {1: "box-dst", 2: "vertex-block-jacobi", 3: "vertex-block-jacobi+coarse"}.get(int(o[0]), "jacobi"), "precond_L": [int(o[1]), int(o[2]), int(o[3])],
                "precond_points": int(o[4]), "dst_avg_s": o[5], "dst_timed": int(o[6]),
                "precond_value_bytes": int(o[7])}

    def spmv(self, x):
        y = np.empty_like(x)
        L.check(L.lib.phx_spmv(self._sys, x.ctypes.data_as(C.c_void_p),
                               y.ctypes.data_as(C.c_void_p), L.HOST))
        return y

    def spmv_bench(self, reps=50):
        o = (C.c_double * 3)()
        L.check(L.lib.phx_spmv_bench(self._sys, int(reps), o))
        return {"ms": o[0], "algorithmic_bytes": o[1], "padded_bytes": o[2]}


class StrongDirichletSolver(PhiFEMSolver):
    """Direct ("strong Dirichlet") phi-FEM: u_h = phi_h w_h with one scalar unknown w_h -- the
    assemble -> solve -> multiply sequence of demo/strong-dirichlet/flower/main.py:83-182 over the
    C ABI (`phx_assemble_poisson_sd`).  `degree` is fe_degree (main.py:39), `levelset_degree` as
    main.py:41; the mesh is the tagged background mesh (mesh_type "bg", main.py:60-65) or the
    sub-mesh (mesh_type "sub", main.py:66-70)."""

    def __init__(self, mesh, stab_coef=1.0, degree=1, levelset_degree=1):
        if degree not in (1, 2) or levelset_degree not in (1, 2):
            raise NotImplementedError("Lagrange degrees 1 and 2 are implemented")
        self.degree, self.levelset_degree = degree, levelset_degree
        self.mesh = mesh
        self.stab_coef = float(stab_coef)
        self._sys = None
        self.stats = {}

    def assemble(self, phi_h, f_h):
        """Bilinear + linear form of main.py:104-129 (nodal data, numpy or device tensors)."""
        self._free()
        nphi = self.mesh.nv if self.levelset_degree == 1 else self.mesh.nv + self.mesh.ne
        phi_h = self._arr(phi_h, nphi)
        f_h = self._arr(f_h, self.ndofs)
        locs = {L.ptr(a)[1] for a in (phi_h, f_h)}
        if len(locs) != 1:
            raise ValueError("phi_h and f_h must both live on the host or both on the device")
        h = C.c_void_p()
        L.check(L.lib.phx_assemble_poisson_sd(
            self.mesh._h, self.stab_coef, self.degree, L.ptr(phi_h)[0], self.levelset_degree,
            L.ptr(f_h)[0], locs.pop(), C.byref(h)))
        self._sys = h
        self._phi = phi_h
        return self.info()

    def split(self, w):
        raise NotImplementedError("one scalar field: solve() returns w_h itself")

    def solution(self, w, solution_degree=None):
        """u_h = w_h phi_h at the nodes of the solution space (main.py:172-182: both factors are
        interpolated into the space of degree `solution_degree`, then multiplied node by node).
        Implemented for solution_degree = degree = levelset_degree, where both interpolations
        are the identity."""
        k = self.degree if solution_degree is None else solution_degree
        if not (k == self.degree == self.levelset_degree):
            raise NotImplementedError("solution_degree must equal degree and levelset_degree")
        return w * self._phi


class NeumannRobinSolver(PhiFEMSolver):
    """Neumann / Robin phi-FEM, mixed (u, y, p) in P1 x P1^d x DG0 with a P2 level-set: the assemble ->
    solve sequence of demo/robin/square/main.py:98-190 over the C ABI (`phx_assemble_poisson_flux`), on
    triangles / tetrahedra.  `robin_coef = 0`, `facet_tag = 3` is the formulation of
    demo/neumann/square/main.py:113-158 (its quadrilateral cells are not covered).  Boundary condition:
    du/dn + robin_coef u = g on {phi = 0}."""

    def __init__(self, mesh, pen_coef=1.0, stab_coef=1.0, robin_coef=0.0, facet_tag=2, quadrature_degree=10):
        self.mesh = mesh
        self.params = np.array([pen_coef, stab_coef, robin_coef], dtype=np.float64)
        self.facet_tag, self.quadrature_degree = int(facet_tag), int(quadrature_degree)
        self.degree, self.levelset_degree = 1, 2
        self._sys = None
        self.stats = {}

    def assemble(self, phi_h, f_h, g_h):
        """phi_h: degree-2 nodal values -- simplices: vertices, then edges (`mesh.p2_dof_points()`);
        quadrilaterals (the cell type of demo/neumann/square/main.py:49-50, Q1 x Q1^2 x DG0 with a Q2 level-set):
        vertices, then edge midpoints by facet id, then cell centres (`mesh.q2_dof_points()`).
        f_h, g_h (u_N / u_R, main.py:103-110): degree-1 nodal values."""
        self._free()
        m = self.mesh
        nphi = m.nv + m.nf + m.nc if m.cell_type == "quadrilateral" else m.nv + m.ne
        phi_h = self._arr(phi_h, nphi)
        f_h, g_h = self._arr(f_h, m.nv), self._arr(g_h, m.nv)
        locs = {L.ptr(a)[1] for a in (phi_h, f_h, g_h)}
        if len(locs) != 1:
            raise ValueError("phi_h, f_h and g_h must all live on the host or all on the device")
        h = C.c_void_p()
        L.check(L.lib.phx_assemble_poisson_flux(
            m._h, self.params.ctypes.data_as(C.c_void_p), self.facet_tag, self.quadrature_degree,
            L.ptr(phi_h)[0], L.ptr(f_h)[0], L.ptr(g_h)[0], locs.pop(), C.byref(h)))
        self._sys = h
        return self.info()

    def split(self, w):
        """solution_wh.split() (main.py:186): u (nv,), y (nv, d), p (nc,)."""
        m = self.mesh
        d, nv = m.gdim, m.nv
        return w[:nv], w[nv:(1 + d) * nv].reshape(d, nv).T, w[(1 + d) * nv:]


class InterfaceElasticitySolver(PhiFEMSolver):
    """Two-material linear elasticity with a level-set interface, 5-field mixed phi-FEM
    (u_in, u_out, y_in, y_out, p), all first-order Lagrange: the assemble -> solve sequence of
    demo/interface-elasticity/main.py:145-289 over the C ABI (`phx_assemble_elasticity_if`).

    The mesh must be tagged in box mode (main.py:115-117).  Solution layout: component-major
    blocks of nv entries, see `blocks()`."""

    def __init__(self, mesh, E_in=1.0, nu_in=0.3, E_out=1.0e-3, nu_out=0.3,
                 penalization_coefficient=1.0, stabilization_coefficient=1.0, deterministic=False, coarse=-1):
        """coarse (PHX_OPT_EL_COARSE): spacing H / h of the coarse-space correction of the solve on generated boxes;
        -1 automatic (on from 80 cubes per axis), 0 off."""
        # material parameters demo/interface-elasticity/data.py:14-22, coefficients param1.yaml:16-17
        super().__init__(mesh, deterministic=deterministic)
        self.coarse = int(coarse)
        self.params = np.array([E_in, nu_in, E_out, nu_out, penalization_coefficient,
                                stabilization_coefficient], dtype=np.float64)

    def assemble(self, phi_h, f_h, u_D, bc_vertices):
        """phi_h: (nv,) nodal level-set; f_h, u_D: (nv, d) nodal vector fields; bc_vertices:
        vertices where u_in = u_D is imposed (the box boundary in the demo, main.py:158-177)."""
        self._free()
        self._apply_options()
        m = self.mesh
        d = m.gdim
        h = C.c_void_p()
        if hasattr(phi_h, "data_ptr") and phi_h.is_cuda:
            # device-resident inputs: f_h, u_D already component-major (d, nv); bc_vertices int32
            f_cm, u_cm, bcv = f_h.contiguous(), u_D.contiguous(), bc_vertices.contiguous()
            if tuple(f_cm.shape) != (d, m.nv) or tuple(u_cm.shape) != (d, m.nv):
                raise ValueError("device f_h and u_D must be component-major (d, nv)")
            self._keep = (phi_h, f_cm, u_cm, bcv)
            L.check(L.lib.phx_assemble_elasticity_if(
                m._h, self.params.ctypes.data_as(C.c_void_p), C.c_void_p(phi_h.data_ptr()),
                C.c_void_p(f_cm.data_ptr()), C.c_void_p(u_cm.data_ptr()), C.c_void_p(bcv.data_ptr()),
                bcv.numel(), L.DEVICE, C.byref(h)))
        else:
            phi_h = np.ascontiguousarray(phi_h, dtype=np.float64)
            f_cm = np.ascontiguousarray(np.asarray(f_h, dtype=np.float64).T)   # component-major
            u_cm = np.ascontiguousarray(np.asarray(u_D, dtype=np.float64).T)
            bcv = np.ascontiguousarray(bc_vertices, dtype=np.int32)
            if phi_h.shape[0] != m.nv or f_cm.shape != (d, m.nv) or u_cm.shape != (d, m.nv):
                raise ValueError("phi_h must be (nv,), f_h and u_D (nv, d)")
            if bcv.size and (bcv.min() < 0 or bcv.max() >= m.nv):
                raise ValueError("bc_vertices out of range")
            vp = lambda a: a.ctypes.data_as(C.c_void_p)
            L.check(L.lib.phx_assemble_elasticity_if(m._h, vp(self.params), vp(phi_h), vp(f_cm), vp(u_cm),
                                                     vp(bcv), bcv.size, L.HOST, C.byref(h)))
        self._sys = h
        return self.info()

    def blocks(self, w):
        """Split the solution: dict of (nv, d) / (nv, d, d) arrays (solution_wh.split(), main.py:291)."""
        m = self.mesh
        d, nv = m.gdim, m.nv
        w = np.asarray(w).reshape(-1, nv)
        return {"u_in": w[0:d].T, "u_out": w[d:2 * d].T,
                "y_in": w[2 * d:2 * d + d * d].T.reshape(nv, d, d),
                "y_out": w[2 * d + d * d:2 * d + 2 * d * d].T.reshape(nv, d, d),
                "p": w[2 * d + 2 * d * d:].T}
