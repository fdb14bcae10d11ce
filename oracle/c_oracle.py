"""ctypes access to the C/OpenMP restatement (oracle/phifem_oracle.c); test infrastructure."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libphifem_oracle.so")


def load():
    if not os.path.exists(_SO):
        subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
    lib = C.CDLL(_SO)
    lib.orc_poisson_sphere.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int64, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p]
    lib.orc_poisson_sphere.restype = C.c_int
    lib.orc_poisson_sphere2.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int64, C.c_int, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p]
    lib.orc_poisson_sphere2.restype = C.c_int
    return lib


def poisson_sphere(n, threads=0, rtol=1e-8, max_iter=20000, want_fields=False, precond=0):
    """precond = 0: Jacobi-BiCGStab; 1: that, then a second solve with the box sine-transform preconditioner of
    the GPU path (keys *_pc); 2: only the preconditioned solve."""
    lib = load()
    stats = np.zeros(14)
    ct = ft = u = None
    if want_fields:
        ct = np.empty(6 * n ** 3, dtype=np.int32)
        ft = np.empty(12 * n ** 3 + 6 * n ** 2, dtype=np.int32)
        u = np.empty(2 * (n + 1) ** 3)
    p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    lib.orc_poisson_sphere2(n, threads, rtol, max_iter, int(precond), p(stats), p(ct), p(ft), p(u))
    keys = ("n_active", "n_active_u", "nnz", "iterations", "relres", "t_tag", "t_assemble",
            "t_solve", "threads", "bad_facets", "iterations_pc", "relres_pc", "t_solve_pc", "pc_built")
    out = dict(zip(keys, stats))
    if want_fields:
        out.update(cell_tags=ct, facet_tags=ft, u_full=u)
    return out
