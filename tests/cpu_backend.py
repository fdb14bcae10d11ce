"""CPU stand-in for the phase API of the C ABI (TEST INFRASTRUCTURE): the local slab system is
tagged and assembled by the numpy oracle, and the seven Krylov phases of
phifem_amd/csrc/phx_solve.hip are restated with numpy on torch CPU tensors.  It lets the
multi-rank host logic (slab layout, ghost layers, ownership, halo lists, exchanges, all-reduces)
run under gloo without a GPU."""
import warnings

import numpy as np
import torch

from oracle import assembly as OA, meshgen, tagging as OT
from oracle.topology import Topology

R_OFF = 8
S_RHO, S_ALPHA, S_OMEGA, S_BB, S_RR = 0, 1, 2, 3, 4
R_RV, R_TS, R_TT, R_RHO, R_RR = 0, 1, 2, 4, 5


def capsule_data(x, world):
    """The weak-scaling problem of phifem_amd.distributed.SlabProblem: capsule of radius 1 along z."""
    dz = np.maximum(np.abs(x[:, 2]) - 1.5 * (world - 1), 0.0)
    phi = x[:, 0] ** 2 + x[:, 1] ** 2 + dz ** 2 - 1.0
    uex = np.sin(x[:, 0]) * np.sin(x[:, 1]) * np.sin(x[:, 2])
    f = 3.0 * uex
    return phi, f, uex


def assemble_local(n, world, k0, k1, has_exterior=None, sphere=False):
    """Oracle tag + assemble on the slab [k0, k1) of the n x n x (n*world) box.  sphere=True: the unit
    sphere in the tall box (BASELINE configs[4] in miniature: the end slabs do not touch the domain)."""
    lo, hi = [-1.5, -1.5, -1.5 * world], [1.5, 1.5, 1.5 * world]
    x, cells = meshgen.create_box(lo, hi, [n, n, k1 - k0], offset=[0, 0, k0],
                                  n_global=[n, n, n * world])
    topo = Topology("tetrahedron", cells, x.shape[0])
    phi, f, uex = capsule_data(x, 1 if sphere else world)
    ls = OT.NodalP1(phi)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        cv = OT.tag_cells_values(topo, x, ls, 1, single_layer_cut=True)
        bc = OT.boundary_cell_cut_flags(topo, x, ls, 1)
    no_ext = None if has_exterior is None else (not has_exterior)
    fv, count = OT.tag_facets_values(topo, cv, bc, no_ext=no_ext)
    ds = OT.integration_entities(topo, np.flatnonzero((cv == 1) | (cv == 2)), np.flatnonzero(fv == 4))
    A, b, act = OA.assemble_poisson_wd(topo, x, cv, fv, ds, phi, f, uex)
    return x, topo, cv, A, b, act


class CpuBackend:
    def __init__(self, A, b, act, nv):
        idx = np.flatnonzero(act)
        self.idx = idx
        self.n, self.nv = idx.size, nv
        Aa = A[idx][:, idx].tocsr()
        self.d = Aa.diagonal()
        # an empty slab (n = 0) keeps a 0 x 0 operator and takes part in the collectives with zeros
        self.As = Aa @ __import__("scipy.sparse", fromlist=["diags"]).diags(1.0 / self.d) if idx.size else Aa
        self.rhs = b[idx]
        full_to_act = -np.ones(2 * nv, dtype=np.int64)
        full_to_act[idx] = np.arange(idx.size)
        self.dof_blocks = [torch.from_numpy(full_to_act[:nv].copy()),
                           torch.from_numpy(full_to_act[nv:].copy())]
        self.perm = torch.arange(self.n)

    def attach(self, work, scal, own):
        n = self.n
        w = work.numpy()
        self.r, self.rhat, self.p, self.v, self.s, self.t, self.y, self.bv = (
            w[i * n:(i + 1) * n] for i in range(8))
        self.S = scal.numpy()
        self.own = own.numpy().astype(bool)

    def phase(self, k):
        S, own = self.S, self.own
        if k == 0:
            S[:16] = 0.0
            bi = np.where(own, self.rhs, 0.0)
            for vec in (self.bv, self.r, self.rhat, self.p):
                vec[:] = bi
            self.y[:] = 0.0
            S[R_OFF + R_RHO] = bi @ bi
            S[R_OFF + R_RR] = 1.0   # veto: this stand-in has no box preconditioner (phx_krylov_precond_disable)
        elif k == 1:
            S[S_RHO] = S[S_BB] = S[S_RR] = S[R_OFF + R_RHO]
        elif k == 2:
            self.v[:] = np.where(own, self.As @ self.p, 0.0)
            S[R_OFF + R_RV] = self.v @ self.rhat
        elif k == 3:
            alpha = S[S_RHO] / S[R_OFF + R_RV]
            self.s[:] = np.where(own, self.r - alpha * self.v, 0.0)
            S[S_ALPHA] = alpha
        elif k == 4:
            self.t[:] = np.where(own, self.As @ self.s, 0.0)
            S[R_OFF + R_TS] = self.t @ self.s
            S[R_OFF + R_TT] = self.t @ self.t
        elif k == 5:
            alpha, omega = S[S_ALPHA], S[R_OFF + R_TS] / S[R_OFF + R_TT]
            self.y[own] += alpha * self.p[own] + omega * self.s[own]
            self.r[:] = np.where(own, self.s - omega * self.t, 0.0)
            S[R_OFF + R_RHO] = self.rhat @ self.r
            S[R_OFF + R_RR] = self.r @ self.r
            S[S_OMEGA] = omega
        elif k == 6:
            with np.errstate(all="ignore"):
                beta = (S[R_OFF + R_RHO] / S[S_RHO]) * (S[S_ALPHA] / S[S_OMEGA])
            restart = not (abs(beta) <= 1e300) or not (abs(S[R_OFF + R_RHO]) > 1e-14 * S[R_OFF + R_RR])
            if restart:   # breakdown guard of k_update_p / k_kr_roll
                self.p[:] = self.r
                self.rhat[:] = self.r
                S[S_RHO] = S[R_OFF + R_RR]
            else:
                self.p[:] = np.where(own, self.r + beta * (self.p - S[S_OMEGA] * self.v), 0.0)
                S[S_RHO] = S[R_OFF + R_RHO]
            S[S_RR] = S[R_OFF + R_RR]
        elif k == 11:   # true-residual verification: t = A y
            self.t[:] = np.where(own, self.As @ self.y, 0.0)
        elif k == 12:   # r = b - t, (r, r)
            self.r[:] = np.where(own, self.bv - self.t, 0.0)
            S[R_OFF + R_RR] = self.r @ self.r
        elif k == 13:   # restart from r (k_restart_from_r)
            self.p[:] = self.r
            self.rhat[:] = self.r
            S[S_RHO] = S[S_RR] = S[R_OFF + R_RR]
            S[S_ALPHA] = S[S_OMEGA] = 1.0

    def finish(self, out):
        o = out.numpy()
        o[:] = 0.0
        o[self.idx] = self.y / self.d

    def precond_disable(self):
        pass

    def profile(self, reset):
        return None

    def synchronize(self):
        pass
