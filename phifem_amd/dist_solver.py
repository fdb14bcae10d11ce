"""Multi-GPU Jacobi-BiCGStab over slabs: torch.distributed (RCCL over xGMI, or gloo in CPU tests)
drives the phase kernels of the C ABI (`phx_krylov_phase`).

Per iteration the data path has exactly two kinds of exchange:
  * halo: the entries of p (before phase 2) and s (before phase 4) on the two vertex planes either
    side of a slab interface, point-to-point with the (at most two) neighbours -- slabs talk over
    dedicated xGMI links, no ring;
  * scalars: one all-reduce of 1, 2 and 2 doubles after phases 2, 4 and 5 (the dot products of a
    phase are batched into one call).
Nothing else crosses ranks: tagging and assembly are redundant on the ghost layers.

The loop is written against a small backend interface so that the same code runs on the GPU
(`HipBackend`, C ABI) and in the CPU tests (a numpy stand-in under tests/).
"""
import ctypes as C
import time

import numpy as np

HALO_PLANES = 2   # a row reaches vertices two planes away (facet macro-elements, main.py:129-134)
SCAL_DOUBLES = 16 + 2 * 8 * 64 * 8
R_OFF = 8
R_RV, R_TS, R_TT, R_RHO, R_RR = 0, 1, 2, 4, 5
S_BB = 3


class HipBackend:
    """The assembled local system of a `PhiFEMSolver` seen through the phase API."""

    def __init__(self, solver, torch_device, blocks=None):
        import torch
        from . import _lib as L
        self.L, self.torch = L, torch
        self.sys = solver._sys
        self.mesh = solver.mesh
        info = solver.info()
        self.n, self.nv = info["n_active"], solver.mesh.nv
        self.dev = torch_device
        nent = info["n_full"] // 2 if blocks is None else info["n_full"]
        perm = torch.empty(self.n, dtype=torch.int32, device=self.dev)
        dof_u = torch.empty(nent, dtype=torch.int32, device=self.dev)
        L.check(L.lib.phx_system_get_perm(self.sys, C.c_void_p(perm.data_ptr()),
                                          C.c_void_p(dof_u.data_ptr()), None, L.DEVICE))
        self.perm = perm.long()
        if blocks is None:      # (u, p) mixed scalar system: two blocks over the vertices
            dof_p = torch.empty(nent, dtype=torch.int32, device=self.dev)
            L.check(L.lib.phx_system_get_perm(self.sys, None, None, C.c_void_p(dof_p.data_ptr()), L.DEVICE))
            self.dof_blocks = [dof_u.long(), dof_p.long()]
        else:                   # component-major blocks of nv entries in ONE map (elasticity)
            full = dof_u.long()
            self.dof_blocks = [full[b * self.nv:(b + 1) * self.nv] for b in range(blocks)]

    def use_current_stream(self):
        st = self.torch.cuda.current_stream(self.dev).cuda_stream
        self.L.check(self.L.lib.phx_mesh_set_stream(self.mesh._h, C.c_uint64(st)))

    def attach(self, work, scal, own):
        self._keep = (work, scal, own)
        L = self.L
        L.check(L.lib.phx_krylov_attach(self.sys, C.c_void_p(work.data_ptr()),
                                        C.c_void_p(scal.data_ptr()), C.c_void_p(own.data_ptr())))

    def phase(self, k):
        self.L.check(self.L.lib.phx_krylov_phase(self.sys, k))

    def precond_active(self):
        a = C.c_int(0)
        self.L.check(self.L.lib.phx_krylov_precond_active(self.sys, C.byref(a)))
        return bool(a.value)

    def precond_disable(self):
        self.L.check(self.L.lib.phx_krylov_precond_disable(self.sys))
        self.exact = False

    exact = False   # slab-exact preconditioner set up (phx_precond_setup_global)

    def setup_exact_precond(self, dist, rank, world, zb, stage_cpu=False):
        """Slab-exact box preconditioner (include/phifem_hip.h): one global lattice box around the active
        vertices of all ranks, sine transforms in x / y rank-local, the tridiagonal z solves continued across
        ranks through one all-gather of two carries per lattice column.  Collective; every rank ends up with the
        same answer (`self.exact`)."""
        L, torch = self.L, self.torch
        bb = (C.c_int64 * 6)()
        L.check(L.lib.phx_precond_local_bbox(self.sys, bb))
        ctl = torch.device("cpu") if (stage_cpu or dist.get_backend() != "nccl") else self.dev
        lo = torch.tensor([bb[0], bb[1], bb[2]], dtype=torch.int64, device=ctl)
        hi = torch.tensor([bb[3], bb[4], bb[5]], dtype=torch.int64, device=ctl)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        g = (C.c_int64 * 6)(*[int(v) for v in lo.tolist() + hi.tolist()])
        zbc = (C.c_int64 * (world + 1))(*[int(z) for z in zb])
        ncol = C.c_int64(0)
        L.check(L.lib.phx_precond_setup_global(self.sys, g, world, rank, zbc, C.byref(ncol)))
        self.exact = ncol.value > 0
        if self.exact:
            self.carry_send = torch.zeros(2 * ncol.value, dtype=torch.float64, device=self.dev)
            self.carry_recv = torch.zeros(world * 2 * ncol.value, dtype=torch.float64, device=self.dev)
            L.check(L.lib.phx_precond_set_carry_buffers(self.sys, C.c_void_p(self.carry_send.data_ptr()),
                                                        C.c_void_p(self.carry_recv.data_ptr())))
        return self.exact

    def finish(self, out):
        self.L.check(self.L.lib.phx_krylov_finish(self.sys, C.c_void_p(out.data_ptr()), self.L.DEVICE))

    def profile(self, reset):
        L = self.L
        if reset:
            L.check(L.lib.phx_krylov_profile(self.sys, 1, None, None))
            return None
        a, c = C.c_double(0.0), C.c_int64(0)
        L.check(L.lib.phx_krylov_profile(self.sys, 0, C.byref(a), C.byref(c)))
        return a.value, c.value

    def synchronize(self):
        self.torch.cuda.synchronize(self.dev)


def ownership_and_halos(torch, backend, plane_size, k0, P0, P1, rank, world, n_planes_local):
    """Owned-row mask (solver order) and the per-neighbour halo index lists (solver positions).

    Local vertex v sits on global vertex plane k0 + v // plane_size.  A rank owns planes
    [P0, P1).  Lists are ordered (u before p, ascending global vertex id) so that the sender's
    and the receiver's enumerations coincide."""
    dev = backend.perm.device
    nv, n = backend.nv, backend.n
    plane = torch.arange(nv, device=dev) // plane_size + k0
    owned_v = (plane >= P0) & (plane < P1)
    owned_a = torch.zeros(n, dtype=torch.bool, device=dev)
    nblk = len(backend.dof_blocks)
    for dof in backend.dof_blocks:
        act = dof >= 0
        owned_a[dof[act]] = owned_v[act]
    iperm = torch.empty(n, dtype=torch.long, device=dev)
    iperm[backend.perm] = torch.arange(n, device=dev)
    own_s = owned_a[backend.perm].to(torch.uint8).contiguous()

    def positions(lo, hi):
        """solver positions + global ids of the active DoFs on global planes [lo, hi)."""
        sel = (plane >= lo) & (plane < hi)
        pos, gid = [], []
        for kind, dof in enumerate(backend.dof_blocks):
            v = torch.nonzero(sel & (dof >= 0)).flatten()
            pos.append(iperm[dof[v]])
            gid.append((v + k0 * plane_size) * nblk + kind)
        return torch.cat(pos), torch.cat(gid)

    halos = []
    H = HALO_PLANES
    if rank > 0:         # lower neighbour owns planes < P0
        send = positions(P0, min(P0 + H, P1))
        recv = positions(max(P0 - H, k0), P0)
        halos.append({"peer": rank - 1, "send": send, "recv": recv})
    if rank < world - 1:  # upper neighbour owns planes >= P1
        send = positions(max(P1 - H, P0), P1)
        recv = positions(P1, min(P1 + H, k0 + n_planes_local))
        halos.append({"peer": rank + 1, "send": send, "recv": recv})
    return own_s, halos


class DistributedSolver:
    """BiCGStab (right Jacobi) on a slab-partitioned system."""

    def __init__(self, backend, dist, torch, rank, world, plane_size, k0, P0, P1, n_planes_local,
                 rtol=1e-8, max_iter=20000, check_every=8):
        self.b, self.dist, self.torch = backend, dist, torch
        self.rank, self.world = rank, world
        # rehearsal mode: gloo cannot move CUDA tensors point-to-point, so stage through the host
        self.stage_cpu = bool(world > 1 and dist.get_backend() == "gloo"
                              and backend.perm.device.type == "cuda")
        self.rtol, self.max_iter, self.check_every = rtol, max_iter, check_every
        dev = backend.perm.device
        self.own, self.halos = ownership_and_halos(torch, backend, plane_size, k0, P0, P1, rank,
                                                   world, n_planes_local)
        n = backend.n
        self.work = torch.zeros(10 * n, dtype=torch.float64, device=dev)
        self.scal = torch.zeros(SCAL_DOUBLES, dtype=torch.float64, device=dev)
        self.p = self.work[2 * n:3 * n]
        self.s = self.work[4 * n:5 * n]
        self.phat = self.work[8 * n:9 * n]
        self.shat = self.work[9 * n:10 * n]
        backend.attach(self.work, self.scal, self.own)
        self.n_owned = int(self.own.sum().item())
        self._verify_halos()

    def _verify_halos(self):
        """Both sides of an interface must enumerate the same DoFs: compare the global ids."""
        torch, dist = self.torch, self.dist
        live = []
        for h in self.halos:
            mine = torch.tensor([h["send"][1].numel(), h["recv"][1].numel()], dtype=torch.long,
                                device=h["send"][1].device)
            theirs = torch.empty_like(mine)
            self._sendrecv(h["peer"], mine, theirs)
            if int(theirs[0]) != h["recv"][1].numel() or int(theirs[1]) != h["send"][1].numel():
                raise RuntimeError(
                    f"rank {self.rank}: halo size mismatch with rank {h['peer']}: they send "
                    f"{int(theirs[0])} / expect {int(theirs[1])}, I expect "
                    f"{h['recv'][1].numel()} / send {h['send'][1].numel()}")
            if h["send"][1].numel() == 0 and h["recv"][1].numel() == 0:
                continue   # an interface outside the domain (e.g. next to an empty slab): both sides drop it
            live.append(h)
            got = torch.empty_like(h["recv"][1])
            self._sendrecv(h["peer"], h["send"][1].contiguous(), got)
            if not torch.equal(got, h["recv"][1]):
                raise RuntimeError(f"rank {self.rank}: halo DoF sets differ from rank {h['peer']}")
            h["sbuf"] = torch.empty(h["send"][0].numel(), dtype=torch.float64, device=got.device)
            h["rbuf"] = torch.empty(h["recv"][0].numel(), dtype=torch.float64, device=got.device)
        self.halos = live

    def _sendrecv(self, peer, out_t, in_t):
        dist = self.dist
        so, si = (out_t.cpu(), in_t.cpu()) if self.stage_cpu else (out_t, in_t)
        ops = [dist.P2POp(dist.isend, so, peer), dist.P2POp(dist.irecv, si, peer)]
        for r in dist.batch_isend_irecv(ops):
            r.wait()
        if self.stage_cpu:
            in_t.copy_(si)

    def halo_exchange(self, vec):
        dist, torch = self.dist, self.torch
        ops, staged = [], []
        for h in self.halos:
            torch.index_select(vec, 0, h["send"][0], out=h["sbuf"])
            sb, rb = (h["sbuf"].cpu(), h["rbuf"].cpu()) if self.stage_cpu else (h["sbuf"], h["rbuf"])
            staged.append(rb)
            ops.append(dist.P2POp(dist.isend, sb, h["peer"]))
            ops.append(dist.P2POp(dist.irecv, rb, h["peer"]))
        if ops:
            for r in dist.batch_isend_irecv(ops):
                r.wait()
        for h, rb in zip(self.halos, staged):
            if self.stage_cpu:
                h["rbuf"].copy_(rb)
            vec.index_copy_(0, h["recv"][0], h["rbuf"])

    def _allreduce(self, lo, hi):
        if self.world > 1:
            self.dist.all_reduce(self.scal[R_OFF + lo:R_OFF + hi])

    def _allgather_carries(self):
        """Slab-exact preconditioner: all-gather of the tridiagonal carries between its two halves."""
        b = self.b
        if self.world == 1:
            b.carry_recv.copy_(b.carry_send)
        elif self.stage_cpu:
            send = b.carry_send.cpu()
            recv = self.torch.empty(self.world * send.numel(), dtype=send.dtype)
            self.dist.all_gather_into_tensor(recv, send)
            b.carry_recv.copy_(recv)
        else:
            self.dist.all_gather_into_tensor(b.carry_recv, b.carry_send)

    def solve(self, out, profile_spmv=False):
        b = self.b
        if profile_spmv:
            b.profile(True)
        b.synchronize()
        t0 = time.perf_counter()
        b.phase(0)
        self._allreduce(R_RHO, R_RR + 1)     # (b, b) and the preconditioner vetoes (phx_krylov_precond_disable)
        b.phase(1)
        head = self.scal[:16].cpu()
        bb = float(head[S_BB])
        it, relres = 0, (0.0 if bb == 0.0 else 1.0)
        # The preconditioner is a COLLECTIVE choice: with one veto every rank iterates with Jacobi, so that all
        # ranks exchange the same vectors (phat / shat = P p / P s, or p / s) and test convergence at the
        # same iterations.  A rank without owned u rows (an empty slab) does not veto.
        pc = float(head[R_OFF + R_RR]) == 0.0
        if not pc:
            b.precond_disable()
        # the SpMV inputs are phat / shat whenever the system keeps them apart from p / s: box preconditioner,
        # or the u-block Jacobi of a structured system (the same on every rank: all slabs are built alike)
        hat = pc or getattr(b, "precond_active", lambda: False)()
        exact = bool(pc and getattr(b, "exact", False))
        vp, vs = (self.phat, self.shat) if hat else (self.p, self.s)
        # convergence checks as in phx_solve / phx_solve_distributed: with the box preconditioner the next check is
        # scheduled from the observed rate; every rank reads the same all-reduced numbers and schedules the same checks
        import math
        base_step = 2 if pc else self.check_every
        next_check, last_check, last_relres, verifications = base_step, 0, 1.0, 0
        vy = self.work[6 * self.b.n:7 * self.b.n]
        while True:
            while bb != 0.0 and it < self.max_iter:
                if hat:
                    b.phase(7)
                    if exact:
                        self._allgather_carries()
                        b.phase(9)
                self.halo_exchange(vp)
                b.phase(2)
                self._allreduce(R_RV, R_RV + 1)
                b.phase(3)
                if hat:
                    b.phase(8)
                    if exact:
                        self._allgather_carries()
                        b.phase(10)
                self.halo_exchange(vs)
                b.phase(4)
                self._allreduce(R_TS, R_TT + 1)
                b.phase(5)
                self._allreduce(R_RHO, R_RR + 1)
                it += 1
                if it >= next_check or it == self.max_iter:
                    rr = float(self.scal[R_OFF + R_RR].item())
                    if not np.isfinite(rr):
                        raise ArithmeticError(f"BiCGStab breakdown at iteration {it}")
                    relres = (rr / bb) ** 0.5
                    if relres <= self.rtol:
                        break
                    step = base_step
                    if pc and 0.0 < relres < last_relres:
                        rate = math.log(last_relres / relres) / (it - last_check)
                        remaining = math.log(relres / self.rtol) / rate
                        step = max(2, min(12, int(0.5 * remaining))) & ~1
                    last_check, last_relres, next_check = it, relres, it + step
                b.phase(6)
            if bb == 0.0 or not relres <= self.rtol:
                break
            # the recurrences say converged: verify the TRUE residual b - A y, restart from it should it miss rtol
            self.halo_exchange(vy)
            b.phase(11)
            b.phase(12)
            self._allreduce(R_RR, R_RR + 1)
            rr = float(self.scal[R_OFF + R_RR].item())
            if not np.isfinite(rr):
                raise ArithmeticError("non-finite true residual")
            relres = (rr / bb) ** 0.5
            verifications += 1
            if relres <= self.rtol or verifications > 8 or it >= self.max_iter:
                break
            b.phase(13)
            last_check, last_relres, next_check = it, relres, it + 2
        b.finish(out)
        b.synchronize()
        dt = time.perf_counter() - t0
        st = {"iterations": it, "relres": relres, "seconds": dt, "n_owned": self.n_owned,
              "converged": bool(relres <= self.rtol), "precond_all": pc, "precond_exact": exact}
        if profile_spmv:
            prof = b.profile(False)
            if prof:
                st["spmv_avg_s"], st["spmv_timed"] = prof
        return st


WATCHDOG_EXIT_CODE = 87


def dist_timeout_s():
    """The ONE time limit of the multi-GPU path, seconds (0 = off): PHX_DIST_TIMEOUT_S, default 300 -- the variable the
    library reads for the host waits of its loop.  PHIFEM_DIST_TIMEOUT_S (the name rounds 2-3 used on the Python side)
    is accepted and handed on to the library when the new name is not set."""
    import os
    if "PHX_DIST_TIMEOUT_S" not in os.environ and "PHIFEM_DIST_TIMEOUT_S" in os.environ:
        os.environ["PHX_DIST_TIMEOUT_S"] = os.environ["PHIFEM_DIST_TIMEOUT_S"]
    return float(os.environ.get("PHX_DIST_TIMEOUT_S", "300"))


class Watchdog:
    """Bounds a call into RCCL.  A collective whose partner never arrives blocks inside the library where no exception
    can reach it; the rank then says so and EXITS with code 87.  Nothing is re-exec'd: this process has initialised
    the GPU and its stream is wedged.  A launcher that has not touched the GPU (`bench.py --gpus N` started without
    one) may start fresh ranks with PHIFEM_NATIVE_LOOP=0 (bench.py does, once); under an external launcher the
    non-zero exit ends the job instead of hanging it.

    Two kinds of call: communicator set-up and the halo self-test have no bound of their own, so a
    wall-clock limit on the whole call stands around them (`PHX_DIST_TIMEOUT_S`, default 300 s, 0 = off;
    `PHIFEM_DIST_TIMEOUT_S` is read as an older name of the same variable).  `phx_solve_distributed` bounds EVERY host
    wait of its loop itself with the same variable (`stream_sync_watchdog`, phx_dist.inc.hip) -- a limit on PROGRESS, so
    a long but healthy solve is never killed; around it (`bounded_inside=True`) this class runs no timer and only turns
    the library's PHX_ERR_TIMEOUT into the same exit."""

    def __init__(self, what, rank, bounded_inside=False):
        import os
        import threading
        self.what, self.rank = what, rank
        self.limit = 0.0 if bounded_inside else float(dist_timeout_s())
        self._done = threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True) if self.limit > 0 else None

    @staticmethod
    def _leave():
        import os
        mark = os.environ.get("PHIFEM_WATCHDOG_FILE")
        if mark:
            try:
                open(mark, "w").close()
            except OSError:
                pass
        os._exit(WATCHDOG_EXIT_CODE)

    def _run(self):
        import sys
        if not self._done.wait(self.limit):
            print(f"phifem_amd: rank {self.rank}: {self.what} did not return within {self.limit:g} s "
                  f"(PHX_DIST_TIMEOUT_S) -- exiting with code {WATCHDOG_EXIT_CODE}", file=sys.stderr, flush=True)
            self._leave()

    def __enter__(self):
        if self._thread:
            self._thread.start()
        return self

    def __exit__(self, et, ev, tb):
        self._done.set()
        if et is not None and issubclass(et, TimeoutError):
            # PHX_ERR_TIMEOUT: the library's own bound on a host wait fired first -- same situation, same way out
            import sys
            print(f"phifem_amd: rank {self.rank}: {self.what}: {ev} -- exiting with code {WATCHDOG_EXIT_CODE}",
                  file=sys.stderr, flush=True)
            self._leave()
        return False


class DistributedKrylov:
    """Glue between `SlabProblem` (phifem_amd/distributed.py) and the solver loops.

    With the nccl backend the loop runs NATIVELY in the library (`phx_solve_distributed`: RCCL
    send/recv + all-reduce on the solver's stream, no Python in the iteration).  The Python-driven
    `DistributedSolver` is the reference implementation of the same protocol: it is what the gloo
    tests exercise, and the fallback whenever the native path cannot be set up or its halo
    self-test disagrees (decided collectively, so all ranks take the same path)."""

    def __init__(self, prob):
        import torch
        import torch.distributed as dist
        self.prob, self.torch, self.dist = prob, torch, dist
        self.dev = torch.device("cuda", prob.device)
        self.comm = None
        self.native = None   # None: not tried yet, True/False afterwards
        self.path = "python"
        self.library = None  # file the library's RCCL entry points are bound to (native loop)
        self.overlap = False  # halo exchanges overlapped with the SpMV (the self-test has seen that path deliver)

    def agree_on_exterior(self):
        """`len(exterior_cells) == 0` (mesh_scripts.py:469) must be decided over ALL slabs."""
        from . import _lib as L
        hist = (C.c_int64 * 4)()
        L.check(L.lib.phx_mesh_tag_histogram(self.prob.mesh._h, hist, None))
        flag = self.torch.tensor([1 if hist[3] > 0 else 0], dtype=self.torch.int32, device=self._ctl_device())
        self.dist.all_reduce(flag, op=self.dist.ReduceOp.MAX)
        L.check(L.lib.phx_set_option(self.prob.mesh._h, L.OPT_HAS_EXTERIOR, int(flag.item())))

    def _ctl_device(self):
        """Device of the small control tensors: the GPU with nccl, the host with gloo."""
        return self.dev if self.dist.get_backend() == "nccl" else self.torch.device("cpu")

    def _all_ok(self, ok):
        t = self.torch.tensor([1 if ok else 0], dtype=self.torch.int32, device=self._ctl_device())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return bool(t.item())

    def _init_native(self):
        """RCCL communicator of the library: id from rank 0, broadcast with torch.distributed.
        The native loop is the default with the nccl backend; PHIFEM_NATIVE_LOOP=1 asks for it under
        gloo as well (tests: several ranks on one GPU with PHX_RCCL_LIB naming the host-staged stand-in
        of tests/fake_rccl), PHIFEM_NATIVE_LOOP=0 forces the Python loop."""
        import os
        from . import _lib as L
        torch, dist = self.torch, self.dist
        want = os.environ.get("PHIFEM_NATIVE_LOOP", "")
        if want == "0" or (dist.get_backend() != "nccl" and want != "1"):
            self.native = False
            return
        ok = True
        ctl = self._ctl_device()
        uid = torch.zeros(128, dtype=torch.uint8, device=ctl)
        try:
            if self.prob.rank == 0:
                buf = (C.c_ubyte * 128)()
                L.check(L.lib.phx_comm_unique_id(buf))
                uid.copy_(torch.tensor(list(buf), dtype=torch.uint8))
        except Exception:
            ok = False
        if not self._all_ok(ok):
            self.native = False
            return
        dist.broadcast(uid, src=0)
        host = (C.c_ubyte * 128)(*uid.cpu().tolist())
        h = C.c_void_p()
        rc = L.lib.phx_comm_create(self.prob.world, self.prob.rank, host, self.prob.device, C.byref(h))
        ok = rc == 0
        if ok:
            self.comm = h
            buf = C.create_string_buffer(512)
            if L.lib.phx_comm_library(buf, 512) == 0:
                self.library = buf.value.decode(errors="replace")
        self.native = self._all_ok(ok)

    def _halo_arrays(self, ds):
        torch = self.torch
        np_ = len(ds.halos)
        peers = (C.c_int * max(np_, 1))(*[h["peer"] for h in ds.halos])
        counts = (C.c_int64 * max(2 * np_, 1))()
        idx = (C.c_void_p * max(2 * np_, 1))()
        keep = []
        for p, h in enumerate(ds.halos):
            sp, rp = h["send"][0].contiguous(), h["recv"][0].contiguous()
            for t in (sp, rp):  # the kernels index the work vectors with these: never out of range
                if t.numel() and (int(t.min()) < 0 or int(t.max()) >= ds.b.n):
                    raise RuntimeError("halo position out of range")
            keep += [sp, rp]
            counts[2 * p], counts[2 * p + 1] = sp.numel(), rp.numel()
            idx[2 * p], idx[2 * p + 1] = sp.data_ptr(), rp.data_ptr()
        return np_, peers, counts, idx, keep

    def _selftest(self, backend, ds, arrays):
        """Send the global DoF ids through ncclSend/ncclRecv + the pack/unpack kernels and compare
        with the ids this rank expects: proves the native wiring before any solve uses it."""
        from . import _lib as L
        torch = self.torch
        np_, peers, counts, idx, keep = arrays
        vec = torch.full((backend.n,), -1.0, dtype=torch.float64, device=self.dev)
        for h in ds.halos:
            vec[h["send"][0]] = h["send"][1].to(torch.float64)
        torch.cuda.synchronize(self.dev)
        rc = L.lib.phx_halo_selftest(backend.sys, self.comm, np_, peers, counts, idx,
                                     C.c_void_p(vec.data_ptr()))
        ok = rc == 0
        if ok:
            for h in ds.halos:
                ok = ok and bool(torch.equal(vec[h["recv"][0]], h["recv"][1].to(torch.float64)))
        return ok

    def solve(self, out, profile_spmv=False):
        from . import _lib as L
        prob = self.prob
        lay = prob.lay
        backend = HipBackend(prob.solver, self.dev, blocks=getattr(prob, "n_blocks", None))
        backend.use_current_stream()
        L.check(L.lib.phx_set_option(prob.mesh._h, L.OPT_PROFILE_SPMV, int(profile_spmv)))
        plane = getattr(prob, "plane_vertices", None) or (prob.nxy + 1) * (prob.nxy + 1)
        ds = DistributedSolver(backend, self.dist, self.torch, prob.rank, prob.world, plane,
                               lay["k0"], lay["P0"], lay["P1"], lay["k1"] - lay["k0"] + 1,
                               rtol=prob.rtol, max_iter=prob.max_iter)
        # slab-exact preconditioner (collective): PHIFEM_PRECOND_EXACT=0 keeps the rank-local block Jacobi
        import os
        if os.environ.get("PHIFEM_PRECOND_EXACT", "1") != "0":
            n_per = lay["L1"] - lay["L0"]
            zb = [r * n_per for r in range(prob.world)] + [lay["nz"] + 1]
            backend.setup_exact_precond(self.dist, prob.rank, prob.world, zb, stage_cpu=ds.stage_cpu)
        if self.native is None:
            with Watchdog("RCCL communicator set-up + halo self-test", prob.rank):
                self._init_native()
                if self.native:
                    self.native = self._all_ok(self._selftest(backend, ds, self._halo_arrays(ds)))
                if self.native:
                    ov = C.c_int(0)
                    if L.lib.phx_comm_overlap(self.comm, C.byref(ov)) == 0:
                        self.overlap = bool(ov.value)
        if not self.native:
            self.path = "python"
            return ds.solve(out, profile_spmv=profile_spmv)
        self.path = "native"
        np_, peers, counts, idx, keep = self._halo_arrays(ds)
        st = (C.c_double * 8)()
        self.torch.cuda.synchronize(self.dev)
        t0 = time.perf_counter()
        with Watchdog("phx_solve_distributed", prob.rank, bounded_inside=True):
            L.check(L.lib.phx_solve_distributed(backend.sys, self.comm, np_, peers, counts, idx,
                                                float(prob.rtol), int(prob.max_iter),
                                                C.c_void_p(out.data_ptr()), L.DEVICE, st))
            self.torch.cuda.synchronize(self.dev)
        return {"iterations": int(st[0]), "relres": st[1], "seconds": time.perf_counter() - t0,
                "n_owned": ds.n_owned, "spmv_avg_s": st[4], "spmv_timed": int(st[5]),
                "converged": bool(st[6]), "precond_all": bool(st[7]),
                "precond_exact": bool(st[7]) and backend.exact}
