// Native multi-GPU Krylov loop: RCCL (ncclSend/ncclRecv halo + ncclAllReduce of the batched dot
// products) on the solver's HIP stream; included by phx_solve.hip.  RCCL is bound at run time with
// dlopen (the process already holds PyTorch's librccl; a C host may hold /opt/rocm's), so the
// library itself has no link-time dependency on it.  SURVEY 8(e): slabs talk point-to-point to at
// most two neighbours over dedicated xGMI links; the scalar all-reduces carry 1, 2 and 2 doubles.
#include <dlfcn.h>
#include <time.h>

typedef struct { char internal[128]; } phx_nccl_uid;
typedef void *phx_nccl_comm;
struct NcclApi {
  int (*GetUniqueId)(phx_nccl_uid *);
  int (*CommInitRank)(phx_nccl_comm *, int, phx_nccl_uid, int);
  int (*CommDestroy)(phx_nccl_comm);
  int (*Send)(const void *, size_t, int, int, phx_nccl_comm, hipStream_t);
  int (*Recv)(void *, size_t, int, int, phx_nccl_comm, hipStream_t);
  int (*AllReduce)(const void *, void *, size_t, int, int, phx_nccl_comm, hipStream_t);
  int (*AllGather)(const void *, void *, size_t, int, phx_nccl_comm, hipStream_t);
  int (*GroupStart)();
  int (*GroupEnd)();
  const char *(*GetErrorString)(int);
  bool ok = false;
};
static NcclApi g_nccl;
static char g_nccl_path[512] = "";   // file the ten entry points were bound from (phx_comm_library)
enum { PHX_NCCL_FLOAT64 = 8, PHX_NCCL_SUM = 0 };  // ncclDataType_t / ncclRedOp_t values of nccl.h

static int nccl_bind() {
  if (g_nccl.ok) return PHX_OK;
  void *h = nullptr;
  // PHX_RCCL_LIB: an explicit library path (the tests load a host-staged stand-in that lets several ranks share
  // the one GPU of the test box, which RCCL itself refuses)
  const char *names[] = {getenv("PHX_RCCL_LIB"), "librccl.so.1", "librccl.so", "libnccl.so.2"};
  for (const char *n : names) { if (!n || !*n) continue; h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
  PHX_REQUIRE(h != nullptr, PHX_ERR_HIP, "librccl not found: %s", dlerror());
#define BIND(field, sym) \
  *(void **)(&g_nccl.field) = dlsym(h, sym); \
  PHX_REQUIRE(g_nccl.field != nullptr, PHX_ERR_HIP, "librccl lacks %s", sym)
  BIND(GetUniqueId, "ncclGetUniqueId");
  BIND(CommInitRank, "ncclCommInitRank");
  BIND(CommDestroy, "ncclCommDestroy");
  BIND(Send, "ncclSend");
  BIND(Recv, "ncclRecv");
  BIND(AllReduce, "ncclAllReduce");
  BIND(AllGather, "ncclAllGather");
  BIND(GroupStart, "ncclGroupStart");
  BIND(GroupEnd, "ncclGroupEnd");
  BIND(GetErrorString, "ncclGetErrorString");
#undef BIND
  {
    // which file the symbols really come from (a soname says little): the object that holds ncclAllReduce.  Said once on
    // stderr when PHX_RCCL_LIB substituted the collective library, so that a stand-in is never bound silently.
    Dl_info di;
    if (dladdr((void *)g_nccl.AllReduce, &di) && di.dli_fname) snprintf(g_nccl_path, sizeof(g_nccl_path), "%s", di.dli_fname);
    const char *e = getenv("PHX_RCCL_LIB");
    if (e && *e) fprintf(stderr, "phifem_hip: collective library taken from PHX_RCCL_LIB: %s\n", g_nccl_path[0] ? g_nccl_path : e);
  }
  g_nccl.ok = true;
  return PHX_OK;
}

// Path of the library the RCCL entry points are bound to (binds them if that has not happened yet).
extern "C" int phx_comm_library(char *out, int64_t len) {
  PHX_CHECK(nccl_bind());
  PHX_REQUIRE(out != nullptr && len > 0, PHX_ERR_VALUE, "phx_comm_library: no buffer");
  snprintf(out, (size_t)len, "%s", g_nccl_path);
  return PHX_OK;
}

#define PHX_NCCL(expr)                                                                     \
  do {                                                                                     \
    int e_ = (expr);                                                                       \
    if (e_ != 0) {                                                                         \
      phx_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, g_nccl.GetErrorString(e_)); \
      return PHX_ERR_HIP;                                                                  \
    }                                                                                      \
  } while (0)

struct phx_comm {
  phx_nccl_comm comm = nullptr;
  int nranks = 1, rank = 0, device = 0;
  // halo exchanges run on their own stream so that the rows of the SpMV that read no halo entry overlap with them
  hipStream_t cs = nullptr;
  hipEvent_t ev_packed = nullptr, ev_recvd = nullptr;
  // the overlapped exchange (second stream, two events) is used once phx_halo_selftest has seen it deliver, on THIS
  // communicator, what the exchange in series delivers (never validated on hardware with several GPUs before)
  bool overlap_ok = false;
};

extern "C" int phx_comm_unique_id(void *out128) {
  PHX_CHECK(nccl_bind());
  phx_nccl_uid id;
  PHX_NCCL(g_nccl.GetUniqueId(&id));
  memcpy(out128, &id, sizeof(id));
  return PHX_OK;
}

extern "C" int phx_comm_create(int nranks, int rank, const void *uid128, int device, phx_comm **out) {
  PHX_CHECK(nccl_bind());
  PHX_HIP(hipSetDevice(device));
  phx_nccl_uid id;
  memcpy(&id, uid128, sizeof(id));
  phx_comm *c = new phx_comm();
  c->nranks = nranks; c->rank = rank; c->device = device;
  const int e = g_nccl.CommInitRank(&c->comm, nranks, id, rank);
  if (e != 0) {
    phx_set_error("ncclCommInitRank failed: %s", g_nccl.GetErrorString(e));
    delete c;
    return PHX_ERR_HIP;
  }
  if (hipStreamCreateWithFlags(&c->cs, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_packed, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_recvd, hipEventDisableTiming) != hipSuccess) {
    phx_set_error("communication stream / events could not be created");
    g_nccl.CommDestroy(c->comm);
    delete c;
    return PHX_ERR_HIP;
  }
  *out = c;
  return PHX_OK;
}

extern "C" int phx_comm_destroy(phx_comm *c) {
  if (!c) return PHX_OK;
  if (c->cs) { (void)hipStreamSynchronize(c->cs); }
  if (c->comm && g_nccl.ok) g_nccl.CommDestroy(c->comm);
  if (c->ev_packed) (void)hipEventDestroy(c->ev_packed);
  if (c->ev_recvd) (void)hipEventDestroy(c->ev_recvd);
  if (c->cs) (void)hipStreamDestroy(c->cs);
  delete c;
  return PHX_OK;
}

// 1 when the halo exchanges of this communicator run overlapped with the SpMV (self-test passed and PHX_DIST_OVERLAP != 0)
extern "C" int phx_comm_overlap(const phx_comm *c, int *out) {
  PHX_REQUIRE(c != nullptr && out != nullptr, PHX_ERR_VALUE, "phx_comm_overlap: null argument");
  const bool env = !(getenv("PHX_DIST_OVERLAP") && atoi(getenv("PHX_DIST_OVERLAP")) == 0);
  *out = (env && c->overlap_ok) ? 1 : 0;
  return PHX_OK;
}

__global__ void k_halo_pack(int64_t n, const int64_t *__restrict__ idx, const double *__restrict__ v,
                            double *__restrict__ buf) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) buf[i] = v[idx[i]];
}
__global__ void k_halo_unpack(int64_t n, const int64_t *__restrict__ idx, const double *__restrict__ buf,
                              double *__restrict__ v) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) v[idx[i]] = buf[i];
}

// entries that differ bit for bit (NaN-safe) between the vector and a snapshot
__global__ void k_halo_compare(int64_t n, const int64_t *__restrict__ idx, const double *__restrict__ v,
                               const double *__restrict__ snap, unsigned long long *__restrict__ bad) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n && __double_as_longlong(v[idx[i]]) != __double_as_longlong(snap[i])) atomicAdd(bad, 1ull);
}

struct HaloSpec {
  int npeers;
  int peer[2];
  int64_t nsend[2], nrecv[2];
  const int64_t *send_idx[2], *recv_idx[2];
  double *sbuf[2], *rbuf[2];
};

// pack on the solver stream `st`, send / receive on `on` (st itself, or the communicator's stream behind an event)
static int halo_begin(phx_system *s, phx_comm *c, const HaloSpec &H, const double *vec, bool overlap) {
  hipStream_t st = s->mesh->stream;
  for (int p = 0; p < H.npeers; ++p)
    if (H.nsend[p] > 0)
      k_halo_pack<<<dim3((unsigned)phx_div_up(H.nsend[p], 256)), dim3(256), 0, st>>>(
          H.nsend[p], H.send_idx[p], vec, H.sbuf[p]);
  PHX_HIP(hipGetLastError());
  hipStream_t on = st;
  if (overlap) {
    PHX_HIP(hipEventRecord(c->ev_packed, st));
    PHX_HIP(hipStreamWaitEvent(c->cs, c->ev_packed, 0));
    on = c->cs;
  }
  PHX_NCCL(g_nccl.GroupStart());
  for (int p = 0; p < H.npeers; ++p) {
    if (H.nsend[p] > 0) PHX_NCCL(g_nccl.Send(H.sbuf[p], (size_t)H.nsend[p], PHX_NCCL_FLOAT64, H.peer[p], c->comm, on));
    if (H.nrecv[p] > 0) PHX_NCCL(g_nccl.Recv(H.rbuf[p], (size_t)H.nrecv[p], PHX_NCCL_FLOAT64, H.peer[p], c->comm, on));
  }
  PHX_NCCL(g_nccl.GroupEnd());
  if (overlap) PHX_HIP(hipEventRecord(c->ev_recvd, c->cs));
  return PHX_OK;
}
// ... and the unpack on the solver stream, once the receives have landed
static int halo_end(phx_system *s, phx_comm *c, const HaloSpec &H, double *vec, bool overlap) {
  hipStream_t st = s->mesh->stream;
  if (overlap) PHX_HIP(hipStreamWaitEvent(st, c->ev_recvd, 0));
  for (int p = 0; p < H.npeers; ++p)
    if (H.nrecv[p] > 0)
      k_halo_unpack<<<dim3((unsigned)phx_div_up(H.nrecv[p], 256)), dim3(256), 0, st>>>(
          H.nrecv[p], H.recv_idx[p], H.rbuf[p], vec);
  PHX_HIP(hipGetLastError());
  return PHX_OK;
}
static int halo_exchange(phx_system *s, phx_comm *c, const HaloSpec &H, double *vec) {
  PHX_CHECK(halo_begin(s, c, H, vec, false));
  return halo_end(s, c, H, vec, false);
}

// slab-exact preconditioner: the zero-inflow carries of every rank's tridiagonal z recurrences (2 doubles per
// lattice column and rank) between the two halves of an application
static int allgather_carries(phx_system *s, phx_comm *c) {
  phx_box_precond *bp = s->precond;
  const size_t count = (size_t)(2 * bp->g.pitch * bp->g.m[1]);
  PHX_REQUIRE(bp->carry_send && bp->carry_recv, PHX_ERR_VALUE, "carry buffers of the slab-exact preconditioner are not set");
  if (c->nranks == 1) {
    PHX_HIP(hipMemcpyAsync(bp->carry_recv, bp->carry_send, sizeof(double) * count, hipMemcpyDeviceToDevice, s->mesh->stream));
    return PHX_OK;
  }
  PHX_NCCL(g_nccl.AllGather(bp->carry_send, bp->carry_recv, count, PHX_NCCL_FLOAT64, c->comm, s->mesh->stream));
  return PHX_OK;
}

static int allreduce_R(phx_system *s, phx_comm *c, int lo, int hi) {
  if (c->nranks == 1) return PHX_OK;
  double *R = kr_scal(s) + R_OFF + lo;
  PHX_NCCL(g_nccl.AllReduce(R, R, (size_t)(hi - lo), PHX_NCCL_FLOAT64, PHX_NCCL_SUM, c->comm, s->mesh->stream));
  return PHX_OK;
}

// Watchdog of the host synchronisations inside the distributed loop: a collective whose partner never arrives would
// otherwise block in hipStreamSynchronize for ever.  PHX_DIST_TIMEOUT_S (default 300; 0 = wait without limit).
static double dist_timeout_s() {
  static const double t = getenv("PHX_DIST_TIMEOUT_S") ? atof(getenv("PHX_DIST_TIMEOUT_S")) : 300.0;   // (the Python side: dist_solver.dist_timeout_s)
  return t;
}
static int stream_sync_watchdog(hipStream_t st, const char *what) {
  const double limit = dist_timeout_s();
  if (!(limit > 0.0)) { PHX_HIP(hipStreamSynchronize(st)); return PHX_OK; }
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (long spins = 0;; ++spins) {
    const hipError_t e = hipStreamQuery(st);
    if (e == hipSuccess) return PHX_OK;
    if (e != hipErrorNotReady) { phx_set_error("%s: %s", what, hipGetErrorString(e)); return PHX_ERR_HIP; }
    if ((spins & 1023) == 1023) {
      clock_gettime(CLOCK_MONOTONIC, &t1);
      const double dt = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
      if (dt > limit) {
        phx_set_error("%s: the stream did not drain within %g s -- a collective is waiting for a rank that never "
                      "arrived (PHX_DIST_TIMEOUT_S)", what, limit);
        return PHX_ERR_TIMEOUT;
      }
      if (dt > 0.01) { struct timespec nap = {0, 50000}; nanosleep(&nap, nullptr); }
    }
  }
}

// peers[npeers], counts[2*npeers] = {nsend, nrecv} per peer, idx[2*npeers] device pointers
// {send_idx, recv_idx} (int64 solver positions), work/scal/own as phx_krylov_attach (already
// attached).  stats[8] as phx_solve: relres / converged refer to the TRUE residual b - A x (verified with one more
// halo exchange + SpMV + all-reduce when the recurrences announce convergence, restart from it when it misses rtol);
// stats[7]: 1 = every rank kept the box preconditioner.
// PHX_DIST_OVERLAP=0: halo exchanges in series on the solver stream (default: overlapped with the rows of the SpMV
// that read no halo entry, on the communicator's own stream -- on a communicator whose phx_halo_selftest has seen the
// overlapped exchange deliver the same entries as the exchange in series; without a self-test: in series).
extern "C" int phx_solve_distributed(phx_system *s, phx_comm *c, int npeers, const int *peers,
                                     const int64_t *counts, const int64_t *const *idx, double rtol,
                                     int64_t max_iter, double *x_out, int loc, double *stats) {
  phx_mesh *m = s->mesh;
  PHX_HIP(hipSetDevice(m->device));
  PHX_REQUIRE(npeers >= 0 && npeers <= 2, PHX_ERR_VALUE, "a slab has at most two neighbours");
  hipStream_t st = m->stream;
  HaloSpec H;
  memset(&H, 0, sizeof(H));
  H.npeers = npeers;
  for (int p = 0; p < npeers; ++p) {
    H.peer[p] = peers[p];
    H.nsend[p] = counts[2 * p]; H.nrecv[p] = counts[2 * p + 1];
    H.send_idx[p] = idx[2 * p]; H.recv_idx[p] = idx[2 * p + 1];
    PHX_HIP(phx_malloc(&H.sbuf[p], sizeof(double) * (size_t)(H.nsend[p] > 0 ? H.nsend[p] : 1)));
    PHX_HIP(phx_malloc(&H.rbuf[p], sizeof(double) * (size_t)(H.nrecv[p] > 0 ? H.nrecv[p] : 1)));
  }
  static const bool overlap_env = !(getenv("PHX_DIST_OVERLAP") && atoi(getenv("PHX_DIST_OVERLAP")) == 0);
  // a local matter: sends and receives pair up whatever stream each side issues them on
  const bool overlap = overlap_env && c->overlap_ok && c->nranks > 1 && npeers > 0 && c->cs != nullptr;
  double *S = kr_scal(s);
  int rc = PHX_OK;
  auto body = [&]() -> int {
    if (overlap) {
      const int64_t *rl[2] = {H.recv_idx[0], H.recv_idx[1]};
      const int64_t rn[2] = {H.nrecv[0], H.nrecv[1]};
      PHX_CHECK(phx_spmv_flag_rows(s, npeers, rl, rn));
    }
    PHX_CHECK(prof_reset(s));
    PHX_CHECK(phx_begin_timing(m));
    PHX_CHECK(phx_krylov_phase(s, 0));
    PHX_CHECK(allreduce_R(s, c, R_RHO, R_RR + 1));  // (b, b) and the preconditioner vetoes
    PHX_CHECK(phx_krylov_phase(s, 1));
    PHX_HIP(hipMemcpyAsync(s->scal_h, S, sizeof(double) * 16, hipMemcpyDeviceToHost, st));
    PHX_CHECK(stream_sync_watchdog(st, "start of the distributed solve"));
    // the preconditioner is a COLLECTIVE choice: one veto and every rank iterates with Jacobi, so that all
    // ranks exchange the same vectors and test convergence at the same iterations
    const bool pc_all = s->scal_h[R_OFF + R_RR] == 0.0;
    if (!pc_all) PHX_CHECK(phx_krylov_precond_disable(s));
    // interface elasticity: the coarse correction on top of the vertex blocks, built collectively (phx_coarse.inc.hip);
    // a rank without rows of its own still joins the reductions
    if (s->el_nblk > 0 && m->precond != 0 && !s->cc && !s->cc_tried) {
      s->cc_tried = true;
      const CcReduce red = [&](double *buf, size_t count) -> int {
        if (c->nranks == 1) return PHX_OK;
        PHX_NCCL(g_nccl.AllReduce(buf, buf, count, PHX_NCCL_FLOAT64, PHX_NCCL_SUM, c->comm, st));
        return PHX_OK;
      };
      PHX_CHECK(coarse_build(s, s->el_nblk, &s->cc, &red));
    }
    const bool coarse = s->cc != nullptr && s->cc->dist;
    auto coarse_step = [&](int ph_restrict, int ph_add) -> int {
      PHX_CHECK(phx_krylov_phase(s, ph_restrict));
      if (c->nranks > 1) PHX_NCCL(g_nccl.AllReduce(s->cc->gc, s->cc->gc, (size_t)s->cc->nc, PHX_NCCL_FLOAT64, PHX_NCCL_SUM, c->comm, st));
      return phx_krylov_phase(s, ph_add);
    };
    const KrVecs V = kr_vecs(s);  // after the vote: the preconditioner decides where phat / shat live
    const bool exact = s->precond_state == 1 && s->precond->dist;   // the same on every rank (set up from all-reduced numbers)
    const double bb = s->scal_h[S_BB];
    // convergence checks as in phx_solve: scheduled from the observed rate (every rank reads the same all-reduced
    // numbers, so every rank schedules the same checks)
    int64_t it = 0, spmvs = 0, next_check = pc_all ? 2 : 8, last_check = 0;
    double relres = bb == 0.0 ? 0.0 : 1.0, last_relres = 1.0;
    int verifications = 0;
    for (;;) {
      while (bb != 0.0 && it < max_iter) {
        PHX_CHECK(phx_krylov_phase(s, 7));
        if (exact) { PHX_CHECK(allgather_carries(s, c)); PHX_CHECK(phx_krylov_phase(s, 9)); }
        if (coarse) PHX_CHECK(coarse_step(30, 31));
        if (overlap) {
          PHX_CHECK(halo_begin(s, c, H, V.phat, true));
          PHX_CHECK(phx_krylov_phase(s, 20));
          PHX_CHECK(halo_end(s, c, H, V.phat, true));
          PHX_CHECK(phx_krylov_phase(s, 21));
        } else {
          PHX_CHECK(halo_exchange(s, c, H, V.phat));
          PHX_CHECK(phx_krylov_phase(s, 2));
        }
        PHX_CHECK(allreduce_R(s, c, R_RV, R_RV + 1));
        PHX_CHECK(phx_krylov_phase(s, 3));
        PHX_CHECK(phx_krylov_phase(s, 8));
        if (exact) { PHX_CHECK(allgather_carries(s, c)); PHX_CHECK(phx_krylov_phase(s, 10)); }
        if (coarse) PHX_CHECK(coarse_step(32, 33));
        if (overlap) {
          PHX_CHECK(halo_begin(s, c, H, V.shat, true));
          PHX_CHECK(phx_krylov_phase(s, 40));
          PHX_CHECK(halo_end(s, c, H, V.shat, true));
          PHX_CHECK(phx_krylov_phase(s, 41));
        } else {
          PHX_CHECK(halo_exchange(s, c, H, V.shat));
          PHX_CHECK(phx_krylov_phase(s, 4));
        }
        PHX_CHECK(allreduce_R(s, c, R_TS, R_TT + 1));
        PHX_CHECK(phx_krylov_phase(s, 5));
        PHX_CHECK(allreduce_R(s, c, R_RHO, R_RR + 1));
        spmvs += 2;
        ++it;
        if (it >= next_check || it == max_iter) {
          PHX_HIP(hipMemcpyAsync(s->scal_h, S, sizeof(double) * 16, hipMemcpyDeviceToHost, st));
          PHX_CHECK(stream_sync_watchdog(st, "convergence check of the distributed solve"));
          const double rr = s->scal_h[R_OFF + R_RR];
          relres = sqrt(rr / bb);
          if (!(rr == rr) || !(fabs(rr) <= 1.0e300)) {
            phx_set_error("BiCGStab breakdown at iteration %lld (rr=%g)", (long long)it, rr);
            return PHX_ERR_BREAKDOWN;
          }
          if (relres <= rtol) break;
          int64_t step = pc_all ? 2 : 8;
          if (pc_all && relres < last_relres && relres > 0.0) {
            const double rate = log(last_relres / relres) / (double)(it - last_check);
            const double remaining = log(relres / rtol) / rate;
            step = std::max<int64_t>(2, std::min<int64_t>(12, (int64_t)(0.5 * remaining)));
            step &= ~(int64_t)1;
          }
          last_check = it;
          last_relres = relres;
          next_check = it + step;
        }
        PHX_CHECK(phx_krylov_phase(s, 6));
      }
      if (bb == 0.0 || !(relres <= rtol)) break;
      // the recurrences say converged: verify b - A y (one halo exchange, one SpMV, one all-reduce), restart from it
      // should it miss the tolerance
      PHX_CHECK(halo_exchange(s, c, H, V.y));
      PHX_CHECK(phx_krylov_phase(s, 11));
      PHX_CHECK(phx_krylov_phase(s, 12));
      PHX_CHECK(allreduce_R(s, c, R_RR, R_RR + 1));
      PHX_HIP(hipMemcpyAsync(s->scal_h, S, sizeof(double) * 16, hipMemcpyDeviceToHost, st));
      PHX_CHECK(stream_sync_watchdog(st, "true-residual check of the distributed solve"));
      spmvs += 1;
      const double rr_true = s->scal_h[R_OFF + R_RR];
      if (!(rr_true == rr_true)) { phx_set_error("non-finite true residual"); return PHX_ERR_BREAKDOWN; }
      relres = sqrt(rr_true / bb);
      if (relres <= rtol || ++verifications > 8 || it >= max_iter) break;
      PHX_CHECK(phx_krylov_phase(s, 13));
      last_relres = relres;
      last_check = it;
      next_check = it + 2;
    }
    PHX_CHECK(phx_krylov_finish(s, x_out, loc));
    PHX_CHECK(phx_end_timing(m, 3));
    double pavg = 0.0;
    int pcount = 0;
    PHX_CHECK(prof_collect(s, &pavg, &pcount));
    if (stats) {
      stats[0] = (double)it; stats[1] = relres; stats[2] = m->timings[3];
      stats[3] = (double)spmvs; stats[4] = pavg; stats[5] = (double)pcount;
      stats[6] = relres <= rtol ? 1.0 : 0.0; stats[7] = pc_all ? 1.0 : 0.0;
    }
    return PHX_OK;
  };
  rc = body();
  if (rc != PHX_ERR_TIMEOUT) {   // a wedged stream would block here for ever
    (void)hipStreamSynchronize(st);
    if (c->cs) (void)hipStreamSynchronize(c->cs);
    for (int p = 0; p < npeers; ++p) { (void)phx_free(H.sbuf[p]); (void)phx_free(H.rbuf[p]); }
  }
  return rc;
}

// Wiring self-test: every rank sends each peer `tags` gathered at its send positions and receives
// into its recv positions; used by the host to prove both sides enumerate the same DoFs through
// the SAME code path the solver uses (ncclSend/ncclRecv + pack/unpack kernels).
extern "C" int phx_halo_selftest(phx_system *s, phx_comm *c, int npeers, const int *peers,
                                 const int64_t *counts, const int64_t *const *idx, double *vec) {
  PHX_HIP(hipSetDevice(s->mesh->device));
  HaloSpec H;
  memset(&H, 0, sizeof(H));
  H.npeers = npeers;
  for (int p = 0; p < npeers; ++p) {
    H.peer[p] = peers[p];
    H.nsend[p] = counts[2 * p]; H.nrecv[p] = counts[2 * p + 1];
    H.send_idx[p] = idx[2 * p]; H.recv_idx[p] = idx[2 * p + 1];
    PHX_HIP(phx_malloc(&H.sbuf[p], sizeof(double) * (size_t)(H.nsend[p] > 0 ? H.nsend[p] : 1)));
    PHX_HIP(phx_malloc(&H.rbuf[p], sizeof(double) * (size_t)(H.nrecv[p] > 0 ? H.nrecv[p] : 1)));
  }
  int rc = halo_exchange(s, c, H, vec);
  if (rc == PHX_OK) rc = stream_sync_watchdog(s->mesh->stream, "halo self-test");
  // ... and once more through the OVERLAPPED path: the received entries are cleared, exchanged again on the
  // communicator's stream behind the two events, and must come back bit for bit.  Every rank makes both exchanges (they
  // pair up), each decides for itself (`overlap_ok`: a local matter, see phx_solve_distributed).
  if (rc == PHX_OK && c->nranks > 1 && npeers > 0 && c->cs != nullptr) {
    hipStream_t st = s->mesh->stream;
    double *snap[2] = {nullptr, nullptr};
    unsigned long long *bad = nullptr;
    bool ok = phx_malloc(&bad, sizeof(unsigned long long)) == hipSuccess;
    for (int p = 0; p < npeers && ok; ++p) ok = phx_malloc(&snap[p], sizeof(double) * (size_t)(H.nrecv[p] > 0 ? H.nrecv[p] : 1)) == hipSuccess;
    if (ok) {
      (void)hipMemsetAsync(bad, 0, sizeof(unsigned long long), st);
      for (int p = 0; p < npeers; ++p)
        if (H.nrecv[p] > 0) {
          const dim3 g((unsigned)phx_div_up(H.nrecv[p], 256)), b(256);
          k_halo_pack<<<g, b, 0, st>>>(H.nrecv[p], H.recv_idx[p], vec, snap[p]);
          (void)hipMemsetAsync(H.rbuf[p], 0xff, sizeof(double) * (size_t)H.nrecv[p], st);   // NaNs: a stale buffer shows
          k_halo_unpack<<<g, b, 0, st>>>(H.nrecv[p], H.recv_idx[p], H.rbuf[p], vec);
        }
      rc = halo_begin(s, c, H, vec, true);
      if (rc == PHX_OK) rc = halo_end(s, c, H, vec, true);
      for (int p = 0; p < npeers && rc == PHX_OK; ++p)
        if (H.nrecv[p] > 0)
          k_halo_compare<<<dim3((unsigned)phx_div_up(H.nrecv[p], 256)), dim3(256), 0, st>>>(H.nrecv[p], H.recv_idx[p], vec, snap[p], bad);
      unsigned long long hbad = 1;
      if (rc == PHX_OK) rc = hipMemcpyAsync(&hbad, bad, sizeof(hbad), hipMemcpyDeviceToHost, st) == hipSuccess ? PHX_OK : PHX_ERR_HIP;
      if (rc == PHX_OK) rc = stream_sync_watchdog(st, "halo self-test (overlapped exchange)");
      if (rc == PHX_OK && c->cs) rc = stream_sync_watchdog(c->cs, "halo self-test (communication stream)");
      c->overlap_ok = rc == PHX_OK && hbad == 0;
      if (rc == PHX_OK && hbad != 0) {
        // the caller checks the entries of the exchange in series: put them back
        for (int p = 0; p < npeers; ++p)
          if (H.nrecv[p] > 0)
            k_halo_unpack<<<dim3((unsigned)phx_div_up(H.nrecv[p], 256)), dim3(256), 0, st>>>(H.nrecv[p], H.recv_idx[p], snap[p], vec);
        (void)hipStreamSynchronize(st);
      }
      if (rc == PHX_OK && hbad != 0)
        fprintf(stderr, "phifem_hip: rank %d: the overlapped halo exchange delivered %llu entries that differ from the exchange in "
                        "series -- halo exchanges stay on the solver stream\n", c->rank, hbad);
    }
    (void)phx_free(bad);
    for (int p = 0; p < npeers; ++p) (void)phx_free(snap[p]);
  }
  (void)hipStreamSynchronize(s->mesh->stream);
  for (int p = 0; p < npeers; ++p) { (void)phx_free(H.sbuf[p]); (void)phx_free(H.rbuf[p]); }
  return rc;
}
