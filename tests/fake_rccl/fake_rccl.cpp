// fake_rccl.cpp -- TEST INFRASTRUCTURE (never shipped, never loaded by the product unless PHX_RCCL_LIB names it).
//
// A host-staged stand-in for the ten RCCL entry points phifem_amd/csrc/phx_dist.inc.hip binds with dlopen.
// RCCL refuses two ranks on one device, and the test box has exactly one GPU, so the library's NATIVE
// multi-GPU loop (phx_solve_distributed: pack kernel -> ncclSend/ncclRecv group -> unpack kernel,
// ncclAllReduce of the batched dot products, ncclAllGather of the preconditioner carries) could otherwise only
// ever run with a one-rank communicator.  With this library every rank is a process sharing the one GPU;
// "communication" goes through a POSIX shared-memory segment.
//
// Stream semantics (round 3): like RCCL, every call only ENQUEUES work on the caller's stream and returns.
//   send       = async D2H copy into pinned staging -> host function (hipLaunchHostFunc) that moves the staging
//                into the mailbox of the ordered pair
//   recv       = host function that waits for the mailbox and moves it into pinned staging -> async H2D copy
//   all-reduce = async D2H -> host function (deposit, barrier, sum in rank order, barrier) -> async H2D
//   all-gather = async D2H -> host function (deposit, barrier, collect, barrier) -> async H2D
// Nothing is complete when the call returns, so a consumer that forgets a stream dependency (a buffer reused
// before the enqueued send has run, a kernel on another stream without an event, a host read without a
// synchronisation) reads stale data here exactly as it would with RCCL.  PHX_FAKE_RCCL_SYNC=1 restores the
// round-2 behaviour (every call complete on return), for bisecting.
// Every wait inside a host function is bounded (120 s) and aborts the process with a message.
//
// build: hipcc -O2 -fPIC -shared -o libfake_rccl.so fake_rccl.cpp -lrt   (tests/fake_rccl/Makefile)
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <mutex>
#include <vector>

typedef struct { char internal[128]; } fake_uid;

#define MAXR 8
#define MBOX_BYTES (32u << 20)  // per ordered pair; pages are only committed when touched (1024 x 1024 slabs: 9.4 MB of carries, 17 MB of halo)
#define RED_MAX 64
#define WAIT_LIMIT_S 120.0

struct Shared {
  std::atomic<int> attached;
  std::atomic<int> bar_count;
  std::atomic<int> bar_sense;
  std::atomic<unsigned> full[MAXR][MAXR];      // mailbox src -> dst holds `full - 1` bytes (0: free)
  double red[MAXR][RED_MAX];
  char mbox[MAXR][MAXR][MBOX_BYTES];
};

struct FakeComm {
  Shared *sh = nullptr;
  int nranks = 1, rank = 0;
  int local_sense = 0;     // touched by host functions only (they run one at a time per process and stream order)
  char name[64];
};

static bool g_sync = getenv("PHX_FAKE_RCCL_SYNC") && atoi(getenv("PHX_FAKE_RCCL_SYNC")) != 0;

static double now_s() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
static void nap() { struct timespec ts = {0, 20000}; nanosleep(&ts, nullptr); }
static int fail(const char *what) { fprintf(stderr, "fake_rccl: %s\n", what); return 1; }
[[noreturn]] static void die(const char *what) { fprintf(stderr, "fake_rccl: %s -- aborting\n", what); fflush(stderr); abort(); }

template <typename F>
static void wait_until(F cond, const char *what) {
  const double t0 = now_s();
  while (!cond()) {
    nap();
    if (now_s() - t0 > WAIT_LIMIT_S) die(what);
  }
}

static void barrier(FakeComm *c) {
  Shared *s = c->sh;
  c->local_sense ^= 1;
  const int sense = c->local_sense;
  if (s->bar_count.fetch_add(1) + 1 == c->nranks) {
    s->bar_count.store(0);
    s->bar_sense.store(sense);
  } else {
    wait_until([&] { return s->bar_sense.load() == sense; }, "barrier: a rank never arrived");
  }
}

static size_t dtype_bytes(int dt) { return dt == 8 ? 8 : (dt == 7 ? 4 : (dt == 2 || dt == 3 ? 4 : (dt == 4 || dt == 5 ? 8 : 1))); }

// ---- pinned staging blocks, recycled once the stream has passed the event recorded behind their last use ----------
struct Block { void *p; size_t cap; hipEvent_t ev; int state; };   // 0 free, 1 handed out (no event yet), 2 waiting for its event
static std::mutex g_pool_mu;
static std::vector<Block> g_pool;

static void *stage_get(size_t bytes, size_t *slot) {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  for (size_t i = 0; i < g_pool.size(); ++i) {
    Block &b = g_pool[i];
    if (b.state == 2 && hipEventQuery(b.ev) == hipSuccess) b.state = 0;
    if (b.state == 0 && b.cap >= bytes && b.cap <= 4 * (bytes < 256 ? 256 : bytes)) { b.state = 1; *slot = i; return b.p; }
  }
  Block b{nullptr, bytes < 256 ? 256 : bytes, nullptr, 1};
  if (hipHostMalloc(&b.p, b.cap, hipHostMallocDefault) != hipSuccess) die("hipHostMalloc of a staging block");
  if (hipEventCreateWithFlags(&b.ev, hipEventDisableTiming) != hipSuccess) die("hipEventCreate");
  g_pool.push_back(b);
  *slot = g_pool.size() - 1;
  return b.p;
}
static void stage_release_after(size_t slot, hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  if (hipEventRecord(g_pool[slot].ev, st) != hipSuccess) die("hipEventRecord");
  g_pool[slot].state = 2;
}

// ---- host functions ------------------------------------------------------------------------------------------------
struct Ctx {
  int kind;            // 0 send, 1 recv, 2 all-reduce, 3 all-gather
  FakeComm *c;
  int peer;
  size_t bytes, count;
  char *in, *out;      // pinned staging
};

static void host_fn(void *arg) {
  Ctx *x = (Ctx *)arg;
  FakeComm *c = x->c;
  Shared *s = c->sh;
  if (x->kind == 0) {
    wait_until([&] { return s->full[c->rank][x->peer].load() == 0; }, "send: the peer never emptied the mailbox");
    memcpy(s->mbox[c->rank][x->peer], x->in, x->bytes);
    s->full[c->rank][x->peer].store((unsigned)x->bytes + 1u);
  } else if (x->kind == 1) {
    unsigned f = 0;
    wait_until([&] { return (f = s->full[x->peer][c->rank].load()) != 0; }, "recv: the peer never sent");
    if ((size_t)(f - 1u) != x->bytes) die("send / recv size mismatch");
    memcpy(x->out, s->mbox[x->peer][c->rank], x->bytes);
    s->full[x->peer][c->rank].store(0);
  } else if (x->kind == 2) {
    const double *mine = (const double *)x->in;
    double *out = (double *)x->out;
    if (x->count <= RED_MAX) {
      for (size_t i = 0; i < x->count; ++i) s->red[c->rank][i] = mine[i];
      barrier(c);
      for (size_t i = 0; i < x->count; ++i) {
        double sum = 0.0;
        for (int r = 0; r < c->nranks; ++r) sum += s->red[r][i];   // rank order: every rank obtains the same bits
        out[i] = sum;
      }
      barrier(c);   // nobody overwrites its deposit before everyone has read it
    } else {
      // long vectors (the coarse matrix of the elasticity solve): through the rank's own mailbox, a chunk at a time
      const size_t chunk = MBOX_BYTES / sizeof(double);
      for (size_t o = 0; o < x->count; o += chunk) {
        const size_t k = x->count - o < chunk ? x->count - o : chunk;
        memcpy(s->mbox[c->rank][c->rank], mine + o, k * sizeof(double));
        barrier(c);
        for (size_t i = 0; i < k; ++i) {
          double sum = 0.0;
          for (int r = 0; r < c->nranks; ++r) sum += ((const double *)s->mbox[r][r])[i];
          out[o + i] = sum;
        }
        barrier(c);
      }
    }
  } else {
    memcpy(s->mbox[c->rank][c->rank], x->in, x->bytes);
    barrier(c);
    for (int r = 0; r < c->nranks; ++r) memcpy(x->out + (size_t)r * x->bytes, s->mbox[r][r], x->bytes);
    barrier(c);
  }
  delete x;
}

// which HIP call failed, with the runtime's own words and the operands (a launch error of the CALLER's earlier kernel
// surfaces in whatever HIP call comes next: the message names it)
static int fail_hip(const char *what, hipError_t e, const void *p, size_t bytes) {
  fprintf(stderr, "fake_rccl: %s: %s (pointer %p, %zu bytes)\n", what, hipGetErrorString(e), p, bytes);
  return 1;
}
static int enqueue(Ctx *x, const void *src, void *dst, size_t in_bytes, size_t out_bytes, hipStream_t st) {
  size_t si = 0, so = 0;
  hipError_t e;
  if (in_bytes) {
    x->in = (char *)stage_get(in_bytes, &si);
    if ((e = hipMemcpyAsync(x->in, src, in_bytes, hipMemcpyDeviceToHost, st)) != hipSuccess) return fail_hip("D2H", e, src, in_bytes);
  }
  if (out_bytes) x->out = (char *)stage_get(out_bytes, &so);
  // host_fn deletes its context when it is done, and it may be done before the next line runs: nothing of *x is read
  // after the launch (reading x->out there was a use-after-free that failed one H2D copy in a few hundred runs)
  char *const out = x->out;
  if ((e = hipLaunchHostFunc(st, host_fn, x)) != hipSuccess) return fail_hip("hipLaunchHostFunc", e, x, 0);
  if (out_bytes && (e = hipMemcpyAsync(dst, out, out_bytes, hipMemcpyHostToDevice, st)) != hipSuccess) return fail_hip("H2D", e, dst, out_bytes);
  if (in_bytes) stage_release_after(si, st);
  if (out_bytes) stage_release_after(so, st);
  if (g_sync && hipStreamSynchronize(st) != hipSuccess) return fail("stream sync");
  return 0;
}

struct Op { int kind; void *buf; size_t bytes; int peer; FakeComm *c; hipStream_t st; };
static thread_local int g_depth = 0;
static thread_local std::vector<Op> g_ops;

static int run_ops(std::vector<Op> &ops) {
  // all sends first (a mailbox per ordered pair: a send never waits for the peer's recv of the SAME exchange), then
  // the receives
  for (int kind = 0; kind < 2; ++kind)
    for (auto &o : ops) {
      if (o.kind != kind) continue;
      if (o.bytes > MBOX_BYTES) return fail("message larger than the mailbox");
      Ctx *x = new Ctx{kind, o.c, o.peer, o.bytes, 0, nullptr, nullptr};
      const int rc = kind == 0 ? enqueue(x, o.buf, nullptr, o.bytes, 0, o.st) : enqueue(x, nullptr, o.buf, 0, o.bytes, o.st);
      if (rc) return rc;
    }
  return 0;
}

extern "C" {

int ncclGetUniqueId(fake_uid *id) {
  memset(id, 0, sizeof(*id));
  struct timespec ts;
  clock_gettime(CLOCK_REALTIME, &ts);
  snprintf(id->internal, sizeof(id->internal), "/phxfake_%d_%ld", (int)getpid(), (long)ts.tv_nsec);
  return 0;
}

int ncclCommInitRank(void **comm, int nranks, fake_uid id, int rank) {
  if (nranks > MAXR) return fail("too many ranks");
  FakeComm *c = new FakeComm();
  c->nranks = nranks; c->rank = rank;
  strncpy(c->name, id.internal, sizeof(c->name) - 1);
  int fd = -1;
  if (rank == 0) {
    fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, sizeof(Shared)) != 0) return fail("shm create");
  } else {
    for (int tries = 0; tries < 100000 && fd < 0; ++tries) {
      fd = shm_open(c->name, O_RDWR, 0600);
      struct stat sb;
      if (fd >= 0 && (fstat(fd, &sb) != 0 || (size_t)sb.st_size < sizeof(Shared))) { close(fd); fd = -1; }
      if (fd < 0) nap();
    }
    if (fd < 0) return fail("shm open");
  }
  void *p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return fail("mmap");
  c->sh = (Shared *)p;   // a fresh segment is zero-filled: counters, senses and flags start at 0
  c->sh->attached.fetch_add(1);
  wait_until([&] { return c->sh->attached.load() >= nranks; }, "communicator set-up: a rank never attached");
  barrier(c);
  *comm = c;
  return 0;
}

int ncclCommDestroy(void *comm) {
  FakeComm *c = (FakeComm *)comm;
  if (!c) return 0;
  (void)hipDeviceSynchronize();   // no host function of this communicator is left in any stream
  barrier(c);
  munmap(c->sh, sizeof(Shared));
  if (c->rank == 0) shm_unlink(c->name);
  delete c;
  return 0;
}

int ncclGroupStart() { ++g_depth; return 0; }
int ncclGroupEnd() {
  if (--g_depth > 0) return 0;
  const int rc = run_ops(g_ops);
  g_ops.clear();
  return rc;
}

int ncclSend(const void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t st) {
  g_ops.push_back(Op{0, (void *)buf, count * dtype_bytes(dtype), peer, (FakeComm *)comm, st});
  if (g_depth == 0) { const int rc = run_ops(g_ops); g_ops.clear(); return rc; }
  return 0;
}
int ncclRecv(void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t st) {
  g_ops.push_back(Op{1, buf, count * dtype_bytes(dtype), peer, (FakeComm *)comm, st});
  if (g_depth == 0) { const int rc = run_ops(g_ops); g_ops.clear(); return rc; }
  return 0;
}

// f64 SUM only (what the solver uses)
int ncclAllReduce(const void *send, void *recv, size_t count, int dtype, int op, void *comm, hipStream_t st) {
  if (dtype != 8 || op != 0) return fail("all-reduce: only f64 SUM");
  Ctx *x = new Ctx{2, (FakeComm *)comm, -1, count * 8, count, nullptr, nullptr};
  return enqueue(x, send, recv, count * 8, count * 8, st);
}

// every rank deposits its block in its own diagonal mailbox, all ranks read all blocks
int ncclAllGather(const void *send, void *recv, size_t count, int dtype, void *comm, hipStream_t st) {
  FakeComm *c = (FakeComm *)comm;
  const size_t bytes = count * dtype_bytes(dtype);
  if (bytes > MBOX_BYTES) return fail("all-gather block larger than the mailbox");
  Ctx *x = new Ctx{3, c, -1, bytes, count, nullptr, nullptr};
  return enqueue(x, send, recv, bytes, bytes * (size_t)c->nranks, st);
}

const char *ncclGetErrorString(int e) { return e == 0 ? "ok" : "fake_rccl failure (see stderr)"; }

}  // extern "C"
