// fake_rccl.cpp -- TEST INFRASTRUCTURE (never shipped, never loaded by the product unless PHX_RCCL_LIB names it).
//
// A host-staged stand-in for the ten RCCL entry points phifem_amd/csrc/phx_dist.inc.hip binds with dlopen.
// RCCL refuses two ranks on one device, and the test box has exactly one GPU, so the library's NATIVE
// multi-GPU loop (phx_solve_distributed: pack kernel -> ncclSend/ncclRecv group -> unpack kernel,
// ncclAllReduce of the batched dot products) could otherwise only ever run with a one-rank communicator.
// With this library every rank is a process sharing the one GPU; "communication" goes through a POSIX
// shared-memory segment: send = stream sync + D2H copy into a mailbox, recv = wait + H2D copy, all-reduce =
// every rank deposits its values, all ranks add them in rank order (so every rank obtains the same bits).
// Stream semantics: each call completes on the host before it returns, which is stronger than RCCL's
// enqueue-on-stream contract, so everything the loop enqueues afterwards sees the data.
//
// build: hipcc -O2 -fPIC -shared -o libfake_rccl.so fake_rccl.cpp -lrt   (tests/fake_rccl/Makefile)
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <vector>

typedef struct { char internal[128]; } fake_uid;

#define MAXR 8
#define MBOX_BYTES (8u << 20)   // per ordered pair; pages are only committed when touched
#define RED_MAX 64

struct Shared {
  std::atomic<int> attached;
  std::atomic<int> bar_count;
  std::atomic<int> bar_sense;
  std::atomic<unsigned> full[MAXR][MAXR];      // mailbox src -> dst holds `full` bytes (0: free)
  double red[MAXR][RED_MAX];
  char mbox[MAXR][MAXR][MBOX_BYTES];
};

struct FakeComm {
  Shared *sh = nullptr;
  int nranks = 1, rank = 0;
  int local_sense = 0;
  char name[64];
};

struct Op { int kind; void *buf; size_t bytes; int peer; FakeComm *c; hipStream_t st; };
static thread_local int g_depth = 0;
static thread_local std::vector<Op> g_ops;

static void nap() { struct timespec ts = {0, 20000}; nanosleep(&ts, nullptr); }

static int fail(const char *what) { fprintf(stderr, "fake_rccl: %s\n", what); return 1; }

static void barrier(FakeComm *c) {
  Shared *s = c->sh;
  c->local_sense ^= 1;
  if (s->bar_count.fetch_add(1) + 1 == c->nranks) {
    s->bar_count.store(0);
    s->bar_sense.store(c->local_sense);
  } else {
    while (s->bar_sense.load() != c->local_sense) nap();
  }
}

static size_t dtype_bytes(int dt) { return dt == 8 ? 8 : (dt == 7 ? 4 : (dt == 2 || dt == 3 ? 4 : (dt == 4 || dt == 5 ? 8 : 1))); }

static int run_ops(std::vector<Op> &ops) {
  if (ops.empty()) return 0;
  if (hipStreamSynchronize(ops[0].st) != hipSuccess) return fail("stream sync");
  // all sends first (a mailbox per ordered pair: never blocks on the peer's recv order), then the receives
  for (auto &o : ops) {
    if (o.kind != 0) continue;
    if (o.bytes > MBOX_BYTES) return fail("message larger than the mailbox");
    Shared *s = o.c->sh;
    while (s->full[o.c->rank][o.peer].load() != 0) nap();
    if (hipMemcpy(s->mbox[o.c->rank][o.peer], o.buf, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return fail("D2H");
    s->full[o.c->rank][o.peer].store((unsigned)o.bytes + 1u);
  }
  for (auto &o : ops) {
    if (o.kind != 1) continue;
    Shared *s = o.c->sh;
    unsigned f;
    while ((f = s->full[o.peer][o.c->rank].load()) == 0) nap();
    if ((size_t)(f - 1u) != o.bytes) return fail("send / recv size mismatch");
    if (hipMemcpy(o.buf, s->mbox[o.peer][o.c->rank], o.bytes, hipMemcpyHostToDevice) != hipSuccess) return fail("H2D");
    s->full[o.peer][o.c->rank].store(0);
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
  while (c->sh->attached.load() < nranks) nap();
  barrier(c);
  *comm = c;
  return 0;
}

int ncclCommDestroy(void *comm) {
  FakeComm *c = (FakeComm *)comm;
  if (!c) return 0;
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
  FakeComm *c = (FakeComm *)comm;
  if (dtype != 8 || op != 0 || count > RED_MAX) return fail("all-reduce: only f64 SUM of <= 64 values");
  if (hipStreamSynchronize(st) != hipSuccess) return fail("stream sync");
  double mine[RED_MAX], out[RED_MAX];
  if (hipMemcpy(mine, send, count * 8, hipMemcpyDeviceToHost) != hipSuccess) return fail("D2H");
  for (size_t i = 0; i < count; ++i) c->sh->red[c->rank][i] = mine[i];
  barrier(c);
  for (size_t i = 0; i < count; ++i) {
    double s = 0.0;
    for (int r = 0; r < c->nranks; ++r) s += c->sh->red[r][i];
    out[i] = s;
  }
  barrier(c);   // nobody overwrites its deposit before everyone has read it
  if (hipMemcpy(recv, out, count * 8, hipMemcpyHostToDevice) != hipSuccess) return fail("H2D");
  return 0;
}

// every rank deposits its block in its own diagonal mailbox, all ranks read all blocks
int ncclAllGather(const void *send, void *recv, size_t count, int dtype, void *comm, hipStream_t st) {
  FakeComm *c = (FakeComm *)comm;
  const size_t bytes = count * dtype_bytes(dtype);
  if (bytes > MBOX_BYTES) return fail("all-gather block larger than the mailbox");
  if (hipStreamSynchronize(st) != hipSuccess) return fail("stream sync");
  if (hipMemcpy(c->sh->mbox[c->rank][c->rank], send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return fail("D2H");
  barrier(c);
  for (int r = 0; r < c->nranks; ++r)
    if (hipMemcpy((char *)recv + (size_t)r * bytes, c->sh->mbox[r][r], bytes, hipMemcpyHostToDevice) != hipSuccess) return fail("H2D");
  barrier(c);
  return 0;
}

const char *ncclGetErrorString(int e) { return e == 0 ? "ok" : "fake_rccl failure (see stderr)"; }

}  // extern "C"
