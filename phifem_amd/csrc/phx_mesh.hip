// Mesh handle: host topology builder for unstructured meshes, device generator for structured
// Kuhn boxes, detection-point tables.  Replaces the dolfinx mesh/topology services the reference
// calls at src/phifem/mesh_scripts.py:151-153,308-315,419-422,430 and
// demo/weak-dirichlet/flower/main.py:45-46.  gfx950 only.
#include "phx_prim.h"
#include <stdarg.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "phx_common.h"
#include "phx_select.h"

static thread_local char g_err[1024] = "";

void phx_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

#include <map>
#include <mutex>
#include <unordered_map>
namespace {
struct Pool {
  std::mutex mu;
  std::multimap<std::pair<int, size_t>, void *> free_blocks;   // (device, bytes) -> ptr
  std::unordered_map<void *, std::pair<int, size_t>> live;      // ptr -> (device, bytes)
  size_t cached = 0;
  size_t live_bytes = 0;        // handed out and not yet released
  size_t device_total = 0;      // of the device the limit was derived from
};
Pool &pool() { static Pool p; return p; }
size_t pool_limit() {
  static size_t lim = [] {
    const char *e = getenv("PHX_POOL_LIMIT_GB");
    if (e) return (size_t)(atof(e) * (double)(1ull << 30));
    // default: 60 % of the device (173 GB of 288) -- the transient buffers of one 256^3 elasticity pass add up to
    // ~110 GB, and with the former 96 GB every pass gave the largest ones back and paid hipMalloc for them again
    // (assembly 0.52 -> 2.5 s).  A failing allocation still empties the cache and retries.
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess || tot == 0) { (void)hipGetLastError(); return (size_t)96 << 30; }
    pool().device_total = tot;
    return (size_t)(0.6 * (double)tot);
  }();
  return lim;
}
void trim_locked(Pool &P) {
  for (auto &kv : P.free_blocks) (void)hipFree(kv.second);
  P.free_blocks.clear();
  P.cached = 0;
}
}  // namespace

// Size classes of the cached blocks: 32 per octave (at most 3 % larger than asked).  Many sizes here depend on the data
// in their last digits -- the value-indexed slices of a matrix assembled with atomics, selections, coarse spaces --
// and with exact-size matching such a block was never found again: the cache filled up with near-duplicates and every
// pass paid hipMalloc / hipFree for its largest buffers (256^3 elasticity: assembly 0.5 -> 2.1 s after a few passes).
static inline size_t pool_size_class(size_t bytes) {
  if (bytes < (1u << 20)) return bytes;
  int lg = 63 - __builtin_clzll((unsigned long long)bytes);
  const size_t gran = (size_t)1 << (lg - 5);
  return (bytes + gran - 1) / gran * gran;
}

hipError_t phx_pool_malloc(void **p, size_t bytes) {
  if (bytes == 0) bytes = 16;
  bytes = pool_size_class(bytes);
  int dev = 0;
  (void)hipGetDevice(&dev);
  Pool &P = pool();
  std::lock_guard<std::mutex> lk(P.mu);
  auto it = P.free_blocks.find({dev, bytes});
  if (it != P.free_blocks.end()) {
    *p = it->second;
    P.free_blocks.erase(it);
    P.cached -= bytes;
    P.live[*p] = {dev, bytes};
    P.live_bytes += bytes;
    return hipSuccess;
  }
  hipError_t e = hipMalloc(p, bytes);
  if (e != hipSuccess) {  // give the cache back and retry once
    (void)hipGetLastError();
    trim_locked(P);
    e = hipMalloc(p, bytes);
  }
  if (e == hipSuccess) { P.live[*p] = {dev, bytes}; P.live_bytes += bytes; }
  return e;
}

hipError_t phx_pool_free(void *p) {
  if (!p) return hipSuccess;
  Pool &P = pool();
  std::lock_guard<std::mutex> lk(P.mu);
  auto it = P.live.find(p);
  if (it == P.live.end()) return hipFree(p);  // not ours (allocated before the pool existed)
  const auto key = it->second;
  P.live.erase(it);
  P.live_bytes -= std::min(P.live_bytes, key.second);
  if (key.second < (1u << 20) || P.cached + key.second > pool_limit()) return hipFree(p);
  // Large problems (this library's blocks, handed out or cached, beyond half of the device): other allocators of the
  // process (torch's) cannot ask this cache to shrink, so it keeps a tenth of the device free for them
  if (P.device_total > 0 && 2 * (P.live_bytes + P.cached + key.second) > P.device_total) {
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) == hipSuccess && 10 * fr < tot) return hipFree(p);
  }
  // hipFree would have synchronised the device; a cached block may be handed to another stream
  (void)hipDeviceSynchronize();
  P.free_blocks.insert({key, p});
  P.cached += key.second;
  return hipSuccess;
}

void phx_pool_trim(void) {
  Pool &P = pool();
  std::lock_guard<std::mutex> lk(P.mu);
  trim_locked(P);
}

extern "C" int phx_pool_release(void) { phx_pool_trim(); return PHX_OK; }

extern "C" int phx_version(void) { return 1; }
extern "C" const char *phx_last_error(void) { return g_err; }
extern "C" int phx_device_count(int *n) {
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess) c = 0;
  *n = c;
  return PHX_OK;
}

int phx_get_cell_info(int cell_type, phx_cell_info *ci) {
  memset(ci, 0, sizeof(*ci));
  if (cell_type == PHX_TRIANGLE) {
    ci->tdim = 2; ci->nvpc = 3; ci->nfpc = 3; ci->nvpf = 2;
    const int fv[3][2] = {{1, 2}, {0, 2}, {0, 1}};
    for (int f = 0; f < 3; ++f) for (int k = 0; k < 2; ++k) ci->fv[f][k] = fv[f][k];
  } else if (cell_type == PHX_QUADRILATERAL) {
    ci->tdim = 2; ci->nvpc = 4; ci->nfpc = 4; ci->nvpf = 2;
    const int fv[4][2] = {{0, 1}, {0, 2}, {1, 3}, {2, 3}};
    for (int f = 0; f < 4; ++f) for (int k = 0; k < 2; ++k) ci->fv[f][k] = fv[f][k];
  } else if (cell_type == PHX_TETRAHEDRON) {
    ci->tdim = 3; ci->nvpc = 4; ci->nfpc = 4; ci->nvpf = 3;
    const int fv[4][3] = {{1, 2, 3}, {0, 2, 3}, {0, 1, 3}, {0, 1, 2}};
    for (int f = 0; f < 4; ++f) for (int k = 0; k < 3; ++k) ci->fv[f][k] = fv[f][k];
  } else {
    // mesh_scripts.py:326-329
    phx_set_error("Mesh tags computation does not support other cell types than 'triangle', "
                  "'quadrilateral' or 'tetrahedron'");
    return PHX_ERR_NOT_IMPLEMENTED;
  }
  return PHX_OK;
}

// ------------------------------------------------------------------------------------------
// Detection points (a1): mesh_scripts.py:28-92, extended to the tetrahedron.  The arithmetic
// is spelled out so that it is bit-identical to oracle/points.py (numpy.linspace semantics).
// ------------------------------------------------------------------------------------------
static void lattice_1d(int N, std::vector<double> &t) {
  t.resize(N + 1);
  const double step = 1.0 / (double)N;
  for (int i = 0; i <= N; ++i) t[i] = (double)i * step;
  t[N] = 1.0;
}

static void cell_points(int cell_type, int N, std::vector<double> &p) {
  p.clear();
  std::vector<double> t;
  if (N > 0) lattice_1d(N, t);
  if (cell_type == PHX_TRIANGLE) {
    if (N <= 0) { p = {1.0 / 3.0, 1.0 / 3.0}; return; }
    for (int i = 0; i <= N; ++i) { p.push_back(t[i]); p.push_back(0.0); }
    for (int i = 1; i <= N; ++i) { p.push_back(1.0 - t[i]); p.push_back(t[i]); }
    for (int i = 1; i < N; ++i) { p.push_back(0.0); p.push_back(1.0 - t[i]); }
  } else if (cell_type == PHX_QUADRILATERAL) {
    if (N <= 0) { p = {0.5, 0.5}; return; }
    for (int i = 0; i <= N; ++i) { p.push_back(t[i]); p.push_back(0.0); }
    for (int i = 1; i <= N; ++i) { p.push_back(1.0); p.push_back(t[i]); }
    for (int i = 1; i <= N; ++i) { p.push_back(1.0 - t[i]); p.push_back(1.0); }
    for (int i = 1; i < N; ++i) { p.push_back(0.0); p.push_back(1.0 - t[i]); }
  } else {  // tetrahedron: boundary lattice points
    if (N <= 0) { p = {0.25, 0.25, 0.25}; return; }
    for (int k = 0; k <= N; ++k)
      for (int j = 0; j <= N - k; ++j)
        for (int i = 0; i <= N - k - j; ++i) {
          const int l = N - i - j - k;
          if (i == 0 || j == 0 || k == 0 || l == 0) {
            p.push_back(t[i]); p.push_back(t[j]); p.push_back(t[k]);
          }
        }
  }
}

static void facet_points(int cell_type, int N, std::vector<double> &p) {
  p.clear();
  std::vector<double> t;
  if (N > 0) lattice_1d(N, t);
  if (cell_type == PHX_TETRAHEDRON) {  // closed triangle lattice
    if (N <= 0) { p = {1.0 / 3.0, 1.0 / 3.0}; return; }
    for (int j = 0; j <= N; ++j)
      for (int i = 0; i <= N - j; ++i) { p.push_back(t[i]); p.push_back(t[j]); }
  } else {  // segment, mesh_scripts.py:28-40
    if (N <= 0) { p = {0.5}; return; }
    for (int i = 0; i <= N; ++i) p.push_back(t[i]);
  }
}

// First-order shape functions, same operation order as oracle/points.py:shape_functions.
// which = 0: the cell's, 1: the facet's.
int phx_shape_table(int cell_type, int degree, int which, std::vector<double> &tab, int *npts,
                    int *nfun) {
  phx_cell_info ci;
  PHX_CHECK(phx_get_cell_info(cell_type, &ci));
  std::vector<double> p;
  int dim;
  int kind;  // 1 interval, 2 triangle, 3 tetrahedron, 4 quadrilateral
  if (which == 0) {
    cell_points(cell_type, degree, p);
    dim = ci.tdim;
    kind = cell_type == PHX_TRIANGLE ? 2 : (cell_type == PHX_QUADRILATERAL ? 4 : 3);
  } else {
    facet_points(cell_type, degree, p);
    dim = ci.tdim - 1;
    kind = cell_type == PHX_TETRAHEDRON ? 2 : 1;
  }
  const int n = (int)(p.size() / dim);
  const int nf = kind == 1 ? 2 : (kind == 2 ? 3 : 4);
  tab.assign((size_t)n * nf, 0.0);
  for (int q = 0; q < n; ++q) {
    const double *c = &p[(size_t)q * dim];
    double *N = &tab[(size_t)q * nf];
    if (kind == 1) { N[0] = 1.0 - c[0]; N[1] = c[0]; }
    else if (kind == 2) { N[0] = (1.0 - c[0]) - c[1]; N[1] = c[0]; N[2] = c[1]; }
    else if (kind == 3) { N[0] = ((1.0 - c[0]) - c[1]) - c[2]; N[1] = c[0]; N[2] = c[1]; N[3] = c[2]; }
    else {
      N[0] = (1.0 - c[0]) * (1.0 - c[1]); N[1] = c[0] * (1.0 - c[1]);
      N[2] = (1.0 - c[0]) * c[1]; N[3] = c[0] * c[1];
    }
  }
  *npts = n;
  *nfun = nf;
  PHX_REQUIRE(n <= PHX_MAX_PTS, PHX_ERR_NOT_IMPLEMENTED, "detection degree %d too large", degree);
  return PHX_OK;
}

extern "C" int phx_detection_points(int cell_type, int degree, int which, double *out,
                                    int64_t *npts) {
  phx_cell_info ci;
  PHX_CHECK(phx_get_cell_info(cell_type, &ci));
  std::vector<double> p;
  if (which == 0) cell_points(cell_type, degree, p);
  else facet_points(cell_type, degree, p);
  const int dim = which == 0 ? ci.tdim : ci.tdim - 1;
  *npts = (int64_t)(p.size() / dim);
  if (out) memcpy(out, p.data(), p.size() * sizeof(double));
  return PHX_OK;
}

// ------------------------------------------------------------------------------------------
// Host topology builder (unstructured meshes)
// ------------------------------------------------------------------------------------------
namespace {
struct FacetRec {
  int32_t v[3];
  int32_t cell;
  int32_t lf;
};
}  // namespace

extern "C" int phx_topology_build_host(int cell_type, int64_t nv, int64_t nc,
                                       const int32_t *cells, int32_t *c2f, int32_t *f2c,
                                       int64_t *nf_out) {
  phx_cell_info ci;
  PHX_CHECK(phx_get_cell_info(cell_type, &ci));
  PHX_REQUIRE(nc > 0 && nv > 0, PHX_ERR_VALUE, "empty mesh");
  std::vector<FacetRec> recs((size_t)nc * ci.nfpc);
  for (int64_t c = 0; c < nc; ++c)
    for (int lf = 0; lf < ci.nfpc; ++lf) {
      FacetRec &r = recs[(size_t)c * ci.nfpc + lf];
      r.v[2] = -1;
      for (int k = 0; k < ci.nvpf; ++k) {
        const int32_t v = cells[c * ci.nvpc + ci.fv[lf][k]];
        PHX_REQUIRE(v >= 0 && v < nv, PHX_ERR_VALUE, "cell %lld has vertex %d out of range",
                    (long long)c, v);
        r.v[k] = v;
      }
      std::sort(r.v, r.v + ci.nvpf);
      r.cell = (int32_t)c;
      r.lf = lf;
    }
  std::sort(recs.begin(), recs.end(), [](const FacetRec &a, const FacetRec &b) {
    if (a.v[0] != b.v[0]) return a.v[0] < b.v[0];
    if (a.v[1] != b.v[1]) return a.v[1] < b.v[1];
    if (a.v[2] != b.v[2]) return a.v[2] < b.v[2];
    return a.cell < b.cell;
  });
  int64_t nf = 0;
  size_t i = 0;
  while (i < recs.size()) {
    size_t j = i + 1;
    while (j < recs.size() && recs[j].v[0] == recs[i].v[0] && recs[j].v[1] == recs[i].v[1] &&
           recs[j].v[2] == recs[i].v[2])
      ++j;
    PHX_REQUIRE(j - i <= 2, PHX_ERR_VALUE, "facet shared by %zu cells (non-manifold mesh)", j - i);
    f2c[2 * nf] = recs[i].cell;
    f2c[2 * nf + 1] = (j - i == 2) ? recs[i + 1].cell : -1;
    for (size_t k = i; k < j; ++k) c2f[(size_t)recs[k].cell * ci.nfpc + recs[k].lf] = (int32_t)nf;
    ++nf;
    i = j;
  }
  *nf_out = nf;
  return PHX_OK;
}

// ------------------------------------------------------------------------------------------
// Common allocation / timing helpers
// ------------------------------------------------------------------------------------------
int phx_begin_timing(phx_mesh *m) {
  PHX_HIP(hipEventRecord(m->ev0, m->stream));
  return PHX_OK;
}
int phx_end_timing(phx_mesh *m, int slot) {
  PHX_HIP(hipEventRecord(m->ev1, m->stream));
  PHX_HIP(hipEventSynchronize(m->ev1));
  float ms = 0.f;
  PHX_HIP(hipEventElapsedTime(&ms, m->ev0, m->ev1));
  m->timings[slot] = (double)ms * 1e-3;
  return PHX_OK;
}
// the same in two halves, for callers that synchronise the stream anyway a few launches later: mark the end now,
// read the interval after that synchronisation (one host round trip instead of two)
int phx_end_timing_mark(phx_mesh *m) {
  PHX_HIP(hipEventRecord(m->ev1, m->stream));
  return PHX_OK;
}
int phx_end_timing_read(phx_mesh *m, int slot) {
  PHX_HIP(hipEventSynchronize(m->ev1));
  float ms = 0.f;
  PHX_HIP(hipEventElapsedTime(&ms, m->ev0, m->ev1));
  m->timings[slot] = (double)ms * 1e-3;
  return PHX_OK;
}

static int mesh_init_device(phx_mesh *m, int device) {
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  PHX_REQUIRE(e == hipSuccess && ndev > 0, PHX_ERR_HIP,
              "no usable HIP device (libphifem_hip has no CPU fallback): %s",
              e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
  PHX_REQUIRE(device >= 0 && device < ndev, PHX_ERR_VALUE, "device %d out of range (%d)", device,
              ndev);
  m->device = device;
  if (const char *e = getenv("PHX_PRECOND")) {   // default of PHX_OPT_PRECOND for this process (0 / 1 / 2)
    const int v = atoi(e);
    if (v >= 0 && v <= 2) m->precond = v;
  }
  PHX_HIP(hipSetDevice(device));
  PHX_HIP(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
  PHX_HIP(hipEventCreate(&m->ev0));
  PHX_HIP(hipEventCreate(&m->ev1));
  return PHX_OK;
}

int phx_mesh_alloc_common(phx_mesh *m) {
  PHX_HIP(phx_malloc(&m->cell_tags, (size_t)m->nc));
  PHX_HIP(phx_malloc(&m->facet_tags, (size_t)m->nf));
  PHX_HIP(hipMemsetAsync(m->cell_tags, 0, (size_t)m->nc, m->stream));
  PHX_HIP(hipMemsetAsync(m->facet_tags, 0, (size_t)m->nf, m->stream));
  return PHX_OK;
}

// boundary facets: compaction of {f : f2c[f][1] < 0} in ascending order
struct IsBoundary {
  const int32_t *f2c;
  __host__ __device__ bool operator()(const int32_t &f) const { return f2c[2 * (int64_t)f + 1] < 0; }
};

__global__ void k_bfacet_pairs(int64_t nbf, const int32_t *__restrict__ bf,
                               const int32_t *__restrict__ f2c, const int32_t *__restrict__ c2f,
                               int nfpc, int32_t *__restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= nbf) return;
  const int32_t f = bf[i];
  const int32_t c = f2c[2 * (int64_t)f];
  int lf = 0;
  for (int k = 0; k < nfpc; ++k)
    if (c2f[(int64_t)c * nfpc + k] == f) lf = k;
  out[2 * i] = c;
  out[2 * i + 1] = lf;
}

static int build_boundary_list(phx_mesh *m) {
  int32_t *sel = nullptr;
  int64_t nbf = 0;
  PHX_CHECK(phx_select_indices(m->stream, m->nf, IsBoundary{m->f2c}, &sel, &nbf));
  m->nbf = nbf;
  PHX_HIP(phx_malloc(&m->bfacets, sizeof(int32_t) * 2 * (size_t)(nbf > 0 ? nbf : 1)));
  if (nbf > 0)
    k_bfacet_pairs<<<dim3((unsigned)phx_div_up(nbf, 256)), dim3(256), 0, m->stream>>>(
        nbf, sel, m->f2c, m->c2f, m->ci.nfpc, m->bfacets);
  PHX_HIP(hipStreamSynchronize(m->stream));
  PHX_HIP(phx_malloc(&m->bfacet_ids, sizeof(int32_t) * (size_t)(nbf > 0 ? nbf : 1)));
  if (nbf > 0) {
    // device-to-device hipMemcpy does not wait on the host side: keep it on the mesh's stream and wait before `sel` goes
    PHX_HIP(hipMemcpyAsync(m->bfacet_ids, sel, sizeof(int32_t) * (size_t)nbf, hipMemcpyDeviceToDevice, m->stream));
    PHX_HIP(hipStreamSynchronize(m->stream));
  }
  PHX_HIP(phx_free(sel));
  return PHX_OK;
}

// Does a caller-supplied simplicial mesh sit on a uniform tensor lattice -- what dolfinx's create_rectangle /
// create_box (the meshes every reference demo builds, demo/weak-dirichlet/flower/main.py:45-46) hand over, in
// whatever vertex order?  Then vertex v <-> lattice point (i, j, k), and the fictitious-domain preconditioner
// applies to it as it does to a sub-mesh of a generated box (on_box_lattice + the two vertex maps).
// Conditions: per axis the coordinates take n_a + 1 equispaced values (1e-9 of the extent), prod (n_a + 1) = nv,
// vertex -> lattice point is a bijection, and every cell lies inside one lattice cube.  Host work, O(nv log nv).
static bool detect_lattice(int gdim, int nvpc, int64_t nv, const double *coords, int64_t nc, const int32_t *cells,
                           int64_t n_out[3], double h_out[3], std::vector<int32_t> &v2lat, std::vector<int32_t> &lat2v,
                           double lo_out[3], double hi_out[3]) {
  if (nv < 8 || nv >= INT32_MAX) return false;
  double lo[3] = {0, 0, 0}, h[3] = {0, 0, 0};
  int64_t n[3] = {1, 1, 1};
  std::vector<double> tmp((size_t)nv);
  for (int a = 0; a < gdim; ++a) {
    for (int64_t v = 0; v < nv; ++v) tmp[(size_t)v] = coords[v * gdim + a];
    std::sort(tmp.begin(), tmp.end());
    const double ext = tmp.back() - tmp.front();
    if (!(ext > 0.0)) return false;
    const double tol = 1e-9 * ext;
    int64_t distinct = 1;
    for (int64_t v = 1; v < nv; ++v) if (tmp[(size_t)v] - tmp[(size_t)v - 1] > tol) ++distinct;
    if (distinct < 2) return false;
    n[a] = distinct - 1;
    lo[a] = tmp.front();
    h[a] = ext / (double)n[a];
    lo_out[a] = tmp.front();
    hi_out[a] = tmp.back();
  }
  int64_t prod = 1;
  for (int a = 0; a < gdim; ++a) prod *= n[a] + 1;
  if (prod != nv) return false;
  v2lat.assign((size_t)nv, -1);
  lat2v.assign((size_t)nv, -1);
  const int64_t s1 = n[0] + 1, s2 = (n[0] + 1) * (n[1] + 1);
  for (int64_t v = 0; v < nv; ++v) {
    int64_t id = 0;
    for (int a = 0; a < gdim; ++a) {
      const double q = (coords[v * gdim + a] - lo[a]) / h[a];
      const int64_t i = (int64_t)llround(q);
      if (i < 0 || i > n[a]) return false;
      // the system is assembled on a GENERATED box (x = lo + (hi - lo) (i / n)): the caller's vertex has to sit there to
      // within a few ulps of the box size, or the mesh keeps the generic path (1e-6 h let a visibly perturbed
      // mesh be replaced by the uniform one)
      const double gen = lo[a] + (hi_out[a] - lo[a]) * ((double)i / (double)n[a]);
      const double scale = std::max(std::max(fabs(lo[a]), fabs(hi_out[a])), hi_out[a] - lo[a]);
      if (fabs(coords[v * gdim + a] - gen) > 16.0 * 2.220446049250313e-16 * scale) return false;
      id += i * (a == 0 ? 1 : (a == 1 ? s1 : s2));
    }
    if (lat2v[(size_t)id] != -1) return false;
    lat2v[(size_t)id] = (int32_t)v;
    v2lat[(size_t)v] = (int32_t)id;
  }
  for (int64_t c = 0; c < nc; ++c) {   // a cell may not span more than one lattice cube
    int64_t mn[3] = {INT64_MAX, INT64_MAX, INT64_MAX}, mx[3] = {-1, -1, -1};
    for (int k = 0; k < nvpc; ++k) {
      int64_t id = v2lat[(size_t)cells[c * nvpc + k]];
      const int64_t idx[3] = {id % s1, gdim == 3 ? (id / s1) % (n[1] + 1) : id / s1, gdim == 3 ? id / s2 : 0};
      for (int a = 0; a < gdim; ++a) { mn[a] = std::min(mn[a], idx[a]); mx[a] = std::max(mx[a], idx[a]); }
    }
    for (int a = 0; a < gdim; ++a) if (mx[a] - mn[a] > 1) return false;
  }
  for (int a = 0; a < 3; ++a) { n_out[a] = a < gdim ? n[a] : 1; h_out[a] = a < gdim ? h[a] : 0.0; }
  return true;
}

#include "phx_topology.inc.hip"

// ---- caller-supplied Kuhn boxes ---------------------------------------------------------------------------------------
// cell c of the caller's mesh -> cell of the generated box with the same lattice: cube = lower corner of its vertices,
// t = the axis permutation its vertices walk (o, o + e_p0, o + e_p0 + e_p1, ...).  bad: some cell is not such a path.
__global__ void k_inner_cmap(int64_t nc, int d, const int32_t *__restrict__ cells, const int32_t *__restrict__ v2lat,
                             int64_t n0, int64_t n1, int64_t n2, int32_t *__restrict__ cmap, int *__restrict__ bad) {
  const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (c >= nc) return;
  const int N = d + 1;
  int64_t idx[4][3];
  int64_t o[3] = {INT64_MAX, INT64_MAX, INT64_MAX};
  for (int k = 0; k < N; ++k) {
    const int64_t id = v2lat[cells[c * N + k]];
    idx[k][0] = id % (n0 + 1);
    idx[k][1] = d == 3 ? (id / (n0 + 1)) % (n1 + 1) : id / (n0 + 1);
    idx[k][2] = d == 3 ? id / ((n0 + 1) * (n1 + 1)) : 0;
    for (int a = 0; a < 3; ++a) o[a] = min(o[a], idx[k][a]);
  }
  // vertex with m ones in its offset from o is path vertex m; the axis that appears at step m is p[m-1]
  int bits_of[4] = {-1, -1, -1, -1};
  bool ok = true;
  for (int k = 0; k < N; ++k) {
    int bits = 0, cnt = 0;
    for (int a = 0; a < d; ++a) {
      const int64_t off = idx[k][a] - o[a];
      if (off < 0 || off > 1) ok = false;
      if (off == 1) { bits |= 1 << a; ++cnt; }
    }
    if (cnt > d || bits_of[cnt] != -1) ok = false; else bits_of[cnt] = bits;
  }
  int p[3] = {0, 1, 2};
  for (int mstep = 1; ok && mstep <= d; ++mstep) {
    const int add = bits_of[mstep] & ~bits_of[mstep - 1];
    if ((bits_of[mstep] & bits_of[mstep - 1]) != bits_of[mstep - 1] || __popc(add) != 1) ok = false;
    else p[mstep - 1] = __ffs(add) - 1;
  }
  if (!ok || o[0] >= n0 || o[1] >= n1 || (d == 3 && o[2] >= n2)) { atomicOr(bad, 1); cmap[c] = 0; return; }
  const int t = d == 3 ? p[0] * 2 + (p[1] > p[2] ? 1 : 0) : p[0];
  const int64_t cube = o[0] + n0 * (o[1] + (d == 3 ? n1 * o[2] : 0));
  cmap[c] = (int32_t)(cube * (d == 3 ? 6 : 2) + t);
}

// facet f of the caller's mesh -> facet of the generated box: the facet opposite local vertex lf of cell c is the facet
// opposite the same lattice vertex in cell cmap[c]
__global__ void k_inner_fmap(int64_t nc, int N, const int32_t *__restrict__ cells, const int32_t *__restrict__ c2f,
                             const int32_t *__restrict__ v2lat, const int32_t *__restrict__ cmap,
                             const int32_t *__restrict__ icells, const int32_t *__restrict__ ic2f,
                             int32_t *__restrict__ fmap, int *__restrict__ bad) {
  const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (c >= nc) return;
  const int64_t ci = cmap[c];
  for (int lf = 0; lf < N; ++lf) {
    const int32_t lat = v2lat[cells[c * N + lf]];
    int k = -1;
    for (int q = 0; q < N; ++q) if (icells[ci * N + q] == lat) k = q;
    if (k < 0) { atomicOr(bad, 1); continue; }
    fmap[c2f[c * N + lf]] = ic2f[ci * N + k];
  }
}

extern "C" int phx_mesh_create_box(int gdim, const double *lo, const double *hi, const int64_t *n,
                                   const int64_t *offset, const int64_t *n_global, int device, phx_mesh **out);

static int mesh_attach_inner_box(phx_mesh *m, const double lo[3], const double hi[3]) {
  const int d = m->gdim, N = d + 1;
  int *bad = nullptr, hbad = 0;
  PHX_HIP(phx_malloc(&m->in_cmap, sizeof(int32_t) * (size_t)m->nc));
  PHX_HIP(phx_malloc(&bad, sizeof(int)));
  PHX_HIP(hipMemsetAsync(bad, 0, sizeof(int), m->stream));
  const dim3 block(256), gc((unsigned)phx_div_up(m->nc, 256));
  k_inner_cmap<<<gc, block, 0, m->stream>>>(m->nc, d, m->cells, m->v2lat, m->box_n[0], m->box_n[1], d == 3 ? m->box_n[2] : 1,
                                            m->in_cmap, bad);
  PHX_HIP(hipMemcpyAsync(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost, m->stream));
  PHX_HIP(hipStreamSynchronize(m->stream));
  int64_t ncube = 1;
  for (int a = 0; a < d; ++a) ncube *= m->box_n[a];
  if (hbad || m->nc != ncube * (d == 3 ? 6 : 2)) {   // another triangulation of the lattice: the generic path serves it
    PHX_HIP(phx_free(bad)); PHX_HIP(phx_free(m->in_cmap));
    m->in_cmap = nullptr;
    return PHX_OK;
  }
  phx_mesh *in = nullptr;
  PHX_CHECK(phx_mesh_create_box(d, lo, hi, m->box_n, nullptr, nullptr, m->device, &in));
  // one stream for both: the pushes of tags and nodal data and the kernels that read them stay ordered
  PHX_HIP(hipStreamDestroy(in->stream));
  in->stream = m->stream;
  in->own_stream = false;
  m->inner = in;
  PHX_HIP(phx_malloc(&m->in_fmap, sizeof(int32_t) * (size_t)m->nf));
  PHX_HIP(hipMemsetAsync(m->in_fmap, 0xff, sizeof(int32_t) * (size_t)m->nf, m->stream));
  k_inner_fmap<<<gc, block, 0, m->stream>>>(m->nc, N, m->cells, m->c2f, m->v2lat, m->in_cmap, in->cells, in->c2f, m->in_fmap, bad);
  PHX_HIP(hipMemcpyAsync(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost, m->stream));
  PHX_HIP(hipStreamSynchronize(m->stream));
  PHX_HIP(phx_free(bad));
  if (hbad || in->nf != m->nf || in->nv != m->nv) {
    phx_mesh_destroy(in);
    m->inner = nullptr;
    PHX_HIP(phx_free(m->in_cmap)); PHX_HIP(phx_free(m->in_fmap));
    m->in_cmap = nullptr; m->in_fmap = nullptr;
  }
  return PHX_OK;
}

// coords / cells at `loc` (host arrays from a caller, or device arrays of a parent mesh: phx_submesh.hip).  The facet
// numbering and both connectivities are built on the device (phx_topology.inc.hip); PHX_TOPOLOGY_HOST=1 takes the
// host sort of phx_topology_build_host instead (same numbering; A/B aid, host inputs only).
int phx_mesh_create_from(int gdim, int cell_type, int64_t nv, const double *coords, int64_t nc,
                         const int32_t *cells, int loc, int device, phx_mesh **out) {
  phx_cell_info ci;
  PHX_CHECK(phx_get_cell_info(cell_type, &ci));
  PHX_REQUIRE(gdim == ci.tdim, PHX_ERR_VALUE, "gdim %d does not match the cell type", gdim);
  PHX_REQUIRE(nc > 0 && nv > 0, PHX_ERR_VALUE, "empty mesh");
  PHX_REQUIRE(nc * (int64_t)ci.nfpc < INT32_MAX, PHX_ERR_VALUE, "mesh too large for 32-bit local ids");
  static const bool host_topo = getenv("PHX_TOPOLOGY_HOST") && atoi(getenv("PHX_TOPOLOGY_HOST")) != 0;
  phx_mesh *m = new phx_mesh();
  int rc = mesh_init_device(m, device);
  if (rc != PHX_OK) { delete m; return rc; }
  m->gdim = gdim; m->cell_type = cell_type; m->ci = ci;
  m->nv = nv; m->nc = nc;
  const hipMemcpyKind kind = loc == PHX_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  auto fail = [&](int code) { phx_mesh_destroy(m); return code; };
  // (a device-to-device hipMemcpy returns before the copy has run and the mesh's stream does not wait for the null
  // stream: device sources are copied ON the mesh's stream, ahead of the kernels that read them)
  if (phx_malloc(&m->x, sizeof(double) * (size_t)nv * gdim) != hipSuccess ||
      phx_malloc(&m->cells, sizeof(int32_t) * (size_t)nc * ci.nvpc) != hipSuccess ||
      hipMemcpyAsync(m->x, coords, sizeof(double) * (size_t)nv * gdim, kind, m->stream) != hipSuccess ||
      hipMemcpyAsync(m->cells, cells, sizeof(int32_t) * (size_t)nc * ci.nvpc, kind, m->stream) != hipSuccess ||
      hipStreamSynchronize(m->stream) != hipSuccess) {
    phx_set_error("mesh upload failed");
    return fail(PHX_ERR_HIP);
  }
  if (host_topo && loc != PHX_DEVICE) {
    std::vector<int32_t> c2f((size_t)nc * ci.nfpc), f2c((size_t)nc * ci.nfpc * 2);
    int64_t nf = 0;
    rc = phx_topology_build_host(cell_type, nv, nc, cells, c2f.data(), f2c.data(), &nf);
    if (rc != PHX_OK) return fail(rc);
    m->nf = nf;
    if (phx_malloc(&m->c2f, sizeof(int32_t) * (size_t)nc * ci.nfpc) != hipSuccess ||
        phx_malloc(&m->f2c, sizeof(int32_t) * (size_t)nf * 2) != hipSuccess ||
        hipMemcpy(m->c2f, c2f.data(), sizeof(int32_t) * (size_t)nc * ci.nfpc, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(m->f2c, f2c.data(), sizeof(int32_t) * (size_t)nf * 2, hipMemcpyHostToDevice) != hipSuccess) {
      phx_set_error("mesh upload failed");
      return fail(PHX_ERR_HIP);
    }
  } else {
    rc = phx_topology_build_device(m);
    if (rc != PHX_OK) return fail(rc);
  }
  rc = build_boundary_list(m);
  if (rc == PHX_OK) rc = phx_mesh_alloc_common(m);
  if (rc != PHX_OK) return fail(rc);
  if (loc != PHX_DEVICE && (cell_type == PHX_TRIANGLE || cell_type == PHX_TETRAHEDRON)) {
    std::vector<int32_t> v2lat, lat2v;
    int64_t ln[3];
    double lh[3], llo[3] = {0, 0, 0}, lhi[3] = {0, 0, 0};
    if (detect_lattice(gdim, ci.nvpc, nv, coords, nc, cells, ln, lh, v2lat, lat2v, llo, lhi)) {
      m->on_box_lattice = true;
      for (int a = 0; a < 3; ++a) { m->box_n[a] = ln[a]; m->box_h[a] = lh[a]; }
      if (phx_malloc(&m->v2lat, sizeof(int32_t) * (size_t)nv) != hipSuccess ||
          phx_malloc(&m->lat2v, sizeof(int32_t) * (size_t)nv) != hipSuccess ||
          hipMemcpy(m->v2lat, v2lat.data(), sizeof(int32_t) * (size_t)nv, hipMemcpyHostToDevice) != hipSuccess ||
          hipMemcpy(m->lat2v, lat2v.data(), sizeof(int32_t) * (size_t)nv, hipMemcpyHostToDevice) != hipSuccess) {
        phx_set_error("mesh upload failed");
        return fail(PHX_ERR_HIP);
      }
      static const bool no_inner = getenv("PHX_INNER_BOX") && atoi(getenv("PHX_INNER_BOX")) == 0;   // A/B aid
      if (!no_inner) {
        rc = mesh_attach_inner_box(m, llo, lhi);   // leaves m->inner == nullptr when the cells are not the Kuhn split
        if (rc != PHX_OK) return fail(rc);
      }
    }
  }
  if (hipStreamSynchronize(m->stream) != hipSuccess) return fail(PHX_ERR_HIP);
  *out = m;
  return PHX_OK;
}

extern "C" int phx_mesh_create(int gdim, int cell_type, int64_t nv, const double *coords,
                               int64_t nc, const int32_t *cells, int device, phx_mesh **out) {
  return phx_mesh_create_from(gdim, cell_type, nv, coords, nc, cells, PHX_HOST, device, out);
}

// ------------------------------------------------------------------------------------------
// Device generator for Kuhn boxes.  Every connectivity is a closed form of (cube, permutation):
//   simplex t of cube o walks  o, o+e_p0, o+e_p0+e_p1(, o+e_p0+e_p1+e_p2)  for the t-th
//   lexicographic axis permutation p.  Facet classes ("types"):
//     A(a,s)   : facets lying in a hyperplane x_a = const, 2-D: type a; 3-D: type 2a+s
//     B        : facets through the cube's main diagonal, 2-D: type 2; 3-D: 6+c (pair first,
//                single axis c last) and 9+a (single axis a first, pair last)
//   facet id = base[type] + linear index of its anchor cube inside the type's extent.
// ------------------------------------------------------------------------------------------
struct BoxDesc {
  int d;
  int64_t n[3];       // cubes
  int64_t ntypes;
  int64_t base[13];   // facet id offsets per type (base[ntypes] = nf)
  int64_t ext[12][3]; // extent per type
  double lo[3], hi[3];
  int64_t off[3], nglob[3];
};

__constant__ int c_perm3[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
__constant__ int c_perm2[2][2] = {{0, 1}, {1, 0}};

__device__ __forceinline__ int perm_index3(int p0, int p1, int p2) { return p0 * 2 + (p1 > p2 ? 1 : 0); }

__device__ __forceinline__ int64_t anchor_index(const BoxDesc &b, int type, const int64_t o[3]) {
  return o[0] + b.ext[type][0] * (o[1] + b.ext[type][1] * o[2]);
}

__global__ void k_box_coords(BoxDesc b, int64_t nv, double *__restrict__ x) {
  const int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (v >= nv) return;
  int64_t idx[3];
  const int64_t n0 = b.n[0] + 1, n1 = b.n[1] + 1;
  idx[0] = v % n0;
  idx[1] = (v / n0) % n1;
  idx[2] = b.d == 3 ? v / (n0 * n1) : 0;
  for (int a = 0; a < b.d; ++a) {
    const double t = (double)(b.off[a] + idx[a]) / (double)b.nglob[a];
    x[v * b.d + a] = b.lo[a] + (b.hi[a] - b.lo[a]) * t;
  }
}

__global__ void k_box_cells(BoxDesc b, int64_t nc, int32_t *__restrict__ cells,
                            int32_t *__restrict__ c2f) {
  const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (c >= nc) return;
  const int d = b.d;
  const int nper = d == 3 ? 6 : 2;
  const int t = (int)(c % nper);
  const int64_t cube = c / nper;
  int64_t o[3];
  o[0] = cube % b.n[0];
  o[1] = (cube / b.n[0]) % b.n[1];
  o[2] = d == 3 ? cube / (b.n[0] * b.n[1]) : 0;
  const int64_t stride[3] = {1, b.n[0] + 1, (b.n[0] + 1) * (b.n[1] + 1)};
  int p[3] = {0, 0, 0};
  for (int s = 0; s < d; ++s) p[s] = d == 3 ? c_perm3[t][s] : c_perm2[t][s];
  int64_t v = o[0] * stride[0] + o[1] * stride[1] + o[2] * stride[2];
  cells[c * (d + 1)] = (int32_t)v;
  for (int s = 0; s < d; ++s) {
    v += stride[p[s]];
    cells[c * (d + 1) + s + 1] = (int32_t)v;
  }
  // local facet m is opposite path vertex m
  for (int m = 0; m <= d; ++m) {
    int type;
    int64_t an[3] = {o[0], o[1], o[2]};
    if (d == 3) {
      if (m == 0) { type = p[0] * 2 + (p[1] > p[2] ? 1 : 0); an[p[0]] += 1; }
      else if (m == 3) { type = p[2] * 2 + (p[0] > p[1] ? 1 : 0); }
      else if (m == 1) { type = 6 + p[2]; }
      else { type = 9 + p[0]; }
    } else {
      if (m == 0) { type = p[0]; an[p[0]] += 1; }
      else if (m == 2) { type = p[1]; }
      else { type = 2; }
    }
    c2f[c * (d + 1) + m] = (int32_t)(b.base[type] + anchor_index(b, type, an));
  }
}

__global__ void k_box_f2c(BoxDesc b, int64_t nf, int32_t *__restrict__ f2c) {
  const int64_t f = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (f >= nf) return;
  const int d = b.d;
  int type = 0;
  while (type + 1 < b.ntypes && f >= b.base[type + 1]) ++type;
  int64_t r = f - b.base[type];
  int64_t o[3];
  o[0] = r % b.ext[type][0];
  o[1] = (r / b.ext[type][0]) % b.ext[type][1];
  o[2] = d == 3 ? r / (b.ext[type][0] * b.ext[type][1]) : 0;
  const int nper = d == 3 ? 6 : 2;
  int64_t c0 = -1, c1 = -1;
  auto cube_of = [&](const int64_t q[3]) { return q[0] + b.n[0] * (q[1] + b.n[1] * q[2]); };
  if (d == 3) {
    if (type < 6) {
      const int a = type / 2, s = type % 2;
      const int r0 = a == 0 ? 1 : 0, r1 = a == 2 ? 1 : 2;  // remaining axes, ascending
      const int s0 = s ? r1 : r0, s1 = s ? r0 : r1;
      if (o[a] > 0) {
        int64_t q[3] = {o[0], o[1], o[2]};
        q[a] -= 1;
        c0 = cube_of(q) * nper + perm_index3(a, s0, s1);  // local facet 0
      }
      if (o[a] < b.n[a]) {
        const int64_t cc = cube_of(o) * nper + perm_index3(s0, s1, a);  // local facet 3
        if (c0 < 0) c0 = cc; else c1 = cc;
      }
    } else if (type < 9) {
      const int c = type - 6;
      const int r0 = c == 0 ? 1 : 0, r1 = c == 2 ? 1 : 2;
      c0 = cube_of(o) * nper + perm_index3(r0, r1, c);
      c1 = cube_of(o) * nper + perm_index3(r1, r0, c);
    } else {
      const int a = type - 9;
      const int r0 = a == 0 ? 1 : 0, r1 = a == 2 ? 1 : 2;
      c0 = cube_of(o) * nper + perm_index3(a, r0, r1);
      c1 = cube_of(o) * nper + perm_index3(a, r1, r0);
    }
  } else {
    if (type < 2) {
      const int a = type, bx = 1 - a;
      if (o[a] > 0) {
        int64_t q[3] = {o[0], o[1], 0};
        q[a] -= 1;
        c0 = cube_of(q) * nper + a;  // perm (a,b), local facet 0
      }
      if (o[a] < b.n[a]) {
        const int64_t cc = cube_of(o) * nper + bx;  // perm (b,a), local facet 2
        if (c0 < 0) c0 = cc; else c1 = cc;
      }
    } else {
      c0 = cube_of(o) * nper;
      c1 = c0 + 1;
    }
  }
  f2c[2 * f] = (int32_t)c0;
  f2c[2 * f + 1] = (int32_t)c1;
}

extern "C" int phx_mesh_create_box(int gdim, const double *lo, const double *hi, const int64_t *n,
                                   const int64_t *offset, const int64_t *n_global, int device,
                                   phx_mesh **out) {
  PHX_REQUIRE(gdim == 2 || gdim == 3, PHX_ERR_VALUE, "gdim must be 2 or 3");
  BoxDesc b;
  memset(&b, 0, sizeof(b));
  b.d = gdim;
  b.n[2] = 1;
  for (int a = 0; a < gdim; ++a) {
    PHX_REQUIRE(n[a] > 0, PHX_ERR_VALUE, "n[%d] must be positive", a);
    b.n[a] = n[a];
    b.lo[a] = lo[a];
    b.hi[a] = hi[a];
    b.off[a] = offset ? offset[a] : 0;
    b.nglob[a] = n_global ? n_global[a] : n[a];
  }
  const int ntypes = gdim == 3 ? 12 : 3;
  b.ntypes = ntypes;
  for (int t = 0; t < ntypes; ++t) {
    for (int a = 0; a < 3; ++a) b.ext[t][a] = a < gdim ? b.n[a] : 1;
    if (gdim == 3 && t < 6) b.ext[t][t / 2] += 1;
    if (gdim == 2 && t < 2) b.ext[t][t] += 1;
  }
  b.base[0] = 0;
  for (int t = 0; t < ntypes; ++t) b.base[t + 1] = b.base[t] + b.ext[t][0] * b.ext[t][1] * b.ext[t][2];
  const int64_t nf = b.base[ntypes];
  int64_t nv = 1, ncube = 1;
  for (int a = 0; a < gdim; ++a) { nv *= (b.n[a] + 1); ncube *= b.n[a]; }
  const int64_t nc = ncube * (gdim == 3 ? 6 : 2);
  PHX_REQUIRE(nf < INT32_MAX && nc < INT32_MAX, PHX_ERR_VALUE,
              "box too large for 32-bit local ids (nc=%lld nf=%lld): partition it", (long long)nc,
              (long long)nf);
  phx_mesh *m = new phx_mesh();
  int rc = mesh_init_device(m, device);
  if (rc != PHX_OK) { delete m; return rc; }
  m->gdim = gdim;
  m->cell_type = gdim == 3 ? PHX_TETRAHEDRON : PHX_TRIANGLE;
  PHX_CHECK(phx_get_cell_info(m->cell_type, &m->ci));
  m->nv = nv; m->nc = nc; m->nf = nf;
  m->is_box = true;
  m->box_plane = gdim == 3 ? (b.n[0] + 1) * (b.n[1] + 1) : (b.n[0] + 1);
  m->box_nlast = b.n[gdim - 1];
  for (int a = 0; a < 3; ++a) m->box_n[a] = a < gdim ? b.n[a] : 1;
  for (int a = 0; a < 3; ++a) m->box_off[a] = a < gdim ? b.off[a] : 0;
  for (int a = 0; a < 3; ++a) m->box_nglob[a] = a < gdim ? b.nglob[a] : 1;
  for (int a = 0; a < gdim; ++a) m->box_h[a] = (b.hi[a] - b.lo[a]) / (double)b.nglob[a];
  PHX_HIP(phx_malloc(&m->x, sizeof(double) * (size_t)nv * gdim));
  PHX_HIP(phx_malloc(&m->cells, sizeof(int32_t) * (size_t)nc * (gdim + 1)));
  PHX_HIP(phx_malloc(&m->c2f, sizeof(int32_t) * (size_t)nc * (gdim + 1)));
  PHX_HIP(phx_malloc(&m->f2c, sizeof(int32_t) * (size_t)nf * 2));
  const int T = 256;
  k_box_coords<<<dim3((unsigned)phx_div_up(nv, T)), dim3(T), 0, m->stream>>>(b, nv, m->x);
  k_box_cells<<<dim3((unsigned)phx_div_up(nc, T)), dim3(T), 0, m->stream>>>(b, nc, m->cells, m->c2f);
  k_box_f2c<<<dim3((unsigned)phx_div_up(nf, T)), dim3(T), 0, m->stream>>>(b, nf, m->f2c);
  PHX_HIP(hipGetLastError());
  PHX_CHECK(build_boundary_list(m));
  PHX_CHECK(phx_mesh_alloc_common(m));
  PHX_HIP(hipStreamSynchronize(m->stream));
  *out = m;
  return PHX_OK;
}

// Slab of a partitioned box: the end planes of the LAST axis that are cuts through the global
// mesh (not part of its boundary) are marked; see phx_common.h.
__global__ void k_mark_exempt(int64_t nbf, const int32_t *__restrict__ bfacets,
                              const int32_t *__restrict__ bfacet_ids,
                              const int32_t *__restrict__ cells, int nvpc, int nvpf, int lower,
                              int upper, int64_t plane, int64_t nlast,
                              uint8_t *__restrict__ exempt) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= nbf) return;
  const int32_t c = bfacets[2 * i];
  const int lf = bfacets[2 * i + 1];
  // simplex: local facet lf is opposite local vertex lf
  bool all_lo = true, all_hi = true;
  for (int j = 0; j < nvpc; ++j) {
    if (j == lf) continue;
    const int64_t k = cells[(int64_t)c * nvpc + j] / plane;
    all_lo = all_lo && (k == 0);
    all_hi = all_hi && (k == nlast);
  }
  if ((lower && all_lo) || (upper && all_hi)) exempt[bfacet_ids[i]] = 1;
}

extern "C" int phx_mesh_set_slab_faces(phx_mesh *m, int lower_is_cut, int upper_is_cut) {
  PHX_HIP(hipSetDevice(m->device));
  PHX_REQUIRE(m->is_box, PHX_ERR_VALUE, "slab faces exist on device-generated boxes only");
  if (!m->facet_exempt) PHX_HIP(phx_malloc(&m->facet_exempt, (size_t)m->nf));
  PHX_HIP(hipMemsetAsync(m->facet_exempt, 0, (size_t)m->nf, m->stream));
  if ((lower_is_cut || upper_is_cut) && m->nbf > 0)
    k_mark_exempt<<<dim3((unsigned)phx_div_up(m->nbf, 256)), dim3(256), 0, m->stream>>>(
        m->nbf, m->bfacets, m->bfacet_ids, m->cells, m->ci.nvpc, m->ci.nvpf, lower_is_cut,
        upper_is_cut, m->box_plane, m->box_nlast, m->facet_exempt);
  PHX_HIP(hipGetLastError());
  PHX_HIP(hipStreamSynchronize(m->stream));
  m->have_facet_tags = false;
  m->have_entities = false;
  return PHX_OK;
}


// ------------------------------------------------------------------------------------------
// Edges (the extra DoFs of P2).  Local edge k of a cell follows basix:
//   triangle (1,2),(0,2),(0,1) -- identical to the local facets, so in 2-D c2e IS c2f;
//   tetrahedron (2,3),(1,3),(1,2),(0,3),(0,2),(0,1).
// Kuhn boxes: 7 edge classes per cube, id = base[class] + anchor index inside the class extent:
//   0,1,2 axis edges x,y,z;  3,4,5 face diagonals (x,y),(x,z),(y,z);  6 body diagonal.
// ------------------------------------------------------------------------------------------
__constant__ int c_tet_edge[6][2] = {{2, 3}, {1, 3}, {1, 2}, {0, 3}, {0, 2}, {0, 1}};

struct EdgeDesc {
  int64_t n[3];
  int64_t base[8];
  int64_t ext[7][3];
};

__global__ void k_box_c2e(EdgeDesc E, int64_t nc, int32_t *__restrict__ c2e) {
  const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (c >= nc) return;
  const int t = (int)(c % 6);
  const int64_t cube = c / 6;
  int64_t o[3] = {cube % E.n[0], (cube / E.n[0]) % E.n[1], cube / (E.n[0] * E.n[1])};
  int p[3] = {c_perm3[t][0], c_perm3[t][1], c_perm3[t][2]};
  // path vertices w0..w3 as offsets from o
  int w[4][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  for (int s = 0; s < 3; ++s) {
    for (int a = 0; a < 3; ++a) w[s + 1][a] = w[s][a];
    w[s + 1][p[s]] += 1;
  }
  for (int k = 0; k < 6; ++k) {
    const int a = c_tet_edge[k][0], b = c_tet_edge[k][1];  // a < b along the path
    int dir[3], nd = 0, cls;
    for (int q = 0; q < 3; ++q) { dir[q] = w[b][q] - w[a][q]; nd += dir[q]; }
    if (nd == 1) cls = dir[0] ? 0 : (dir[1] ? 1 : 2);
    else if (nd == 2) cls = !dir[2] ? 3 : (!dir[1] ? 4 : 5);
    else cls = 6;
    const int64_t an[3] = {o[0] + w[a][0], o[1] + w[a][1], o[2] + w[a][2]};
    c2e[c * 6 + k] = (int32_t)(E.base[cls] + an[0] + E.ext[cls][0] * (an[1] + E.ext[cls][1] * an[2]));
  }
}

__global__ void k_edge_vertices(int64_t nc, int nvpc, int nepc, int is_tet,
                                const int32_t *__restrict__ cells,
                                const int32_t *__restrict__ c2e, int32_t *__restrict__ edges) {
  const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (c >= nc) return;
  for (int k = 0; k < nepc; ++k) {
    int a, b;
    if (is_tet) { a = c_tet_edge[k][0]; b = c_tet_edge[k][1]; }
    else { a = k == 0 ? 1 : 0; b = k == 2 ? 1 : 2; }
    const int32_t va = cells[c * nvpc + a], vb = cells[c * nvpc + b];
    const int64_t e = c2e[c * nepc + k];
    edges[2 * e] = va < vb ? va : vb;       // every writer of an edge stores the same pair
    edges[2 * e + 1] = va < vb ? vb : va;
  }
}

int phx_mesh_build_edges(phx_mesh *m) {
  if (m->edges) return PHX_OK;
  PHX_REQUIRE(m->cell_type == PHX_TRIANGLE || m->cell_type == PHX_TETRAHEDRON,
              PHX_ERR_NOT_IMPLEMENTED, "edges (P2) are implemented on simplices only");
  const dim3 block(256), grid((unsigned)phx_div_up(m->nc, 256));
  if (m->cell_type == PHX_TRIANGLE) {
    m->ne = m->nf;
    m->c2e = m->c2f;
    m->c2e_is_alias = true;
  } else if (m->is_box) {
    EdgeDesc E;
    for (int a = 0; a < 3; ++a) E.n[a] = m->box_n[a];
    for (int cls = 0; cls < 7; ++cls) {
      // an edge class spans (n_a) cubes along every axis it moves in, (n_a + 1) otherwise
      const bool mv[7][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {1, 1, 0}, {1, 0, 1}, {0, 1, 1}, {1, 1, 1}};
      for (int a = 0; a < 3; ++a) E.ext[cls][a] = mv[cls][a] ? E.n[a] : E.n[a] + 1;
    }
    E.base[0] = 0;
    for (int cls = 0; cls < 7; ++cls) E.base[cls + 1] = E.base[cls] + E.ext[cls][0] * E.ext[cls][1] * E.ext[cls][2];
    m->ne = E.base[7];
    PHX_REQUIRE(m->ne < INT32_MAX, PHX_ERR_VALUE, "too many edges for 32-bit local ids");
    PHX_HIP(phx_malloc(&m->c2e, sizeof(int32_t) * (size_t)m->nc * 6));
    k_box_c2e<<<grid, block, 0, m->stream>>>(E, m->nc, m->c2e);
  } else {
    // unstructured tetrahedra: number the edges by the rank of their sorted vertex pair (host)
    std::vector<int32_t> cells((size_t)m->nc * 4);
    PHX_HIP(hipMemcpy(cells.data(), m->cells, cells.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    const int te[6][2] = {{2, 3}, {1, 3}, {1, 2}, {0, 3}, {0, 2}, {0, 1}};
    struct Rec { int32_t a, b; int64_t slot; };
    std::vector<Rec> recs((size_t)m->nc * 6);
    for (int64_t c = 0; c < m->nc; ++c)
      for (int k = 0; k < 6; ++k) {
        const int32_t va = cells[c * 4 + te[k][0]], vb = cells[c * 4 + te[k][1]];
        recs[c * 6 + k] = {std::min(va, vb), std::max(va, vb), c * 6 + k};
      }
    std::sort(recs.begin(), recs.end(), [](const Rec &x, const Rec &y) {
      return x.a != y.a ? x.a < y.a : (x.b != y.b ? x.b < y.b : x.slot < y.slot);
    });
    std::vector<int32_t> c2e((size_t)m->nc * 6);
    int64_t ne = 0;
    for (size_t i = 0; i < recs.size(); ++i) {
      if (i > 0 && (recs[i].a != recs[i - 1].a || recs[i].b != recs[i - 1].b)) ++ne;
      c2e[recs[i].slot] = (int32_t)ne;
    }
    m->ne = recs.empty() ? 0 : ne + 1;
    PHX_HIP(phx_malloc(&m->c2e, sizeof(int32_t) * c2e.size()));
    PHX_HIP(hipMemcpy(m->c2e, c2e.data(), sizeof(int32_t) * c2e.size(), hipMemcpyHostToDevice));
  }
  PHX_HIP(phx_malloc(&m->edges, sizeof(int32_t) * 2 * (size_t)m->ne));
  k_edge_vertices<<<grid, block, 0, m->stream>>>(m->nc, m->ci.nvpc, m->cell_type == PHX_TETRAHEDRON ? 6 : 3,
                                                 m->cell_type == PHX_TETRAHEDRON ? 1 : 0, m->cells,
                                                 m->c2e, m->edges);
  PHX_HIP(hipGetLastError());
  PHX_HIP(hipStreamSynchronize(m->stream));
  return PHX_OK;
}

extern "C" int phx_mesh_edge_count(phx_mesh *m, int64_t *ne) {
  PHX_HIP(hipSetDevice(m->device));
  PHX_CHECK(phx_mesh_build_edges(m));
  *ne = m->ne;
  return PHX_OK;
}

extern "C" int phx_mesh_destroy(phx_mesh *m) {
  if (!m) return PHX_OK;
  (void)hipSetDevice(m->device);
  (void)hipDeviceSynchronize();
  void *ptrs[] = {m->x, m->cells, m->c2f, m->f2c, m->bfacets, m->bfacet_ids, m->cell_tags,
                  m->facet_tags, m->ent_buf[0], m->ent_buf[1], m->facet_exempt, m->v2c_ptr, m->v2c_idx,
                  m->edges, m->c2e_is_alias ? nullptr : (void *)m->c2e};
  for (void *p : ptrs) (void)phx_free(p);
  free(m->c_map_h); free(m->v_map_h);
  (void)phx_free(m->v2lat); (void)phx_free(m->lat2v);
  (void)phx_free(m->in_cmap); (void)phx_free(m->in_fmap);
  if (m->inner) { phx_mesh_destroy(m->inner); m->inner = nullptr; }
  if (m->scal_h) (void)hipHostFree(m->scal_h);
  for (auto &pe : m->prof_ev) for (auto &e : pe) (void)hipEventDestroy(e);
  for (int w = 0; w < 2; ++w) (void)phx_free(m->sel_counts[w]);
  (void)phx_free(m->sel_counts_cut);
  (void)phx_free(m->act_in); (void)phx_free(m->act_cut);
  if (m->ev0) (void)hipEventDestroy(m->ev0);
  if (m->ev1) (void)hipEventDestroy(m->ev1);
  if (m->stream && m->own_stream) (void)hipStreamDestroy(m->stream);
  delete m;
  return PHX_OK;
}

extern "C" int phx_mesh_counts(const phx_mesh *m, int64_t *counts) {
  counts[0] = m->gdim; counts[1] = m->cell_type; counts[2] = m->nv;
  counts[3] = m->nc; counts[4] = m->nf; counts[5] = m->nbf;
  return PHX_OK;
}

struct RbArgs { const uint8_t *src[8]; int bytes[8], off[8], n; };
__global__ void k_rb_pack(RbArgs a, uint8_t *__restrict__ out) {
  for (int i = 0; i < a.n; ++i)
    for (int b = threadIdx.x; b < a.bytes[i]; b += blockDim.x) out[a.off[i] + b] = a.src[i][b];
}
int phx_read_back(hipStream_t st, const phx_rb_item *items, int n) {
  PHX_REQUIRE(n >= 1 && n <= 8, PHX_ERR_VALUE, "read-back of %d items", n);
  // staging blocks per (device, stream): a handful per process, alive until it ends
  struct Stage { uint8_t *dev = nullptr, *pin = nullptr; };
  static std::mutex mu;
  static std::map<std::pair<int, hipStream_t>, Stage> stages;
  int dev = 0;
  PHX_HIP(hipGetDevice(&dev));
  Stage sg;
  {
    std::lock_guard<std::mutex> lk(mu);
    Stage &ref = stages[{dev, st}];
    if (!ref.dev) {
      PHX_HIP(hipMalloc(&ref.dev, 512));
      PHX_HIP(hipHostMalloc(&ref.pin, 512));
    }
    sg = ref;
  }
  RbArgs a;
  memset(&a, 0, sizeof(a));
  a.n = n;
  int tot = 0;
  for (int i = 0; i < n; ++i) {
    PHX_REQUIRE(items[i].bytes > 0 && items[i].bytes <= 64, PHX_ERR_VALUE, "read-back item of %d bytes", items[i].bytes);
    a.src[i] = static_cast<const uint8_t *>(items[i].dev);
    a.bytes[i] = items[i].bytes;
    a.off[i] = tot;
    tot += (items[i].bytes + 7) & ~7;
  }
  k_rb_pack<<<1, 64, 0, st>>>(a, sg.dev);
  PHX_HIP(hipGetLastError());
  PHX_HIP(hipMemcpyAsync(sg.pin, sg.dev, (size_t)tot, hipMemcpyDeviceToHost, st));
  PHX_HIP(hipStreamSynchronize(st));
  for (int i = 0; i < n; ++i) memcpy(items[i].host, sg.pin + a.off[i], (size_t)items[i].bytes);
  return PHX_OK;
}

int phx_mesh_pinned_scalars(phx_mesh *m, double **out) {
  if (!m->scal_h) PHX_HIP(hipHostMalloc(&m->scal_h, sizeof(double) * 16));
  *out = m->scal_h;
  return PHX_OK;
}

extern "C" int phx_mesh_stream(phx_mesh *m, uint64_t *stream) {
  *stream = (uint64_t)(uintptr_t)m->stream;
  return PHX_OK;
}

extern "C" int phx_mesh_set_stream(phx_mesh *m, uint64_t stream) {
  PHX_HIP(hipSetDevice(m->device));
  PHX_HIP(hipStreamSynchronize(m->stream));
  if (m->own_stream) PHX_HIP(hipStreamDestroy(m->stream));
  m->stream = (hipStream_t)(uintptr_t)stream;
  m->own_stream = false;
  if (m->inner) m->inner->stream = m->stream;   // (the inner box never owns a stream)
  return PHX_OK;
}

extern "C" int phx_set_option(phx_mesh *m, int option, int64_t value) {
  if (m->inner) PHX_CHECK(phx_set_option(m->inner, option, value));
  switch (option) {
    case PHX_OPT_PROFILE_SPMV:
      PHX_REQUIRE(value >= 0 && value <= 1024, PHX_ERR_VALUE, "SpMV profiling stride out of range");
      m->profile_spmv = (int)value;
      return PHX_OK;
    case PHX_OPT_HAS_EXTERIOR: m->has_exterior_override = (int)value; return PHX_OK;
    case PHX_OPT_SPMV_VALUE_INDEX: m->spmv_value_index = value != 0; return PHX_OK;
    case PHX_OPT_ALLOW_EMPTY: m->allow_empty = value != 0; return PHX_OK;
    case PHX_OPT_EXPORT_CSR: m->export_csr = value != 0; return PHX_OK;
    case PHX_OPT_STRUCTURED: m->structured = value != 0; return PHX_OK;
    case PHX_OPT_DETERMINISTIC: m->deterministic = value != 0; return PHX_OK;
    case PHX_OPT_EL_COARSE:
      PHX_REQUIRE(value == -1 || value == 0 || (value >= 5 && value <= 4096), PHX_ERR_VALUE, "coarse ratio %lld: -1, 0 or >= 5", (long long)value);
      m->el_coarse = (int)value;
      return PHX_OK;
    case PHX_OPT_PRECOND:
      PHX_REQUIRE(value >= 0 && value <= 2, PHX_ERR_VALUE, "unknown preconditioner %lld", (long long)value);
      m->precond = (int)value;
      return PHX_OK;
    case PHX_OPT_SPMV_XCD_GROUP:
      PHX_REQUIRE(value >= 0 && value < (1 << 24), PHX_ERR_VALUE, "XCD group size out of range");
      m->spmv_xcd_group = (int)value;
      return PHX_OK;
    case PHX_OPT_STENCIL_PLANE_ROWS:
      PHX_REQUIRE(value >= 0, PHX_ERR_VALUE, "rows per plane out of range");
      m->stencil_plane_rows = value;
      return PHX_OK;
    default: phx_set_error("unknown option %d", option); return PHX_ERR_VALUE;
  }
}

extern "C" int phx_mesh_tag_histogram(const phx_mesh *m, int64_t *cells4, int64_t *facets7) {
  if (cells4) for (int i = 0; i < 4; ++i) cells4[i] = m->tag_hist[i];
  if (facets7) for (int i = 0; i < 7; ++i) facets7[i] = m->ftag_hist[i];
  return PHX_OK;
}

extern "C" int phx_mesh_synchronize(phx_mesh *m) {
  PHX_HIP(hipSetDevice(m->device));
  PHX_HIP(hipStreamSynchronize(m->stream));
  return PHX_OK;
}

__global__ void k_widen_tags(int64_t n, const int8_t *__restrict__ in, int32_t *__restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = (int32_t)(in[i] & PHX_TAG_MASK);
}

extern "C" int phx_mesh_get_array(phx_mesh *m, int which, void *out, int loc) {
  PHX_HIP(hipSetDevice(m->device));
  const hipMemcpyKind kind = loc == PHX_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  const void *src = nullptr;
  size_t bytes = 0;
  switch (which) {
    case PHX_ARR_COORDS: src = m->x; bytes = sizeof(double) * m->nv * m->gdim; break;
    case PHX_ARR_CELLS: src = m->cells; bytes = sizeof(int32_t) * m->nc * m->ci.nvpc; break;
    case PHX_ARR_C2F: src = m->c2f; bytes = sizeof(int32_t) * m->nc * m->ci.nfpc; break;
    case PHX_ARR_F2C: src = m->f2c; bytes = sizeof(int32_t) * m->nf * 2; break;
    case PHX_ARR_BFACETS: src = m->bfacets; bytes = sizeof(int32_t) * m->nbf * 2; break;
    case PHX_ARR_C2E:
      PHX_CHECK(phx_mesh_build_edges(m));
      src = m->c2e; bytes = sizeof(int32_t) * m->nc * (m->cell_type == PHX_TETRAHEDRON ? 6 : 3); break;
    case PHX_ARR_EDGES:
      PHX_CHECK(phx_mesh_build_edges(m));
      src = m->edges; bytes = sizeof(int32_t) * m->ne * 2; break;
    case PHX_ARR_CELL_TAGS:
    case PHX_ARR_FACET_TAGS: {
      const bool fac = which == PHX_ARR_FACET_TAGS;
      PHX_REQUIRE(fac ? m->have_facet_tags : m->have_cell_tags, PHX_ERR_VALUE, "tags not computed yet");
      const int64_t n = fac ? m->nf : m->nc;
      int32_t *dst = (int32_t *)out;
      int32_t *tmp = nullptr;
      if (loc != PHX_DEVICE) { PHX_HIP(phx_malloc(&tmp, sizeof(int32_t) * (size_t)n)); dst = tmp; }
      k_widen_tags<<<dim3((unsigned)phx_div_up(n, 256)), dim3(256), 0, m->stream>>>(
          n, fac ? m->facet_tags : m->cell_tags, dst);
      if (tmp) PHX_HIP(hipMemcpyAsync(out, tmp, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, m->stream));
      PHX_HIP(hipStreamSynchronize(m->stream));  // the caller may read `out` from another stream
      if (tmp) PHX_HIP(phx_free(tmp));
      return PHX_OK;
    }
    default:
      phx_set_error("unknown array selector %d", which);
      return PHX_ERR_VALUE;
  }
  if (bytes) PHX_HIP(hipMemcpyAsync(out, src, bytes, kind, m->stream));
  PHX_HIP(hipStreamSynchronize(m->stream));
  return PHX_OK;
}

extern "C" int phx_last_timings(const phx_mesh *m, double *t) {
  for (int i = 0; i < 8; ++i) t[i] = m->timings[i];
  if (m->inner) for (int i = 2; i < 8; ++i) if (m->inner->timings[i] != 0.0) t[i] = m->inner->timings[i];   // assemble / solve ran there
  return PHX_OK;
}
