// Weak-Dirichlet phi-FEM Poisson assembly on the device (gfx950): element integration, scatter
// into per-row slots with f64 atomics, compaction to CSR, conversion to SELL-64 for the solver.
// Replaces the FFCx-generated tabulate_tensor kernels + dolfinx assemble_matrix/assemble_vector +
// PETSc MatSetValuesLocal behind demo/weak-dirichlet/flower/main.py:112-139 (bilinear form) and
// :142-154 (linear form) [3P].  P1 x P1 on affine simplices: every integrand is a polynomial, the
// closed-form simplex integrals below equal any exact quadrature to round-off.
//
// Only ACTIVE DoFs get rows (u on vertices of cells tagged 1/2, p on vertices of cut cells): the
// reference's matrix has empty rows elsewhere and relies on MUMPS null-pivot detection
// (main.py:169-173); restricting to the active set is the same solution (hard part 2).
#include "phx_prim.h"
#include <string.h>

#include <algorithm>
#include <vector>

#include "phx_common.h"
#include "phx_select.h"

int phx_collect_entities(phx_mesh *m);
int phx_system_build_sell(phx_system *s);

// ---------------------------------------------------------------------------------------------
// active DoF numbering
// ---------------------------------------------------------------------------------------------
// fu / fp from the vertex flags the single-layer tagging pass left behind (phx_common.h: act_in, act_cut)
__global__ void k_flags_from_tagging(int64_t nv, const uint8_t *__restrict__ vin, const uint8_t *__restrict__ vcut,
                                     uint8_t *__restrict__ fu, uint8_t *__restrict__ fp) {
  const int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (v >= nv) return;
  const uint8_t c = vcut[v] ? 1 : 0;
  fu[v] = (vin[v] || c) ? 1 : 0;
  fp[v] = c;
}

template <int NVPC>
__global__ void k_mark_active(int64_t nc, const int32_t *__restrict__ cells,
                              const int8_t *__restrict__ tags, uint8_t *__restrict__ fu,
                              uint8_t *__restrict__ fp) {
  const int64_t c0 = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) * 4;   // four cells per thread (phx_tag_word)
  if (c0 >= nc) return;
  const uint32_t w = phx_tag_word(tags, c0, nc);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int t = (int)((w >> (8 * j)) & PHX_TAG_MASK);
    if (t != 1 && t != 2) continue;
    const int64_t c = c0 + j;
    for (int i = 0; i < NVPC; ++i) {
      const int32_t v = cells[c * NVPC + i];
      fu[v] = 1;
      if (t == 2) fp[v] = 1;
    }
  }
}

__global__ void k_finish_numbering(int64_t nv, const uint8_t *__restrict__ fu,
                                   const uint8_t *__restrict__ fp,
                                   const int32_t *__restrict__ su, const int32_t *__restrict__ sp,
                                   int32_t nu, int32_t *__restrict__ du, int32_t *__restrict__ dp,
                                   int64_t *__restrict__ full_of_active) {
  const int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (v >= nv) return;
  int32_t a = -1, b = -1;
  if (fu[v]) { a = su[v]; full_of_active[a] = v; }
  if (fp[v]) { b = nu + sp[v]; full_of_active[b] = nv + v; }
  du[v] = a;
  dp[v] = b;
}
// sup[v]: ranks of vertex v among the flagged u (low word) and p (high word) vertices -- ONE 64-bit scan of both flags
__global__ void k_finish_numbering_packed(int64_t nv, const uint8_t *__restrict__ fu,
                                   const uint8_t *__restrict__ fp,
                                   const unsigned long long *__restrict__ sup,
                                   int32_t nu, int32_t *__restrict__ du, int32_t *__restrict__ dp,
                                   int64_t *__restrict__ full_of_active, int u_or_p) {
  const int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (v >= nv) return;
  int32_t a = -1, b = -1;
  const unsigned long long r = sup[v];
  if (fu[v] || (u_or_p && fp[v])) { a = (int32_t)(r & 0xffffffffull); full_of_active[a] = v; }
  if (fp[v]) { b = nu + (int32_t)(r >> 32); full_of_active[b] = nv + v; }
  du[v] = a;
  dp[v] = b;
}

// ---------------------------------------------------------------------------------------------
// row slots: a small open-addressing table of W (col,val) pairs per row.  A column claims a free
// slot with a CAS; values accumulate with hardware f64 atomics (global_atomic_add_f64).
// ---------------------------------------------------------------------------------------------
struct Slots {
  int32_t *cols;
  double *vals;
  int W;
  int *overflow;
  // clean[row] = c > 0: the row has one writer (the gather kernel) that stored its c entries at slots
  // 0 .. c-1 in ascending column order: no hashing, no sort at compaction.  nullptr: no such rows.
  uint8_t *clean = nullptr;
  // per-row capacities (interface elasticity: 64 slots for the bulk rows, 256+ for the rows of cut-cell
  // vertices): row r owns slots [off[r], off[r] + (1 << wlog[r])).  nullptr: every row has W slots.
  const int64_t *off = nullptr;
  const uint8_t *wlog = nullptr;
  // Deterministic accumulation (PHX_OPT_DETERMINISTIC; P2 and elasticity assemblies): f64 atomics add in an order that
  // changes from run to run, the last bits of the matrix with it, and BiCGStab amplifies them over hundreds of
  // iterations (694-892 for the same P2 problem).  The element kernels then run TWICE: pass 1 records per slot the
  // largest exponent among its contributions (integer atomicMax: order-free), pass 2 splits every contribution v into
  // hi = v rounded to 2^(E-40) and lo = the rest rounded to 2^(E-81) and adds them to two accumulators -- sums of
  // fewer than 2^11 such terms are EXACT in f64, so any order gives the same bits.  The slot value is hi + lo (one
  // rounding; what is dropped is below 2^-81 of the largest contribution).  emax == nullptr: plain atomic adds.
  int32_t *emax = nullptr;   // [slots]
  double *lo = nullptr;      // [slots]
  int32_t *remax = nullptr;  // [rows] the same for the right-hand side
  double *rlo = nullptr;     // [rows]
  int pass = 0;
};

__device__ __forceinline__ int det_expo(double v) { return (int)((__double_as_longlong(v) >> 52) & 0x7ff); }
__device__ __forceinline__ void det_split(double v, int e, double &hi, double &lo) {
  const int E = e - 1023;                       // every |contribution| < 2^(E+1)
  hi = ldexp(rint(ldexp(v, 40 - E)), E - 40);
  lo = ldexp(rint(ldexp(v - hi, 81 - E)), E - 81);
}
// accumulate v into acc[i] (plain atomics, or the two-pass exact scheme above)
__device__ __forceinline__ void det_add(const Slots &s, double *acc, int32_t *emax, double *lo, int64_t i, double v) {
  if (!emax) { unsafeAtomicAdd(&acc[i], v); return; }
  if (s.pass == 1) { atomicMax(&emax[i], det_expo(v)); return; }
  double hi, l;
  det_split(v, emax[i], hi, l);
  unsafeAtomicAdd(&acc[i], hi);
  unsafeAtomicAdd(&lo[i], l);
}
__device__ __forceinline__ void slot_rhs_add(const Slots &s, double *rhs, int32_t row, double v) {
  det_add(s, rhs, s.remax, s.rlo, row, v);
}

__device__ __forceinline__ int64_t slot_base(const Slots &s, int32_t row, int *W) {
  if (s.off) { *W = 1 << s.wlog[row]; return s.off[row]; }
  *W = s.W;
  return (int64_t)row * s.W;
}

// ANYKEY: the key may be negative (structured P2 systems encode a p column e as -2 - e: 2 (nv + ne) exceeds 2^31 at
// 512^3); -1 stays the empty marker
template <bool ANYKEY = false>
__device__ __forceinline__ void slot_add(const Slots &s, int32_t row, int32_t col, double v) {
  if (row < 0 || (ANYKEY ? col == -1 : col < 0)) return;  // inactive DoF (only reachable through user-overwritten tags)
  int W;
  const int64_t base = slot_base(s, row, &W);
  int32_t *rc = s.cols + base;
  double *rv = s.vals + base;
  // open addressing inside the row: start at a hash of the column (W is a power of two), so a
  // lookup costs ~1-2 probes instead of a walk over the row's whole prefix
  const int mask = W - 1;
  int k = (int)(((uint32_t)col * 2654435761u) >> 16) & mask;
  for (int t = 0; t < W; ++t, k = (k + 1) & mask) {
    int32_t cur = __hip_atomic_load(&rc[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == -1) {
      cur = atomicCAS(&rc[k], -1, col);
      if (cur == -1) cur = col;
    }
    if (cur == col) {
      // the slot is claimed either way (the structural pattern is dolfinx's); an exactly vanishing contribution -- on a
      // Kuhn box most products of two normal-derivative jumps of the ghost penalty are: the gradients of non-neighbouring
      // path vertices are orthogonal, the zeros are exact -- needs no f64 atomic
      if (v != 0.0) det_add(s, s.vals, s.emax, s.lo, base + k, v);
      return;
    }
  }
  atomicOr(s.overflow, 1);
}

// ---------------------------------------------------------------------------------------------
// affine simplex geometry: gradients of the barycentric coordinates, volume, diameter
// ---------------------------------------------------------------------------------------------
template <int D>
struct Geo {
  double g[D + 1][D];
  double vol, h;
};

template <int D>
__device__ __forceinline__ void simplex_geometry(const double (*X)[D], Geo<D> &G) {
  if constexpr (D == 2) {
    const double a = X[1][0] - X[0][0], b = X[2][0] - X[0][0];
    const double c = X[1][1] - X[0][1], d = X[2][1] - X[0][1];
    const double det = a * d - b * c;
    const double id = 1.0 / det;
    G.g[1][0] = d * id;  G.g[1][1] = -b * id;
    G.g[2][0] = -c * id; G.g[2][1] = a * id;
    G.g[0][0] = -(G.g[1][0] + G.g[2][0]);
    G.g[0][1] = -(G.g[1][1] + G.g[2][1]);
    G.vol = 0.5 * fabs(det);
  } else {
    double e[3][3];  // e[k] = X[k+1]-X[0]
    for (int k = 0; k < 3; ++k)
      for (int d = 0; d < 3; ++d) e[k][d] = X[k + 1][d] - X[0][d];
    // rows of J^-1 are cross products / det  (J columns = e[k])
    double cr[3][3];
    cr[0][0] = e[1][1] * e[2][2] - e[1][2] * e[2][1];
    cr[0][1] = e[1][2] * e[2][0] - e[1][0] * e[2][2];
    cr[0][2] = e[1][0] * e[2][1] - e[1][1] * e[2][0];
    cr[1][0] = e[2][1] * e[0][2] - e[2][2] * e[0][1];
    cr[1][1] = e[2][2] * e[0][0] - e[2][0] * e[0][2];
    cr[1][2] = e[2][0] * e[0][1] - e[2][1] * e[0][0];
    cr[2][0] = e[0][1] * e[1][2] - e[0][2] * e[1][1];
    cr[2][1] = e[0][2] * e[1][0] - e[0][0] * e[1][2];
    cr[2][2] = e[0][0] * e[1][1] - e[0][1] * e[1][0];
    const double det = e[0][0] * cr[0][0] + e[0][1] * cr[0][1] + e[0][2] * cr[0][2];
    const double id = 1.0 / det;
    for (int k = 0; k < 3; ++k)
      for (int d = 0; d < 3; ++d) G.g[k + 1][d] = cr[k][d] * id;
    for (int d = 0; d < 3; ++d) G.g[0][d] = -(G.g[1][d] + G.g[2][d] + G.g[3][d]);
    G.vol = fabs(det) * (1.0 / 6.0);
  }
  // ufl.CellDiameter (main.py:108): largest vertex-vertex distance
  double h2 = 0.0;
  for (int i = 0; i <= D; ++i)
    for (int j = i + 1; j <= D; ++j) {
      double s = 0.0;
      for (int d = 0; d < D; ++d) { const double t = X[i][d] - X[j][d]; s += t * t; }
      h2 = fmax(h2, s);
    }
  G.h = sqrt(h2);
}

template <int D>
__device__ __forceinline__ void load_cell(const int32_t *__restrict__ cells,
                                          const double *__restrict__ x, int64_t c, int32_t *v,
                                          double (*X)[D]) {
  for (int i = 0; i <= D; ++i) {
    v[i] = cells[c * (D + 1) + i];
    for (int d = 0; d < D; ++d) X[i][d] = x[(int64_t)v[i] * D + d];
  }
}

// Column keys inside the slot tables are FULL DoF indices (u: v, p: nv + v), straight from the
// connectivity; the compaction kernel translates them to active rows (one dof-map lookup per
// stored entry instead of one per contribution).
struct AsmArgs {
  const uint8_t *touched = nullptr;  // [nv] vertex gets contributions from the scattering kernels (nullptr: all do)
  int32_t nv;
  const int32_t *cells;
  const double *x;
  const int8_t *ctags;
  const int8_t *ftags;
  const int32_t *c2f;
  const int32_t *f2c;
  const int32_t *du;
  const int32_t *dp;
  const double *phi, *f, *ud;
  double gamma, sigma;
  double *rhs;
  Slots slots;
  // structured systems (phx_common.h): rows of C0 are not stored -- the row kernel leaves their diagonal in
  // `diag`, one of them the stencil coefficients in `stencil`; store_c0 != 0 (PHX_OPT_EXPORT_CSR) stores them too
  const uint8_t *c0 = nullptr;
  double *diag = nullptr;
  double *stencil = nullptr;
  int store_c0 = 1;
};

// (1/|K|) int N_i N_j N_k N_l = D! alpha! / (D+4)!
__device__ __forceinline__ double mult4(int i, int j, int k, int l) {
  int cnt[4] = {0, 0, 0, 0};
  cnt[i]++; cnt[j]++; cnt[k]++; cnt[l]++;
  double r = 1.0;
  for (int a = 0; a < 4; ++a) r *= (cnt[a] == 2 ? 2.0 : (cnt[a] == 3 ? 6.0 : (cnt[a] == 4 ? 24.0 : 1.0)));
  return r;
}

// Work lists: the element kernels run over COMPACTED lists (inside cells, cut cells, stabilised
// facets) with one lane per entry of the element tensor, so a wavefront issues 64 independent
// slot updates instead of one lane walking a whole element matrix.
struct SelOmega { const int8_t *t; __host__ __device__ bool operator()(const int32_t &c) const { const int v = t[c] & PHX_TAG_MASK; return v == 1 || v == 2; } };
struct SelCut {
  const int8_t *t;
  __host__ __device__ bool operator()(const int32_t &c) const { return (t[c] & PHX_TAG_MASK) == 2; }
  __host__ __device__ const int8_t *bytes() const { return t; }           // byte fast path of phx_select.h
  __host__ __device__ bool test(int tag, int32_t) const { return (tag & PHX_TAG_MASK) == 2; }
};
struct SelGhostFacet {
  const int8_t *ft; const int32_t *f2c;
  __host__ __device__ bool operator()(const int32_t &f) const {
    const int t = ft[f];
    return (t == 2 || t == 3) && f2c[2 * (int64_t)f + 1] >= 0;  // dS: interior facets only
  }
  __host__ __device__ const int8_t *bytes() const { return ft; }
  __host__ __device__ bool test(int t, int32_t f) const { return (t == 2 || t == 3) && f2c[2 * (int64_t)f + 1] >= 0; }
};

// --- vertex -> cell adjacency (once per mesh) ---------------------------------------------------
__global__ void k_v2c_count(int64_t nc, int nvpc, const int32_t *__restrict__ cells,
                            unsigned long long *__restrict__ cnt) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= nc * nvpc) return;
  atomicAdd(&cnt[cells[i]], 1ull);
}
__global__ void k_v2c_fill(int64_t nc, int nvpc, const int32_t *__restrict__ cells,
                           const int64_t *__restrict__ ptr, unsigned long long *__restrict__ cursor,
                           int32_t *__restrict__ idx) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= nc * nvpc) return;
  const int32_t v = cells[i];
  const unsigned long long k = atomicAdd(&cursor[v], 1ull);
  idx[ptr[v] + (int64_t)k] = (int32_t)(i / nvpc);
}

// plain (non-atomic) insert into a row this thread owns exclusively; same hash as slot_add
__device__ __forceinline__ void slot_add_owned(const Slots &s, int32_t row, int32_t col, double v) {
  if (s.emax) { slot_add(s, row, col, v); return; }   // deterministic mode: a contribution like any other (both passes)
  int W;
  const int64_t base = slot_base(s, row, &W);
  int32_t *rc = s.cols + base;
  double *rv = s.vals + base;
  const int mask = W - 1;
  int k = (int)(((uint32_t)col * 2654435761u) >> 16) & mask;
  for (int t = 0; t < W; ++t, k = (k + 1) & mask) {
    const int32_t cur = rc[k];
    if (cur == -1) { rc[k] = col; rv[k] = v; return; }
    if (cur == col) { rv[k] += v; return; }
  }
  atomicOr(s.overflow, 1);
}

// --- stiffness + source rows, main.py:113 + :143 over dx((1,2)): one thread per ACTIVE VERTEX
// gathers row i of every incident element tensor into a thread-private table in LDS (32 slots,
// slot-major so that lanes sit on consecutive banks), then writes the row once.  No atomics: the
// kernel runs first on the cleared slot tables and every row has exactly one writer. --------------
#define ROW_LDS_SLOTS 32
#define ROW_THREADS 128
struct RowAcc {
  int32_t *col;   // [ROW_LDS_SLOTS][ROW_THREADS]
  double *val;
  int t;
  const Slots *spill;
  int32_t row;
  __device__ __forceinline__ void add(int32_t key, double v) {
    int k = (int)(((uint32_t)key * 2654435761u) >> 16) & (ROW_LDS_SLOTS - 1);
    for (int q = 0; q < ROW_LDS_SLOTS; ++q, k = (k + 1) & (ROW_LDS_SLOTS - 1)) {
      const int32_t cur = col[k * ROW_THREADS + t];
      if (cur == -1) { col[k * ROW_THREADS + t] = key; val[k * ROW_THREADS + t] = v; return; }
      if (cur == key) { val[k * ROW_THREADS + t] += v; return; }
    }
    slot_add_owned(*spill, row, key, v);  // vertex of very high valence: straight to the row
  }
};

template <int D>
__global__ void __launch_bounds__(ROW_THREADS)
k_assemble_rows(int64_t nv, const int64_t *__restrict__ v2c_ptr, const int32_t *__restrict__ v2c_idx, AsmArgs A) {
  __shared__ int32_t lcol[ROW_LDS_SLOTS * ROW_THREADS];
  __shared__ double lval[ROW_LDS_SLOTS * ROW_THREADS];
  const int64_t vtx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (vtx >= nv) return;
  const int32_t row = A.du[vtx];
  if (row < 0) return;
  RowAcc acc{lcol, lval, (int)threadIdx.x, &A.slots, row};
  for (int k = 0; k < ROW_LDS_SLOTS; ++k) lcol[k * ROW_THREADS + threadIdx.x] = -1;
  constexpr int N = D + 1;
  constexpr double c2 = D == 3 ? 1.0 / 20.0 : 1.0 / 12.0;
  double rhs = 0.0;
  for (int64_t e = v2c_ptr[vtx]; e < v2c_ptr[vtx + 1]; ++e) {
    const int64_t c = v2c_idx[e];
    const int t = A.ctags[c] & PHX_TAG_MASK;
    if (t != 1 && t != 2) continue;
    int32_t v[N];
    double X[N][D];
    load_cell<D>(A.cells, A.x, c, v, X);
    Geo<D> G;
    simplex_geometry<D>(X, G);
    int i = 0;
    double sf = 0.0;
    for (int q = 0; q < N; ++q) { if (v[q] == (int32_t)vtx) i = q; sf += A.f[v[q]]; }
    for (int j = 0; j < N; ++j) {
      double k = 0.0;
      for (int d = 0; d < D; ++d) k += G.g[i][d] * G.g[j][d];
      acc.add(v[j], k * G.vol);
    }
    rhs += G.vol * c2 * (sf + A.f[vtx]);  // int f_h N_i
  }
  A.rhs[row] = rhs;
  for (int k = 0; k < ROW_LDS_SLOTS; ++k) {
    const int32_t key = lcol[k * ROW_THREADS + threadIdx.x];
    if (key != -1) slot_add_owned(A.slots, row, key, lval[k * ROW_THREADS + threadIdx.x]);
  }
}

// --- the same rows on a Kuhn box (phx_mesh_create_box): the star of a vertex is known in closed form.
// Simplex t of cube o walks o, o+e_p0, o+e_p0+e_p1(, ...) for the t-th axis permutation p
// (k_box_cells), so vertex V is path vertex m of simplex t of the cube at V - e_p0 - ... - e_p(m-1):
// 24 tetrahedra / 6 triangles and 14 / 6 neighbours at fixed lattice offsets.  With every loop
// unrolled the accumulators are registers addressed at compile time: no v2c list, no connectivity
// loads, no LDS hash.  The simplices are built from the lattice spacing h_a = (hi_a - lo_a) / n_a, not from
// differences of the stored (rounded) coordinates: a box IS uniform, and with translation-invariant
// geometry every interior row runs the very same arithmetic on the very same numbers for ANY n, which
// is what lets k_sell_index fold the interior values into a dictionary (entries differ from a
// coordinate-based evaluation by a few ulp).
struct BoxDims { int64_t n[3]; double h[3]; };

// C0: active vertices strictly inside the box that no scattering kernel touches and whose whole star (24 tets,
// 2-D: 6 triangles) is tagged inside.  Their rows are the translation-invariant lattice Laplacian row.
template <int D>
__global__ void __launch_bounds__(256)
k_mark_c0(int64_t nv, BoxDims bd, const int32_t *__restrict__ du, const int8_t *__restrict__ ctags,
          const uint8_t *__restrict__ touched, uint8_t *__restrict__ c0) {
  constexpr int N = D + 1, NPERM = D == 3 ? 6 : 2;
  constexpr int P[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
  constexpr int P2[2][3] = {{0, 1, 0}, {1, 0, 0}};
  const int64_t vtx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (vtx >= nv) return;
  const int32_t row = du[vtx];
  if (row < 0) return;
  const int64_t n0 = bd.n[0] + 1, n1 = bd.n[1] + 1;
  const int64_t idx[3] = {vtx % n0, D == 3 ? (vtx / n0) % n1 : vtx / n0, D == 3 ? vtx / (n0 * n1) : 0};
  const int64_t cstride[3] = {1, bd.n[0], bd.n[0] * bd.n[1]};
  bool ok = !touched[vtx];
  for (int a = 0; a < D; ++a) ok = ok && idx[a] >= 1 && idx[a] <= bd.n[a] - 1;
  if (ok) {
#pragma unroll
    for (int t = 0; t < NPERM; ++t) {
#pragma unroll
      for (int m = 0; m < N; ++m) {
        int dd[3] = {0, 0, 0};
        for (int q = 0; q < m; ++q) dd[D == 3 ? P[t][q] : P2[t][q]] -= 1;
        int64_t cube = 0;
        for (int a = 0; a < D; ++a) cube += (idx[a] + dd[a]) * cstride[a];
        ok = ok && (ctags[cube * NPERM + t] & PHX_TAG_MASK) == 1;
      }
    }
  }
  c0[row] = ok ? 1 : 0;
}

__global__ void k_not_flags(int64_t n, const uint8_t *__restrict__ in, uint8_t *__restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] ? 0 : 1;
}

// slot offsets of the rows that are stored: rank[row] = number of stored rows before it (exclusive scan of 1 - c0)
__global__ void k_slot_offsets(int64_t n, int W, int wl, const uint8_t *__restrict__ c0, const int32_t *__restrict__ rank,
                               int64_t *__restrict__ off, uint8_t *__restrict__ wlog) {
  const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r >= n) return;
  off[r] = c0[r] ? 0 : (int64_t)rank[r] * W;   // a C0 row is never written: any in-range offset
  wlog[r] = (uint8_t)wl;
}

// The row of a vertex whose whole star lies inside (a C0 row) is the same for every such vertex of the box: one
// thread evaluates it once -- the very arithmetic k_assemble_rows_box runs for a stored row -- and leaves
// {diagonal, x, y, z neighbour coefficient} in stencil[0..3].  The C0 rows then only compute their right-hand side.
template <int D>
__global__ void k_box_stencil(BoxDims bd, double *__restrict__ stencil) {
  constexpr int N = D + 1, NPERM = D == 3 ? 6 : 2, NCODE = D == 3 ? 27 : 9;
  constexpr int P[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
  constexpr int P2[2][3] = {{0, 1, 0}, {1, 0, 0}};
  constexpr int POW3[3] = {1, 3, 9};
  constexpr int SELF = D == 3 ? 13 : 4;
  double acc[NCODE];
  for (int c = 0; c < NCODE; ++c) acc[c] = 0.0;
  for (int t = 0; t < NPERM; ++t)
    for (int m = 0; m < N; ++m) {
      int dd[3] = {0, 0, 0};
      for (int q = 0; q < m; ++q) dd[D == 3 ? P[t][q] : P2[t][q]] -= 1;
      int code[N];
      double X[N][D];
      for (int q = 0; q < N; ++q) {
        code[q] = 0;
        for (int a = 0; a < D; ++a) code[q] += (dd[a] + 1) * POW3[a];
        for (int a = 0; a < D; ++a) X[q][a] = (double)dd[a] * bd.h[a];
        if (q < D) dd[D == 3 ? P[t][q] : P2[t][q]] += 1;
      }
      Geo<D> G;
      simplex_geometry<D>(X, G);
      for (int j = 0; j < N; ++j) {
        double k = 0.0;
        for (int a = 0; a < D; ++a) k += G.g[m][a] * G.g[j][a];
        acc[code[j]] += k * G.vol;
      }
      if (t == 0 && m == 0) stencil[5] = G.vol;   // every Kuhn simplex of the box has this volume
    }
  stencil[0] = acc[SELF];
  stencil[1] = acc[SELF + 1];
  stencil[2] = acc[SELF + 3];
  stencil[3] = D == 3 ? acc[SELF + 9] : 0.0;
}

// the stored rows as a list (rank = exclusive scan of the flags)
__global__ void k_flagged_list(int64_t n, const uint8_t *__restrict__ flags, const int32_t *__restrict__ rank,
                               int32_t *__restrict__ list) {
  const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r < n && flags[r]) list[rank[r]] = (int32_t)r;
}

// mode 0: one thread per vertex, every active u row.  Structured systems split the work: mode 1 -- one thread per
// vertex, the C0 rows only (right-hand side, a few loads each); mode 2 -- one thread per entry of the list of STORED
// rows (`rows`, active indices; p rows skipped), the full closed-form row: the 2e5 stored u rows sit 3-6 to a
// wavefront of consecutive vertices, and a launch over all vertices ran the whole 24-simplex code for those few
// lanes (0.96 ms at 256^3; list-driven the waves are dense).
template <int D>
__global__ void __launch_bounds__(256)
k_assemble_rows_box(int64_t nthreads, BoxDims bd, AsmArgs A, int mode, const int32_t *__restrict__ rows,
                    const int64_t *__restrict__ full, int64_t nu) {
  constexpr int N = D + 1, NPERM = D == 3 ? 6 : 2, NCODE = D == 3 ? 27 : 9;
  constexpr int P[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};  // c_perm3 / c_perm2
  constexpr int P2[2][3] = {{0, 1, 0}, {1, 0, 0}};
  constexpr int POW3[3] = {1, 3, 9};
  constexpr double c2 = D == 3 ? 1.0 / 20.0 : 1.0 / 12.0;
  const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (tid >= nthreads) return;
  int64_t vtx = tid;
  int32_t row;
  if (mode == 2) {
    row = rows[tid];
    if (row >= nu) return;
    vtx = full[row];
  } else {
    row = A.du[vtx];
    if (row < 0) return;
    if (mode == 1 && !A.c0[row]) return;
  }
  const int64_t n0 = bd.n[0] + 1, n1 = bd.n[1] + 1;
  int64_t idx[3] = {vtx % n0, D == 3 ? (vtx / n0) % n1 : vtx / n0, D == 3 ? vtx / (n0 * n1) : 0};
  const int64_t vstride[3] = {1, n0, n0 * n1};
  const int64_t cstride[3] = {1, bd.n[0], bd.n[0] * bd.n[1]};
  // neighbour source values, once (offset code = sum (d_a + 1) 3^a)
  double nf[NCODE], acc[NCODE];
  bool seen[NCODE];
#pragma unroll
  for (int code = 0; code < NCODE; ++code) {
    int d[3] = {code % 3 - 1, (code / 3) % 3 - 1, D == 3 ? code / 9 - 1 : 0};
    // Kuhn neighbours: all non-zero components share one sign
    bool pos = false, neg = false;
    for (int a = 0; a < D; ++a) { pos |= d[a] > 0; neg |= d[a] < 0; }
    acc[code] = 0.0;
    seen[code] = false;
    nf[code] = 0.0;
    if (pos && neg) continue;
    bool in = true;
    int64_t w = vtx;
    for (int a = 0; a < D; ++a) {
      const int64_t q = idx[a] + d[a];
      in = in && q >= 0 && q <= bd.n[a];
      w += d[a] * vstride[a];
    }
    if (!in) continue;
    nf[code] = A.f[w];
  }
  constexpr int SELF = D == 3 ? 13 : 4;
  double rhs = 0.0;
  if (A.c0 && !A.store_c0 && A.c0[row]) {
    // translation-invariant interior row (k_mark_c0: the whole star is inside): applied from the stencil of
    // k_box_stencil, not stored -- only its right-hand side int f_h N_i = |K| c2 sum_star (sum_q f_q + f_i) is
    // left to do: no tag loads, no geometry (2.4e6 of 2.6e6 u rows at 256^3)
    double sf = 0.0;
#pragma unroll
    for (int t = 0; t < NPERM; ++t) {
#pragma unroll
      for (int m = 0; m < N; ++m) {
        int dd[3] = {0, 0, 0};
#pragma unroll
        for (int q = 0; q < m; ++q) dd[D == 3 ? P[t][q] : P2[t][q]] -= 1;
#pragma unroll
        for (int q = 0; q < N; ++q) {
          int code = 0;
#pragma unroll
          for (int a = 0; a < D; ++a) code += (dd[a] + 1) * POW3[a];
          sf += nf[code];
          if (q < D) dd[D == 3 ? P[t][q] : P2[t][q]] += 1;
        }
      }
    }
    A.rhs[row] = A.stencil[5] * c2 * (sf + (double)(NPERM * N) * nf[SELF]);
    A.diag[row] = A.stencil[0];
    return;
  }
#pragma unroll
  for (int t = 0; t < NPERM; ++t) {
#pragma unroll
    for (int m = 0; m < N; ++m) {
      int dd[3] = {0, 0, 0};
      for (int q = 0; q < m; ++q) dd[D == 3 ? P[t][q] : P2[t][q]] -= 1;
      bool in = true;
      int64_t cube = 0;
      for (int a = 0; a < D; ++a) {
        const int64_t o = idx[a] + dd[a];
        in = in && o >= 0 && o < bd.n[a];
        cube += o * cstride[a];
      }
      if (!in) continue;
      const int tag = A.ctags[cube * NPERM + t] & PHX_TAG_MASK;
      if (tag != 1 && tag != 2) continue;
      int code[N];
      double X[N][D];
      double sf = 0.0;
      for (int q = 0; q < N; ++q) {
        code[q] = 0;
        for (int a = 0; a < D; ++a) code[q] += (dd[a] + 1) * POW3[a];
        for (int a = 0; a < D; ++a) X[q][a] = (double)dd[a] * bd.h[a];
        sf += nf[code[q]];
        if (q < D) dd[D == 3 ? P[t][q] : P2[t][q]] += 1;
      }
      Geo<D> G;
      simplex_geometry<D>(X, G);
      for (int j = 0; j < N; ++j) {
        double k = 0.0;
        for (int a = 0; a < D; ++a) k += G.g[m][a] * G.g[j][a];
        acc[code[j]] += k * G.vol;
        seen[code[j]] = true;
      }
      rhs += G.vol * c2 * (sf + nf[SELF]);  // int f_h N_i
    }
  }
  A.rhs[row] = rhs;
  if (A.c0 && A.c0[row]) {
    // translation-invariant interior row: applied from the stencil, not stored (unless the CSR is exported)
    A.diag[row] = acc[SELF];
    if (!A.store_c0) return;
  }
  // a row no scattering kernel will touch is stored densely and already sorted (codes ascend with the
  // vertex index, hence with the column): it skips the hash table and the sort of the compaction
  const bool clean = A.slots.clean && A.touched && !A.touched[vtx];
  int slotW;
  const int64_t sbase = slot_base(A.slots, row, &slotW);
  int cnt = 0;
#pragma unroll
  for (int code = 0; code < NCODE; ++code) {
    if (!seen[code]) continue;
    int64_t w = vtx;
    w += (code % 3 - 1) * vstride[0] + ((code / 3) % 3 - 1) * vstride[1];
    if (D == 3) w += (code / 9 - 1) * vstride[2];
    if (clean) {
      A.slots.cols[sbase + cnt] = (int32_t)w;
      A.slots.vals[sbase + cnt] = acc[code];
      ++cnt;
    } else {
      slot_add_owned(A.slots, row, (int32_t)w, acc[code]);
    }
  }
  if (clean) A.slots.clean[row] = (uint8_t)cnt;
}

// vertices whose rows the scattering kernels (cut cells, one-sided boundary term, ghost penalty) add to
template <int N>
__global__ void k_mark_cells(int64_t n, const int32_t *__restrict__ list, const int32_t *__restrict__ cells,
                             uint8_t *__restrict__ touched) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t c = list[i];
  for (int k = 0; k < N; ++k) touched[cells[c * N + k]] = 1;
}
template <int N>
__global__ void k_mark_facet_cells(int64_t n, const int32_t *__restrict__ list, const int32_t *__restrict__ f2c,
                                   const int32_t *__restrict__ cells, uint8_t *__restrict__ touched) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int side = 0; side < 2; ++side) {
    const int64_t c = f2c[2 * (int64_t)list[i] + side];
    if (c < 0) continue;
    for (int k = 0; k < N; ++k) touched[cells[c * N + k]] = 1;
  }
}
template <int N>
__global__ void k_mark_entity_cells(int64_t n, const int64_t *__restrict__ ent_packed, const int32_t *__restrict__ cells,
                                    uint8_t *__restrict__ touched) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t c = ent_packed[2 * i + 1] >> 8;
  for (int k = 0; k < N; ++k) touched[cells[c * N + k]] = 1;
}

// --- cut cells, main.py:115-122,144-149 (penalisation): 64 lanes per cell, lane = (a, b) of the mixed
// (u,p) x (u,p) element tensor, a = i (u_i) or N + i (p_i) ---------------------------------------
template <int D>
__global__ void __launch_bounds__(256) k_assemble_cut(int64_t nlist, const int32_t *__restrict__ list, AsmArgs A) {
  const int64_t gid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t e = gid >> 6;
  if (e >= nlist) return;
  constexpr int N = D + 1;
  const int a = (int)(gid & 63) >> 3, b = (int)(gid & 7);
  if (a >= 2 * N || b >= 2 * N) return;
  const int64_t c = list[e];
  int32_t v[N];
  double X[N][D];
  load_cell<D>(A.cells, A.x, c, v, X);
  Geo<D> G;
  simplex_geometry<D>(X, G);
  constexpr double c2 = D == 3 ? 1.0 / 20.0 : 1.0 / 12.0;    // (1+d_ij) c2  = int N_i N_j / |K|
  constexpr double c3 = D == 3 ? 1.0 / 120.0 : 1.0 / 60.0;   // alpha! c3
  constexpr double c4 = D == 3 ? 1.0 / 840.0 : 1.0 / 360.0;  // alpha! c4
  const int i = a % N, j = b % N;
  const bool ap = a >= N, bp = b >= N;
  double ph[N], sp = 0.0;
  for (int q = 0; q < N; ++q) { ph[q] = A.phi[v[q]]; sp += ph[q]; }
  const double h1 = 1.0 / G.h;
  const double gam = A.gamma * G.vol;
  double val;
  if (!ap && !bp) {
    val = gam * h1 * h1 * c2 * (i == j ? 2.0 : 1.0);                      // :115-122 (u,v); :113 is
                                                                           // in k_assemble_rows
  } else if (ap != bp) {
    // int N_i N_j phi_h = |K| c3 (1+d_ij)(S + phi_i + phi_j)
    val = -gam * h1 * h1 * h1 * c3 * (i == j ? 2.0 : 1.0) * (sp + ph[i] + ph[j]);   // (u,q),(p,v)
  } else {
    double m4 = 0.0;
    for (int k = 0; k < N; ++k)
      for (int l = 0; l < N; ++l) m4 += mult4(i, j, k, l) * ph[k] * ph[l];
    val = gam * h1 * h1 * h1 * h1 * c4 * m4;                               // (p,q)
  }
  const int32_t row = ap ? A.dp[v[i]] : A.du[v[i]];
  const int32_t col = bp ? A.nv + v[j] : v[j];
  slot_add(A.slots, row, col, val);
  if (b == 0) {
    double ud[N], sud = 0.0;
    for (int q = 0; q < N; ++q) { ud[q] = A.ud[v[q]]; sud += ud[q]; }
    double r;
    if (!ap) {
      r = gam * h1 * h1 * c2 * (sud + ud[i]);                              // :147 (v part)
    } else {
      double bq = 0.0;
      for (int q = 0; q < N; ++q) bq += ud[q] * (i == q ? 2.0 : 1.0) * (sp + ph[i] + ph[q]);
      r = -gam * h1 * h1 * h1 * c3 * bq;                                   // :147 (q part)
    }
    unsafeAtomicAdd(&A.rhs[row], r);
  }
  // main.py:123-128,150: div(grad(.)) of a P1 function is identically zero.
}

// --- cut-cell penalisation as ROW GATHERS on a Kuhn box (the terms of k_assemble_cut, main.py:115-122,144-149):
// one thread per vertex of a cut cell walks the 24 (6) simplices of its star in closed form (as
// k_assemble_rows_box does), keeps the rows u_v and p_v of every cut one in registers -- 3 x 27 accumulators: (u,u),
// (u,p) = (p,u), (p,p) per lattice neighbour -- and adds them to the two rows once, without atomics (the kernel
// is the only writer of its rows while it runs).  The scatter form issued 64 hashed atomics per cut cell
// (7e7 at 256^3, 2.1 ms); this one evaluates every cut cell once per vertex (4x the arithmetic, which is cheap).
template <int D>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2)))   // 81 accumulators: registers over occupancy
k_assemble_cut_rows_box(int64_t np, const int64_t *__restrict__ full_p, BoxDims bd, AsmArgs A) {
  constexpr int N = D + 1, NPERM = D == 3 ? 6 : 2, NCODE = D == 3 ? 27 : 9;
  constexpr int P[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
  constexpr int P2[2][3] = {{0, 1, 0}, {1, 0, 0}};
  constexpr int POW3[3] = {1, 3, 9};
  constexpr double c2 = D == 3 ? 1.0 / 20.0 : 1.0 / 12.0;
  constexpr double c3 = D == 3 ? 1.0 / 120.0 : 1.0 / 60.0;
  constexpr double c4 = D == 3 ? 1.0 / 840.0 : 1.0 / 360.0;
  // p lives on the vertices of the cut cells and its DoFs are numbered contiguously behind the u DoFs: thread k
  // takes the vertex of the k-th p DoF, so every lane of a wavefront has work (a launch over ALL vertices left
  // 3-6 live lanes per wave and a chain of 24 dependent tag loads: 3 ms)
  const int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (k >= np) return;
  const int64_t vtx = full_p[k] - A.nv;
  const int32_t rowp = A.dp[vtx], rowu = A.du[vtx];
  const int64_t n0 = bd.n[0] + 1, n1 = bd.n[1] + 1;
  const int64_t idx[3] = {vtx % n0, D == 3 ? (vtx / n0) % n1 : vtx / n0, D == 3 ? vtx / (n0 * n1) : 0};
  const int64_t vstride[3] = {1, n0, n0 * n1};
  const int64_t cstride[3] = {1, bd.n[0], bd.n[0] * bd.n[1]};
  double auu[NCODE], aup[NCODE], app[NCODE];
  uint32_t seen = 0u;
#pragma unroll
  for (int c = 0; c < NCODE; ++c) { auu[c] = aup[c] = app[c] = 0.0; }
  double ru = 0.0, rp = 0.0;
  // tags of the whole star first: 24 independent loads in flight instead of 24 load -> branch round trips
  int8_t stag[NPERM][N];
#pragma unroll
  for (int t = 0; t < NPERM; ++t) {
#pragma unroll
    for (int m = 0; m < N; ++m) {
      int dd[3] = {0, 0, 0};
#pragma unroll
      for (int q = 0; q < m; ++q) dd[D == 3 ? P[t][q] : P2[t][q]] -= 1;
      bool in = true;
      int64_t cube = 0;
#pragma unroll
      for (int a = 0; a < D; ++a) {
        const int64_t o = idx[a] + dd[a];
        in = in && o >= 0 && o < bd.n[a];
        cube += o * cstride[a];
      }
      stag[t][m] = in ? (int8_t)(A.ctags[cube * NPERM + t] & PHX_TAG_MASK) : (int8_t)0;
    }
  }
  // every Kuhn simplex of the box has the volume h_x h_y (h_z) / D! and contains the main diagonal of its cube
  // (ufl.CellDiameter, main.py:108): the three coefficients are constants of the mesh
  double vol = bd.h[0] * bd.h[1] * (D == 3 ? bd.h[2] * (1.0 / 6.0) : 0.5), hd2 = 0.0;
  for (int a = 0; a < D; ++a) hd2 += bd.h[a] * bd.h[a];
  const double h1 = 1.0 / sqrt(hd2), gam = A.gamma * vol;
  const double kuu = gam * h1 * h1 * c2, kup = -gam * h1 * h1 * h1 * c3, kpp = gam * h1 * h1 * h1 * h1 * c4;
#pragma unroll
  for (int t = 0; t < NPERM; ++t) {
#pragma unroll
    for (int m = 0; m < N; ++m) {
      int dd[3] = {0, 0, 0};
#pragma unroll
      for (int q = 0; q < m; ++q) dd[D == 3 ? P[t][q] : P2[t][q]] -= 1;
      if (stag[t][m] != 2) continue;
      int code[N];
      double ph[N], ud[N], sp = 0.0, sp2 = 0.0, sud = 0.0;
#pragma unroll
      for (int q = 0; q < N; ++q) {
        code[q] = 0;
        int64_t w = vtx;
#pragma unroll
        for (int a = 0; a < D; ++a) { code[q] += (dd[a] + 1) * POW3[a]; w += dd[a] * vstride[a]; }
        ph[q] = A.phi[w]; ud[q] = A.ud[w];
        sp += ph[q]; sp2 += ph[q] * ph[q]; sud += ud[q];
        if (q < D) dd[D == 3 ? P[t][q] : P2[t][q]] += 1;
      }
      double bq = 0.0;
#pragma unroll
      for (int j = 0; j < N; ++j) {
        const double d2 = j == m ? 2.0 : 1.0;
        auu[code[j]] += kuu * d2;
        aup[code[j]] += kup * d2 * (sp + ph[m] + ph[j]);
        // sum_kl alpha! phi_k phi_l = c! [T^2 + sum_k (c_k + 1) phi_k^2], c = e_i + e_j, T = sum_k (c_k + 1) phi_k
        const double T = sp + ph[m] + ph[j];
        const double m4 = d2 * (T * T + sp2 + ph[m] * ph[m] + ph[j] * ph[j]);
        app[code[j]] += kpp * m4;
        seen |= 1u << code[j];
        bq += ud[j] * d2 * (sp + ph[m] + ph[j]);
      }
      ru += kuu * (sud + ud[m]);
      rp += kup * bq;
    }
  }
  const int32_t nvi = A.nv;
#pragma unroll
  for (int code = 0; code < NCODE; ++code) {
    if (!((seen >> code) & 1u)) continue;
    int64_t w = vtx;
    w += (code % 3 - 1) * vstride[0] + ((code / 3) % 3 - 1) * vstride[1];
    if (D == 3) w += (code / 9 - 1) * vstride[2];
    slot_add_owned(A.slots, rowu, (int32_t)w, auu[code]);
    slot_add_owned(A.slots, rowu, nvi + (int32_t)w, aup[code]);
    slot_add_owned(A.slots, rowp, (int32_t)w, aup[code]);
    slot_add_owned(A.slots, rowp, nvi + (int32_t)w, app[code]);
  }
  A.rhs[rowu] += ru;
  A.rhs[rowp] += rp;
}

// --- one-sided boundary term, main.py:114:  -int_F (grad u . n) v  over (cell, local facet) ---
// With n = -g_lf/|g_lf|, |F| = D |K| |g_lf| and int_F N_i = |F|/D (i on F) the entry is
// |K| (g_j . g_lf) for every row i != lf and every column j.  16 lanes per entity.
template <int D>
__global__ void k_assemble_ds(int64_t nent, const int64_t *__restrict__ ent_packed,
                              const int32_t *__restrict__ ent_pairs, AsmArgs A) {
  const int64_t gid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t e = gid >> 4;
  if (e >= nent) return;
  constexpr int N = D + 1;
  const int i = (int)(gid & 15) >> 2, j = (int)(gid & 3);
  if (i >= N || j >= N) return;
  int64_t c;
  int lf;
  if (ent_packed) { c = ent_packed[2 * e + 1] >> 8; lf = (int)(ent_packed[2 * e + 1] & 0xff); }
  else { c = ent_pairs[2 * e]; lf = ent_pairs[2 * e + 1]; }
  if (i == lf) return;
  int32_t v[N];
  double X[N][D];
  load_cell<D>(A.cells, A.x, c, v, X);
  Geo<D> G;
  simplex_geometry<D>(X, G);
  double k = 0.0;
  for (int d = 0; d < D; ++d) k += G.g[j][d] * G.g[lf][d];
  slot_add(A.slots, A.du[v[i]], v[j], k * G.vol);
}

// --- ghost penalty, main.py:129-134:  sigma avg(h) int_F [grad u . n][grad v . n] on dS((2,3)) ---
// The two cells share the D vertices of F, so the macro-element has D+2 distinct vertices: the
// jump coefficient of a shared vertex is the sum of its two one-sided ones: 25 (16) atomics per facet,
// not 64 (36).
// One lane per facet: the geometry of the two cells is evaluated once per facet, then the lane walks the
// (D+2)^2 entries.  Measured at 256^3 (2e6 facets): 32 lanes per facet (one entry each, every load issued
// once per TWO facets of a wavefront) 2.8 ms, 8 lanes per facet (one tensor row each) 1.8 ms, this 1.5 ms.
// macro-element of interior facet f: its D + 2 distinct vertices vd (the D + 1 of the first cell, then the vertex of
// the second cell opposite f), their jump coefficients Jd (a shared vertex carries the sum of its two one-sided
// normal derivatives) and the weight sigma avg(h) |F|: entry (a, b) of the facet tensor is w Jd[a] Jd[b]
template <int D>
__device__ __forceinline__ void facet_macro(const AsmArgs &A, int64_t f, int32_t *vd, double *Jd, double *w_out) {
  constexpr int N = D + 1;
  int32_t vp[N], vm[N];
  double Jp[N], Jm[N], hsum = 0.0, area = 0.0;
  int lfm = 0;
#pragma unroll
  for (int side = 0; side < 2; ++side) {
    const int64_t c = A.f2c[2 * f + side];
    int32_t *v = side == 0 ? vp : vm;
    double *J = side == 0 ? Jp : Jm;
    double X[N][D];
    load_cell<D>(A.cells, A.x, c, v, X);
    Geo<D> G;
    simplex_geometry<D>(X, G);
    int lf = 0;
#pragma unroll
    for (int k = 0; k < N; ++k)
      if (A.c2f[c * N + k] == (int32_t)f) lf = k;
    if (side == 1) lfm = lf;
    double gl[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
      gl[d] = 0.0;
#pragma unroll
      for (int k = 0; k < N; ++k) gl[d] = k == lf ? G.g[k][d] : gl[d];  // G.g[lf] without dynamic indexing
    }
    double gn = 0.0;
    for (int d = 0; d < D; ++d) gn += gl[d] * gl[d];
    gn = sqrt(gn);
    if (side == 0) area = D * G.vol * gn;
    hsum += G.h;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      double sj = 0.0;
      for (int d = 0; d < D; ++d) sj += G.g[j][d] * gl[d];
      J[j] = -sj / gn;
    }
  }
#pragma unroll
  for (int j = 0; j < N; ++j) { vd[j] = vp[j]; Jd[j] = Jp[j]; }
  vd[N] = 0;
  Jd[N] = 0.0;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    if (j == lfm) { vd[N] = vm[j]; Jd[N] = Jm[j]; continue; }
#pragma unroll
    for (int q = 0; q < N; ++q)
      if (vp[q] == vm[j]) Jd[q] += Jm[j];
  }
  *w_out = A.sigma * 0.5 * hsum * area;
}

template <int D>
__global__ void __launch_bounds__(256) k_assemble_facets(int64_t nlist, const int32_t *__restrict__ list, AsmArgs A) {
  const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= nlist) return;
  constexpr int M = D + 2;
  int32_t vd[M];
  double Jd[M], w;
  facet_macro<D>(A, list[e], vd, Jd, &w);
  int32_t rows[M];
#pragma unroll
  for (int a = 0; a < M; ++a) rows[a] = A.du[vd[a]];
#pragma unroll
  for (int a = 0; a < M; ++a)
#pragma unroll
    for (int b = 0; b < M; ++b) slot_add(A.slots, rows[a], vd[b], w * Jd[a] * Jd[b]);
}

// deterministic accumulation: value = hi + lo
__global__ void k_det_finalize(int64_t n, double *__restrict__ acc, const double *__restrict__ lo) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) acc[i] += lo[i];
}
// second accumulators of PHX_OPT_DETERMINISTIC for `nslots` slots and `nrows` right-hand-side entries; false when the
// extra 12 bytes per slot do not fit the budget (then the assembly keeps plain atomics)
static int det_alloc(phx_mesh *m, Slots &sl, int64_t nslots, int64_t nrows, bool *on) {
  *on = false;
  if (!m->deterministic) return PHX_OK;
  // default: a fifth of the device (57.6 GB of 288: the 256^3 elasticity system needs 52 GB, the 512^3 P2 system more)
  static const double limit_gb = [] {
    if (const char *e = getenv("PHX_DET_LIMIT_GB")) return atof(e);
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess || tot == 0) { (void)hipGetLastError(); return 48.0; }
    return 0.2 * (double)tot / 1073741824.0;
  }();
  if (12.0 * (double)nslots > limit_gb * 1073741824.0) return PHX_OK;
  PHX_HIP(phx_malloc(&sl.emax, sizeof(int32_t) * (size_t)nslots));
  PHX_HIP(phx_malloc(&sl.lo, sizeof(double) * (size_t)nslots));
  PHX_HIP(phx_malloc(&sl.remax, sizeof(int32_t) * (size_t)nrows));
  PHX_HIP(phx_malloc(&sl.rlo, sizeof(double) * (size_t)nrows));
  PHX_HIP(hipMemsetAsync(sl.emax, 0, sizeof(int32_t) * (size_t)nslots, m->stream));
  PHX_HIP(hipMemsetAsync(sl.lo, 0, sizeof(double) * (size_t)nslots, m->stream));
  PHX_HIP(hipMemsetAsync(sl.remax, 0, sizeof(int32_t) * (size_t)nrows, m->stream));
  PHX_HIP(hipMemsetAsync(sl.rlo, 0, sizeof(double) * (size_t)nrows, m->stream));
  *on = true;
  return PHX_OK;
}
static int det_finish(phx_mesh *m, Slots &sl, int64_t nslots, int64_t nrows, double *rhs) {
  if (!sl.emax) return PHX_OK;
  k_det_finalize<<<dim3((unsigned)phx_div_up(nslots, 256)), dim3(256), 0, m->stream>>>(nslots, sl.vals, sl.lo);
  k_det_finalize<<<dim3((unsigned)phx_div_up(nrows, 256)), dim3(256), 0, m->stream>>>(nrows, rhs, sl.rlo);
  PHX_HIP(hipGetLastError());
  PHX_HIP(hipStreamSynchronize(m->stream));
  PHX_HIP(phx_free(sl.emax)); PHX_HIP(phx_free(sl.lo)); PHX_HIP(phx_free(sl.remax)); PHX_HIP(phx_free(sl.rlo));
  sl.emax = nullptr; sl.lo = nullptr; sl.remax = nullptr; sl.rlo = nullptr;
  return PHX_OK;
}

template <typename Pred>
static int build_list(phx_mesh *m, int64_t n, Pred pred, int32_t **list, int64_t *count,
                      std::vector<void *> *later = nullptr, const int32_t *known_counts = nullptr,
                      int64_t known_total = -1) {
  return phx_select_indices(m->stream, n, pred, list, count, later, known_counts, known_total);
}

// ---------------------------------------------------------------------------------------------
// compaction: one wave per row; lanes hold the slots, sorted by column with a bitonic network
// ---------------------------------------------------------------------------------------------
__global__ void k_row_counts(int64_t n, int W, const int32_t *__restrict__ cols,
                             const uint8_t *__restrict__ clean, int64_t *__restrict__ counts) {
  // a wavefront walks many rows: a wave per row (2.9 M of them at 256^3) is launch-bound
  const int lane = threadIdx.x & 63;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t row = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6; row < n; row += nwaves) {
    if (clean && clean[row]) {
      if (lane == 0) counts[row] = clean[row];
      continue;
    }
    int cnt = 0;
    for (int k = lane; k < W; k += 64) cnt += cols[row * W + k] != -1;
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if (lane == 0) counts[row] = cnt;
  }
}

__global__ void k_row_fill(int64_t n, int W, const int32_t *__restrict__ cols,
                           const double *__restrict__ vals, const uint8_t *__restrict__ clean,
                           const int64_t *__restrict__ rowptr,
                           int32_t nv, const int32_t *__restrict__ du,
                           const int32_t *__restrict__ dp, int32_t *__restrict__ ocol,
                           double *__restrict__ oval, double *__restrict__ diag,
                           int32_t *__restrict__ row_nz) {
  const int lane = threadIdx.x & 63;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t row = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6; row < n; row += nwaves) {
    int32_t c = 0x7fffffff;
    double v = 0.0;
    const int nclean = clean ? clean[row] : 0;  // dense, sorted row: translate and copy
    if (lane < (nclean ? nclean : W)) {
      const int32_t cc = cols[row * W + lane];
      if (cc != -1) { c = cc < nv ? du[cc] : dp[cc - nv]; v = vals[row * W + lane]; }
    }
    if (c == (int32_t)row) diag[row] = v;
    if (!nclean)
    for (int k = 2; k <= 64; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        const int32_t oc = __shfl_xor(c, j);
        const double ov = __shfl_xor(v, j);
        const bool up = ((lane & k) == 0);
        const bool lower = ((lane & j) == 0);
        const bool take = (lower == up) ? (oc < c) : (oc > c);
        if (take) { c = oc; v = ov; }
      }
    const int64_t base = rowptr[row];
    const int64_t cnt = rowptr[row + 1] - base;
    if (lane < cnt) { ocol[base + lane] = c; oval[base + lane] = v; }
    // what the SELL copy will keep of this row (explicit zeros dropped, the diagonal always kept)
    const unsigned long long keep = __ballot(lane < cnt && (v != 0.0 || c == (int32_t)row));
    if (lane == 0) row_nz[row] = __popcll(keep);
  }
}

// rows wider than one wavefront (P2): one block of W threads per row, bitonic sort in LDS
template <int W>
__global__ void __launch_bounds__(W)
k_row_fill_block(int64_t n, const int32_t *__restrict__ cols, const double *__restrict__ vals,
                 const int64_t *__restrict__ rowptr, int32_t nent, const int32_t *__restrict__ du,
                 const int32_t *__restrict__ dp, int32_t *__restrict__ ocol,
                 double *__restrict__ oval, double *__restrict__ diag) {
  __shared__ int32_t sc[W];
  __shared__ double sv[W];
  const int t = threadIdx.x;
  // a block walks rows with a grid stride: a HIP grid holds fewer than 2^32 threads, and 2.3e7 rows of a
  // 256^3 P2 system times 256 threads is more (the launch then covers only part of the rows -- silently)
  for (int64_t row = blockIdx.x; row < n; row += gridDim.x) {
    int32_t c = 0x7fffffff;
    double v = 0.0;
    const int32_t cc = cols[row * W + t];
    if (cc != -1) { c = cc < nent ? du[cc] : dp[cc - nent]; v = vals[row * W + t]; }
    if (c == (int32_t)row) diag[row] = v;
    sc[t] = c;
    sv[t] = v;
    __syncthreads();
    for (int k = 2; k <= W; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        const int p = t ^ j;
        if (p > t) {
          const bool up = ((t & k) == 0);
          const int32_t a = sc[t], b = sc[p];
          if ((a > b) == up) {
            sc[t] = b; sc[p] = a;
            const double x = sv[t]; sv[t] = sv[p]; sv[p] = x;
          }
        }
        __syncthreads();
      }
    const int64_t base = rowptr[row];
    const int64_t cnt = rowptr[row + 1] - base;
    if (t < cnt) { ocol[base + t] = sc[t]; oval[base + t] = sv[t]; }
    __syncthreads();
  }
}

// ---- compaction of slot tables with per-row capacities (Slots::off / wlog)
__global__ void k_row_counts_var(int64_t n, const int64_t *__restrict__ off, const uint8_t *__restrict__ wlog,
                                 const int32_t *__restrict__ cols, const uint8_t *__restrict__ clean,
                                 int64_t *__restrict__ counts) {
  const int lane = threadIdx.x & 63;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t row = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6; row < n; row += nwaves) {
    if (clean && clean[row]) {
      if (lane == 0) counts[row] = clean[row];
      continue;
    }
    const int W = 1 << wlog[row];
    const int32_t *rc = cols + off[row];
    int cnt = 0;
    for (int k = lane; k < W; k += 64) cnt += rc[k] != -1;
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if (lane == 0) counts[row] = cnt;
  }
}

// rows of up to 64 slots: one wave per row (as k_row_fill); wider rows are left to k_row_fill_list
__global__ void k_row_fill_var(int64_t n, const int64_t *__restrict__ off, const uint8_t *__restrict__ wlog,
                               const int32_t *__restrict__ cols, const double *__restrict__ vals,
                               const uint8_t *__restrict__ clean,
                               const int64_t *__restrict__ rowptr, int32_t nent, const int32_t *__restrict__ du,
                               const int32_t *__restrict__ dp, int32_t *__restrict__ ocol,
                               double *__restrict__ oval, double *__restrict__ diag, int32_t *__restrict__ row_nz) {
  const int64_t row = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (row >= n || wlog[row] > 6) return;
  const int W = 1 << wlog[row];
  const int64_t sb = off[row];
  const int nclean = clean ? clean[row] : 0;  // dense, sorted row: translate and copy
  int32_t c = 0x7fffffff;
  double v = 0.0;
  if (lane < (nclean ? nclean : W)) {
    const int32_t cc = cols[sb + lane];
    if (cc != -1) { c = cc < nent ? du[cc] : dp[cc - nent]; v = vals[sb + lane]; }
  }
  if (c == (int32_t)row) diag[row] = v;
  if (!nclean)
  for (int k = 2; k <= 64; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      const int32_t oc = __shfl_xor(c, j);
      const double ov = __shfl_xor(v, j);
      const bool up = ((lane & k) == 0);
      const bool lower = ((lane & j) == 0);
      const bool take = (lower == up) ? (oc < c) : (oc > c);
      if (take) { c = oc; v = ov; }
    }
  const int64_t base = rowptr[row];
  const int64_t cnt = rowptr[row + 1] - base;
  if (lane < cnt) { ocol[base + lane] = c; oval[base + lane] = v; }
  const unsigned long long keep = __ballot(lane < cnt && (v != 0.0 || c == (int32_t)row));
  if (lane == 0) row_nz[row] = __popcll(keep);
}

struct SelWideRow { const uint8_t *wlog; __host__ __device__ bool operator()(const int32_t &r) const { return wlog[r] > 6; } };

// wide rows (capacity W > 64) from a list: one block of W threads per row, bitonic sort in LDS
template <int W>
__global__ void __launch_bounds__(W)
k_row_fill_list(int64_t nlist, const int32_t *__restrict__ list, const int64_t *__restrict__ off,
                const int32_t *__restrict__ cols, const double *__restrict__ vals,
                const int64_t *__restrict__ rowptr, int32_t nent, const int32_t *__restrict__ du,
                const int32_t *__restrict__ dp, int32_t *__restrict__ ocol, double *__restrict__ oval,
                double *__restrict__ diag, int32_t *__restrict__ row_nz) {
  __shared__ int32_t sc[W];
  __shared__ double sv[W];
  __shared__ int nz;
  const int64_t row = list[blockIdx.x];
  const int t = threadIdx.x;
  if (t == 0) nz = 0;
  int32_t c = 0x7fffffff;
  double v = 0.0;
  const int64_t sb = off[row];
  const int32_t cc = cols[sb + t];
  if (cc != -1) { c = cc < nent ? du[cc] : dp[cc - nent]; v = vals[sb + t]; }
  if (c == (int32_t)row) diag[row] = v;
  sc[t] = c;
  sv[t] = v;
  __syncthreads();
  if (c != 0x7fffffff && (v != 0.0 || c == (int32_t)row)) atomicAdd(&nz, 1);
  for (int k = 2; k <= W; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      const int p = t ^ j;
      if (p > t) {
        const bool up = ((t & k) == 0);
        const int32_t a = sc[t], b = sc[p];
        if ((a > b) == up) {
          sc[t] = b; sc[p] = a;
          const double x = sv[t]; sv[t] = sv[p]; sv[p] = x;
        }
      }
      __syncthreads();
    }
  const int64_t base = rowptr[row];
  const int64_t cnt = rowptr[row + 1] - base;
  if (t < cnt) { ocol[base + t] = sc[t]; oval[base + t] = sv[t]; }
  if (t == 0) row_nz[row] = nz;
}

template <typename T>
static int exclusive_sum(phx_mesh *m, const T *in, T *out, int64_t n) {
  size_t bytes = 0;
  PHX_HIP(phx_exclusive_sum(nullptr, bytes, in, out, (size_t)(n), m->stream));
  void *tmp = nullptr;
  PHX_HIP(phx_malloc(&tmp, bytes ? bytes : 16));
  PHX_HIP(phx_exclusive_sum(tmp, bytes, in, out, (size_t)(n), m->stream));
  PHX_HIP(hipStreamSynchronize(m->stream));
  PHX_HIP(phx_free(tmp));
  return PHX_OK;
}

struct U8ToI32 {
  __host__ __device__ int32_t operator()(const uint8_t &a) const { return (int32_t)a; }
};

static int scan_flags(phx_mesh *m, const uint8_t *flags, int32_t *out, int64_t n, int32_t *total) {
  auto it = rocprim::make_transform_iterator(flags, U8ToI32());
  size_t bytes = 0;
  PHX_HIP(phx_exclusive_sum(nullptr, bytes, it, out, (size_t)(n), m->stream));
  void *tmp = nullptr;
  PHX_HIP(phx_malloc(&tmp, bytes ? bytes : 16));
  PHX_HIP(phx_exclusive_sum(tmp, bytes, it, out, (size_t)(n), m->stream));
  int32_t last = 0;
  uint8_t lastf = 0;
  const phx_rb_item rb[2] = {{out + (n - 1), 4, &last}, {flags + (n - 1), 1, &lastf}};
  PHX_CHECK(phx_read_back(m->stream, rb, 2));
  PHX_HIP(phx_free(tmp));
  *total = last + (int32_t)lastf;
  return PHX_OK;
}

// two flag arrays of the same entities as ONE scan of 64-bit words (low word: a, high word: b; the totals stay below
// 2^31), both totals with one host round trip
// a_or_b: the first flag is fa | fb (u DoFs: vertices of inside OR cut cells, from the flag arrays of the tagging pass)
struct PackFlags2 {
  const uint8_t *fa, *fb;
  int a_or_b;
  __host__ __device__ unsigned long long operator()(const int64_t &i) const {
    const unsigned long long b = fb[i] ? 1ull : 0ull, a = (fa[i] || (a_or_b && b)) ? 1ull : 0ull;
    return a | (b << 32);
  }
};
static int scan_flags_packed(phx_mesh *m, const uint8_t *fa, const uint8_t *fb, unsigned long long *out, int32_t *ta,
                             int32_t *tb, int64_t n, int a_or_b = 0) {
  auto it = rocprim::make_transform_iterator(rocprim::make_counting_iterator<int64_t>(0), PackFlags2{fa, fb, a_or_b});
  size_t bytes = 0;
  PHX_HIP(phx_exclusive_sum(nullptr, bytes, it, out, (size_t)(n), m->stream));
  void *tmp = nullptr;
  PHX_HIP(phx_malloc(&tmp, bytes ? bytes : 16));
  PHX_HIP(phx_exclusive_sum(tmp, bytes, it, out, (size_t)(n), m->stream));
  unsigned long long last = 0;
  uint8_t la = 0, lb = 0;
  const phx_rb_item rb[3] = {{out + (n - 1), 8, &last}, {fa + (n - 1), 1, &la}, {fb + (n - 1), 1, &lb}};
  PHX_CHECK(phx_read_back(m->stream, rb, 3));
  PHX_HIP(phx_free(tmp));
  *ta = (int32_t)(last & 0xffffffffull) + ((la || (a_or_b && lb)) ? 1 : 0);
  *tb = (int32_t)(last >> 32) + (lb ? 1 : 0);
  return PHX_OK;
}

// two flag arrays scanned behind each other, both totals with ONE host round trip
static int scan_flags2(phx_mesh *m, const uint8_t *fa, int32_t *oa, int32_t *ta, const uint8_t *fb, int32_t *ob,
                       int32_t *tb, int64_t n) {
  auto ia = rocprim::make_transform_iterator(fa, U8ToI32());
  auto ib = rocprim::make_transform_iterator(fb, U8ToI32());
  size_t bytes = 0;
  PHX_HIP(phx_exclusive_sum(nullptr, bytes, ia, oa, (size_t)(n), m->stream));
  void *tmp = nullptr;
  PHX_HIP(phx_malloc(&tmp, bytes ? bytes : 16));
  PHX_HIP(phx_exclusive_sum(tmp, bytes, ia, oa, (size_t)(n), m->stream));
  PHX_HIP(phx_exclusive_sum(tmp, bytes, ib, ob, (size_t)(n), m->stream));
  int32_t last[2] = {0, 0};
  uint8_t lastf[2] = {0, 0};
  const phx_rb_item rb[4] = {{oa + (n - 1), 4, &last[0]}, {fa + (n - 1), 1, &lastf[0]}, {ob + (n - 1), 4, &last[1]},
                             {fb + (n - 1), 1, &lastf[1]}};
  PHX_CHECK(phx_read_back(m->stream, rb, 4));
  PHX_HIP(phx_free(tmp));
  *ta = last[0] + (int32_t)lastf[0];
  *tb = last[1] + (int32_t)lastf[1];
  return PHX_OK;
}

extern "C" int phx_system_destroy(phx_system *s) {
  if (!s) return PHX_OK;
  // the mesh handle may already be gone (interpreter shutdown destroys in arbitrary order)
  (void)hipSetDevice(s->device);
  (void)hipDeviceSynchronize();
  void *ptrs[] = {s->dof_of_vertex_u, s->dof_of_vertex_p, s->full_of_active, s->rowptr, s->col,
                  s->val, s->rhs, s->diag, s->slice_ptr, s->sell_col, s->sell_val,
                  s->sell_val_raw, s->sell_kind, s->perm, s->iperm, s->work, s->scal, s->row_nz,
                  s->c0, s->stencil, s->seg, s->slice_seg, s->sell_rows, s->cscale, s->pvec, s->bnd, s->bnd_rec, s->dpart, s->st_map};
  for (void *p : ptrs) (void)phx_free(p);
  if (s->p2s) {
    (void)phx_free(s->p2s->coef); (void)phx_free(s->p2s->mask); (void)phx_free(s->p2s->runs);
    (void)phx_free(s->p2s->tabE); (void)phx_free(s->p2s->tabO); (void)phx_free(s->p2s->linemask);
    delete s->p2s;
  }
  phx_box_precond_destroy(s->precond);
  phx_blockjac_destroy(s->bj);
  phx_coarse_destroy(s->cc);
  delete s;
  return PHX_OK;
}

static int free_slots(Slots &sl) {
  PHX_HIP(phx_free(sl.cols)); PHX_HIP(phx_free(sl.vals)); PHX_HIP(phx_free(sl.overflow)); PHX_HIP(phx_free(sl.clean));
  PHX_HIP(phx_free((void *)sl.off)); PHX_HIP(phx_free((void *)sl.wlog));
  sl.cols = nullptr; sl.vals = nullptr; sl.overflow = nullptr; sl.clean = nullptr; sl.off = nullptr; sl.wlog = nullptr;
  return PHX_OK;
}

static int check_overflow(phx_mesh *m, Slots &sl) {
  int overflow = 0;
  PHX_HIP(hipMemcpyAsync(&overflow, sl.overflow, sizeof(int), hipMemcpyDeviceToHost, m->stream));
  PHX_HIP(hipStreamSynchronize(m->stream));
  if (overflow) {
    PHX_CHECK(free_slots(sl));
    phx_set_error("row-slot capacity %d exceeded", sl.W);
    return PHX_ERR_CAPACITY;
  }
  return PHX_OK;
}

static int csr_from_slots(phx_system *s, Slots &sl, int32_t nent);

// overflow check -> compaction of the slot tables into CSR (sorted columns) -> SELL
int phx_finish_system(phx_system *s, Slots &sl, int32_t nent) {
  PHX_CHECK(check_overflow(s->mesh, sl));
  PHX_CHECK(csr_from_slots(s, sl, nent));
  PHX_CHECK(free_slots(sl));
  return phx_system_build_sell(s);
}

// Structured systems: the solver formats come straight from the slots (stencil segments + SELL over the stored
// rows); the CSR copy only when PHX_OPT_EXPORT_CSR asks for it (then every row, C0 included, sits in the slots).
static int finish_structured(phx_system *s, Slots &sl, int32_t nent, bool with_csr) {
  PHX_CHECK(check_overflow(s->mesh, sl));
  if (with_csr) PHX_CHECK(csr_from_slots(s, sl, nent));
  const phx_slot_view sv{sl.cols, sl.vals, sl.W, sl.clean, sl.off, sl.wlog};
  const int rc = phx_system_build_structured(s, sv, nent);
  PHX_CHECK(free_slots(sl));
  return rc;
}

static int csr_from_slots(phx_system *s, Slots &sl, int32_t nent) {
  phx_mesh *m = s->mesh;
  const int W = sl.W;
  const dim3 block(256);
  int64_t *counts = nullptr;
  PHX_HIP(phx_malloc(&counts, sizeof(int64_t) * (size_t)(s->n + 1)));
  PHX_HIP(hipMemsetAsync(counts, 0, sizeof(int64_t) * (size_t)(s->n + 1), m->stream));
  PHX_HIP(phx_malloc(&s->rowptr, sizeof(int64_t) * (size_t)(s->n + 1)));
  // 8 waves per SIMD x 4 SIMDs x 256 CUs, each walking rows with a wave stride
  const dim3 growave((unsigned)std::min<int64_t>(phx_div_up(s->n * 64, 256), 2048));
  if (sl.off) k_row_counts_var<<<growave, block, 0, m->stream>>>(s->n, sl.off, sl.wlog, sl.cols, sl.clean, counts);
  else k_row_counts<<<growave, block, 0, m->stream>>>(s->n, W, sl.cols, sl.clean, counts);
  PHX_CHECK(exclusive_sum<int64_t>(m, counts, s->rowptr, s->n + 1));
  int64_t nnz = 0;
  PHX_HIP(hipMemcpy(&nnz, s->rowptr + s->n, sizeof(int64_t), hipMemcpyDeviceToHost));
  s->nnz = nnz;
  PHX_HIP(phx_malloc(&s->col, sizeof(int32_t) * (size_t)nnz));
  PHX_HIP(phx_malloc(&s->val, sizeof(double) * (size_t)nnz));
  if (!s->diag) {
    PHX_HIP(phx_malloc(&s->diag, sizeof(double) * (size_t)s->n));
    PHX_HIP(hipMemsetAsync(s->diag, 0, sizeof(double) * (size_t)s->n, m->stream));
  }
  if (W <= 64 || sl.off) PHX_HIP(phx_malloc(&s->row_nz, sizeof(int32_t) * (size_t)s->n));
  if (sl.off) {
    // per-row capacities: narrow rows one wave each, the wide ones (list) one block of W threads each
    k_row_fill_var<<<dim3((unsigned)phx_div_up(s->n * 64, 256)), block, 0, m->stream>>>(
        s->n, sl.off, sl.wlog, sl.cols, sl.vals, sl.clean, s->rowptr, nent, s->dof_of_vertex_u, s->dof_of_vertex_p,
        s->col, s->val, s->diag, s->row_nz);
    int32_t *wide = nullptr;
    int64_t nwide = 0;
    PHX_CHECK(phx_select_indices(m->stream, s->n, SelWideRow{sl.wlog}, &wide, &nwide));
    if (nwide > 0) {
      const dim3 g((unsigned)nwide);
      if (W == 128) k_row_fill_list<128><<<g, dim3(128), 0, m->stream>>>(nwide, wide, sl.off, sl.cols, sl.vals, s->rowptr, nent, s->dof_of_vertex_u, s->dof_of_vertex_p, s->col, s->val, s->diag, s->row_nz);
      else if (W == 256) k_row_fill_list<256><<<g, dim3(256), 0, m->stream>>>(nwide, wide, sl.off, sl.cols, sl.vals, s->rowptr, nent, s->dof_of_vertex_u, s->dof_of_vertex_p, s->col, s->val, s->diag, s->row_nz);
      else if (W == 512) k_row_fill_list<512><<<g, dim3(512), 0, m->stream>>>(nwide, wide, sl.off, sl.cols, sl.vals, s->rowptr, nent, s->dof_of_vertex_u, s->dof_of_vertex_p, s->col, s->val, s->diag, s->row_nz);
      else if (W == 1024) k_row_fill_list<1024><<<g, dim3(1024), 0, m->stream>>>(nwide, wide, sl.off, sl.cols, sl.vals, s->rowptr, nent, s->dof_of_vertex_u, s->dof_of_vertex_p, s->col, s->val, s->diag, s->row_nz);
      else { phx_set_error("unsupported wide-row capacity %d", W); return PHX_ERR_VALUE; }
    }
    PHX_HIP(hipGetLastError());
    PHX_HIP(hipStreamSynchronize(m->stream));
    PHX_HIP(phx_free(wide));
  } else if (W <= 64 && s->n * 64 >= ((int64_t)1 << 32)) {
    phx_set_error("row compaction: %lld rows exceed one HIP grid -- partition the problem", (long long)s->n);
    return PHX_ERR_VALUE;
  } else if (W <= 64)  // one wave per row here: the wave-stride variant was slower (2.23 vs 1.89 ms), the sort hides nothing
    k_row_fill<<<dim3((unsigned)phx_div_up(s->n * 64, 256)), block, 0, m->stream>>>(s->n, W, sl.cols, sl.vals, sl.clean, s->rowptr, nent,
                                                  s->dof_of_vertex_u, s->dof_of_vertex_p, s->col, s->val, s->diag, s->row_nz);
  else if (W == 128)
    k_row_fill_block<128><<<dim3((unsigned)std::min<int64_t>(s->n, 1 << 21)), dim3(128), 0, m->stream>>>(
        s->n, sl.cols, sl.vals, s->rowptr, nent, s->dof_of_vertex_u, s->dof_of_vertex_p, s->col, s->val, s->diag);
  else if (W == 256)
    k_row_fill_block<256><<<dim3((unsigned)std::min<int64_t>(s->n, 1 << 21)), dim3(256), 0, m->stream>>>(
        s->n, sl.cols, sl.vals, s->rowptr, nent, s->dof_of_vertex_u, s->dof_of_vertex_p, s->col, s->val, s->diag);
  else if (W == 512)
    k_row_fill_block<512><<<dim3((unsigned)std::min<int64_t>(s->n, 1 << 21)), dim3(512), 0, m->stream>>>(
        s->n, sl.cols, sl.vals, s->rowptr, nent, s->dof_of_vertex_u, s->dof_of_vertex_p, s->col, s->val, s->diag);
  else if (W == 1024)
    k_row_fill_block<1024><<<dim3((unsigned)std::min<int64_t>(s->n, 1 << 21)), dim3(1024), 0, m->stream>>>(
        s->n, sl.cols, sl.vals, s->rowptr, nent, s->dof_of_vertex_u, s->dof_of_vertex_p, s->col, s->val, s->diag);
  else {
    phx_set_error("unsupported slot capacity %d", W);
    return PHX_ERR_VALUE;
  }
  PHX_HIP(hipGetLastError());
  PHX_HIP(hipStreamSynchronize(m->stream));
  PHX_HIP(phx_free(counts));
  return PHX_OK;
}

static int assemble_with_capacity(phx_mesh *m, double pen_coef, double stab_coef,
                                  const double *dphi, const double *df, const double *dud, int W,
                                  phx_system **out) {
  phx_system *s = new phx_system();
  s->mesh = m;
  s->device = m->device;
  s->nent = m->nv;
  s->nfull = 2 * m->nv;
  s->slot_cap = W;
  s->u_vertex_block = true;
  const int D = m->gdim;
  const dim3 block(256);
  // ---- active numbering
  uint8_t *fu = nullptr, *fp = nullptr;
  unsigned long long *sup = nullptr;
  PHX_HIP(phx_malloc(&sup, sizeof(unsigned long long) * (size_t)m->nv));
  // the single-layer tagging pass left the vertex flags behind (inside cells; cells that stayed cut): no walk over the cells
  static const bool no_act = getenv("PHX_NO_ACT_FLAGS") != nullptr;   // A/B aid
  const int from_tags = (m->act_valid && m->act_in && m->act_cut && !no_act) ? 1 : 0;
  PHX_HIP(phx_malloc(&fu, (size_t)m->nv));
  PHX_HIP(phx_malloc(&fp, (size_t)m->nv));
  if (from_tags) {
    k_flags_from_tagging<<<dim3((unsigned)phx_div_up(m->nv, 256)), block, 0, m->stream>>>(m->nv, m->act_in, m->act_cut, fu, fp);
  } else {
    PHX_HIP(hipMemsetAsync(fu, 0, (size_t)m->nv, m->stream));
    PHX_HIP(hipMemsetAsync(fp, 0, (size_t)m->nv, m->stream));
    const dim3 gcells((unsigned)phx_div_up(phx_div_up(m->nc, 4), 256));
    if (D == 2) k_mark_active<3><<<gcells, block, 0, m->stream>>>(m->nc, m->cells, m->cell_tags, fu, fp);
    else k_mark_active<4><<<gcells, block, 0, m->stream>>>(m->nc, m->cells, m->cell_tags, fu, fp);
  }
  int32_t nu = 0, np = 0;
  PHX_CHECK(scan_flags_packed(m, fu, fp, sup, &nu, &np, m->nv));
  s->nu = nu;
  s->n = (int64_t)nu + np;
  if (s->n == 0) {
    PHX_HIP(hipStreamSynchronize(m->stream));
    PHX_HIP(phx_free(fu)); PHX_HIP(phx_free(fp)); PHX_HIP(phx_free(sup));
    if (!m->allow_empty) {
      delete s;
      phx_set_error("no active DoF: no cell is tagged 1 or 2");
      return PHX_ERR_VALUE;
    }
    // PHX_OPT_ALLOW_EMPTY (slab drivers): a system without rows that still takes part in the collectives
    int rc = PHX_OK;
    if (phx_malloc(&s->dof_of_vertex_u, sizeof(int32_t) * (size_t)m->nv) != hipSuccess ||
        phx_malloc(&s->dof_of_vertex_p, sizeof(int32_t) * (size_t)m->nv) != hipSuccess ||
        hipMemsetAsync(s->dof_of_vertex_u, 0xff, sizeof(int32_t) * (size_t)m->nv, m->stream) != hipSuccess ||
        hipMemsetAsync(s->dof_of_vertex_p, 0xff, sizeof(int32_t) * (size_t)m->nv, m->stream) != hipSuccess)
      rc = PHX_ERR_HIP;
    if (rc == PHX_OK) rc = phx_system_build_empty(s);
    if (rc != PHX_OK) { phx_system_destroy(s); return rc; }
    *out = s;
    return PHX_OK;
  }
  PHX_HIP(phx_malloc(&s->dof_of_vertex_u, sizeof(int32_t) * (size_t)m->nv));
  PHX_HIP(phx_malloc(&s->dof_of_vertex_p, sizeof(int32_t) * (size_t)m->nv));
  PHX_HIP(phx_malloc(&s->full_of_active, sizeof(int64_t) * (size_t)s->n));
  k_finish_numbering_packed<<<dim3((unsigned)phx_div_up(m->nv, 256)), block, 0, m->stream>>>(
      m->nv, fu, fp, sup, nu, s->dof_of_vertex_u, s->dof_of_vertex_p, s->full_of_active, 0);
  // temporaries whose last kernel is only enqueued: freed behind the synchronisation in front of finish_*
  std::vector<void *> later = {fu, fp, sup};
  // ---- element kernels run over compacted work lists
  int32_t *l_cut = nullptr, *l_fac = nullptr;
  int64_t n_cut = 0, n_fac = 0;
  PHX_CHECK(build_list(m, m->nc, SelCut{m->cell_tags}, &l_cut, &n_cut, &later,
                       m->sel_cut_valid ? m->sel_counts_cut : nullptr,         // counted by the cell tagging kernel:
                       m->sel_cut_valid ? m->tag_hist[2] : -1));               // the cut cells ARE the cells tagged 2
  PHX_CHECK(build_list(m, m->nf, SelGhostFacet{m->facet_tags, m->f2c}, &l_fac, &n_fac, &later,
                       m->sel_counts_valid ? m->sel_counts[0] : nullptr,     // counted by the facet tagging kernel
                       m->sel_counts_valid ? m->sel_total[0] : -1));
  Slots sl;
  sl.W = W;
  sl.cols = nullptr; sl.vals = nullptr; sl.overflow = nullptr;
  PHX_HIP(phx_malloc(&s->rhs, sizeof(double) * (size_t)s->n));
  PHX_HIP(hipMemsetAsync(s->rhs, 0, sizeof(double) * (size_t)s->n, m->stream));
  AsmArgs A;
  A.cells = m->cells; A.x = m->x; A.ctags = m->cell_tags; A.ftags = m->facet_tags;
  A.c2f = m->c2f; A.f2c = m->f2c; A.du = s->dof_of_vertex_u; A.dp = s->dof_of_vertex_p;
  A.phi = dphi; A.f = df; A.ud = dud; A.gamma = pen_coef; A.sigma = stab_coef;
  A.rhs = s->rhs; A.nv = (int32_t)m->nv;
  uint8_t *touched = nullptr;
  int32_t *stored_rows = nullptr;   // structured systems without a CSR copy: active indices of the rows that are stored
  int64_t n_stored_rows = 0;
  const bool structured = m->is_box && !m->is_submesh && m->structured != 0;
  int64_t slot_rows = s->n;
  if (m->is_box && !m->is_submesh) {
    // rows of vertices no scattering kernel reaches are written dense and sorted by the gather kernel
    PHX_CHECK(phx_collect_entities(m));
    PHX_HIP(phx_malloc(&touched, (size_t)m->nv));
    PHX_HIP(phx_malloc(&sl.clean, (size_t)s->n));
    PHX_HIP(hipMemsetAsync(touched, 0, (size_t)m->nv, m->stream));
    PHX_HIP(hipMemsetAsync(sl.clean, 0, (size_t)s->n, m->stream));
    constexpr int TB = 256;
    if (n_cut > 0) {
      const dim3 g((unsigned)phx_div_up(n_cut, TB));
      if (D == 2) k_mark_cells<3><<<g, dim3(TB), 0, m->stream>>>(n_cut, l_cut, m->cells, touched);
      else k_mark_cells<4><<<g, dim3(TB), 0, m->stream>>>(n_cut, l_cut, m->cells, touched);
    }
    if (n_fac > 0) {
      const dim3 g((unsigned)phx_div_up(n_fac, TB));
      if (D == 2) k_mark_facet_cells<3><<<g, dim3(TB), 0, m->stream>>>(n_fac, l_fac, m->f2c, m->cells, touched);
      else k_mark_facet_cells<4><<<g, dim3(TB), 0, m->stream>>>(n_fac, l_fac, m->f2c, m->cells, touched);
    }
    if (m->ent_count[0] > 0) {
      const dim3 g((unsigned)phx_div_up(m->ent_count[0], TB));
      if (D == 2) k_mark_entity_cells<3><<<g, dim3(TB), 0, m->stream>>>(m->ent_count[0], m->ent_buf[0], m->cells, touched);
      else k_mark_entity_cells<4><<<g, dim3(TB), 0, m->stream>>>(m->ent_count[0], m->ent_buf[0], m->cells, touched);
    }
    A.touched = touched;
  }
  if (structured) {
    // C0 rows (translation-invariant interior rows) are applied from a stencil: no slots, no stored entries
    s->structured = true;
    s->u_unscaled = true;
    PHX_HIP(phx_malloc(&s->c0, (size_t)s->n));
    PHX_HIP(phx_malloc(&s->diag, sizeof(double) * (size_t)s->n));
    PHX_HIP(phx_malloc(&s->stencil, sizeof(double) * 8));
    PHX_HIP(hipMemsetAsync(s->c0, 0, (size_t)s->n, m->stream));
    PHX_HIP(hipMemsetAsync(s->diag, 0, sizeof(double) * (size_t)s->n, m->stream));
    PHX_HIP(hipMemsetAsync(s->stencil, 0, sizeof(double) * 8, m->stream));
    const BoxDims bd{{m->box_n[0], m->box_n[1], m->box_n[2]}, {m->box_h[0], m->box_h[1], m->box_h[2]}};
    const dim3 g((unsigned)phx_div_up(m->nv, 256));
    if (D == 2) k_mark_c0<2><<<g, block, 0, m->stream>>>(m->nv, bd, s->dof_of_vertex_u, m->cell_tags, touched, s->c0);
    else k_mark_c0<3><<<g, block, 0, m->stream>>>(m->nv, bd, s->dof_of_vertex_u, m->cell_tags, touched, s->c0);
    if (D == 2) k_box_stencil<2><<<1, 1, 0, m->stream>>>(bd, s->stencil);
    else k_box_stencil<3><<<1, 1, 0, m->stream>>>(bd, s->stencil);
    A.c0 = s->c0; A.diag = s->diag; A.stencil = s->stencil;
    A.store_c0 = m->export_csr ? 1 : 0;
    if (!m->export_csr) {
      // only the stored rows get slots: offsets from the rank among the non-C0 rows
      int32_t *rank = nullptr, nstored_before_last = 0;
      uint8_t *notc0 = nullptr;
      PHX_HIP(phx_malloc(&rank, sizeof(int32_t) * (size_t)s->n));
      PHX_HIP(phx_malloc(&notc0, (size_t)s->n));
      k_not_flags<<<dim3((unsigned)phx_div_up(s->n, 256)), block, 0, m->stream>>>(s->n, s->c0, notc0);
      int32_t nstored = 0;
      PHX_CHECK(scan_flags(m, notc0, rank, s->n, &nstored));
      (void)nstored_before_last;
      slot_rows = nstored;
      int64_t *off = nullptr;
      uint8_t *wl = nullptr;
      PHX_HIP(phx_malloc(&off, sizeof(int64_t) * (size_t)s->n));
      PHX_HIP(phx_malloc(&wl, (size_t)s->n));
      int lg = 0;
      while ((1 << lg) < W) ++lg;
      k_slot_offsets<<<dim3((unsigned)phx_div_up(s->n, 256)), block, 0, m->stream>>>(s->n, W, lg, s->c0, rank, off, wl);
      sl.off = off; sl.wlog = wl;
      if (nstored > 0) {   // the stored rows as a list: work list of the row kernel (mode 2)
        PHX_HIP(phx_malloc(&stored_rows, sizeof(int32_t) * (size_t)nstored));
        k_flagged_list<<<dim3((unsigned)phx_div_up(s->n, 256)), block, 0, m->stream>>>(s->n, notc0, rank, stored_rows);
        n_stored_rows = nstored;
      }
      later.push_back(rank); later.push_back(notc0);
    }
  }
  // ---- slots (of the stored rows)
  {
    const size_t ns = (size_t)std::max<int64_t>(slot_rows, 1) * W;
    PHX_HIP(phx_malloc(&sl.cols, sizeof(int32_t) * ns));
    PHX_HIP(phx_malloc(&sl.vals, sizeof(double) * ns));
    PHX_HIP(phx_malloc(&sl.overflow, sizeof(int)));
    PHX_HIP(hipMemsetAsync(sl.cols, 0xff, sizeof(int32_t) * ns, m->stream));
    PHX_HIP(hipMemsetAsync(sl.vals, 0, sizeof(double) * ns, m->stream));
    PHX_HIP(hipMemsetAsync(sl.overflow, 0, sizeof(int), m->stream));
  }
  A.slots = sl;
  {
    if (m->is_box) {
      const BoxDims bd{{m->box_n[0], m->box_n[1], m->box_n[2]}, {m->box_h[0], m->box_h[1], m->box_h[2]}};
      const dim3 g((unsigned)phx_div_up(m->nv, 256));
      const int mode = stored_rows ? 1 : 0;
      if (D == 2) k_assemble_rows_box<2><<<g, block, 0, m->stream>>>(m->nv, bd, A, mode, nullptr, nullptr, s->nu);
      else k_assemble_rows_box<3><<<g, block, 0, m->stream>>>(m->nv, bd, A, mode, nullptr, nullptr, s->nu);
      if (stored_rows && n_stored_rows > 0) {
        const dim3 g2((unsigned)phx_div_up(n_stored_rows, 256));
        if (D == 2) k_assemble_rows_box<2><<<g2, block, 0, m->stream>>>(n_stored_rows, bd, A, 2, stored_rows, s->full_of_active, s->nu);
        else k_assemble_rows_box<3><<<g2, block, 0, m->stream>>>(n_stored_rows, bd, A, 2, stored_rows, s->full_of_active, s->nu);
      }
    } else {
      const dim3 g((unsigned)phx_div_up(m->nv, ROW_THREADS)), b(ROW_THREADS);
      if (D == 2) k_assemble_rows<2><<<g, b, 0, m->stream>>>(m->nv, m->v2c_ptr, m->v2c_idx, A);
      else k_assemble_rows<3><<<g, b, 0, m->stream>>>(m->nv, m->v2c_ptr, m->v2c_idx, A);
    }
  }
  static const bool cut_scatter = getenv("PHX_CUT_SCATTER") && atoi(getenv("PHX_CUT_SCATTER")) != 0;   // A/B aid
  if (n_cut > 0 && m->is_box && !cut_scatter) {
    // Kuhn box: the cut-cell terms as row gathers (no atomics)
    const BoxDims bd{{m->box_n[0], m->box_n[1], m->box_n[2]}, {m->box_h[0], m->box_h[1], m->box_h[2]}};
    const int64_t npd = s->n - s->nu;
    const dim3 g((unsigned)phx_div_up(std::max<int64_t>(npd, 1), 256));
    if (D == 2) k_assemble_cut_rows_box<2><<<g, block, 0, m->stream>>>(npd, s->full_of_active + s->nu, bd, A);
    else k_assemble_cut_rows_box<3><<<g, block, 0, m->stream>>>(npd, s->full_of_active + s->nu, bd, A);
  } else if (n_cut > 0) {
    // 64 lanes per cut cell.  Measured at 256^3 (1.1e6 cut cells): 64 lanes per cell 2.1 ms, 8 lanes (one
    // tensor row each) 2.6 ms, one lane per cell 2.7 ms -- the dependent hash probes of a lane serialise;
    // the ghost-penalty facets behave the other way round (one lane per facet: 2.8 -> 1.5 ms).
    PHX_REQUIRE_GRID(n_cut * 64, "cut-cell assembly");
    const dim3 g((unsigned)phx_div_up(n_cut * 64, 256));
    if (D == 2) k_assemble_cut<2><<<g, block, 0, m->stream>>>(n_cut, l_cut, A);
    else k_assemble_cut<3><<<g, block, 0, m->stream>>>(n_cut, l_cut, A);
  }
  PHX_HIP(hipGetLastError());
  if (m->is_submesh) {
    // main.py:74: ds = every exterior facet of the sub-mesh
    if (m->nbf > 0) {
      const dim3 g((unsigned)phx_div_up(m->nbf * 16, 256));
      if (D == 2) k_assemble_ds<2><<<g, block, 0, m->stream>>>(m->nbf, nullptr, m->bfacets, A);
      else k_assemble_ds<3><<<g, block, 0, m->stream>>>(m->nbf, nullptr, m->bfacets, A);
    }
  } else {
    PHX_CHECK(phx_collect_entities(m));  // main.py:65: ds = ds_bdy(100)
    if (m->ent_count[0] > 0) {
      const dim3 g((unsigned)phx_div_up(m->ent_count[0] * 16, 256));
      if (D == 2) k_assemble_ds<2><<<g, block, 0, m->stream>>>(m->ent_count[0], m->ent_buf[0], nullptr, A);
      else k_assemble_ds<3><<<g, block, 0, m->stream>>>(m->ent_count[0], m->ent_buf[0], nullptr, A);
    }
  }
  PHX_HIP(hipGetLastError());
  if (n_fac > 0) {
    // (Per-class macro-element tables for generated boxes -- vertices as anchor + class offsets, the tensor w Jd Jd^T from
    // a table, no f2c / cells / c2f / coordinate loads and no simplex geometry per facet -- were measured at 1.46 ms against
    // 1.49 ms for this kernel at 256^3: the 5e7 hashed f64 atomics are the whole cost.  Not kept.)
    const dim3 g((unsigned)phx_div_up(n_fac, 256));
    if (D == 2) k_assemble_facets<2><<<g, block, 0, m->stream>>>(n_fac, l_fac, A);
    else k_assemble_facets<3><<<g, block, 0, m->stream>>>(n_fac, l_fac, A);
  }
  PHX_HIP(hipGetLastError());
  PHX_HIP(hipStreamSynchronize(m->stream));
  for (void *q : later) PHX_HIP(phx_free(q));
  PHX_HIP(phx_free(l_cut)); PHX_HIP(phx_free(l_fac));
  if (touched) PHX_HIP(phx_free(touched));
  if (stored_rows) PHX_HIP(phx_free(stored_rows));
  {
    const int rc = structured ? finish_structured(s, sl, (int32_t)m->nv, m->export_csr != 0)
                              : phx_finish_system(s, sl, (int32_t)m->nv);
    if (rc != PHX_OK) { phx_system_destroy(s); return rc; }
  }
  *out = s;
  return PHX_OK;
}

static int build_v2c(phx_mesh *m) {
  if (m->v2c_ptr) return PHX_OK;
  const int nvpc = m->ci.nvpc;
  const int64_t tot = m->nc * nvpc;
  unsigned long long *cnt = nullptr;
  PHX_HIP(phx_malloc(&cnt, sizeof(unsigned long long) * (size_t)(m->nv + 1)));
  PHX_HIP(hipMemsetAsync(cnt, 0, sizeof(unsigned long long) * (size_t)(m->nv + 1), m->stream));
  PHX_HIP(phx_malloc(&m->v2c_ptr, sizeof(int64_t) * (size_t)(m->nv + 1)));
  PHX_HIP(phx_malloc(&m->v2c_idx, sizeof(int32_t) * (size_t)tot));
  const dim3 block(256), grid((unsigned)phx_div_up(tot, 256));
  k_v2c_count<<<grid, block, 0, m->stream>>>(m->nc, nvpc, m->cells, cnt);
  PHX_CHECK(exclusive_sum<int64_t>(m, (const int64_t *)cnt, m->v2c_ptr, m->nv + 1));
  PHX_HIP(hipMemsetAsync(cnt, 0, sizeof(unsigned long long) * (size_t)(m->nv + 1), m->stream));
  k_v2c_fill<<<grid, block, 0, m->stream>>>(m->nc, nvpc, m->cells, m->v2c_ptr, cnt, m->v2c_idx);
  PHX_HIP(hipGetLastError());
  PHX_HIP(hipStreamSynchronize(m->stream));
  PHX_HIP(phx_free(cnt));
  return PHX_OK;
}

static int to_device(phx_mesh *m, const double *p, int loc, int64_t n, const double **dev,
                     double **owned) {
  *owned = nullptr;
  PHX_REQUIRE(p != nullptr, PHX_ERR_VALUE, "NULL nodal array");
  if (loc == PHX_DEVICE) { *dev = p; return PHX_OK; }
  PHX_HIP(phx_malloc(owned, sizeof(double) * (size_t)n));
  PHX_HIP(hipMemcpyAsync(*owned, p, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, m->stream));
  *dev = *owned;
  return PHX_OK;
}

// caller-supplied Kuhn box (phx_mesh::inner): tags and nodal data into the numbering of the generated box
__global__ void k_push_tags(int64_t n, const int32_t *__restrict__ map, const int8_t *__restrict__ src, int8_t *__restrict__ dst) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n && map[i] >= 0) dst[map[i]] = src[i];
}
__global__ void k_gather_nodal(int64_t n, const int32_t *__restrict__ lat2v, const double *__restrict__ src, double *__restrict__ dst) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[lat2v[i]];
}

static int assemble_poisson_wd_on_inner(phx_mesh *m, double pen_coef, double stab_coef, const double *phi_h,
                                        const double *f_h, const double *u_D, int loc, phx_system **out) {
  phx_mesh *in = m->inner;
  hipStream_t st = m->stream;
  const dim3 block(256);
  k_push_tags<<<dim3((unsigned)phx_div_up(m->nc, 256)), block, 0, st>>>(m->nc, m->in_cmap, m->cell_tags, in->cell_tags);
  k_push_tags<<<dim3((unsigned)phx_div_up(m->nf, 256)), block, 0, st>>>(m->nf, m->in_fmap, m->facet_tags, in->facet_tags);
  in->have_cell_tags = in->have_facet_tags = true;
  in->sel_counts_valid = false;
  in->sel_cut_valid = false;
  in->act_valid = false;
  in->have_entities = false;
  for (int i = 0; i < 4; ++i) in->tag_hist[i] = m->tag_hist[i];
  for (int i = 0; i < 8; ++i) in->ftag_hist[i] = m->ftag_hist[i];
  in->has_exterior_override = m->has_exterior_override;
  const double *src[3], *dev[3];
  double *owned[3] = {nullptr, nullptr, nullptr}, *perm[3] = {nullptr, nullptr, nullptr};
  src[0] = phi_h; src[1] = f_h; src[2] = u_D;
  const dim3 gv((unsigned)phx_div_up(m->nv, 256));
  for (int k = 0; k < 3; ++k) {
    PHX_CHECK(to_device(m, src[k], loc, m->nv, &dev[k], &owned[k]));
    PHX_HIP(phx_malloc(&perm[k], sizeof(double) * (size_t)m->nv));
    k_gather_nodal<<<gv, block, 0, st>>>(m->nv, m->lat2v, dev[k], perm[k]);
  }
  PHX_HIP(hipGetLastError());
  const int rc = phx_assemble_poisson_wd(in, pen_coef, stab_coef, perm[0], perm[1], perm[2], PHX_DEVICE, out);
  PHX_HIP(hipStreamSynchronize(st));
  for (int k = 0; k < 3; ++k) { if (owned[k]) (void)phx_free(owned[k]); (void)phx_free(perm[k]); }
  if (rc == PHX_OK) { (*out)->out_vertex = m->lat2v; (*out)->outer = m; }
  return rc;
}

extern "C" int phx_assemble_poisson_wd(phx_mesh *m, double pen_coef, double stab_coef,
                                       const double *phi_h, const double *f_h, const double *u_D,
                                       int loc, phx_system **out) {
  PHX_HIP(hipSetDevice(m->device));
  PHX_REQUIRE(m->cell_type == PHX_TRIANGLE || m->cell_type == PHX_TETRAHEDRON,
              PHX_ERR_NOT_IMPLEMENTED, "assembly supports simplices (triangle, tetrahedron) only");
  PHX_REQUIRE(m->have_cell_tags && m->have_facet_tags, PHX_ERR_VALUE,
              "cell and facet tags must be computed before assembly");
  // a caller-supplied mesh that is a Kuhn box in disguise: assemble (and later solve) on the generated box behind it
  if (m->inner) return assemble_poisson_wd_on_inner(m, pen_coef, stab_coef, phi_h, f_h, u_D, loc, out);
  if (!m->is_box) PHX_CHECK(build_v2c(m));  // Kuhn boxes enumerate vertex stars in closed form
  const double *dphi, *df, *dud;
  double *o1, *o2, *o3;
  PHX_CHECK(to_device(m, phi_h, loc, m->nv, &dphi, &o1));
  PHX_CHECK(to_device(m, f_h, loc, m->nv, &df, &o2));
  PHX_CHECK(to_device(m, u_D, loc, m->nv, &dud, &o3));
  PHX_CHECK(phx_begin_timing(m));
  int W = m->gdim == 3 ? 64 : 32;
  int rc = assemble_with_capacity(m, pen_coef, stab_coef, dphi, df, dud, W, out);
  if (rc == PHX_ERR_CAPACITY && W < 64) rc = assemble_with_capacity(m, pen_coef, stab_coef, dphi, df, dud, 64, out);
  if (rc == PHX_OK) rc = phx_end_timing(m, 2);
  if (o1) (void)phx_free(o1);
  if (o2) (void)phx_free(o2);
  if (o3) (void)phx_free(o3);
  return rc;
}

extern "C" int phx_system_info(const phx_system *s, int64_t *info) {
  info[0] = s->n; info[1] = s->nu; info[2] = s->nnz; info[3] = s->nfull;
  info[4] = s->sell_nnz; info[5] = s->slot_cap; info[6] = s->sell_true_nnz; info[7] = s->nslices;
  info[8] = s->sell_indexed_slices; info[9] = s->sell_stream_bytes; info[10] = s->sell_indexed_large;
  info[11] = s->nc0; info[12] = s->p2s ? s->p2s->nrun : s->nseg; info[13] = s->rowptr != nullptr;
  return PHX_OK;
}

extern "C" int phx_system_get_perm(phx_system *s, int32_t *perm, int32_t *dof_u, int32_t *dof_p,
                                   int loc) {
  PHX_HIP(hipSetDevice(s->mesh->device));
  const hipMemcpyKind k = loc == PHX_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  hipStream_t st = s->mesh->stream;   // device-to-device copies do not wait on the host side: stream + explicit wait
  if (perm) PHX_HIP(hipMemcpyAsync(perm, s->perm, sizeof(int32_t) * (size_t)s->n, k, st));
  if (dof_u) PHX_HIP(hipMemcpyAsync(dof_u, s->dof_of_vertex_u, sizeof(int32_t) * (size_t)s->nent, k, st));
  if (dof_p) PHX_HIP(hipMemcpyAsync(dof_p, s->dof_of_vertex_p, sizeof(int32_t) * (size_t)s->nent, k, st));
  PHX_HIP(hipStreamSynchronize(st));
  return PHX_OK;
}

extern "C" int phx_system_export(phx_system *s, int64_t *rowptr, int32_t *col, double *val,
                                 double *rhs, int64_t *dof) {
  PHX_HIP(hipSetDevice(s->mesh->device));
  PHX_REQUIRE(s->rowptr != nullptr || (!rowptr && !col && !val), PHX_ERR_VALUE,
              "this system was assembled without its CSR copy: set PHX_OPT_EXPORT_CSR before assembling");
  if (rowptr) PHX_HIP(hipMemcpy(rowptr, s->rowptr, sizeof(int64_t) * (size_t)(s->n + 1), hipMemcpyDeviceToHost));
  if (col) PHX_HIP(hipMemcpy(col, s->col, sizeof(int32_t) * (size_t)s->nnz, hipMemcpyDeviceToHost));
  if (val) PHX_HIP(hipMemcpy(val, s->val, sizeof(double) * (size_t)s->nnz, hipMemcpyDeviceToHost));
  if (rhs) PHX_HIP(hipMemcpy(rhs, s->rhs, sizeof(double) * (size_t)s->n, hipMemcpyDeviceToHost));
  if (dof) PHX_HIP(hipMemcpy(dof, s->full_of_active, sizeof(int64_t) * (size_t)s->n, hipMemcpyDeviceToHost));
  if (dof && s->out_vertex) {   // assembled on the inner box of a caller-supplied mesh: full indices in the caller's numbering
    std::vector<int32_t> map((size_t)s->nent);
    PHX_HIP(hipMemcpy(map.data(), s->out_vertex, sizeof(int32_t) * (size_t)s->nent, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < s->n; ++i) dof[i] = (dof[i] / s->nent) * s->nent + map[(size_t)(dof[i] % s->nent)];
  }
  return PHX_OK;
}

#include "phx_assemble_p2.inc.hip"
#include "phx_assemble_sd.inc.hip"
#include "phx_assemble_el.inc.hip"
#include "phx_assemble_flux.inc.hip"
#include "phx_assemble_flux_quad.inc.hip"
#include "phx_errors.inc.hip"
