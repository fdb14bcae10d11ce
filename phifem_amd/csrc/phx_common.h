// Internal declarations shared by the translation units of libphifem_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "../../include/phifem_hip.h"

#define PHX_MAX_PTS 40   // detection points per cell (tet degree 4 has 34)
#define PHX_MAX_VPC 4

void phx_set_error(const char *fmt, ...);

// Caching device allocator: assembly and solve allocate and release the same multi-GB transient
// buffers on every pass; hipMalloc/hipFree of such blocks costs hundreds of milliseconds at
// BASELINE scale (0.8 s per pass on the 1024x1024x128 slab).  Released blocks are kept per device and
// handed back on an exact-size match; everything cached is returned to the driver when a real
// allocation fails or the cache exceeds PHX_POOL_LIMIT_GB (default: 60 % of the device memory).
hipError_t phx_pool_malloc(void **p, size_t bytes);
hipError_t phx_pool_free(void *p);
void phx_pool_trim(void);
template <typename T>
static inline hipError_t phx_malloc(T **p, size_t bytes) { return phx_pool_malloc((void **)p, bytes); }
static inline hipError_t phx_free(void *p) { return phx_pool_free(p); }

#define PHX_HIP(expr)                                                                       \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) {                                                                 \
      phx_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_));   \
      return PHX_ERR_HIP;                                                                   \
    }                                                                                       \
  } while (0)

#define PHX_CHECK(expr)        \
  do {                         \
    int rc_ = (expr);          \
    if (rc_ != PHX_OK) return rc_; \
  } while (0)

#define PHX_REQUIRE(cond, code, ...) \
  do {                               \
    if (!(cond)) {                   \
      phx_set_error(__VA_ARGS__);    \
      return (code);                 \
    }                                \
  } while (0)

static inline int64_t phx_div_up(int64_t a, int64_t b) { return (a + b - 1) / b; }
// A HIP grid holds fewer than 2^32 threads; a larger launch covers only part of the work without an error.
#define PHX_REQUIRE_GRID(threads_total, what)                                                        \
  PHX_REQUIRE((int64_t)(threads_total) < ((int64_t)1 << 32), PHX_ERR_VALUE,                           \
              "%s: %lld threads exceed one HIP grid -- partition the problem", what, (long long)(threads_total))

// Bits 0-6 of a cell-tag byte: 1 inside, 2 cut, 3 outside (or a user tag <= 127).  Bit 7: the
// `ds` detection says the level-set changes sign over the cell's background-boundary facets.
#define PHX_TAG_MASK 0x7f
#define PHX_BCUT_BIT 0x80
// Kernels that look at ONE tag byte per entity and act on few of them take four entities per thread from one 32-bit
// load (device allocations are 256-byte aligned): `i0` = first of the four, bytes past n read as 0x7f (no tag).
__device__ __forceinline__ uint32_t phx_tag_word(const int8_t *tags, int64_t i0, int64_t n) {
  if (i0 + 3 < n) return *reinterpret_cast<const uint32_t *>(tags + i0);
  uint32_t w = 0x7f7f7f7fu;
  for (int j = 0; j < 4; ++j)
    if (i0 + j < n) w = (w & ~(255u << (8 * j))) | ((uint32_t)(uint8_t)tags[i0 + j] << (8 * j));
  return w;
}

struct phx_cell_info {
  int tdim, nvpc, nfpc, nvpf;
  int fv[4][3];  // local facet -> local vertices
};
int phx_get_cell_info(int cell_type, phx_cell_info *ci);

struct phx_mesh {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int gdim = 0, cell_type = 0;
  phx_cell_info ci{};
  int64_t nv = 0, nc = 0, nf = 0, nbf = 0;
  double *x = nullptr;        // [nv*gdim]
  int32_t *cells = nullptr;   // [nc*nvpc]
  int32_t *c2f = nullptr;     // [nc*nfpc]
  int32_t *f2c = nullptr;     // [nf*2]
  int32_t *bfacets = nullptr; // [nbf*2] (cell, local facet), ascending facet id
  int32_t *bfacet_ids = nullptr; // [nbf] facet ids, ascending
  // slab of a partitioned box: facets on an ARTIFICIAL end plane are not background-boundary
  // facets; they stay untagged (0) and are left out of the `ds` detection
  uint8_t *facet_exempt = nullptr;  // [nf] or NULL
  // vertex -> incident cells (CSR), built on first assembly: rows of the stiffness block are
  // gathered by their owning vertex instead of scattered with atomics
  // edges (P2 DoFs): 2-D edges ARE the facets; 3-D boxes use a closed form, others a host sort
  int64_t ne = 0;
  int32_t *c2e = nullptr;    // [nc*nepc], local edge k per basix: tri (1,2),(0,2),(0,1);
                             // tet (2,3),(1,3),(1,2),(0,3),(0,2),(0,1)
  int32_t *edges = nullptr;  // [ne*2] vertex pairs, ascending
  bool c2e_is_alias = false;
  int64_t box_n[3] = {0, 0, 0};
  double box_h[3] = {0.0, 0.0, 0.0};  // exact lattice spacing (hi - lo) / n_global per axis
  int64_t box_off[3] = {0, 0, 0};     // cube offset of this (slab of a) box in the global box
  int64_t box_nglob[3] = {0, 0, 0};   // cubes per axis of the global box
  int64_t *v2c_ptr = nullptr;  // [nv+1]
  int32_t *v2c_idx = nullptr;  // [nc*nvpc]
  bool is_box = false;
  int64_t box_plane = 0, box_nlast = 0;
  int8_t *cell_tags = nullptr;   // [nc]
  int8_t *facet_tags = nullptr;  // [nf]
  bool have_cell_tags = false, have_facet_tags = false;
  int64_t tag_hist[4] = {0, 0, 0, 0};
  int64_t ftag_hist[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // per-chunk counts (chunks of 2048 facets, phx_select.h) of the two facet selections every assembly starts with,
  // left behind by the facet tagging kernel: [0] ghost-penalty facets (tag 2 / 3, interior), [1] tags 3 / 4.  Valid
  // while sel_counts_valid (the facet tags are the ones k_tag_facets wrote).
  int32_t *sel_counts[2] = {nullptr, nullptr};
  bool sel_counts_valid = false;
  int64_t sel_total[2] = {0, 0};       // the totals of sel_counts[.] (host copies, valid with sel_counts_valid)
  int32_t *sel_counts_cut = nullptr;   // the same for the cut cells (tag 2), left behind by the last cell tagging kernel
  bool sel_cut_valid = false;
  // vertex flags the single-layer tagging pass produces anyway: act_in[v] = v belongs to a cell tagged 1, act_cut[v] =
  // v belongs to a cell tagged 2 (after the demotion).  The P1 assembly numbers its DoFs from them (u: either, p: cut)
  // instead of walking the cells again.  Valid while act_valid.
  uint8_t *act_in = nullptr, *act_cut = nullptr;
  bool act_valid = false;
  // integration entities of the current tags (device, unordered): (key, cell, lf) triples
  int64_t *ent_buf[2] = {nullptr, nullptr};
  int64_t ent_count[2] = {0, 0};
  bool have_entities = false;
  // sub-mesh provenance
  bool is_submesh = false;
  int32_t *c_map_h = nullptr, *v_map_h = nullptr;
  // sub-mesh of a Kuhn box: the lattice of the parent (box_n, box_h are copied) and the two vertex maps on
  // the device, so that the box preconditioner applies to sub-mesh systems too
  bool on_box_lattice = false;
  int32_t *v2lat = nullptr;   // [nv]          sub-mesh vertex -> parent lattice vertex
  int32_t *lat2v = nullptr;   // [parent nv]   parent lattice vertex -> sub-mesh vertex or -1
  double timings[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  bool own_stream = true;
  int profile_spmv = 0;
  int spmv_xcd_group = 0;          // PHX_OPT_SPMV_XCD_GROUP
  int64_t stencil_plane_rows = 32768;  // PHX_OPT_STENCIL_PLANE_ROWS
  int spmv_value_index = 1;        // PHX_OPT_SPMV_VALUE_INDEX
  int precond = 1;                 // PHX_OPT_PRECOND: 0 Jacobi, 1 / 2 box sine transforms in f64 / f32 where applicable
  int has_exterior_override = -1;  // -1: decide from the local tags; 0/1: imposed (multi-GPU)
  // Per-solve host resources live HERE, not in the systems a mesh sees come and go (one per bench step): creating
  // 4096 profiling events and a pinned buffer per system cost 1.2 ms of idle GPU per step.
  std::vector<hipEvent_t> prof_ev[2];  // event pairs of the sampled launches: [0] SpMV, [1] sine-transform y pass
  int prof_used[2] = {0, 0}, prof_seen[2] = {0, 0};
  double *scal_h = nullptr;            // 16 pinned doubles: the Krylov scalars the host looks at
  int export_csr = 0;              // PHX_OPT_EXPORT_CSR: assembly also builds the CSR copy phx_system_export reads
  int structured = 1;              // PHX_OPT_STRUCTURED: stencil-coded interior rows on Kuhn boxes (P1 weak Dirichlet)
  int allow_empty = 0;             // PHX_OPT_ALLOW_EMPTY: assembly returns an EMPTY system when no cell is tagged 1 / 2
  int el_coarse = -1;              // PHX_OPT_EL_COARSE
  int deterministic = 0;           // PHX_OPT_DETERMINISTIC: bit-reproducible P2 / elasticity assembly and Krylov dot products
  // A caller-supplied mesh that IS a Kuhn box in some vertex / cell order (what dolfinx's create_box / create_rectangle
  // hand over, demo/weak-dirichlet/flower/main.py:45-46): `inner` is the generated box with the same lattice, the maps
  // translate.  Tags are computed on THIS mesh (caller numbering, bit-exact as before); the P1 weak-Dirichlet assembly
  // pushes them to `inner` and assembles / solves there -- closed-form rows, stencil operator, SELL-16, lattice
  // preconditioner -- and the solution comes back in the caller's numbering.  Vertex maps: v2lat / lat2v.
  phx_mesh *inner = nullptr;
  int32_t *in_cmap = nullptr;      // [nc] cell -> cell of `inner`
  int32_t *in_fmap = nullptr;      // [nf] facet -> facet of `inner`
};

// ---- P2 on a Kuhn box: the DoFs (vertices and edge midpoints) are exactly the points of the lattice of spacing h / 2
// ("fine lattice", F_a = 2 n_a + 1 points per axis).  Closed-form maps both ways: the edge numbering of a generated box
// is base[class] + anchor index inside the class extent (phx_mesh.hip, k_box_c2e).
struct phx_p2_lattice {
  int64_t n[3];        // cubes per axis
  int64_t F[3];        // fine points per axis
  int64_t nv;          // vertices
  int64_t base[8];     // edge class c holds the ids base[c] .. base[c+1]-1
  int64_t ext[7][3];   // anchor extents of the class
};
// edge class of a direction (a, b, c) in {0,1}^3 \ 0: x, y, z, (x,y), (x,z), (y,z), (x,y,z)
__host__ __device__ __forceinline__ int phx_p2_edge_class(int a, int b, int c) {
  const int nd = a + b + c;
  return nd == 1 ? (a ? 0 : (b ? 1 : 2)) : (nd == 2 ? (!c ? 3 : (!b ? 4 : 5)) : 6);
}
__host__ __device__ __forceinline__ int64_t phx_p2_entity_of_fine(const phx_p2_lattice &L, int64_t I, int64_t J, int64_t K) {
  const int a = (int)(I & 1), b = (int)(J & 1), c = (int)(K & 1);
  const int64_t i = I >> 1, j = J >> 1, k = K >> 1;
  if (!(a | b | c)) return i + (L.n[0] + 1) * (j + (L.n[1] + 1) * k);
  const int cls = phx_p2_edge_class(a, b, c);
  return L.nv + L.base[cls] + i + L.ext[cls][0] * (j + L.ext[cls][1] * k);
}
__host__ __device__ __forceinline__ void phx_p2_fine_of_entity(const phx_p2_lattice &L, int64_t e, int64_t *q) {
  if (e < L.nv) {
    const int64_t n0 = L.n[0] + 1, n1 = L.n[1] + 1, r = e / n0;
    q[0] = 2 * (e - r * n0); q[1] = 2 * (r % n1); q[2] = 2 * (r / n1);
    return;
  }
  const int64_t id = e - L.nv;
  int cls = 0;
  while (cls < 6 && id >= L.base[cls + 1]) ++cls;
  const int64_t rem = id - L.base[cls], r = rem / L.ext[cls][0];
  const int dir[7][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {1, 1, 0}, {1, 0, 1}, {0, 1, 1}, {1, 1, 1}};
  q[0] = 2 * (rem - r * L.ext[cls][0]) + dir[cls][0];
  q[1] = 2 * (r % L.ext[cls][1]) + dir[cls][1];
  q[2] = 2 * (r / L.ext[cls][1]) + dir[cls][2];
}
// Structured P2 system: interior rows (every DoF of the 5 x 5 x 5 fine neighbourhood has its whole support tagged
// inside) are applied from 8 translation-invariant stencils (one per parity class of the fine point) over runs of
// consecutive fine points of one x line; everything else sits in SELL over a row list, as for P1.
#define PHX_P2S_REC 32   // ints per run record: {first position, length, (b + 2 c) | a0 << 2, 25 line offsets, pad}
struct phx_p2_struct {
  phx_p2_lattice lat;
  double *coef = nullptr;      // device [8][125]: class a + 2 b + 4 c, offset (dx+2) + 5 (dy+2) + 25 (dz+2)
  unsigned long long *mask = nullptr;  // device [4][2]: offsets with a non-zero coefficient for a = 0 or a = 1, per (b + 2 c)
  // the same coefficients as k_spmv_p2s reads them: per line type bc = b + 2 c and neighbouring line l = (dy+2) + 5 (dz+2)
  // the five dx coefficients for a = 0 (tabE) and a = 1 (tabO); linemask[bc] bit l = the line has a non-zero coefficient
  double *tabE = nullptr, *tabO = nullptr;   // device [4][25][5]
  unsigned *linemask = nullptr;              // device [4]
  int32_t *runs = nullptr;     // device [nrun][PHX_P2S_REC]
  int64_t nrun = 0, nc0i = 0;
};

struct phx_system {
  phx_mesh *mesh = nullptr;
  int device = 0;
  int64_t n = 0, nu = 0, nnz = 0, nfull = 0;
  int64_t nent = 0;  // DoF entities per field: nv (P1) or nv + ne (P2)
  int slot_cap = 0;
  // original active numbering
  int32_t *dof_of_vertex_u = nullptr;  // [nv]  active index or -1
  int32_t *dof_of_vertex_p = nullptr;  // [nv]
  int64_t *full_of_active = nullptr;   // [n]
  int64_t *rowptr = nullptr;           // [n+1]
  int32_t *col = nullptr;              // [nnz]
  double *val = nullptr;               // [nnz]
  double *rhs = nullptr;               // [n]
  double *diag = nullptr;              // [n]
  int32_t *row_nz = nullptr;           // [n] stored entries per row with explicit zeros dropped (from the compaction; may be null)
  // SELL-64 in solver ordering (rows permuted by window-sorted length)
  int64_t nslices = 0, sell_nnz = 0, sell_true_nnz = 0;
  int64_t *slice_ptr = nullptr;  // [nslices+1] offsets in units of entries
  int32_t *sell_col = nullptr;   // [sell_nnz] solver-order column ids
  double *sell_val = nullptr;    // [sell_nnz] values of A*D^-1 (right Jacobi scaling)
  double *sell_val_raw = nullptr;// [sell_nnz] values of A
  uint8_t *sell_kind = nullptr;      // [nslices] dictionary size of a value-indexed slice of sell_val, 0 = raw doubles
  uint8_t *sell_kind_raw = nullptr;  // same for sell_val_raw (second half of the sell_kind allocation)
  int64_t sell_indexed_slices = 0, sell_indexed_large = 0, sell_stream_bytes = 0;
  int32_t *perm = nullptr;       // [n] solver position -> original active row
  int32_t *iperm = nullptr;      // [n] original active row -> solver position
  // solver workspace
  double *work = nullptr;        // 8 vectors of n
  double *scal = nullptr;        // device scalars
  double *scal_h = nullptr;      // pinned; BORROWED from the mesh (phx_mesh_pinned_scalars)
  // externally attached Krylov buffers (multi-GPU driver) and ownership mask (solver order)
  double *kr_work = nullptr, *kr_scal = nullptr;
  const uint8_t *own = nullptr;
  // fictitious-domain preconditioner (phx_precond.inc.hip): built on demand for P1 Poisson systems on 3-D boxes
  bool u_vertex_block = false;     // rows [0, nu) are one scalar u DoF per active vertex
  bool u_weighted = false;         // u block ~ S K S with a nodal weight S (strong Dirichlet, S ~ |phi_h|)
  bool u_p2_block = false;         // rows [0, nu): P2, one DoF per active vertex and edge (entities nv + ne)
  struct phx_box_precond *precond = nullptr;
  int precond_state = 0;           // 0 not tried, 1 built, -1 not applicable
  bool precond_veto = false;       // multi-GPU vote: this rank cannot run the box preconditioner although it has u rows
  // --- structured systems (P1 weak-Dirichlet on a Kuhn box).  C0 = rows of vertices interior to the box, untouched
  // by the scattering kernels, whose whole star is tagged inside: the translation-invariant 7-point (2-D: 5-point)
  // row.  Solver order: C0 rows in lattice order, then the other u rows, then the p rows.  C0 rows whose axis
  // neighbours are C0 rows too are APPLIED from `stencil` over runs of consecutive positions of one x line
  // (`seg`), never stored; all other rows sit in a SELL-16 copy over a row list (`sell_rows`).
  bool structured = false;
  bool u_unscaled = false;         // columns of u DoFs carry A (the u preconditioner is K_box^-1 or D_u^-1), p columns A D^-1
  uint8_t *c0 = nullptr;           // [n] 1: row applied by the stencil kernel
  int64_t nc0 = 0;                 // rows the stencil applies
  int64_t nstencil_pos = 0;        // solver positions [0, nstencil_pos) hold the C0 rows (the stencil slices cover them)
  double *stencil = nullptr;       // device [8]: {diag, x, y, z off-diagonal entries} of a C0 row, [4] = claimed flag
  int32_t *seg = nullptr;          // [nseg][6] {first row, end row, offset of the +y, -y, +z, -z neighbour rows}
  int32_t nseg = 0;
  int32_t *slice_seg = nullptr;    // [ceil(nu / 64)][16] slice records: {runs in the slice, first run, two runs inline}
  // stencil blocks of LARGE lattice planes (three planes of x exceed an XCD's L2): block (XCD k, sequence j) -> {first
  // slice, slices} so that XCD k walks the k-th eighth of EVERY plane, plane after plane (phx_solve.hip, k_stmap_build)
  int32_t *st_map = nullptr;       // [8][st_chunk][2]
  int64_t st_chunk = 0;
  int32_t *sell_rows = nullptr;    // [nslices * 64] row held by SELL slot (slice, lane), -1: padding
  int64_t n_sell_rows = 0;
  double *cscale = nullptr;        // [n] x = cscale * y when the iteration ends (1 for unscaled columns, else 1 / diag)
  double *pvec = nullptr;          // [2 n] phat / shat of the library-owned workspace when no box preconditioner holds them
  // --- multi-GPU overlap (phx_dist.inc.hip): rows that reference halo entries.  The SpMV of an iteration runs in two
  // launches: every other row while the halo is in flight, these rows (k_spmv_bnd over `bnd_rec`) after the unpack.
  uint8_t *bnd = nullptr;          // [n] 1: row references an entry some neighbour sends
  int32_t *bnd_rec = nullptr;      // [nbnd][6] {row, kind (0 stencil / 1 SELL-16 / 2 SELL-64), 4 kind-specific ints}
  int64_t nbnd = 0;
  phx_p2_struct *p2s = nullptr;    // structured P2 system (3-D Kuhn boxes), else nullptr
  // interface elasticity: vertex-block Jacobi (phx_blockjac.inc.hip), built on the first solve from the CSR copy
  int el_nblk = 0;                 // > 0: block-major system with this many blocks of nv entries (27 / 14)
  struct phx_blockjac *bj = nullptr;
  bool bj_tried = false;
  struct phx_coarse *cc = nullptr; // coarse correction on top of the vertex blocks (phx_coarse.inc.hip)
  bool cc_tried = false;
  // PHX_OPT_DETERMINISTIC: every block of a dot-product kernel leaves its partial sum in its own entry of `dpart`
  // ([2][dpart_cap]) instead of adding it to a slot atomically; k_fold_partials sums them in a fixed order
  double *dpart = nullptr;
  int64_t dpart_cap = 0, dpart_used = 0;
  // system assembled on the `inner` box of a caller-supplied mesh: vertex of s->mesh -> vertex of the caller's mesh
  // (applied where full DoF indices leave the library: the solution vector, phx_system_export's dof map)
  const int32_t *out_vertex = nullptr;
  phx_mesh *outer = nullptr;       // the caller's mesh (timings are mirrored there)
};

// helpers implemented in phx_mesh.hip
int phx_mesh_alloc_common(phx_mesh *m);
int phx_begin_timing(phx_mesh *m);
int phx_end_timing(phx_mesh *m, int slot);
int phx_end_timing_mark(phx_mesh *m);
int phx_end_timing_read(phx_mesh *m, int slot);
int phx_mesh_build_edges(phx_mesh *m);
int phx_mesh_create_from(int gdim, int cell_type, int64_t nv, const double *coords, int64_t nc,
                         const int32_t *cells, int loc, int device, phx_mesh **out);
int phx_mesh_pinned_scalars(phx_mesh *m, double **out);
// Several small device values to the host with ONE round trip on `st`: a one-wave kernel packs them into a staging
// block, one copy lands in pinned memory.  (Every hipMemcpyAsync into a pageable host variable is a round trip of its
// own, ~20 us each: the totals of two scans cost four.)  At most 8 items of at most 64 bytes.
struct phx_rb_item { const void *dev; int bytes; void *host; };
int phx_read_back(hipStream_t st, const phx_rb_item *items, int n);

int phx_system_build_empty(phx_system *s);  // phx_solve.hip
// row slots of the assembly as the SELL builder of structured systems reads them (phx_assemble.hip: Slots)
struct phx_slot_view {
  const int32_t *cols;
  const double *vals;
  int W;
  const uint8_t *clean;   // clean[row] = c > 0: the row's c entries sit at slots 0 .. c-1
  const int64_t *off;     // per-row slot offsets (nullptr: row * W)
  const uint8_t *wlog;
  bool pneg = false;      // column key of a p DoF e: -2 - e (structured P2) instead of nent + e
};
int phx_system_build_structured(phx_system *s, const phx_slot_view &sv, int32_t nent);  // phx_solve.hip
// structured P2: lattice flags of the C0 rows / of the rows the stencils apply (uint8 [F0 F1 F2]), s->c0 = the latter
// in active numbering, s->p2s with lattice + coefficient tables set (phx_solve.hip)
int phx_system_build_structured_p2(phx_system *s, const phx_slot_view &sv, int32_t nent, const uint8_t *latc0,
                                   const uint8_t *latc0i);
struct phx_box_precond;
void phx_box_precond_destroy(phx_box_precond *bp);  // phx_solve.hip
struct phx_blockjac;
void phx_blockjac_destroy(phx_blockjac *b);         // phx_solve.hip
struct phx_coarse;
void phx_coarse_destroy(phx_coarse *c);             // phx_solve.hip
