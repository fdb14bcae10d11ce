// Cell / facet classification against the level-set on the device (gfx950).
// Replaces src/phifem/mesh_scripts.py:95-134 (_compute_detection_vector), :284-390 (_tag_cells),
// :393-558 (_tag_facets), :137-192 (_compute_integration_entities), :561-568 (_overwrite_tags).
//
// All kernels are HBM-bound streaming passes (one thread per cell / facet, 16-byte connectivity
// loads, one byte written per entity).  This file is compiled with -ffp-contract=off: the tags
// hang on exact float compares (mesh_scripts.py:343-347), so every sum is evaluated in the order
// the oracle spells out and without fused multiply-add.
#include "phx_prim.h"
#include <string.h>

#include <algorithm>
#include <vector>

#include "phx_common.h"
#include "phx_select.h"

int phx_shape_table(int cell_type, int degree, int which, std::vector<double> &tab, int *npts,
                    int *nfun);

struct DetTab {
  int npts, nfun;
  double N[PHX_MAX_PTS * 4];
};
struct Quadric {
  double c[3], s[3], c0;
};
struct FacetVerts {
  int nfpc, nvpf;
  int fv[4][3];
};

template <int GDIM>
__device__ __forceinline__ double quadric_eval(const Quadric &q, const double *xq) {
  const double t0 = q.s[0] * xq[0] - q.c[0];
  const double t1 = q.s[1] * xq[1] - q.c[1];
  double r = t0 * t0 + t1 * t1;
  if (GDIM == 3) {
    const double t2 = q.s[2] * xq[2] - q.c[2];
    r = r + t2 * t2;
  }
  return r + q.c0;
}

// phi at detection point q of an entity with NV vertices `v`
template <int KIND, int GDIM>
__device__ __forceinline__ double phi_at(const DetTab &tab, int q, int nvert, const int32_t *v,
                                         const double *__restrict__ phi,
                                         const double *__restrict__ x, const Quadric &quad,
                                         int64_t point_base) {
  const double *N = &tab.N[q * tab.nfun];
  if (KIND == PHX_PHI_NODAL_P1) {
    double a = N[0] * phi[v[0]];
    for (int i = 1; i < nvert; ++i) a = a + N[i] * phi[v[i]];
    return a;
  } else if (KIND == PHX_PHI_POINTS) {
    return phi[point_base + q];
  } else {
    double xq[3] = {0.0, 0.0, 0.0};
    for (int d = 0; d < GDIM; ++d) {
      double a = N[0] * x[(int64_t)v[0] * GDIM + d];
      for (int i = 1; i < nvert; ++i) a = a + N[i] * x[(int64_t)v[i] * GDIM + d];
      xq[d] = a;
    }
    return quadric_eval<GDIM>(quad, xq);
  }
}

// --- a2 + a3: detection ratio and classification, one thread per cell -----------------------
template <int KIND, int GDIM, int NVPC>
__global__ void __launch_bounds__(256)
k_tag_cells(int64_t nc, DetTab tab, const int32_t *__restrict__ cells,
            const double *__restrict__ phi, const double *__restrict__ x, Quadric quad,
            int8_t *__restrict__ tags, int *__restrict__ warn) {
  const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (c >= nc) return;
  int32_t v[NVPC];
  if constexpr (NVPC == 4) {
    const int4 cv = *reinterpret_cast<const int4 *>(cells + c * 4);
    v[0] = cv.x; v[1] = cv.y; v[2] = cv.z; v[3] = cv.w;
  } else {
    for (int i = 0; i < NVPC; ++i) v[i] = cells[c * NVPC + i];
  }
  double num = 0.0, den = 0.0;
  bool pos = false, neg = false;
  for (int q = 0; q < tab.npts; ++q) {
    const double p = phi_at<KIND, GDIM>(tab, q, NVPC, v, phi, x, quad, c * (int64_t)tab.npts);
    num = num + p;
    den = den + fabs(p);
    pos |= p > 0.0;
    neg |= p < 0.0;
  }
  const double den0 = den;
  if (pos && neg) {
    // Samples of both signs: FFCx scales every term of a dx sum by |det J| (x weight 1), and whether a tiny term is
    // absorbed by the rounding of the running sum depends on that factor (oracle/tagging.py:_ratio -- with it the 8
    // ellipse_in_square degree-3 goldens are reproduced).  One sign only: num = +-den term by term whatever the factor,
    // so the (few) cells that need the coordinate gathers are the cut ones.  |det J|: edge vectors from vertex 0,
    // cofactor expansion along the first row, left to right (oracle/tagging.py:cell_scale).
    double e[GDIM][GDIM];
    for (int a = 0; a < GDIM; ++a)
      for (int dd = 0; dd < GDIM; ++dd) e[a][dd] = x[(int64_t)v[a + 1] * GDIM + dd] - x[(int64_t)v[0] * GDIM + dd];
    double s;
    if (GDIM == 2) {
      s = fabs(e[0][0] * e[1][1] - e[0][1] * e[1][0]);
    } else {
      const double c0 = e[1][1] * e[GDIM - 1][GDIM - 1] - e[1][GDIM - 1] * e[GDIM - 1][1];
      const double c1 = e[1][0] * e[GDIM - 1][GDIM - 1] - e[1][GDIM - 1] * e[GDIM - 1][0];
      const double c2 = e[1][0] * e[GDIM - 1][1] - e[1][1] * e[GDIM - 1][0];
      s = fabs((e[0][0] * c0 - e[0][1] * c1) + e[0][GDIM - 1] * c2);
    }
    num = 0.0;
    den = 0.0;
    for (int q = 0; q < tab.npts; ++q) {
      const double t = phi_at<KIND, GDIM>(tab, q, NVPC, v, phi, x, quad, c * (int64_t)tab.npts) * s;
      num = num + t;
      den = den + fabs(t);
    }
  }
  // mesh_scripts.py:124-128: 0.5 wherever the denominator is not > 0 (zero or NaN)
  const double d = (den > 0.0) ? num / den : 0.5;
  int8_t t = 0;
  if (d > -1.0 && d < 1.0) t = 2;   // :343-345
  if (d == 1.0) t = 3;              // :346
  if (d == -1.0) t = 1;             // :347
  tags[c] = t;
  if (fabs(den0) <= 1.0e-8) atomicOr(warn, 1);  // numpy.isclose(den, 0.0), :129 (on the unscaled sum)
}

// --- single layer (mesh_scripts.py:349-358) -------------------------------------------------
// "none of the cells sharing a vertex with this cut cell is inside" == "no vertex of this cut
// cell belongs to an inside cell": mark, then test.  No vertex->cell adjacency is needed.
template <int NVPC>
__global__ void k_mark_inside_vertices(int64_t nc, const int32_t *__restrict__ cells,
                                       const int8_t *__restrict__ tags,
                                       uint8_t *__restrict__ touched) {
  const int64_t c0 = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) * 4;   // four cells per thread (phx_tag_word)
  if (c0 >= nc) return;
  const uint32_t w = phx_tag_word(tags, c0, nc);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (((w >> (8 * j)) & PHX_TAG_MASK) != 1u) continue;
    const int64_t c = c0 + j;
    for (int i = 0; i < NVPC; ++i) touched[cells[c * NVPC + i]] = 1;
  }
}

// The last kernel that writes cell tags also counts them (as k_tag_facets does for the facets): hist_part[bin][block]
// for the bins 0..3 and, per chunk of PHX_SEL_CHUNK cells, the cut cells (tag 2) the assembly will select.
// Launched with 256 threads; touched == nullptr: no demotion (single_layer_cut off), counting only.
template <int NVPC>
__global__ void __launch_bounds__(256)
k_demote_isolated_cut(int64_t nc, const int32_t *__restrict__ cells, int8_t *__restrict__ tags,
                      const uint8_t *__restrict__ touched, uint32_t *__restrict__ hist_part,
                      int32_t *__restrict__ sel_cut, uint8_t *__restrict__ vcut) {
  __shared__ uint32_t lh[4][4];
  const int64_t c0 = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) * 4;
  int tj[4] = {0x7f, 0x7f, 0x7f, 0x7f};
  if (c0 < nc) {
    const uint32_t w = phx_tag_word(tags, c0, nc);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (c0 + j >= nc) continue;
      int t = (int)((w >> (8 * j)) & PHX_TAG_MASK);
      if (t == 2 && touched) {
        const int64_t c = c0 + j;
        int32_t v[NVPC];
        bool keep = false;
        for (int i = 0; i < NVPC; ++i) { v[i] = cells[c * NVPC + i]; keep = keep || touched[v[i]]; }
        if (!keep) { tags[c] = 3; t = 3; }
        else if (vcut) for (int i = 0; i < NVPC; ++i) vcut[v[i]] = 1;   // vertices of the cells that stay cut
      }
      tj[j] = t;
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t cnt[4] = {0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int b = 0; b < 4; ++b) cnt[b] += (uint32_t)__popcll(__ballot(tj[j] == b));
  if (lane == 0) {
    const int64_t wbase = (blockIdx.x * (int64_t)blockDim.x + (threadIdx.x & ~63)) * 4;
    if (wbase < nc && cnt[2]) atomicAdd(&sel_cut[wbase / PHX_SEL_CHUNK], (int32_t)cnt[2]);
    for (int b = 0; b < 4; ++b) lh[wv][b] = cnt[b];
  }
  __syncthreads();
  if (threadIdx.x < 4)
    hist_part[(size_t)threadIdx.x * gridDim.x + blockIdx.x] =
        lh[0][threadIdx.x] + lh[1][threadIdx.x] + lh[2][threadIdx.x] + lh[3][threadIdx.x];
}

// Tag histogram: four tag bytes per lane and load, counted with ballots (the counters are
// wave-uniform registers; contended LDS atomics made the first version 10x slower than the read).
__global__ void __launch_bounds__(256)
k_tag_hist(int64_t n, const int8_t *__restrict__ tags, int nbins, unsigned long long *__restrict__ hist) {
  __shared__ unsigned int lh[8];
  if (threadIdx.x < 8) lh[threadIdx.x] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int64_t nwords = n >> 2;
  const uint32_t *tw = reinterpret_cast<const uint32_t *>(tags);  // device allocations are 256-B aligned
  unsigned int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t base = wave * 64; base < nwords; base += nwaves * 64) {
    const int64_t i = base + lane;
    const uint32_t w = i < nwords ? tw[i] : 0x7f7f7f7fu;  // 0x7f is no tag: falls through every bin
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int t = (int)((w >> (8 * j)) & PHX_TAG_MASK);
#pragma unroll
      for (int b = 0; b < 8; ++b)
        if (b < nbins) cnt[b] += (unsigned int)__popcll(__ballot(t == b));
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (int64_t i = nwords << 2; i < n; ++i) {
      const int t = tags[i] & PHX_TAG_MASK;
      if (t < nbins) atomicAdd(&lh[t], 1u);
    }
  if (lane == 0)
    for (int b = 0; b < 8; ++b)
      if (cnt[b]) atomicAdd(&lh[b], cnt[b]);
  __syncthreads();
  if (threadIdx.x < nbins && lh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)lh[threadIdx.x]);
}

// --- `ds` detection of the background-boundary facets (mesh_scripts.py:434-452) --------------
// One thread per boundary facet; the thread owning a cell's FIRST boundary facet judges the cell
// on all of its boundary facets together (one partial sum per facet, added in local order).
__device__ __forceinline__ int64_t lower_bound_i32(const int32_t *a, int64_t n, int32_t key) {
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (a[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

template <int KIND, int GDIM>
__global__ void k_boundary_cell_cut(int64_t nbf, DetTab tab, FacetVerts fvs, int nvpc,
                                    const int32_t *__restrict__ bfacets,
                                    const int32_t *__restrict__ bfacet_ids,
                                    const int32_t *__restrict__ cells,
                                    const int32_t *__restrict__ c2f,
                                    const int32_t *__restrict__ f2c,
                                    const double *__restrict__ phi,
                                    const double *__restrict__ x, Quadric quad,
                                    const uint8_t *__restrict__ exempt,
                                    int8_t *__restrict__ tags) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= nbf) return;
  const int32_t c = bfacets[2 * i];
  const int lf0 = bfacets[2 * i + 1];
  if (exempt && exempt[bfacet_ids[i]]) return;
  for (int k = 0; k < lf0; ++k) {
    const int32_t fk = c2f[(int64_t)c * fvs.nfpc + k];
    if (f2c[2 * (int64_t)fk + 1] < 0 && !(exempt && exempt[fk])) return;  // not the first
  }
  double num = 0.0, den = 0.0;
  for (int k = lf0; k < fvs.nfpc; ++k) {
    const int32_t f = c2f[(int64_t)c * fvs.nfpc + k];
    if (f2c[2 * (int64_t)f + 1] >= 0) continue;
    if (exempt && exempt[f]) continue;
    int32_t v[3];
    for (int j = 0; j < fvs.nvpf; ++j) v[j] = cells[(int64_t)c * nvpc + fvs.fv[k][j]];
    int64_t base = 0;
    if (KIND == PHX_PHI_POINTS) base = lower_bound_i32(bfacet_ids, nbf, f) * (int64_t)tab.npts;
    double pn = 0.0, pd = 0.0;
    for (int q = 0; q < tab.npts; ++q) {
      const double p = phi_at<KIND, GDIM>(tab, q, fvs.nvpf, v, phi, x, quad, base);
      pn = pn + p;
      pd = pd + fabs(p);
    }
    num = num + pn;
    den = den + pd;
  }
  const double d = (den > 0.0) ? num / den : 0.5;
  if (d > -1.0 && d < 1.0) tags[c] = (int8_t)(tags[c] | PHX_BCUT_BIT);
}

// --- a4: facet tags; per-facet predicates equivalent to the set algebra of :454-496 ----------
// the tag of one facet from the tags of its two cells; returns how many of the reference's facet sets hold it
__device__ __forceinline__ int facet_rule(int2 cc, const int8_t *__restrict__ ctags, int no_ext, int8_t *tag) {
  const int b0 = (int)(uint8_t)ctags[cc.x];
  const int t0 = b0 & PHX_TAG_MASK;
  const int t1 = cc.y >= 0 ? (ctags[cc.y] & PHX_TAG_MASK) : 0;
  const bool I = (t0 == 1) || (t1 == 1);
  const bool C = (t0 == 2) || (t1 == 2);
  const bool E = (t0 == 3) || (t1 == 3);
  const bool B = cc.y < 0;
  const bool cellcut = (b0 & PHX_BCUT_BIT) != 0;
  const bool CB = B && cellcut;                       // :454-456
  const bool UB = B && !cellcut && !E && !I;          // :457-461
  const bool IB = I && C;                             // :464-466
  bool BF = no_ext ? B : ((E && C) || UB);            // :469-474
  const bool DI = E && I;                             // :476-478
  const bool cut = (C && !(BF || IB || DI || UB)) || CB;  // :480-484
  const bool inte = I && !(IB || BF || DI);           // :487-489
  const bool ext = E && !(IB || BF || DI);            // :492-494
  BF = BF && !cut;                                    // :496
  int8_t t = 0;
  if (ext) t = 5;
  if (inte) t = 1;
  if (IB) t = 3;
  if (cut) t = 2;
  if (BF) t = 4;
  if (DI) t = 6;
  *tag = t;
  return (int)ext + (int)inte + (int)IB + (int)cut + (int)BF + (int)DI;
}

// four facets per thread: two 16-byte loads of f2c, one 4-byte store of the tags (a byte per lane and store left the
// kernel at 1.9 TB/s of its 11 B per facet)
// The kernel that writes the tags also counts them: hist_part[bin][block] (7 bins, summed by k_hist_fold: the separate
// histogram pass re-read 2e8 tag bytes) and, per chunk of PHX_SEL_CHUNK facets, how many facets the two selections every
// assembly starts with will keep (sel0: ghost-penalty facets = tag 2 / 3 with two cells; sel1: tags 3 / 4) -- their
// counting passes re-read the tags (and f2c) as well.  A wavefront covers 256 consecutive facets of ONE chunk.
__global__ void __launch_bounds__(256)
k_tag_facets(int64_t nf, const int32_t *__restrict__ f2c, const int8_t *__restrict__ ctags,
             int no_ext, const uint8_t *__restrict__ exempt, int8_t *__restrict__ ftags,
             unsigned long long *__restrict__ bad, uint32_t *__restrict__ hist_part,
             int32_t *__restrict__ sel0, int32_t *__restrict__ sel1) {
  __shared__ uint32_t lh[4][9];
  const int64_t f0 = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) * 4;
  int nbad = 0;
  int8_t tj[4] = {0x7f, 0x7f, 0x7f, 0x7f};   // 0x7f: no facet here
  bool two[4] = {false, false, false, false};
  if (f0 + 3 < nf) {
    const int4 a = *reinterpret_cast<const int4 *>(f2c + 2 * f0), b = *reinterpret_cast<const int4 *>(f2c + 2 * f0 + 4);
    const int2 cc[4] = {make_int2(a.x, a.y), make_int2(a.z, a.w), make_int2(b.x, b.y), make_int2(b.z, b.w)};
    uint32_t w = 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int8_t t = 0;
      if (!(exempt && exempt[f0 + j])) nbad += facet_rule(cc[j], ctags, no_ext, &t) != 1;   // exempt: cut through the global mesh, no tag
      w |= (uint32_t)(uint8_t)t << (8 * j);
      tj[j] = t;
      two[j] = cc[j].y >= 0;
    }
    *reinterpret_cast<uint32_t *>(ftags + f0) = w;
  } else {
    for (int64_t f = f0; f < nf; ++f) {
      int8_t t = 0;
      const int2 cc = *reinterpret_cast<const int2 *>(f2c + 2 * f);
      if (!(exempt && exempt[f])) nbad += facet_rule(cc, ctags, no_ext, &t) != 1;
      ftags[f] = t;
      tj[f - f0] = t;
      two[f - f0] = cc.y >= 0;
    }
  }
  if (nbad) atomicAdd(bad, (unsigned long long)nbad);
  // ---- counts (wave-uniform: every lane of the block takes part in the ballots)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t cnt[7] = {0, 0, 0, 0, 0, 0, 0}, c0 = 0, c1 = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int b = 0; b < 7; ++b) cnt[b] += (uint32_t)__popcll(__ballot(tj[j] == b));
    c0 += (uint32_t)__popcll(__ballot((tj[j] == 2 || tj[j] == 3) && two[j]));
    c1 += (uint32_t)__popcll(__ballot(tj[j] == 3 || tj[j] == 4));
  }
  if (lane == 0) {
    const int64_t wbase = (blockIdx.x * (int64_t)blockDim.x + (threadIdx.x & ~63)) * 4;   // first facet of the wave
    if (wbase < nf) {
      const int64_t chunk = wbase / PHX_SEL_CHUNK;
      if (c0) atomicAdd(&sel0[chunk], (int32_t)c0);
      if (c1) atomicAdd(&sel1[chunk], (int32_t)c1);
    }
    for (int b = 0; b < 7; ++b) lh[wv][b] = cnt[b];
    lh[wv][7] = c0;   // "bins" 7 / 8: the totals of the two selections
    lh[wv][8] = c1;
  }
  __syncthreads();
  if (threadIdx.x < 9)
    hist_part[(size_t)threadIdx.x * gridDim.x + blockIdx.x] =
        lh[0][threadIdx.x] + lh[1][threadIdx.x] + lh[2][threadIdx.x] + lh[3][threadIdx.x];
}

// hist[b] += sum over blocks of part[b][.]   (grid: slices x bins; hist zeroed by the caller)
__global__ void __launch_bounds__(256) k_hist_fold(int64_t nblocks, const uint32_t *__restrict__ part,
                                                   unsigned long long *__restrict__ hist) {
  __shared__ unsigned long long red[256];
  const int b = blockIdx.y;
  unsigned long long a = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nblocks; i += (int64_t)gridDim.x * blockDim.x)
    a += part[(size_t)b * nblocks + i];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0 && red[0]) atomicAdd(&hist[b], red[0]);
}

// --- a6: (facet, cell) incidences of a one-sided measure -------------------------------------
// key = 2*facet + position of the cell in the reversed link list (mesh_scripts.py:210-213), so a
// host-side sort by key reproduces the reference's first-seen order.
// Facets tagged 3 or 4 are compacted first (ordered select); a count / scan / fill over that short
// list then places the entries without any contended counter (a single global append counter
// serialises: the matches are ~1 per wavefront, 4 ms at 2*10^8 facets).
//   which 0 -> ds(100): facets tagged 4 seen from cells {1,2}   (mesh_scripts.py:619-622)
//   which 1 -> ds(101): facets tagged 3 seen from cells {2,3}   (mesh_scripts.py:623-626)
struct SelTag34 {
  const int8_t *ft;
  __host__ __device__ bool operator()(const int32_t &f) const { return ft[f] == 3 || ft[f] == 4; }
  __host__ __device__ const int8_t *bytes() const { return ft; }
  __host__ __device__ bool test(int t, int32_t) const { return t == 3 || t == 4; }
};

template <bool FILL>
__global__ void k_entities(int64_t nlist, const int32_t *__restrict__ list,
                           const int32_t *__restrict__ f2c, const int32_t *__restrict__ c2f,
                           int nfpc, const int8_t *__restrict__ ctags,
                           const int8_t *__restrict__ ftags, int32_t *__restrict__ cnt0,
                           int32_t *__restrict__ cnt1, const int32_t *__restrict__ off0,
                           const int32_t *__restrict__ off1, int64_t *__restrict__ out0,
                           int64_t *__restrict__ out1) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= nlist) return;
  const int64_t f = list[i];
  const int w = ftags[f] == 4 ? 0 : 1;
  const int cell_mask = w == 0 ? ((1 << 1) | (1 << 2)) : ((1 << 2) | (1 << 3));
  const int32_t c0 = f2c[2 * f], c1 = f2c[2 * f + 1];
  int n = 0;
  for (int pos = 0; pos < 2; ++pos) {
    const int32_t c = c1 >= 0 ? (pos == 0 ? c1 : c0) : (pos == 0 ? c0 : -1);
    if (c < 0) continue;
    const int t = ctags[c] & PHX_TAG_MASK;
    if (t > 30 || !((cell_mask >> t) & 1)) continue;
    if (FILL) {
      int lf = 0;
      for (int k = 0; k < nfpc; ++k)
        if (c2f[(int64_t)c * nfpc + k] == (int32_t)f) lf = k;
      int64_t *out = w == 0 ? out0 : out1;
      const int64_t slot = (w == 0 ? off0[i] : off1[i]) + n;
      out[2 * slot] = 2 * f + pos;
      out[2 * slot + 1] = ((int64_t)c << 8) | lf;
    }
    ++n;
  }
  if (!FILL) { cnt0[i] = w == 0 ? n : 0; cnt1[i] = w == 1 ? n : 0; }
}

// ------------------------------------------------------------------------------------------
static int make_tab(const phx_mesh *m, int degree, int which, DetTab *tab) {
  std::vector<double> t;
  int npts = 0, nfun = 0;
  PHX_CHECK(phx_shape_table(m->cell_type, degree, which, t, &npts, &nfun));
  memset(tab, 0, sizeof(*tab));
  tab->npts = npts;
  tab->nfun = nfun;
  for (size_t i = 0; i < t.size(); ++i) tab->N[i] = t[i];
  return PHX_OK;
}

// Stage the level-set on the device; returns the device pointer (and an owned temp, if any).
static int stage_phi(phx_mesh *m, int phi_kind, const double *phi, int loc, int64_t count,
                     const double **dev, double **owned, Quadric *quad) {
  *owned = nullptr;
  *dev = nullptr;
  memset(quad, 0, sizeof(*quad));
  PHX_REQUIRE(phi != nullptr, PHX_ERR_VALUE, "phi is NULL");
  if (phi_kind == PHX_PHI_QUADRIC) {
    double h[7];
    if (loc == PHX_DEVICE) PHX_HIP(hipMemcpy(h, phi, sizeof(h), hipMemcpyDeviceToHost));
    else memcpy(h, phi, sizeof(h));
    for (int a = 0; a < 3; ++a) { quad->c[a] = h[a]; quad->s[a] = h[3 + a]; }
    quad->c0 = h[6];
    return PHX_OK;
  }
  if (loc == PHX_DEVICE) { *dev = phi; return PHX_OK; }
  PHX_HIP(phx_malloc(owned, sizeof(double) * (size_t)(count > 0 ? count : 1)));
  PHX_HIP(hipMemcpyAsync(*owned, phi, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, m->stream));
  *dev = *owned;
  return PHX_OK;
}

static int64_t cell_points_count(const phx_mesh *m, int degree) {
  DetTab t;
  if (make_tab(m, degree, 0, &t) != PHX_OK) return 0;
  return t.npts;
}

// extra (nullable): `extra_bytes` more device bytes fetched with the same host round trip (flags of the tagging kernels)
static int read_hist(phx_mesh *m, const int8_t *tags, int64_t n, int nbins, int64_t *out, const void *extra = nullptr,
                     size_t extra_bytes = 0, void *extra_host = nullptr) {
  unsigned long long *d = nullptr;
  PHX_HIP(phx_malloc(&d, sizeof(unsigned long long) * 8));
  PHX_HIP(hipMemsetAsync(d, 0, sizeof(unsigned long long) * 8, m->stream));
  const int blocks = (int)std::min<int64_t>(phx_div_up(n, 256), 2048);
  k_tag_hist<<<dim3(blocks), dim3(256), 0, m->stream>>>(n, tags, nbins, d);
  unsigned long long h[8];
  const phx_rb_item rb[2] = {{d, (int)sizeof(h), h}, {extra, (int)extra_bytes, extra_host}};
  PHX_CHECK(phx_read_back(m->stream, rb, extra ? 2 : 1));
  PHX_HIP(phx_free(d));
  for (int i = 0; i < nbins; ++i) out[i] = (int64_t)h[i];
  return PHX_OK;
}

template <int KIND>
static int launch_tag_cells(phx_mesh *m, const DetTab &tab, const double *dphi, const Quadric &q,
                            int *dwarn) {
  const dim3 grid((unsigned)phx_div_up(m->nc, 256)), block(256);
  if (m->cell_type == PHX_TRIANGLE)
    k_tag_cells<KIND, 2, 3><<<grid, block, 0, m->stream>>>(m->nc, tab, m->cells, dphi, m->x, q, m->cell_tags, dwarn);
  else if (m->cell_type == PHX_QUADRILATERAL)
    k_tag_cells<KIND, 2, 4><<<grid, block, 0, m->stream>>>(m->nc, tab, m->cells, dphi, m->x, q, m->cell_tags, dwarn);
  else
    k_tag_cells<KIND, 3, 4><<<grid, block, 0, m->stream>>>(m->nc, tab, m->cells, dphi, m->x, q, m->cell_tags, dwarn);
  PHX_HIP(hipGetLastError());
  return PHX_OK;
}

extern "C" int phx_tag_cells(phx_mesh *m, int phi_kind, const double *phi, int loc,
                             int detection_degree, int single_layer_cut,
                             int *warn_zero_denominator) {
  PHX_HIP(hipSetDevice(m->device));
  PHX_REQUIRE(phi_kind >= 0 && phi_kind <= 2, PHX_ERR_VALUE, "unknown phi_kind %d", phi_kind);
  DetTab tab;
  PHX_CHECK(make_tab(m, detection_degree, 0, &tab));
  const double *dphi = nullptr;
  double *owned = nullptr;
  Quadric quad;
  const int64_t count = phi_kind == PHX_PHI_NODAL_P1 ? m->nv : m->nc * (int64_t)tab.npts;
  PHX_CHECK(stage_phi(m, phi_kind, phi, loc, count, &dphi, &owned, &quad));
  int *dwarn = nullptr;
  PHX_HIP(phx_malloc(&dwarn, sizeof(int)));
  PHX_HIP(hipMemsetAsync(dwarn, 0, sizeof(int), m->stream));
  PHX_CHECK(phx_begin_timing(m));
  if (phi_kind == PHX_PHI_NODAL_P1) PHX_CHECK(launch_tag_cells<PHX_PHI_NODAL_P1>(m, tab, dphi, quad, dwarn));
  else if (phi_kind == PHX_PHI_POINTS) PHX_CHECK(launch_tag_cells<PHX_PHI_POINTS>(m, tab, dphi, quad, dwarn));
  else PHX_CHECK(launch_tag_cells<PHX_PHI_QUADRIC>(m, tab, dphi, quad, dwarn));
  uint8_t *touched = nullptr, *vcut = nullptr;
  const dim3 grid4((unsigned)phx_div_up(phx_div_up(m->nc, 4), 256)), block4(256);   // four cells per thread
  m->act_valid = false;
  if (single_layer_cut) {
    // the two vertex flag arrays stay with the mesh: the P1 assembly numbers its DoFs from them
    if (!m->act_in) PHX_HIP(phx_malloc(&m->act_in, (size_t)m->nv));
    if (!m->act_cut) PHX_HIP(phx_malloc(&m->act_cut, (size_t)m->nv));
    touched = m->act_in;
    vcut = m->act_cut;
    PHX_HIP(hipMemsetAsync(vcut, 0, (size_t)m->nv, m->stream));
    PHX_HIP(hipMemsetAsync(touched, 0, (size_t)m->nv, m->stream));
    if (m->ci.nvpc == 3) k_mark_inside_vertices<3><<<grid4, block4, 0, m->stream>>>(m->nc, m->cells, m->cell_tags, touched);
    else k_mark_inside_vertices<4><<<grid4, block4, 0, m->stream>>>(m->nc, m->cells, m->cell_tags, touched);
  }
  // demotion of isolated cut cells (if asked for) + histogram + per-chunk counts of the cut cells, in one pass
  unsigned long long *dres = nullptr;   // [0..3] histogram, [4] warning flag (low word)
  uint32_t *part = nullptr;
  const int64_t nchunks = phx_div_up(m->nc > 0 ? m->nc : 1, (int64_t)PHX_SEL_CHUNK);
  PHX_HIP(phx_malloc(&dres, sizeof(unsigned long long) * 4));
  PHX_HIP(phx_malloc(&part, sizeof(uint32_t) * 4 * (size_t)grid4.x));
  PHX_HIP(hipMemsetAsync(dres, 0, sizeof(unsigned long long) * 4, m->stream));
  if (!m->sel_counts_cut) PHX_HIP(phx_malloc(&m->sel_counts_cut, sizeof(int32_t) * (size_t)(nchunks + 1)));
  PHX_HIP(hipMemsetAsync(m->sel_counts_cut, 0, sizeof(int32_t) * (size_t)(nchunks + 1), m->stream));
  if (m->ci.nvpc == 3) k_demote_isolated_cut<3><<<grid4, block4, 0, m->stream>>>(m->nc, m->cells, m->cell_tags, touched, part, m->sel_counts_cut, vcut);
  else k_demote_isolated_cut<4><<<grid4, block4, 0, m->stream>>>(m->nc, m->cells, m->cell_tags, touched, part, m->sel_counts_cut, vcut);
  PHX_HIP(hipGetLastError());
  PHX_CHECK(phx_end_timing_mark(m));
  k_hist_fold<<<dim3(64, 4), dim3(256), 0, m->stream>>>((int64_t)grid4.x, part, dres);
  // one host round trip for the histogram, the warning flag and the timing events
  int hwarn = 0;
  unsigned long long hh[4] = {0, 0, 0, 0};
  const phx_rb_item rb[2] = {{dres, (int)sizeof(hh), hh}, {dwarn, (int)sizeof(int), &hwarn}};
  PHX_CHECK(phx_read_back(m->stream, rb, 2));
  for (int i = 0; i < 4; ++i) m->tag_hist[i] = (int64_t)hh[i];
  m->sel_cut_valid = true;
  m->act_valid = single_layer_cut != 0;
  PHX_CHECK(phx_end_timing_read(m, 0));
  PHX_HIP(phx_free(dres));
  PHX_HIP(phx_free(part));
  PHX_HIP(phx_free(dwarn));
  if (owned) PHX_HIP(phx_free(owned));
  if (warn_zero_denominator) *warn_zero_denominator = hwarn;
  m->have_cell_tags = true;
  m->have_facet_tags = false;
  m->sel_counts_valid = false;
  m->have_entities = false;
  return PHX_OK;
}

template <int KIND>
static int launch_bcut(phx_mesh *m, const DetTab &tab, const FacetVerts &fvs, const double *dphi,
                       const Quadric &q) {
  if (m->nbf == 0) return PHX_OK;
  const dim3 grid((unsigned)phx_div_up(m->nbf, 256)), block(256);
  if (m->gdim == 2)
    k_boundary_cell_cut<KIND, 2><<<grid, block, 0, m->stream>>>(m->nbf, tab, fvs, m->ci.nvpc, m->bfacets, m->bfacet_ids, m->cells, m->c2f, m->f2c, dphi, m->x, q, m->facet_exempt, m->cell_tags);
  else
    k_boundary_cell_cut<KIND, 3><<<grid, block, 0, m->stream>>>(m->nbf, tab, fvs, m->ci.nvpc, m->bfacets, m->bfacet_ids, m->cells, m->c2f, m->f2c, dphi, m->x, q, m->facet_exempt, m->cell_tags);
  PHX_HIP(hipGetLastError());
  return PHX_OK;
}

__global__ void k_clear_bcut(int64_t n, int8_t *tags) {
  const int64_t i0 = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) * 4;   // four tags per thread
  if (i0 + 3 < n) {
    uint32_t *w = reinterpret_cast<uint32_t *>(tags + i0);
    *w &= 0x01010101u * (uint32_t)PHX_TAG_MASK;
  } else {
    for (int64_t i = i0; i < n; ++i) tags[i] = (int8_t)(tags[i] & PHX_TAG_MASK);
  }
}

static int run_facet_rule(phx_mesh *m) {
  unsigned long long *dbad = nullptr;   // [0] bad facets, [1..7] the histogram, [8..9] the totals of the two selections
  uint32_t *part = nullptr;
  const int64_t nblocks = phx_div_up(phx_div_up(m->nf, 4), 256);
  const int64_t nchunks = phx_div_up(m->nf > 0 ? m->nf : 1, (int64_t)PHX_SEL_CHUNK);
  PHX_HIP(phx_malloc(&dbad, sizeof(unsigned long long) * 10));
  PHX_HIP(phx_malloc(&part, sizeof(uint32_t) * 9 * (size_t)nblocks));
  PHX_HIP(hipMemsetAsync(dbad, 0, sizeof(unsigned long long) * 10, m->stream));
  for (int w = 0; w < 2; ++w) {
    if (!m->sel_counts[w]) PHX_HIP(phx_malloc(&m->sel_counts[w], sizeof(int32_t) * (size_t)(nchunks + 1)));
    PHX_HIP(hipMemsetAsync(m->sel_counts[w], 0, sizeof(int32_t) * (size_t)(nchunks + 1), m->stream));
  }
  k_tag_facets<<<dim3((unsigned)nblocks), dim3(256), 0, m->stream>>>(
      m->nf, m->f2c, m->cell_tags,
      m->has_exterior_override >= 0 ? (m->has_exterior_override ? 0 : 1) : (m->tag_hist[3] == 0 ? 1 : 0),
      m->facet_exempt, m->facet_tags, dbad, part, m->sel_counts[0], m->sel_counts[1]);
  k_hist_fold<<<dim3(64, 9), dim3(256), 0, m->stream>>>(nblocks, part, dbad + 1);
  PHX_HIP(hipGetLastError());
  PHX_CHECK(phx_end_timing_mark(m));
  unsigned long long hb[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ht[2] = {0, 0};
  const phx_rb_item rb[2] = {{dbad, (int)sizeof(hb), hb}, {dbad + 8, (int)sizeof(ht), ht}};
  PHX_CHECK(phx_read_back(m->stream, rb, 2));
  const unsigned long long bad = hb[0];
  for (int i = 0; i < 7; ++i) m->ftag_hist[i] = (int64_t)hb[1 + i];
  m->sel_total[0] = (int64_t)ht[0];
  m->sel_total[1] = (int64_t)ht[1];
  m->sel_counts_valid = true;
  PHX_CHECK(phx_end_timing_read(m, 1));
  PHX_HIP(phx_free(dbad));
  PHX_HIP(phx_free(part));
  m->have_facet_tags = true;
  m->have_entities = false;
  PHX_REQUIRE(bad == 0, PHX_ERR_PARTITION,
              "%llu facets belong to none or several of the reference's facet sets "
              "(dolfinx MeshTags would reject duplicated entities)", bad);
  return PHX_OK;
}

extern "C" int phx_tag_facets(phx_mesh *m, int phi_kind, const double *phi, int loc,
                              int detection_degree) {
  PHX_HIP(hipSetDevice(m->device));
  PHX_REQUIRE(m->have_cell_tags, PHX_ERR_VALUE, "phx_tag_cells must run before phx_tag_facets");
  PHX_REQUIRE(phi_kind >= 0 && phi_kind <= 2, PHX_ERR_VALUE, "unknown phi_kind %d", phi_kind);
  DetTab tab;
  PHX_CHECK(make_tab(m, detection_degree, 1, &tab));
  FacetVerts fvs;
  fvs.nfpc = m->ci.nfpc;
  fvs.nvpf = m->ci.nvpf;
  for (int f = 0; f < 4; ++f) for (int k = 0; k < 3; ++k) fvs.fv[f][k] = m->ci.fv[f][k];
  const double *dphi = nullptr;
  double *owned = nullptr;
  Quadric quad;
  int64_t count = m->nv;
  const double *src = phi;
  if (phi_kind == PHX_PHI_POINTS) {
    // layout: phi_cells[nc*npts_cell] followed by phi_bfacets[nbf*npts_facet]
    const int64_t skip = m->nc * cell_points_count(m, detection_degree);
    src = phi + skip;
    count = m->nbf * (int64_t)tab.npts;
  }
  PHX_CHECK(stage_phi(m, phi_kind, src, loc, count, &dphi, &owned, &quad));
  PHX_CHECK(phx_begin_timing(m));
  k_clear_bcut<<<dim3((unsigned)phx_div_up(phx_div_up(m->nc, 4), 256)), dim3(256), 0, m->stream>>>(m->nc, m->cell_tags);
  if (phi_kind == PHX_PHI_NODAL_P1) PHX_CHECK(launch_bcut<PHX_PHI_NODAL_P1>(m, tab, fvs, dphi, quad));
  else if (phi_kind == PHX_PHI_POINTS) PHX_CHECK(launch_bcut<PHX_PHI_POINTS>(m, tab, fvs, dphi, quad));
  else PHX_CHECK(launch_bcut<PHX_PHI_QUADRIC>(m, tab, fvs, dphi, quad));
  const int rc = run_facet_rule(m);
  if (owned) PHX_HIP(phx_free(owned));
  return rc;
}

// ------------------------------------------------------------------------------------------
__global__ void k_scatter_tags(int64_t n, const int32_t *__restrict__ idx,
                               const int32_t *__restrict__ val, int8_t *__restrict__ tags,
                               int keep_mask) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) tags[idx[i]] = (int8_t)((tags[idx[i]] & keep_mask) | val[i]);
}

extern "C" int phx_overwrite_tags(phx_mesh *m, int entity_is_facet, int64_t n,
                                  const int32_t *indices, const int32_t *values) {
  PHX_HIP(hipSetDevice(m->device));
  PHX_REQUIRE(entity_is_facet ? m->have_facet_tags : m->have_cell_tags, PHX_ERR_VALUE,
              "tags must be computed before they can be overwritten");
  const int64_t nent = entity_is_facet ? m->nf : m->nc;
  for (int64_t i = 0; i < n; ++i) {
    const int32_t v = values[i];
    if (entity_is_facet) {
      // mesh_scripts.py:613-614
      PHX_REQUIRE(!((v >= 1 && v <= 6) || v == 100 || v == 101), PHX_ERR_VALUE,
                  "Cannot overwrite facets tags with values 1, 2, 3, 4, 5, 6, 100 or 101.");
    } else {
      // mesh_scripts.py:608-609
      PHX_REQUIRE(!(v >= 1 && v <= 3), PHX_ERR_VALUE,
                  "Cannot overwrite cells tags with values 1, 2 or 3.");
    }
    PHX_REQUIRE(v >= 0 && v <= 127, PHX_ERR_NOT_IMPLEMENTED, "user tag values must lie in 0..127");
    PHX_REQUIRE(indices[i] >= 0 && indices[i] < nent, PHX_ERR_VALUE, "entity index out of range");
  }
  if (n == 0) return PHX_OK;
  int32_t *di = nullptr, *dv = nullptr;
  PHX_HIP(phx_malloc(&di, sizeof(int32_t) * (size_t)n));
  PHX_HIP(phx_malloc(&dv, sizeof(int32_t) * (size_t)n));
  PHX_HIP(hipMemcpyAsync(di, indices, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, m->stream));
  PHX_HIP(hipMemcpyAsync(dv, values, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, m->stream));
  k_scatter_tags<<<dim3((unsigned)phx_div_up(n, 256)), dim3(256), 0, m->stream>>>(
      n, di, dv, entity_is_facet ? m->facet_tags : m->cell_tags, entity_is_facet ? 0 : PHX_BCUT_BIT);
  PHX_HIP(hipStreamSynchronize(m->stream));
  PHX_HIP(phx_free(di));
  PHX_HIP(phx_free(dv));
  if (entity_is_facet) PHX_CHECK(read_hist(m, m->facet_tags, m->nf, 7, m->ftag_hist));
  else PHX_CHECK(read_hist(m, m->cell_tags, m->nc, 4, m->tag_hist));
  if (entity_is_facet) m->sel_counts_valid = false;
  else { m->sel_cut_valid = false; m->act_valid = false; }
  m->have_entities = false;
  return PHX_OK;
}

__global__ void k_narrow_tags(int64_t n, const int32_t *__restrict__ in, int8_t *__restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = (int8_t)(in[i] & PHX_TAG_MASK);
}

extern "C" int phx_set_tags(phx_mesh *m, int entity_is_facet, const int32_t *values, int loc) {
  PHX_HIP(hipSetDevice(m->device));
  const int64_t n = entity_is_facet ? m->nf : m->nc;
  const int32_t *src = values;
  int32_t *tmp = nullptr;
  if (loc != PHX_DEVICE) {
    PHX_HIP(phx_malloc(&tmp, sizeof(int32_t) * (size_t)n));
    PHX_HIP(hipMemcpyAsync(tmp, values, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, m->stream));
    src = tmp;
  }
  k_narrow_tags<<<dim3((unsigned)phx_div_up(n, 256)), dim3(256), 0, m->stream>>>(
      n, src, entity_is_facet ? m->facet_tags : m->cell_tags);
  PHX_HIP(hipStreamSynchronize(m->stream));
  if (tmp) PHX_HIP(phx_free(tmp));
  if (entity_is_facet) {
    PHX_CHECK(read_hist(m, m->facet_tags, m->nf, 7, m->ftag_hist));
    m->have_facet_tags = true;
    m->sel_counts_valid = false;
  } else {
    PHX_CHECK(read_hist(m, m->cell_tags, m->nc, 4, m->tag_hist));
    m->have_cell_tags = true;
    m->sel_cut_valid = false;
    m->act_valid = false;
  }
  m->have_entities = false;
  return PHX_OK;
}

// ------------------------------------------------------------------------------------------
// Integration entities: device collection (unordered), cached per tag state.
int phx_collect_entities(phx_mesh *m) {
  if (m->have_entities) return PHX_OK;
  PHX_REQUIRE(m->have_cell_tags && m->have_facet_tags, PHX_ERR_VALUE,
              "cell and facet tags must be computed first");
  hipStream_t st = m->stream;
  const int64_t nmax = m->ftag_hist[3] + m->ftag_hist[4];
  for (int w = 0; w < 2; ++w) {
    if (m->ent_buf[w]) { PHX_HIP(phx_free(m->ent_buf[w])); m->ent_buf[w] = nullptr; }
    m->ent_count[w] = 0;
  }
  if (nmax == 0) {
    for (int w = 0; w < 2; ++w) PHX_HIP(phx_malloc(&m->ent_buf[w], 16));
    m->have_entities = true;
    return PHX_OK;
  }
  int32_t *list = nullptr, *cnt = nullptr, *off = nullptr;
  std::vector<void *> later;   // freed behind the synchronisation at the end
  {
    int64_t nsel = 0;
    PHX_CHECK(phx_select_indices(st, m->nf, SelTag34{m->facet_tags}, &list, &nsel, &later,
                                 m->sel_counts_valid ? m->sel_counts[1] : nullptr,
                                 m->sel_counts_valid ? m->sel_total[1] : -1));
    PHX_REQUIRE(nsel == nmax, PHX_ERR_VALUE, "facet tag histogram and selection disagree");
  }
  PHX_HIP(phx_malloc(&cnt, sizeof(int32_t) * 2 * (size_t)(nmax + 1)));
  PHX_HIP(phx_malloc(&off, sizeof(int32_t) * 2 * (size_t)(nmax + 1)));
  PHX_HIP(hipMemsetAsync(cnt, 0, sizeof(int32_t) * 2 * (size_t)(nmax + 1), st));
  int32_t *cnt0 = cnt, *cnt1 = cnt + (nmax + 1), *off0 = off, *off1 = off + (nmax + 1);
  const dim3 grid((unsigned)phx_div_up(nmax, 256)), block(256);
  k_entities<false><<<grid, block, 0, st>>>(nmax, list, m->f2c, m->c2f, m->ci.nfpc, m->cell_tags,
                                            m->facet_tags, cnt0, cnt1, nullptr, nullptr, nullptr, nullptr);
  {
    // both scans, then ONE host round trip for the two totals
    int32_t tot[2] = {0, 0};
    size_t bytes = 0;
    PHX_HIP(phx_exclusive_sum(nullptr, bytes, cnt0, off0, (size_t)(nmax + 1), st));
    void *tmp = nullptr;
    PHX_HIP(phx_malloc(&tmp, bytes ? bytes : 16));
    for (int w = 0; w < 2; ++w) {
      int32_t *ci = w == 0 ? cnt0 : cnt1, *oi = w == 0 ? off0 : off1;
      PHX_HIP(phx_exclusive_sum(tmp, bytes, ci, oi, (size_t)(nmax + 1), st));
    }
    const phx_rb_item rb[2] = {{off0 + nmax, 4, &tot[0]}, {off1 + nmax, 4, &tot[1]}};
    PHX_CHECK(phx_read_back(st, rb, 2));
    PHX_HIP(phx_free(tmp));
    for (int w = 0; w < 2; ++w) {
      m->ent_count[w] = tot[w];
      PHX_HIP(phx_malloc(&m->ent_buf[w], sizeof(int64_t) * 2 * (size_t)(tot[w] > 0 ? tot[w] : 1)));
    }
  }
  k_entities<true><<<grid, block, 0, st>>>(nmax, list, m->f2c, m->c2f, m->ci.nfpc, m->cell_tags,
                                           m->facet_tags, nullptr, nullptr, off0, off1,
                                           m->ent_buf[0], m->ent_buf[1]);
  PHX_HIP(hipGetLastError());
  PHX_HIP(hipStreamSynchronize(st));
  for (void *q : later) PHX_HIP(phx_free(q));
  PHX_HIP(phx_free(list)); PHX_HIP(phx_free(cnt)); PHX_HIP(phx_free(off));
  m->have_entities = true;
  return PHX_OK;
}

extern "C" int phx_integration_entities(phx_mesh *m, int which, int32_t *out, int64_t *n_pairs) {
  PHX_HIP(hipSetDevice(m->device));
  PHX_REQUIRE(which == 100 || which == 101, PHX_ERR_VALUE, "which must be 100 or 101");
  PHX_CHECK(phx_collect_entities(m));
  const int w = which == 100 ? 0 : 1;
  const int64_t n = m->ent_count[w];
  *n_pairs = n;
  if (!out || n == 0) return PHX_OK;
  std::vector<int64_t> h(2 * (size_t)n);
  PHX_HIP(hipMemcpy(h.data(), m->ent_buf[w], sizeof(int64_t) * 2 * (size_t)n, hipMemcpyDeviceToHost));
  struct Rec { int64_t key; int32_t cell; int32_t lf; int64_t first; };
  std::vector<Rec> r((size_t)n);
  for (int64_t i = 0; i < n; ++i) {
    r[i].key = h[2 * i];
    r[i].cell = (int32_t)(h[2 * i + 1] >> 8);
    r[i].lf = (int32_t)(h[2 * i + 1] & 0xff);
  }
  // first-seen order of the cells (np.unique(..., return_index) at mesh_scripts.py:168-170),
  // then ascending local facet inside a cell (:173-185)
  std::sort(r.begin(), r.end(), [](const Rec &a, const Rec &b) {
    return a.cell != b.cell ? a.cell < b.cell : a.key < b.key;
  });
  for (size_t i = 0; i < r.size();) {
    size_t j = i;
    while (j < r.size() && r[j].cell == r[i].cell) { r[j].first = r[i].key; ++j; }
    i = j;
  }
  std::sort(r.begin(), r.end(), [](const Rec &a, const Rec &b) {
    return a.first != b.first ? a.first < b.first : a.lf < b.lf;
  });
  for (int64_t i = 0; i < n; ++i) { out[2 * i] = r[i].cell; out[2 * i + 1] = r[i].lf; }
  return PHX_OK;
}

#include "phx_levelset.inc.hip"
