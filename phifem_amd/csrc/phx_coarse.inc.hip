// Coarse-space correction for the 5-field interface-elasticity system (included by phx_solve.hip, after the
// vertex-block Jacobi it complements).
//
// The vertex blocks (phx_blockjac.inc.hip) settle the penalty couplings of the band, but nothing in them sees the
// smooth displacement modes of the bulk: the count grows with the box (280 iterations at 96^3, 488 at 160^3, 624 at
// 256^3).  Two-level additive Schwarz:   P = D ( B^-1 + R Ac^-1 R^T ),   Ac = R^T A R   (Galerkin),
// R = trilinear functions on a coarse lattice of spacing H = ratio * h, one set per displacement block (u_in[a],
// u_out[a]) restricted to the active, unconstrained DoFs of that block; the y / p fields have no coarse part.
// CPU prototype against the oracle matrices (tools/experiments/elasticity_coarse.py, E_out = 1e-3): 90 / 98 / 129 / 180
// iterations with the vertex blocks alone for n = 16 / 24 / 32 / 48 become 81 / 83 / 89 / 86 with H = 4h, 92 at
// n = 48 with H = 8h, 140 with H = 16h: the count follows H / h, not n.
//   * Ac is found by PROBING: coarse functions of one block whose nodes are 3 apart in every axis have disjoint
//     images under A, so 27 colours x 2 D blocks SpMVs with the solver's own operator give every column;
//   * Ac is dense and small (<= ~17 000: the ratio is chosen for that), inverted once per system by the library's own
//     blocked Gauss-Jordan elimination (phx_dense.inc.hip; rounds 2-3: rocSOLVER through dlopen) and applied as a
//     matrix-vector product;
//   * R^T and R run as three 1-D passes over lattice lines (no atomics: bit-reproducible).
// Partitioned boxes (native RCCL loop, phx_dist.inc.hip): the coarse lattice is the one of the GLOBAL box; every rank
// restricts its owned rows, the coarse right-hand side is all-reduced (a few thousand doubles per application), every
// rank holds Ac^-1 (its own rows' share of Ac summed over the ranks once per system) and prolongs onto its owned rows.
#include <functional>

struct phx_coarse {
  int d = 3, nblk_u = 6, ratio = 16;
  bool dist = false;           // partitioned box: restriction and prolongation are two phases with an all-reduce between
  int off[3] = {0, 0, 0};      // lattice offset of this rank's vertices in the global box
  int64_t nf[3] = {1, 1, 1};   // fine vertices per axis (this rank)
  int m[3] = {1, 1, 1};        // coarse nodes per axis
  int64_t M = 1;               // coarse nodes per block
  int nc = 0;                  // compact coarse DoFs
  int32_t *cpos = nullptr;     // [nblk_u * nv] solver position of an eligible fine DoF, else -1
  double *dpos = nullptr;      // [n] D by solver position
  int32_t *cmap = nullptr;     // [nblk_u * M] compact index or -1
  int32_t *node_of = nullptr;  // [nc] block * M + node
  double *Ainv = nullptr;      // [nc * nc] row-major Ac^-1
  double *X1 = nullptr, *X2 = nullptr, *X3 = nullptr;   // line-restricted arrays ([blk][k][j][ci], [blk][k][cj][ci], [blk][ck][cj][ci])
  double *gc = nullptr, *zc = nullptr;                  // [nc]
};

static void coarse_free(phx_coarse *c) {
  if (!c) return;
  (void)phx_free(c->cpos); (void)phx_free(c->dpos); (void)phx_free(c->cmap); (void)phx_free(c->node_of); (void)phx_free(c->Ainv);
  (void)phx_free(c->X1); (void)phx_free(c->X2); (void)phx_free(c->X3); (void)phx_free(c->gc); (void)phx_free(c->zc);
  delete c;
}
void phx_coarse_destroy(phx_coarse *c) { coarse_free(c); }

// ---- set-up kernels ---------------------------------------------------------------------------------------------
// eligible fine DoFs: active rows of the displacement blocks that are not Dirichlet rows (a constrained u_in row is the
// unit row `1 * u = u_D`: a single entry on the diagonal)
__global__ void k_cc_positions(int64_t n, int64_t nv, int nblk_u, int d, const int64_t *__restrict__ full_of_active,
                               const int64_t *__restrict__ rowptr, const int32_t *__restrict__ iperm,
                               const double *__restrict__ diag, const uint8_t *__restrict__ own, int32_t *__restrict__ cpos,
                               double *__restrict__ dpos) {
  const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r >= n) return;
  const int32_t pos = iperm[r];
  dpos[pos] = diag[r];
  if (own && !own[pos]) return;   // a row another rank owns
  const int64_t f = full_of_active[r];
  const int blk = (int)(f / nv);
  if (blk >= nblk_u) return;
  if (blk < d && rowptr[r + 1] - rowptr[r] == 1) return;
  cpos[f] = pos;
}

struct CcDims {
  int64_t nf[3];
  int m[3];
  int ratio;
  int64_t M;
  int off[3];
};
// coarse node and weight of the two hat functions that cover fine index i along an axis with `m` coarse nodes
__device__ __forceinline__ void cc_axis(int64_t i, int ratio, int m, int c[2], double w[2]) {
  int c0 = (int)(i / ratio);
  if (c0 > m - 2) c0 = m - 2 > 0 ? m - 2 : 0;
  const double t = (double)i / (double)ratio - (double)c0;
  c[0] = c0; c[1] = c0 + 1 < m ? c0 + 1 : c0;
  w[0] = 1.0 - t; w[1] = c0 + 1 < m ? t : 0.0;
}

__global__ void k_cc_used(int64_t total, int64_t nv, CcDims g, const int32_t *__restrict__ cpos, uint8_t *__restrict__ used) {
  const int64_t f = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (f >= total || cpos[f] < 0) return;
  const int blk = (int)(f / nv);
  const int64_t v = f % nv;
  const int64_t i[3] = {v % g.nf[0], (v / g.nf[0]) % g.nf[1], v / (g.nf[0] * g.nf[1])};
  int c[3][2];
  double w[3][2];
  for (int a = 0; a < 3; ++a) cc_axis(i[a] + g.off[a], g.ratio, g.m[a], c[a], w[a]);
  for (int q = 0; q < 8; ++q) {
    const double ww = w[0][q & 1] * w[1][(q >> 1) & 1] * w[2][q >> 2];
    if (ww > 0.0) used[blk * g.M + c[0][q & 1] + (int64_t)g.m[0] * (c[1][(q >> 1) & 1] + (int64_t)g.m[1] * c[2][q >> 2])] = 1;
  }
}

// probing vector of (block, colour): D R e, e = the sum of the coarse functions of the block whose node has the colour
__global__ void k_cc_probe(int64_t nv, int blk, int col3, CcDims g, const int32_t *__restrict__ cpos,
                           const double *__restrict__ dpos, double *__restrict__ wvec) {
  const int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (v >= nv) return;
  const int32_t pos = cpos[(int64_t)blk * nv + v];
  if (pos < 0) return;
  const int64_t i[3] = {v % g.nf[0], (v / g.nf[0]) % g.nf[1], v / (g.nf[0] * g.nf[1])};
  const int cc[3] = {col3 % 3, (col3 / 3) % 3, col3 / 9};
  double s[3];
  for (int a = 0; a < 3; ++a) {
    int c[2];
    double w[2];
    cc_axis(i[a] + g.off[a], g.ratio, g.m[a], c, w);
    s[a] = (c[0] % 3 == cc[a] ? w[0] : 0.0) + ((c[1] != c[0] && c[1] % 3 == cc[a]) ? w[1] : 0.0);
  }
  wvec[pos] = dpos[pos] * s[0] * s[1] * s[2];
}

// ---- R^T in three line passes -----------------------------------------------------------------------------------
__device__ __forceinline__ int cc_part(int f, int co, int colour, int ratio);
// X1[line][ci] = sum_i w(i, ci) vin[cpos[line][i]]  (line = (blk, k, j)); SPLIT: X1s[line][ci][part], see below.
// One wavefront per line: the line's values are staged in LDS with coalesced loads, then lane ci sums its 2 H - 1
// entries in index order (a thread per output with its own dependent gathers: 75 us instead of 20 at 96^3).
// DIRECT: vin is in full lattice order (the probing images) instead of solver order.
template <bool SPLIT, bool DIRECT>
__global__ void __launch_bounds__(256)
k_cc_restrict_x(int64_t nlines, CcDims g, int colour, const int32_t *__restrict__ cpos, const double *__restrict__ vin,
                double *__restrict__ X1) {
  extern __shared__ double cc_xs[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t line = blockIdx.x * (int64_t)(blockDim.x >> 6) + w;
  if (line >= nlines) return;
  double *xs = cc_xs + (int64_t)w * g.nf[0];
  const int64_t base = line * g.nf[0];
  for (int64_t i = lane; i < g.nf[0]; i += 64) {
    const int32_t p = cpos[base + i];
    xs[i] = p >= 0 ? vin[DIRECT ? base + i : (int64_t)p] : 0.0;
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const double inv_h = 1.0 / (double)g.ratio;
  const int nf0 = (int)g.nf[0];
  for (int ci = lane; ci < g.m[0]; ci += 64) {
    // the hat of node ci: 1 - |i - ci H| / H on (ci - 1) H < i < (ci + 1) H
    const int centre = ci * g.ratio - g.off[0];   // in this rank's indices
    const int lo = centre - g.ratio + 1 > 0 ? centre - g.ratio + 1 : 0;
    const int hi = centre + g.ratio - 1 < nf0 - 1 ? centre + g.ratio - 1 : nf0 - 1;
    double acc[2] = {0.0, 0.0};
    for (int i = lo; i <= hi; ++i) {
      const double wt = 1.0 - fabs((double)(i - centre)) * inv_h;
      acc[SPLIT ? cc_part(i + g.off[0], ci, colour, g.ratio) : 0] += wt * xs[i];
    }
    if (SPLIT) { X1[2 * (line * g.m[0] + ci)] = acc[0]; X1[2 * (line * g.m[0] + ci) + 1] = acc[1]; }
    else X1[line * g.m[0] + ci] = acc[0];
  }
}
// generic middle pass: out[o][co][in] = sum_f w(f, co) in[o][f][in]   (axis of fine length nfa -> ma coarse nodes)
__global__ void __launch_bounds__(256)
k_cc_restrict_axis(int64_t total, int64_t inner, int64_t nfa, int ma, int ratio, int off, const double *__restrict__ in,
                   double *__restrict__ out) {
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int64_t q = t % inner, o = t / (inner * ma);
  const int co = (int)((t / inner) % ma), centre = co * ratio - off;
  const int lo = centre - ratio + 1 > 0 ? centre - ratio + 1 : 0;
  const int hi = centre + ratio - 1 < (int)nfa - 1 ? centre + ratio - 1 : (int)nfa - 1;
  const double inv_h = 1.0 / (double)ratio;
  double acc = 0.0;
  for (int f = lo; f <= hi; ++f) acc += (1.0 - fabs((double)(f - centre)) * inv_h) * in[(o * nfa + f) * inner + q];
  out[t] = acc;
}
__global__ void k_cc_compact(int nc, const int32_t *__restrict__ node_of, const double *__restrict__ X3, double *__restrict__ gc) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nc) gc[i] = X3[node_of[i]];
}
__global__ void k_cc_expand(int64_t total, const int32_t *__restrict__ cmap, const double *__restrict__ zc, double *__restrict__ X3) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < total) X3[i] = cmap[i] >= 0 ? zc[cmap[i]] : 0.0;
}
// middle pass of R: out[o][f][in] = sum_{two co} w(f, co) in[o][co][in]
__global__ void __launch_bounds__(256)
k_cc_prolong_axis(int64_t total, int64_t inner, int64_t nfa, int ma, int ratio, int off, const double *__restrict__ in,
                  double *__restrict__ out) {
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int64_t q = t % inner, f = (t / inner) % nfa, o = t / (inner * nfa);
  int c[2];
  double w[2];
  cc_axis(f + off, ratio, ma, c, w);
  out[t] = w[0] * in[(o * ma + c[0]) * inner + q] + w[1] * in[(o * ma + c[1]) * inner + q];
}
// last pass of R, fused with the scaling and the sum: vout[pos] += D[pos] * sum_{two ci} w X1[blk][k][j][ci]
__global__ void __launch_bounds__(256)
k_cc_prolong_x_add(int64_t total, CcDims g, const int32_t *__restrict__ cpos, const double *__restrict__ dpos,
                   const double *__restrict__ X1, double *__restrict__ vout) {
  const int64_t f = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (f >= total) return;
  const int32_t p = cpos[f];
  if (p < 0) return;
  const int64_t i = f % g.nf[0], line = f / g.nf[0];
  int c[2];
  double w[2];
  cc_axis(i + g.off[0], g.ratio, g.m[0], c, w);
  vout[p] += dpos[p] * (w[0] * X1[line * g.m[0] + c[0]] + w[1] * X1[line * g.m[0] + c[1]]);
}

// ---- probing, all colours of one block in ONE pass over the matrix -------------------------------------------------
// T[c][f] = (A R e_c)(f) for the eligible fine rows f (full lattice order), c = the 9 colours (cx, cy) of block bj with
// the z colour `cz` (nine images at a time: 7 GB at 256^3; all 27 at once cost 22 GB and pushed the pool over its limit).
// Reads the CSR copy (unscaled A, active numbering) once: the probing vectors are analytic -- the value of
// R e_c at a column is a product of per-axis hat weights of its lattice position -- so they are evaluated per entry
// and never stored.  16 lanes per row.  (162 products with the solver's SpMV: 0.66 s at 256^3; this: eighteen passes of ~7 ms.)
__global__ void __launch_bounds__(256)
k_cc_probe_rows(int64_t n, int64_t nv, int bj, int nblk_u, CcDims g, int cz, const int64_t *__restrict__ rowptr,
                const int32_t *__restrict__ col, const double *__restrict__ val, const int32_t *__restrict__ f32,
                const int32_t *__restrict__ cpos, double *__restrict__ T, int64_t tf) {
  const int l16 = threadIdx.x & 15;
  const int64_t r = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 4;
  if (r >= n) return;
  const int32_t fr = f32[r];
  if (fr / nv >= nblk_u || cpos[fr] < 0) return;   // uniform over the 16 lanes of a row
  const int nf0 = (int)g.nf[0], nf1 = (int)g.nf[1];
  const int32_t lo = (int32_t)(bj * nv), hi = (int32_t)((bj + 1) * nv);
  const double inv_h = 1.0 / (double)g.ratio;
  double acc[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) acc[c] = 0.0;
  for (int64_t q = rowptr[r] + l16; q < rowptr[r + 1]; q += 16) {
    const int32_t fc = f32[col[q]];
    if (fc < lo || fc >= hi) continue;
    const int v = fc - lo;
    const int idx[3] = {v % nf0, (v / nf0) % nf1, v / (nf0 * nf1)};
    double sw[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int gi = idx[a] + g.off[a];
      int c0 = gi / g.ratio;
      if (c0 > g.m[a] - 2) c0 = g.m[a] - 2 > 0 ? g.m[a] - 2 : 0;
      const double t = (double)(gi - c0 * g.ratio) * inv_h;
      const int k0 = c0 % 3, k1 = (c0 + 1) % 3;
      const double w1 = c0 + 1 < g.m[a] ? t : 0.0;
      sw[a][0] = (k0 == 0 ? 1.0 - t : 0.0) + (k1 == 0 ? w1 : 0.0);
      sw[a][1] = (k0 == 1 ? 1.0 - t : 0.0) + (k1 == 1 ? w1 : 0.0);
      sw[a][2] = (k0 == 2 ? 1.0 - t : 0.0) + (k1 == 2 ? w1 : 0.0);
    }
    const double az = val[q] * (cz == 0 ? sw[2][0] : (cz == 1 ? sw[2][1] : sw[2][2]));
#pragma unroll
    for (int cy = 0; cy < 3; ++cy) {
      const double ayz = az * sw[1][cy];
#pragma unroll
      for (int cx = 0; cx < 3; ++cx) acc[cx + 3 * cy] = __builtin_fma(ayz, sw[0][cx], acc[cx + 3 * cy]);
    }
  }
#pragma unroll
  for (int c = 0; c < 9; ++c) {
    double x = acc[c];
    x += __shfl_xor(x, 8); x += __shfl_xor(x, 4); x += __shfl_xor(x, 2); x += __shfl_xor(x, 1);
    if (l16 == 0) T[(int64_t)c * tf + fr] = x;
  }
}
__global__ void k_cc_f32(int64_t n, const int64_t *__restrict__ full_of_active, int32_t *__restrict__ f32) {
  const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r < n) f32[r] = (int32_t)full_of_active[r];
}

// ---- probing passes: the same three line passes, but every partial sum is kept apart by the probe function it
// belongs to.  Same-colour nodes are 3 H apart; the images A phi_J of two of them are disjoint sets of fine rows as
// long as H > 4 h (the operator reaches two vertices), and a test function phi_I (support I H +- H) sees at most two
// of them per axis -- split by the plane half a cell beyond its own node on the side of the farther one:
//   delta = (ci - colour) mod 3 = 0: everything belongs to node ci;
//   delta = 1: rows below ci H + H/2 belong to node ci - 1, the rest to node ci + 2;
//   delta = 2: rows below ci H - H/2 belong to node ci - 2, the rest to node ci + 1.
// Two parts per axis, eight per coarse row: every entry of Ac up to two cells away comes out exactly.
__device__ __forceinline__ int cc_part(int f, int co, int colour, int ratio) {
  const int delta = ((co - colour) % 3 + 3) % 3;
  if (delta == 0) return 0;
  const int twice_split = 2 * co * ratio + (delta == 1 ? ratio : -ratio);
  return 2 * f < twice_split ? 0 : 1;
}
__device__ __forceinline__ int cc_part_target(int co, int colour, int part) {
  const int delta = ((co - colour) % 3 + 3) % 3;
  if (delta == 0) return part == 0 ? co : -1;
  if (delta == 1) return part == 0 ? co - 1 : co + 2;
  return part == 0 ? co - 2 : co + 1;
}
// out[((o ma + co) inner + q) 2 + part] = sum_f [part(f) == part] w(f, co) in[(o nfa + f) inner + q]
__global__ void __launch_bounds__(256)
k_cc_restrict_axis_split(int64_t total, int64_t inner, int64_t nfa, int ma, int ratio, int off, int colour,
                         const double *__restrict__ in, double *__restrict__ out) {
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int64_t q = t % inner, o = t / (inner * ma);
  const int co = (int)((t / inner) % ma), centre = co * ratio - off;
  const int lo = centre - ratio + 1 > 0 ? centre - ratio + 1 : 0;
  const int hi = centre + ratio - 1 < (int)nfa - 1 ? centre + ratio - 1 : (int)nfa - 1;
  const double inv_h = 1.0 / (double)ratio;
  double acc[2] = {0.0, 0.0};
  for (int f = lo; f <= hi; ++f)
    acc[cc_part(f + off, co, colour, ratio)] += (1.0 - fabs((double)(f - centre)) * inv_h) * in[(o * nfa + f) * inner + q];
  out[2 * t] = acc[0];
  out[2 * t + 1] = acc[1];
}
// row I of Ac from the eight parts of its node: X3s[((blk m2 + ck) (m1 m0 4) + (cj m0 + ci) 4 + px 2 + py) 2 + pz]
__global__ void k_cc_store_parts(int nc, int bj, int col3, CcDims g, const int32_t *__restrict__ node_of,
                                 const int32_t *__restrict__ cmap, const double *__restrict__ X3s, double *__restrict__ buf) {
  const int I = blockIdx.x * blockDim.x + threadIdx.x;
  if (I >= nc) return;
  const int64_t blk = node_of[I] / g.M, node = node_of[I] % g.M;
  const int ci[3] = {(int)(node % g.m[0]), (int)((node / g.m[0]) % g.m[1]), (int)(node / ((int64_t)g.m[0] * g.m[1]))};
  const int cc[3] = {col3 % 3, (col3 / 3) % 3, col3 / 9};
  const int64_t inner = (int64_t)g.m[1] * g.m[0] * 4;
  const int64_t b0 = ((blk * g.m[2] + ci[2]) * inner + ((int64_t)ci[1] * g.m[0] + ci[0]) * 4) * 2;
  for (int q = 0; q < 8; ++q) {
    const int px = q >> 2, py = (q >> 1) & 1, pz = q & 1;
    const int tx = cc_part_target(ci[0], cc[0], px), ty = cc_part_target(ci[1], cc[1], py), tz = cc_part_target(ci[2], cc[2], pz);
    if (tx < 0 || tx >= g.m[0] || ty < 0 || ty >= g.m[1] || tz < 0 || tz >= g.m[2]) continue;
    const int32_t J = cmap[bj * g.M + tx + (int64_t)g.m[0] * (ty + (int64_t)g.m[1] * tz)];
    if (J >= 0) buf[(int64_t)J + (int64_t)nc * I] = X3s[b0 + q];
  }
}

// column of Ac found by the probe of (block bj, colour) -- the LUMPED variant (PHX_EL_COARSE_LUMPED=1, A/B aid): row I takes the entry of the one node of that colour within
// one coarse cell of its own node.  The buffer holds Ac TRANSPOSED in column-major order = Ac row-major ... of the
// transpose: buf[J + nc * I] = Ac[I][J], so that the column-major inverse rocSOLVER leaves is Ac^-1 in row-major order.
__global__ void k_cc_store_column(int nc, int bj, int col3, CcDims g, const int32_t *__restrict__ node_of,
                                  const int32_t *__restrict__ cmap, const double *__restrict__ gc, double *__restrict__ buf) {
  const int I = blockIdx.x * blockDim.x + threadIdx.x;
  if (I >= nc) return;
  const int64_t node = node_of[I] % g.M;
  const int ci[3] = {(int)(node % g.m[0]), (int)((node / g.m[0]) % g.m[1]), (int)(node / ((int64_t)g.m[0] * g.m[1]))};
  const int cc[3] = {col3 % 3, (col3 / 3) % 3, col3 / 9};
  int cj[3];
  for (int a = 0; a < 3; ++a) {
    int found = -1;
    for (int dlt = -1; dlt <= 1; ++dlt) {
      const int q = ci[a] + dlt;
      if (q >= 0 && q < g.m[a] && q % 3 == cc[a]) found = q;
    }
    if (found < 0) return;
    cj[a] = found;
  }
  const int32_t J = cmap[bj * g.M + cj[0] + (int64_t)g.m[0] * (cj[1] + (int64_t)g.m[1] * cj[2])];
  if (J >= 0) buf[(int64_t)J + (int64_t)nc * I] = gc[I];
}

// zc = Ainv gc, one wavefront per row (fixed summation order)
__global__ void __launch_bounds__(256) k_cc_gemv(int nc, const double *__restrict__ Ainv, const double *__restrict__ x, double *__restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= nc) return;
  const double *a = Ainv + (int64_t)row * nc;
  double acc = 0.0;
  for (int j = lane; j < nc; j += 64) acc = __builtin_fma(a[j], x[j], acc);
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) y[row] = acc;
}

static inline dim3 cc_grid(int64_t n) { return dim3((unsigned)phx_div_up(std::max<int64_t>(n, 1), 256)); }

// gc = R^T vin (compact coarse vector)
static int coarse_restrict(phx_system *s, phx_coarse *c, const double *vin) {
  hipStream_t st = s->mesh->stream;
  if (s->n == 0) { PHX_HIP(hipMemsetAsync(c->gc, 0, sizeof(double) * (size_t)c->nc, st)); return PHX_OK; }
  const CcDims g{{c->nf[0], c->nf[1], c->nf[2]}, {c->m[0], c->m[1], c->m[2]}, c->ratio, c->M, {c->off[0], c->off[1], c->off[2]}};
  const int64_t nb = c->nblk_u;
  const int64_t nl = nb * c->nf[2] * c->nf[1];
  k_cc_restrict_x<false, false><<<dim3((unsigned)phx_div_up(nl, 4)), dim3(256), sizeof(double) * 4 * (size_t)c->nf[0], st>>>(nl, g, 0, c->cpos, vin, c->X1);
  const int64_t t2 = nb * c->nf[2] * c->m[1] * c->m[0];
  k_cc_restrict_axis<<<cc_grid(t2), dim3(256), 0, st>>>(t2, c->m[0], c->nf[1], c->m[1], c->ratio, c->off[1], c->X1, c->X2);
  const int64_t t3 = nb * c->M;
  k_cc_restrict_axis<<<cc_grid(t3), dim3(256), 0, st>>>(t3, (int64_t)c->m[0] * c->m[1], c->nf[2], c->m[2], c->ratio, c->off[2], c->X2, c->X3);
  k_cc_compact<<<cc_grid(c->nc), dim3(256), 0, st>>>(c->nc, c->node_of, c->X3, c->gc);
  PHX_HIP(hipGetLastError());
  return PHX_OK;
}

// the eight parts of R^T vin for the probe of colour `col3` (X1s / X2s / X3s: scratch of 2 / 4 / 8 times the plain sizes)
static int coarse_restrict_split(phx_system *s, phx_coarse *c, int col3, const double *vin, bool direct, double *X1s, double *X2s,
                                 double *X3s) {
  hipStream_t st = s->mesh->stream;
  const CcDims g{{c->nf[0], c->nf[1], c->nf[2]}, {c->m[0], c->m[1], c->m[2]}, c->ratio, c->M, {c->off[0], c->off[1], c->off[2]}};
  const int64_t nb = c->nblk_u;
  const int64_t nl = nb * c->nf[2] * c->nf[1];
  if (direct) k_cc_restrict_x<true, true><<<dim3((unsigned)phx_div_up(nl, 4)), dim3(256), sizeof(double) * 4 * (size_t)c->nf[0], st>>>(nl, g, col3 % 3, c->cpos, vin, X1s);
  else k_cc_restrict_x<true, false><<<dim3((unsigned)phx_div_up(nl, 4)), dim3(256), sizeof(double) * 4 * (size_t)c->nf[0], st>>>(nl, g, col3 % 3, c->cpos, vin, X1s);
  const int64_t t2 = nb * c->nf[2] * c->m[1] * c->m[0] * 2;
  k_cc_restrict_axis_split<<<cc_grid(t2), dim3(256), 0, st>>>(t2, (int64_t)c->m[0] * 2, c->nf[1], c->m[1], c->ratio, c->off[1], (col3 / 3) % 3, X1s, X2s);
  const int64_t t3 = nb * c->M * 4;
  k_cc_restrict_axis_split<<<cc_grid(t3), dim3(256), 0, st>>>(t3, (int64_t)c->m[0] * c->m[1] * 4, c->nf[2], c->m[2], c->ratio, c->off[2], col3 / 9, X2s, X3s);
  PHX_HIP(hipGetLastError());
  return PHX_OK;
}

// vout += D R zc,  zc = Ac^-1 gc  (gc: the restricted vector, all-reduced by the driver on a partitioned box)
static int coarse_apply_end(phx_system *s, phx_coarse *c, double *vout) {
  if (!c || c->nc == 0) return PHX_OK;
  hipStream_t st = s->mesh->stream;
  if (s->n == 0) return PHX_OK;   // a slab outside the domain: nothing to prolong onto
  k_cc_gemv<<<dim3((unsigned)phx_div_up(c->nc, 4)), dim3(256), 0, st>>>(c->nc, c->Ainv, c->gc, c->zc);
  const CcDims g{{c->nf[0], c->nf[1], c->nf[2]}, {c->m[0], c->m[1], c->m[2]}, c->ratio, c->M, {c->off[0], c->off[1], c->off[2]}};
  const int64_t nb = c->nblk_u;
  const int64_t t3 = nb * c->M;
  k_cc_expand<<<cc_grid(t3), dim3(256), 0, st>>>(t3, c->cmap, c->zc, c->X3);
  const int64_t t2 = nb * c->nf[2] * c->m[1] * c->m[0];
  k_cc_prolong_axis<<<cc_grid(t2), dim3(256), 0, st>>>(t2, (int64_t)c->m[0] * c->m[1], c->nf[2], c->m[2], c->ratio, c->off[2], c->X3, c->X2);
  const int64_t t1 = nb * c->nf[2] * c->nf[1] * c->m[0];
  k_cc_prolong_axis<<<cc_grid(t1), dim3(256), 0, st>>>(t1, c->m[0], c->nf[1], c->m[1], c->ratio, c->off[1], c->X2, c->X1);
  const int64_t tf = nb * s->mesh->nv;
  k_cc_prolong_x_add<<<cc_grid(tf), dim3(256), 0, st>>>(tf, g, c->cpos, c->dpos, c->X1, vout);
  PHX_HIP(hipGetLastError());
  return PHX_OK;
}
// vout += D R Ac^-1 R^T vin on one rank
static int coarse_apply_add(phx_system *s, phx_coarse *c, const double *vin, double *vout) {
  if (!c || c->nc == 0 || s->n == 0 || c->dist) return PHX_OK;
  PHX_CHECK(coarse_restrict(s, c, vin));
  return coarse_apply_end(s, c, vout);
}

__global__ void k_cc_flags_to_f64(int64_t n, const uint8_t *__restrict__ f, double *__restrict__ d) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) d[i] = f[i] ? 1.0 : 0.0;
}

// Builds the coarse correction of `s` (*out = nullptr when it does not apply: not a generated box, rocSOLVER missing,
// a singular coarse matrix, PHX_OPT_EL_COARSE = 0).  `reduce` (partitioned boxes): in-place sum of a device buffer of
// doubles over the ranks, enqueued on the solver stream -- EVERY rank then walks the same sequence of reductions
// (a veto of one rank switches the correction off on all of them).
typedef std::function<int(double *, size_t)> CcReduce;
static int coarse_build(phx_system *s, int nblk, phx_coarse **out, const CcReduce *reduce = nullptr) {
  *out = nullptr;
  phx_mesh *m = s->mesh;
  hipStream_t st = m->stream;
  static const int env_req = getenv("PHX_EL_COARSE") ? atoi(getenv("PHX_EL_COARSE")) : -2;   // A/B aid: overrides the option
  const int req = env_req != -2 ? env_req : m->el_coarse;   // -1: automatic, 0: off, > 0: the ratio H / h
  // decided from numbers every rank shares
  if (req == 0 || !m->is_box || m->is_submesh) return PHX_OK;
  if (!reduce && (!s->rowptr || s->n == 0)) return PHX_OK;
  const int d = m->gdim;
  int64_t nglob[3];
  for (int a = 0; a < 3; ++a) nglob[a] = a < d ? (m->box_nglob[a] > 0 ? m->box_nglob[a] : m->box_n[a]) : 1;
  const int64_t nmax = std::max(nglob[0], std::max(nglob[1], nglob[2]));
  // automatic: the probing passes and the inverse cost about as much as 50-100 iterations of the plain loop, which is
  // what the correction saves from ~64 cubes per axis on (48^3: 176 -> 88 iterations but 52 -> 73 ms)
  if (req < 0 && nmax < 80) return PHX_OK;
  // coarse cells per axis: 8 up to 128 cubes (~3 000 coarse DoFs: the dense inverse takes 30 ms), growing to 13 at 256
  // (~10 000: 0.25 s) -- measured at 256^3: H = 16 h 104 iterations / 2.9 s, 20 h 112 / 2.3 s, 24 h 152 / 2.7 s
  const double cells = 8.0 + (nmax > 128 ? (double)(nmax - 128) / 25.0 : 0.0);
  int ratio = req > 0 ? req : (int)ceil((double)nmax / cells);
  if (ratio < 5) ratio = 5;   // the split of the probing passes needs H > 4 h
  if (nmax < 2 * ratio) return PHX_OK;   // nothing coarser than the mesh itself
  // from here on a partitioned box must reach the veto reduction on every rank
  bool veto = false;
  phx_coarse *c = new phx_coarse();
  auto fail = [&](int code) { coarse_free(c); return code; };
  c->d = d; c->nblk_u = 2 * d; c->ratio = ratio; c->dist = reduce != nullptr;
  c->M = 1;
  for (int a = 0; a < 3; ++a) {
    c->nf[a] = a < d ? m->box_n[a] + 1 : 1;
    c->off[a] = a < d ? (int)m->box_off[a] : 0;
    c->m[a] = a < d ? (int)phx_div_up(nglob[a], ratio) + 1 : 1;
    c->M *= c->m[a];
  }
  const CcDims g{{c->nf[0], c->nf[1], c->nf[2]}, {c->m[0], c->m[1], c->m[2]}, c->ratio, c->M, {c->off[0], c->off[1], c->off[2]}};
  const int64_t nv = m->nv, n = s->n, nb = c->nblk_u, tf = nb * nv;
  const bool rows = n > 0 && s->rowptr;   // false: a slab outside the domain, it only joins the reductions
  if (n > 0 && (!s->rowptr || !s->bj)) veto = true;   // without the vertex blocks phat aliases p: nothing to add a correction to
  if (phx_malloc(&c->cpos, sizeof(int32_t) * (size_t)tf) != hipSuccess || phx_malloc(&c->dpos, sizeof(double) * (size_t)std::max<int64_t>(n, 1)) != hipSuccess ||
      phx_malloc(&c->cmap, sizeof(int32_t) * (size_t)(nb * c->M)) != hipSuccess)
    return fail(PHX_ERR_HIP);
  PHX_HIP(hipMemsetAsync(c->cpos, 0xff, sizeof(int32_t) * (size_t)tf, st));
  if (rows) k_cc_positions<<<cc_grid(n), dim3(256), 0, st>>>(n, nv, c->nblk_u, d, s->full_of_active, s->rowptr, s->iperm, s->diag, s->own, c->cpos, c->dpos);
  uint8_t *used = nullptr;
  double *usedf = nullptr;
  if (phx_malloc(&used, (size_t)(nb * c->M)) != hipSuccess || phx_malloc(&usedf, sizeof(double) * (size_t)(nb * c->M)) != hipSuccess) {
    (void)phx_free(used); (void)phx_free(usedf);
    return fail(PHX_ERR_HIP);
  }
  PHX_HIP(hipMemsetAsync(used, 0, (size_t)(nb * c->M), st));
  k_cc_used<<<cc_grid(tf), dim3(256), 0, st>>>(tf, nv, g, c->cpos, used);
  k_cc_flags_to_f64<<<cc_grid(nb * c->M), dim3(256), 0, st>>>(nb * c->M, used, usedf);
  int rrc = reduce ? (*reduce)(usedf, (size_t)(nb * c->M)) : PHX_OK;
  std::vector<double> hused((size_t)(nb * c->M));
  if (rrc == PHX_OK && hipMemcpyAsync(hused.data(), usedf, sizeof(double) * hused.size(), hipMemcpyDeviceToHost, st) != hipSuccess) rrc = PHX_ERR_HIP;
  if (rrc == PHX_OK && hipStreamSynchronize(st) != hipSuccess) rrc = PHX_ERR_HIP;
  (void)phx_free(used); (void)phx_free(usedf);
  if (rrc != PHX_OK) return fail(rrc);
  std::vector<int32_t> hmap(hused.size(), -1), hnode;
  for (size_t q = 0; q < hused.size(); ++q)
    if (hused[q] > 0.0) { hmap[q] = (int32_t)hnode.size(); hnode.push_back((int32_t)q); }
  c->nc = (int)hnode.size();
  static const int nc_max = getenv("PHX_EL_COARSE_MAX") ? atoi(getenv("PHX_EL_COARSE_MAX")) : 24000;
  if (c->nc == 0 || c->nc > nc_max) { coarse_free(c); return PHX_OK; }   // the same numbers on every rank
  const int nc = c->nc;
  const int64_t t1 = nb * c->nf[2] * c->nf[1] * c->m[0], t2 = nb * c->nf[2] * c->m[1] * c->m[0], t3 = nb * c->M;
  double *wv = nullptr, *tv = nullptr, *X1s = nullptr, *X2s = nullptr, *X3s = nullptr, *T = nullptr, *vflag = nullptr;
  int32_t *f32 = nullptr;
  auto drop_scratch = [&]() {
    (void)phx_free(wv); (void)phx_free(tv); (void)phx_free(X1s); (void)phx_free(X2s); (void)phx_free(X3s); (void)phx_free(T);
    (void)phx_free(f32); (void)phx_free(vflag);
  };
  static const bool lumped = getenv("PHX_EL_COARSE_LUMPED") && atoi(getenv("PHX_EL_COARSE_LUMPED")) != 0;   // A/B aids, one rank
  static const bool by_spmv = getenv("PHX_EL_COARSE_SPMV") && atoi(getenv("PHX_EL_COARSE_SPMV")) != 0;
  bool mem_ok = phx_malloc(&c->node_of, sizeof(int32_t) * (size_t)nc) == hipSuccess && phx_malloc(&c->Ainv, sizeof(double) * (size_t)nc * nc) == hipSuccess &&
                phx_malloc(&c->X1, sizeof(double) * (size_t)t1) == hipSuccess && phx_malloc(&c->X2, sizeof(double) * (size_t)t2) == hipSuccess &&
                phx_malloc(&c->X3, sizeof(double) * (size_t)t3) == hipSuccess && phx_malloc(&c->gc, sizeof(double) * (size_t)nc) == hipSuccess &&
                phx_malloc(&c->zc, sizeof(double) * (size_t)nc) == hipSuccess && phx_malloc(&vflag, sizeof(double)) == hipSuccess &&
                phx_malloc(&X1s, sizeof(double) * (size_t)t1 * 2) == hipSuccess && phx_malloc(&X2s, sizeof(double) * (size_t)t2 * 4) == hipSuccess &&
                phx_malloc(&X3s, sizeof(double) * (size_t)t3 * 8) == hipSuccess;
  // nine colours of a block from one pass over the CSR copy when their images fit (nine vectors in full lattice order:
  // 7 GB at 256^3), else -- one rank only -- one product with the solver's SpMV per (block, colour)
  if (mem_ok && rows && !(lumped && !reduce) && !(by_spmv && !reduce) && (int64_t)c->nblk_u * nv < INT32_MAX && s->nent < INT32_MAX) {
    size_t fr = 0, tot = 0;
    (void)hipMemGetInfo(&fr, &tot);
    const size_t need = sizeof(double) * 9 * (size_t)tf + sizeof(int32_t) * (size_t)n;
    if (need + (size_t(4) << 30) < fr && phx_malloc(&f32, sizeof(int32_t) * (size_t)n) == hipSuccess) {
      if (phx_malloc(&T, sizeof(double) * 9 * (size_t)tf) != hipSuccess) { (void)hipGetLastError(); (void)phx_free(f32); f32 = nullptr; T = nullptr; }
    }
  }
  if (mem_ok && rows && !T) {
    if (reduce) veto = true;   // the SpMV probes would need the halo of every probing vector
    else mem_ok = phx_malloc(&wv, sizeof(double) * (size_t)n) == hipSuccess && phx_malloc(&tv, sizeof(double) * (size_t)n) == hipSuccess;
  }
  if (!mem_ok) { (void)hipGetLastError(); if (!reduce) { drop_scratch(); return fail(PHX_ERR_HIP); } veto = true; }
  if (reduce) {   // one veto and nobody corrects
    const double v = veto ? 1.0 : 0.0;
    double hv = 0.0;
    int rc2 = vflag ? PHX_OK : PHX_ERR_HIP;
    if (rc2 == PHX_OK && hipMemcpyAsync(vflag, &v, sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess) rc2 = PHX_ERR_HIP;
    if (rc2 == PHX_OK) rc2 = (*reduce)(vflag, 1);
    if (rc2 == PHX_OK && hipMemcpyAsync(&hv, vflag, sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess) rc2 = PHX_ERR_HIP;
    if (rc2 == PHX_OK && hipStreamSynchronize(st) != hipSuccess) rc2 = PHX_ERR_HIP;
    if (rc2 != PHX_OK) { drop_scratch(); return fail(rc2); }
    if (hv > 0.0) { drop_scratch(); coarse_free(c); return PHX_OK; }
  }
  int rc = PHX_OK;
  if (hipMemcpyAsync(c->cmap, hmap.data(), sizeof(int32_t) * hmap.size(), hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemcpyAsync(c->node_of, hnode.data(), sizeof(int32_t) * hnode.size(), hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemsetAsync(c->Ainv, 0, sizeof(double) * (size_t)nc * nc, st) != hipSuccess)
    rc = PHX_ERR_HIP;
  // ---- Ac (this rank's rows' share of it) by probing
  const int ncol = d == 3 ? 27 : 9;
  if (T && rc == PHX_OK) {
    k_cc_f32<<<cc_grid(n), dim3(256), 0, st>>>(n, s->full_of_active, f32);
    for (int bj = 0; bj < c->nblk_u && rc == PHX_OK; ++bj) {
      for (int cz = 0; cz < 3 && cz < c->m[2] && rc == PHX_OK; ++cz) {
        k_cc_probe_rows<<<cc_grid(n * 16), dim3(256), 0, st>>>(n, nv, bj, c->nblk_u, g, cz, s->rowptr, s->col, s->val, f32, c->cpos, T, tf);
        for (int cxy = 0; cxy < 9 && rc == PHX_OK; ++cxy) {
          if (cxy % 3 >= c->m[0] || cxy / 3 >= c->m[1]) continue;   // a colour no node of the lattice carries
          const int col = cxy + 9 * cz;
          rc = coarse_restrict_split(s, c, col, T + (int64_t)cxy * tf, true, X1s, X2s, X3s);
          if (rc == PHX_OK) k_cc_store_parts<<<cc_grid(nc), dim3(256), 0, st>>>(nc, bj, col, g, c->node_of, c->cmap, X3s, c->Ainv);
        }
      }
    }
  }
  for (int bj = 0; bj < c->nblk_u && rc == PHX_OK && !T && rows; ++bj) {
    for (int col = 0; col < ncol && rc == PHX_OK; ++col) {
      // a colour no node of the lattice carries probes nothing
      bool any = true;
      const int cc3[3] = {col % 3, (col / 3) % 3, col / 9};
      for (int a = 0; a < 3; ++a) any = any && cc3[a] < c->m[a];
      if (!any) continue;
      if (hipMemsetAsync(wv, 0, sizeof(double) * (size_t)n, st) != hipSuccess) { rc = PHX_ERR_HIP; break; }
      k_cc_probe<<<cc_grid(nv), dim3(256), 0, st>>>(nv, bj, col, g, c->cpos, c->dpos, wv);
      rc = launch_spmv(s, s->sell_val, wv, tv, 0, nullptr, nullptr, nullptr, 0);
      if (rc == PHX_OK && lumped) {
        rc = coarse_restrict(s, c, tv);
        if (rc == PHX_OK) k_cc_store_column<<<cc_grid(nc), dim3(256), 0, st>>>(nc, bj, col, g, c->node_of, c->cmap, c->gc, c->Ainv);
      } else if (rc == PHX_OK) {
        rc = coarse_restrict_split(s, c, col, tv, false, X1s, X2s, X3s);
        if (rc == PHX_OK) k_cc_store_parts<<<cc_grid(nc), dim3(256), 0, st>>>(nc, bj, col, g, c->node_of, c->cmap, X3s, c->Ainv);
      }
    }
  }
  if (rc == PHX_OK && hipGetLastError() != hipSuccess) rc = PHX_ERR_HIP;
  if (rc == PHX_OK && reduce) rc = (*reduce)(c->Ainv, (size_t)nc * nc);
  if (rc == PHX_OK && hipStreamSynchronize(st) != hipSuccess) rc = PHX_ERR_HIP;
  drop_scratch();
  if (rc != PHX_OK) return fail(rc);
  // ---- dense inverse (every rank inverts the same matrix): the library's own blocked Gauss-Jordan, phx_dense.inc.hip
  int singular = 0;
  const int rci = dense_inverse_inplace(c->Ainv, nc, st, &singular);
  if (rci != PHX_OK) return fail(rci);
  if (singular) {
    // singular coarse matrix (the same on every rank): the vertex blocks alone -- said, not silent
    fprintf(stderr, "phifem_hip: the Galerkin coarse matrix of the elasticity correction (%d x %d) is singular to working precision: "
                    "the solve keeps the vertex blocks alone\n", nc, nc);
    coarse_free(c);
    return PHX_OK;
  }
  *out = c;
  return PHX_OK;
}
