// Interface linear elasticity, 5-field mixed phi-FEM (u_in, u_out, y_in, y_out, p), all P1:
// included by phx_assemble.hip.  Forms: demo/interface-elasticity/main.py:179-235 (bilinear),
// :255-269 (linear), Dirichlet rows of u_in :158-177,237-239,271-277; material law data.py:5-36.
// Closed-form integrals on affine simplices (every integrand is a polynomial), restated in
// oracle/elasticity.py.  DoF layout: component-major blocks of nv entries,
//   u_in[a] -> a, u_out[a] -> d+a, y_in[a][b] -> 2d+a d+b, y_out[a][b] -> 2d+d^2+a d+b, p[a] -> 2d+2d^2+a.
// This first version scatters every contribution through the row-slot tables (one 256-thread
// block per element, threads walk the element tensor); at BASELINE scale the bulk stiffness
// needs the row-gather treatment of k_assemble_rows.

struct ElArgs {
  const int32_t *cells;
  const double *x;
  const int8_t *ctags;
  const int32_t *c2f, *f2c;
  const int32_t *dofmap;   // [C*nv] full DoF -> active row or -1
  const uint8_t *bc;       // [nv] 1: u_in is prescribed at this vertex
  const double *phi;       // [nv]
  const double *f;         // [d*nv] component-major
  const double *ud;        // [d*nv]
  double lam[2], mu[2], coefW[2];  // coefW[0] = coef_out (multiplies the "in" term), [1] = coef_in
  double gamma, sigma;
  int32_t nv;
  double *rhs;
  Slots slots;
};

template <int D>
struct ElB {
  static constexpr int N = D + 1, C = 2 * D + 2 * D * D + D, R = N * C;
  // kind 0 u_in, 1 u_out, 2 y_in, 3 y_out, 4 p;  a, b components (b = 0 unless y)
  __device__ static __forceinline__ void decode(int blk, int &kind, int &a, int &b) {
    b = 0;
    if (blk < D) { kind = 0; a = blk; }
    else if (blk < 2 * D) { kind = 1; a = blk - D; }
    else if (blk < 2 * D + D * D) { kind = 2; a = (blk - 2 * D) / D; b = (blk - 2 * D) % D; }
    else if (blk < 2 * D + 2 * D * D) { kind = 3; a = (blk - 2 * D - D * D) / D; b = (blk - 2 * D - D * D) % D; }
    else { kind = 4; a = blk - 2 * D - 2 * D * D; }
  }
  __device__ static __forceinline__ int ublk(int side, int a) { return side * D + a; }
  __device__ static __forceinline__ int yblk(int side, int a, int b) { return 2 * D + side * D * D + a * D + b; }
};

// sigma(N_j e_b)[p][q] = lam g_{j,b} d_pq + mu (d_pb g_{j,q} + g_{j,p} d_qb)
template <int D>
__device__ __forceinline__ double sig_pq(const Geo<D> &G, double lam, double mu, int j, int b, int p, int q) {
  return (p == q ? lam * G.g[j][b] : 0.0) + mu * ((p == b ? G.g[j][q] : 0.0) + (q == b ? G.g[j][p] : 0.0));
}

__device__ __forceinline__ bool el_is_bc(const ElArgs &A, int32_t full, int d) {
  return full < d * A.nv && A.bc[full % A.nv];
}

// insert with the Dirichlet treatment of assemble_matrix(bcs) + apply_lifting
template <int D>
__device__ __forceinline__ void el_add(const ElArgs &A, int32_t row_full, int32_t col_full, double v) {
  const bool rbc = el_is_bc(A, row_full, D), cbc = el_is_bc(A, col_full, D);
  if (cbc) {
    if (!rbc) slot_rhs_add(A.slots, A.rhs, A.dofmap[row_full], -v * A.ud[col_full]);
    return;
  }
  if (rbc) return;
  slot_add(A.slots, A.dofmap[row_full], col_full, v);
}
template <int D>
__device__ __forceinline__ void el_rhs(const ElArgs &A, int32_t row_full, double v) {
  if (el_is_bc(A, row_full, D)) return;
  slot_rhs_add(A.slots, A.rhs, A.dofmap[row_full], v);
}

template <int D>
__global__ void k_el_mark_active(int64_t nc, ElArgs A, uint8_t *__restrict__ flags) {
  using B = ElB<D>;
  const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (c >= nc) return;
  const int t = A.ctags[c] & PHX_TAG_MASK;
  for (int i = 0; i < B::N; ++i) {
    const int64_t v = A.cells[c * B::N + i];
    for (int blk = 0; blk < B::C; ++blk) {
      int kind, a, b;
      B::decode(blk, kind, a, b);
      const bool on = kind == 0 ? (t == 1 || t == 2) : (kind == 1 ? (t == 2 || t == 3) : t == 2);
      if (on) flags[(int64_t)blk * A.nv + v] = 1;
    }
  }
}
__global__ void k_el_mark_bc(int64_t nbc, int d, int64_t nv, const int32_t *__restrict__ bcv,
                             uint8_t *__restrict__ flags, uint8_t *__restrict__ bc) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= nbc) return;
  bc[bcv[i]] = 1;
  for (int a = 0; a < d; ++a) flags[(int64_t)a * nv + bcv[i]] = 1;
}
__global__ void k_el_numbering(int64_t nent, const uint8_t *__restrict__ flags,
                               const int32_t *__restrict__ scan, int32_t *__restrict__ dofmap,
                               int64_t *__restrict__ full_of_active) {
  const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= nent) return;
  int32_t a = -1;
  if (flags[e]) { a = scan[e]; full_of_active[a] = e; }
  dofmap[e] = a;
}

// --- stiffness main.py:185-186,226-227 + source :263-264; one block per cell --------------------
template <int D>
__global__ void __launch_bounds__(256) k_el_bulk(int64_t nc, ElArgs A) {
  using B = ElB<D>;
  const int64_t c = blockIdx.x;
  if (c >= nc) return;
  const int t = A.ctags[c] & PHX_TAG_MASK;
  if (t < 1 || t > 3) return;
  int32_t v[B::N];
  double X[B::N][D];
  load_cell<D>(A.cells, A.x, c, v, X);
  Geo<D> G;
  simplex_geometry<D>(X, G);
  constexpr int M = B::N * D;
  constexpr double c2 = D == 3 ? 1.0 / 20.0 : 1.0 / 12.0;
  for (int side = 0; side < 2; ++side) {
    if (side == 0 ? !(t == 1 || t == 2) : !(t == 2 || t == 3)) continue;
    for (int idx = threadIdx.x; idx < M * M; idx += blockDim.x) {
      const int r = idx / M, s = idx % M;
      const int a = r / B::N, i = r % B::N, cc = s / B::N, j = s % B::N;
      // eps(N_i e_a) : sigma(N_j e_c) = sigma(N_j e_c)[a][:] . g_i
      double k = 0.0;
      for (int q = 0; q < D; ++q) k += sig_pq<D>(G, A.lam[side], A.mu[side], j, cc, a, q) * G.g[i][q];
      el_add<D>(A, B::ublk(side, a) * A.nv + v[i], B::ublk(side, cc) * A.nv + v[j], k * G.vol);
    }
    if (threadIdx.x < M) {
      const int a = threadIdx.x / B::N, i = threadIdx.x % B::N;
      double sf = 0.0;
      for (int q = 0; q < B::N; ++q) sf += A.f[(int64_t)a * A.nv + v[q]];
      el_rhs<D>(A, B::ublk(side, a) * A.nv + v[i], G.vol * c2 * (sf + A.f[(int64_t)a * A.nv + v[i]]));
    }
  }
}

// --- the same bulk terms on a Kuhn box, gathered per row: one thread per (vertex, side, component a) walks
// the closed-form star of the vertex (k_assemble_rows_box) and accumulates the 15 neighbours x D components of
// row u_side[a] in registers.  No atomics; rows no scattering kernel reaches are stored dense and sorted
// (component-major, then vertex: the order of the active numbering).  The one-block-per-cell scatter above
// (144 atomics per cell and side) took 172 of the 294 ms of a 256 x 256 x 40 slab.
template <int D>
__global__ void __launch_bounds__(256)
k_el_bulk_box(int64_t nthreads, BoxDims bd, const uint8_t *__restrict__ touched, ElArgs A) {
  using B = ElB<D>;
  constexpr int N = D + 1, NPERM = D == 3 ? 6 : 2, NCODE = D == 3 ? 27 : 9;
  constexpr int P[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
  constexpr int P2[2][3] = {{0, 1, 0}, {1, 0, 0}};
  constexpr int POW3[3] = {1, 3, 9};
  constexpr double c2 = D == 3 ? 1.0 / 20.0 : 1.0 / 12.0;
  const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (tid >= nthreads) return;
  const int64_t vtx = tid % A.nv;
  const int sa = (int)(tid / A.nv);
  const int side = sa / D, a = sa % D;
  const int32_t row_full = (int32_t)(B::ublk(side, a) * (int64_t)A.nv + vtx);
  const int32_t row = A.dofmap[row_full];
  if (row < 0) return;
  if (side == 0 && A.bc[vtx]) return;  // Dirichlet row: k_el_bc_rows writes it
  const int64_t n0 = bd.n[0] + 1, n1 = bd.n[1] + 1;
  int64_t idx[3] = {vtx % n0, D == 3 ? (vtx / n0) % n1 : vtx / n0, D == 3 ? vtx / (n0 * n1) : 0};
  const int64_t vstride[3] = {1, n0, n0 * n1};
  const int64_t cstride[3] = {1, bd.n[0], bd.n[0] * bd.n[1]};
  const double lam = A.lam[side], mu = A.mu[side];
  double acc[NCODE][D], nf[NCODE];
  bool seen[NCODE];
#pragma unroll
  for (int code = 0; code < NCODE; ++code) {
    int d[3] = {code % 3 - 1, (code / 3) % 3 - 1, D == 3 ? code / 9 - 1 : 0};
    bool pos = false, neg = false;
    for (int q = 0; q < D; ++q) { pos |= d[q] > 0; neg |= d[q] < 0; }
    for (int q = 0; q < D; ++q) acc[code][q] = 0.0;
    seen[code] = false;
    nf[code] = 0.0;
    if (pos && neg) continue;
    bool in = true;
    int64_t w = vtx;
    for (int q = 0; q < D; ++q) {
      const int64_t z = idx[q] + d[q];
      in = in && z >= 0 && z <= bd.n[q];
      w += d[q] * vstride[q];
    }
    if (in) nf[code] = A.f[(int64_t)a * A.nv + w];
  }
  constexpr int SELF = D == 3 ? 13 : 4;
  double rhs = 0.0;
#pragma unroll
  for (int t = 0; t < NPERM; ++t) {
#pragma unroll
    for (int m = 0; m < N; ++m) {
      int dd[3] = {0, 0, 0};
      for (int q = 0; q < m; ++q) dd[D == 3 ? P[t][q] : P2[t][q]] -= 1;
      bool in = true;
      int64_t cube = 0;
      for (int q = 0; q < D; ++q) {
        const int64_t o = idx[q] + dd[q];
        in = in && o >= 0 && o < bd.n[q];
        cube += o * cstride[q];
      }
      if (!in) continue;
      const int tag = A.ctags[cube * NPERM + t] & PHX_TAG_MASK;
      if (side == 0 ? !(tag == 1 || tag == 2) : !(tag == 2 || tag == 3)) continue;
      int code[N];
      double X[N][D];
      double sf = 0.0;
      for (int q = 0; q < N; ++q) {
        code[q] = 0;
        for (int r = 0; r < D; ++r) code[q] += (dd[r] + 1) * POW3[r];
        for (int r = 0; r < D; ++r) X[q][r] = (double)dd[r] * bd.h[r];
        sf += nf[code[q]];
        if (q < D) dd[D == 3 ? P[t][q] : P2[t][q]] += 1;
      }
      Geo<D> G;
      simplex_geometry<D>(X, G);
      // eps(N_m e_a) : sigma(N_j e_c) = sigma(N_j e_c)[a][:] . g_m
      for (int j = 0; j < N; ++j) {
        for (int cc = 0; cc < D; ++cc) {
          double k = 0.0;
          for (int q = 0; q < D; ++q) k += sig_pq<D>(G, lam, mu, j, cc, a, q) * G.g[m][q];
          acc[code[j]][cc] += k * G.vol;
        }
        seen[code[j]] = true;
      }
      rhs += G.vol * c2 * (sf + nf[SELF]);
    }
  }
  const bool clean = A.slots.clean && touched && !touched[vtx];
  int W;
  const int64_t sbase = slot_base(A.slots, row, &W);
  int cnt = 0;
  // component-major, then vertex: ascending active column
#pragma unroll
  for (int cc = 0; cc < D; ++cc) {
#pragma unroll
    for (int code = 0; code < NCODE; ++code) {
      if (!seen[code]) continue;
      int64_t w = vtx;
      w += (code % 3 - 1) * vstride[0] + ((code / 3) % 3 - 1) * vstride[1];
      if (D == 3) w += (code / 9 - 1) * vstride[2];
      const int32_t col_full = (int32_t)(B::ublk(side, cc) * (int64_t)A.nv + w);
      if (side == 0 && A.bc[w]) {       // prescribed column: lifting (apply_lifting, main.py:271-274)
        rhs -= acc[code][cc] * A.ud[col_full];
        continue;
      }
      if (clean) {
        A.slots.cols[sbase + cnt] = col_full;
        A.slots.vals[sbase + cnt] = acc[code][cc];
        ++cnt;
      } else {
        slot_add_owned(A.slots, row, col_full, acc[code][cc]);
      }
    }
  }
  if (clean) A.slots.clean[row] = (uint8_t)cnt;
  if (A.slots.remax) slot_rhs_add(A.slots, A.rhs, row, rhs);   // deterministic mode: a contribution like the scattered ones
  else A.rhs[row] += rhs;
}

// --- cut cells: penalization main.py:188-203, cell stabilisation :211-217, rhs :255-260 ------------
// block pairs (rb, cb) of the cut-cell tensor that carry a term at all (255 of the 27^2 = 729 in 3-D): the same
// conditions as in the body of k_el_cut, on the block codes alone
template <int D>
__device__ __forceinline__ bool el_cut_pair_present(int rb, int cb) {
  using B = ElB<D>;
  int kr, ar, br, kc, ac, bc;
  B::decode(rb, kr, ar, br);
  B::decode(cb, kc, ac, bc);
  if (kr <= 1 && kc <= 1) return kr == kc || ar == ac;
  if (kr <= 1 && (kc == 2 || kc == 3)) return kr == kc - 2;
  if ((kr == 2 || kr == 3) && kc <= 1) return kc == kr - 2;
  if ((kr == 2 || kr == 3) && (kc == 2 || kc == 3)) return ar == ac;
  return ar == ac;   // u - p, p - u, p - p
}

// One workgroup per cut cell.  Round 4: the threads walk the entries of the PRESENT block pairs only (listed once per
// workgroup in LDS) instead of all (27 (D+1))^2 = 11664 entries of the tensor, two thirds of which carry no term.
template <int D>
__global__ void __launch_bounds__(256) k_el_cut(int64_t nlist, const int32_t *__restrict__ list, ElArgs A) {
  using B = ElB<D>;
  const int64_t e = blockIdx.x;
  if (e >= nlist) return;
  __shared__ uint16_t present[B::C * B::C];
  __shared__ int npresent;
  if (threadIdx.x == 0) npresent = 0;
  __syncthreads();
  for (int pi = threadIdx.x; pi < B::C * B::C; pi += blockDim.x)
    if (el_cut_pair_present<D>(pi / B::C, pi % B::C)) present[atomicAdd(&npresent, 1)] = (uint16_t)pi;
  __syncthreads();
  const int nent = npresent * B::N * B::N;
  const int64_t c = list[e];
  int32_t v[B::N];
  double X[B::N][D];
  load_cell<D>(A.cells, A.x, c, v, X);
  Geo<D> G;
  simplex_geometry<D>(X, G);
  constexpr double c2 = D == 3 ? 1.0 / 20.0 : 1.0 / 12.0;
  constexpr double c3 = D == 3 ? 1.0 / 120.0 : 1.0 / 60.0;
  constexpr double c4 = D == 3 ? 1.0 / 840.0 : 1.0 / 360.0;
  double ph[B::N], sp = 0.0, gphi[D];
  for (int q = 0; q < B::N; ++q) { ph[q] = A.phi[v[q]]; sp += ph[q]; }
  for (int dd = 0; dd < D; ++dd) {
    double t = 0.0;
    for (int q = 0; q < B::N; ++q) t += ph[q] * G.g[q][dd];
    gphi[dd] = t;
  }
  const double h1 = 1.0 / G.h, gam = A.gamma;
  const double sgn[2] = {1.0, -1.0};
  for (int idx = threadIdx.x; idx < nent; idx += blockDim.x) {
    const int pair = present[idx / (B::N * B::N)], ij = idx % (B::N * B::N);
    const int rb = pair / B::C, cb = pair % B::C, i = ij / B::N, j = ij % B::N;
    int kr, ar, br, kc, ac, bc;
    B::decode(rb, kr, ar, br);
    B::decode(cb, kc, ac, bc);
    const double Mij = G.vol * c2 * (i == j ? 2.0 : 1.0);
    double val = 0.0;
    bool has = false;
    if (kr <= 1 && kc <= 1) {                       // u_s - u_t
      const int sr = kr, sc = kc;
      if (sr == sc) {                               // (y + sigma(u)):(z + sigma(v)), v-u part
        double t = 0.0;
        for (int p = 0; p < D; ++p)
          for (int q = 0; q < D; ++q)
            t += sig_pq<D>(G, A.lam[sr], A.mu[sr], i, ar, p, q) * sig_pq<D>(G, A.lam[sr], A.mu[sr], j, ac, p, q);
        val += gam * A.coefW[sr] * G.vol * t;
        has = true;
      }
      if (ar == ac) { val += gam * sgn[sr] * sgn[sc] * h1 * h1 * Mij; has = true; }
    } else if (kr <= 1 && (kc == 2 || kc == 3)) {   // u_s - y_t : v-y part
      if (kr == kc - 2) { val = gam * A.coefW[kr] * (G.vol / B::N) * sig_pq<D>(G, A.lam[kr], A.mu[kr], i, ar, ac, bc); has = true; }
    } else if ((kr == 2 || kr == 3) && kc <= 1) {   // y_s - u_t : z-u part
      if (kc == kr - 2) { val = gam * A.coefW[kc] * (G.vol / B::N) * sig_pq<D>(G, A.lam[kc], A.mu[kc], j, ac, ar, br); has = true; }
    } else if ((kr == 2 || kr == 3) && (kc == 2 || kc == 3)) {  // y_s - y_t
      const int sr = kr - 2, sc = kc - 2;
      if (ar == ac) {
        val += gam * sgn[sr] * sgn[sc] * h1 * h1 * gphi[br] * gphi[bc] * Mij;   // main.py:193-197
        has = true;
        if (sr == sc) {
          val += A.sigma * G.h * G.h * G.vol * G.g[i][br] * G.g[j][bc];          // main.py:211-217
          if (br == bc) val += gam * A.coefW[sr] * Mij;                          // z-y part
        }
      }
    } else if (kr <= 1 && kc == 4) {                // u_s - p
      if (ar == ac) { val = gam * sgn[kr] * h1 * h1 * h1 * G.vol * c3 * (i == j ? 2.0 : 1.0) * (sp + ph[i] + ph[j]); has = true; }
    } else if (kr == 4 && kc <= 1) {                // p - u_t
      if (ar == ac) { val = gam * sgn[kc] * h1 * h1 * h1 * G.vol * c3 * (i == j ? 2.0 : 1.0) * (sp + ph[i] + ph[j]); has = true; }
    } else if (kr == 4 && kc == 4) {
      if (ar == ac) {
        double m4 = 0.0;
        for (int k = 0; k < B::N; ++k)
          for (int l = 0; l < B::N; ++l) m4 += mult4(i, j, k, l) * ph[k] * ph[l];
        val = gam * h1 * h1 * h1 * h1 * G.vol * c4 * m4;
        has = true;
      }
    }
    if (has) el_add<D>(A, rb * A.nv + v[i], cb * A.nv + v[j], val);
  }
  // rhs: sigma h^2 f . div(z), z = N_i E_ab  ->  h^2 vol fbar_a g_{i,b}
  for (int idx = threadIdx.x; idx < 2 * D * D * B::N; idx += blockDim.x) {
    const int side = idx / (D * D * B::N), rem = idx % (D * D * B::N);
    const int a = rem / (D * B::N), b = (rem / B::N) % D, i = rem % B::N;
    double fb = 0.0;
    for (int q = 0; q < B::N; ++q) fb += A.f[(int64_t)a * A.nv + v[q]];
    fb /= B::N;
    el_rhs<D>(A, B::yblk(side, a, b) * A.nv + v[i], A.sigma * G.h * G.h * G.vol * fb * G.g[i][b]);
  }
}

// --- one-sided boundary terms main.py:182-183: (y_s n) . v_s over d_bdry(100) / d_bdry(101) --------
template <int D>
__global__ void __launch_bounds__(256) k_el_ds(int64_t nent, const int64_t *__restrict__ ent_packed, int side, ElArgs A) {
  using B = ElB<D>;
  const int64_t e = blockIdx.x;
  if (e >= nent) return;
  const int64_t c = ent_packed[2 * e + 1] >> 8;
  const int lf = (int)(ent_packed[2 * e + 1] & 0xff);
  int32_t v[B::N];
  double X[B::N][D];
  load_cell<D>(A.cells, A.x, c, v, X);
  Geo<D> G;
  simplex_geometry<D>(X, G);
  double gn = 0.0;
  for (int dd = 0; dd < D; ++dd) gn += G.g[lf][dd] * G.g[lf][dd];
  gn = sqrt(gn);
  const double area = D * G.vol * gn;
  for (int idx = threadIdx.x; idx < D * D * B::N * B::N; idx += blockDim.x) {
    const int a = idx / (D * B::N * B::N), ee = (idx / (B::N * B::N)) % D, i = (idx / B::N) % B::N, j = idx % B::N;
    if (i == lf || j == lf) continue;
    const double ne = -G.g[lf][ee] / gn;
    el_add<D>(A, B::ublk(side, a) * A.nv + v[i], B::yblk(side, a, ee) * A.nv + v[j],
              area * ne * (i == j ? 2.0 : 1.0) / (D * (D + 1)));
  }
}

// --- facet stabilisation main.py:205-209 (dS(3), in) and :219-223 (dS(4), out) --------------------
template <int D>
__global__ void __launch_bounds__(256) k_el_facets(int64_t nlist, const int32_t *__restrict__ list, int side, ElArgs A) {
  using B = ElB<D>;
  const int64_t e = blockIdx.x;
  if (e >= nlist) return;
  const int64_t f = list[e];
  int32_t vv[2][B::N];
  Geo<D> G[2];
  double nrm[2][D], hsum = 0.0, area = 0.0;
  for (int sd = 0; sd < 2; ++sd) {
    const int64_t c = A.f2c[2 * f + sd];
    double X[B::N][D];
    load_cell<D>(A.cells, A.x, c, vv[sd], X);
    simplex_geometry<D>(X, G[sd]);
    int lf = 0;
    for (int k = 0; k < B::N; ++k)
      if (A.c2f[c * B::N + k] == (int32_t)f) lf = k;
    double gn = 0.0;
    for (int dd = 0; dd < D; ++dd) gn += G[sd].g[lf][dd] * G[sd].g[lf][dd];
    gn = sqrt(gn);
    if (sd == 0) area = D * G[0].vol * gn;
    hsum += G[sd].h;
    for (int dd = 0; dd < D; ++dd) nrm[sd][dd] = -G[sd].g[lf][dd] / gn;
  }
  const double wgt = A.sigma * 0.5 * hsum * area;
  constexpr int M = 2 * B::N * D;   // (cell side, comp, vertex)
  for (int idx = threadIdx.x; idx < M * M; idx += blockDim.x) {
    const int r = idx / M, s = idx % M;
    const int sr = r / (B::N * D), ar = (r / B::N) % D, i = r % B::N;
    const int sc = s / (B::N * D), ac = (s / B::N) % D, j = s % B::N;
    double acc = 0.0;
    for (int p = 0; p < D; ++p) {
      double jr = 0.0, jc = 0.0;   // (sigma(N e) n)_p on each side
      for (int q = 0; q < D; ++q) {
        jr += sig_pq<D>(G[sr], A.lam[side], A.mu[side], i, ar, p, q) * nrm[sr][q];
        jc += sig_pq<D>(G[sc], A.lam[side], A.mu[side], j, ac, p, q) * nrm[sc][q];
      }
      acc += jr * jc;
    }
    el_add<D>(A, B::ublk(side, ar) * A.nv + vv[sr][i], B::ublk(side, ac) * A.nv + vv[sc][j], wgt * acc);
  }
}

// Dirichlet rows: unit diagonal, rhs = prescribed value (main.py:237-239, 275-277)
__global__ void k_el_bc_rows(int64_t nbc, int d, ElArgs A, const int32_t *__restrict__ bcv) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= nbc * d) return;
  const int a = (int)(i / nbc);
  const int32_t full = a * A.nv + bcv[i % nbc];
  const int32_t row = A.dofmap[full];
  slot_add(A.slots, row, full, 1.0);
  A.rhs[row] = A.ud[full];
}

// slot capacity class per active row: rows of DoFs at vertices of cut cells get `wbig` (log2), the bulk
// rows (u_in / u_out away from the interface: at most 15 neighbours x 3 components = 45 entries, 54 next to
// the cut layer) get 64 slots.  Measured at 24^3: longest row 225, 45 % of the rows at most 64.
__global__ void k_el_row_caps(int64_t n, const int64_t *__restrict__ full_of_active, int64_t nv,
                              const uint8_t *__restrict__ cutv, int wbig, uint8_t *__restrict__ wlog,
                              int64_t *__restrict__ cap) {
  const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r > n) return;
  if (r == n) { cap[r] = 0; return; }
  const int w = cutv[full_of_active[r] % nv] ? wbig : 6;
  wlog[r] = (uint8_t)w;
  cap[r] = (int64_t)1 << w;
}

struct SelFacetTagInterior {
  const int8_t *ft; const int32_t *f2c; int tag;
  __host__ __device__ bool operator()(const int32_t &f) const { return ft[f] == tag && f2c[2 * (int64_t)f + 1] >= 0; }
};

static int assemble_el_with_capacity(phx_mesh *m, const double *params, const double *dphi,
                                     const double *df, const double *dud, const int32_t *dbcv,
                                     int64_t nbc, int W, phx_system **out) {
  const int D = m->gdim;
  const int C = 2 * D + 2 * D * D + D;
  const int64_t nent = (int64_t)C * m->nv;
  PHX_REQUIRE(nent < INT32_MAX, PHX_ERR_VALUE, "too many DoFs for 32-bit column keys");
  phx_system *s = new phx_system();
  s->mesh = m; s->device = m->device; s->nfull = nent; s->slot_cap = W; s->nent = nent;
  s->el_nblk = C;
  const dim3 block(256);
  ElArgs A;
  memset(&A, 0, sizeof(A));
  A.cells = m->cells; A.x = m->x; A.ctags = m->cell_tags; A.c2f = m->c2f; A.f2c = m->f2c;
  A.phi = dphi; A.f = df; A.ud = dud; A.nv = (int32_t)m->nv;
  // params = {E_in, nu_in, E_out, nu_out, pen_coef, stab_coef}
  const double E[2] = {params[0], params[2]}, nu[2] = {params[1], params[3]};
  for (int sd = 0; sd < 2; ++sd) {
    A.lam[sd] = E[sd] * nu[sd] / (1.0 + nu[sd]) / (1.0 - 2.0 * nu[sd]);   // data.py:5-10
    A.mu[sd] = E[sd] / 2.0 / (1.0 + nu[sd]);
  }
  const double coef_in = (E[0] / (E[0] + E[1])) * (E[0] / (E[0] + E[1]));  // main.py:188-189
  const double coef_out = (E[1] / (E[0] + E[1])) * (E[1] / (E[0] + E[1]));
  A.coefW[0] = coef_out; A.coefW[1] = coef_in;
  A.gamma = params[4]; A.sigma = params[5];
  uint8_t *flags = nullptr, *bc = nullptr;
  int32_t *scan = nullptr;
  PHX_HIP(phx_malloc(&flags, (size_t)nent)); PHX_HIP(phx_malloc(&bc, (size_t)m->nv));
  PHX_HIP(phx_malloc(&scan, sizeof(int32_t) * (size_t)nent));
  PHX_HIP(hipMemsetAsync(flags, 0, (size_t)nent, m->stream));
  PHX_HIP(hipMemsetAsync(bc, 0, (size_t)m->nv, m->stream));
  A.bc = bc;
  const dim3 gcells((unsigned)phx_div_up(m->nc, 256));
  if (D == 2) k_el_mark_active<2><<<gcells, block, 0, m->stream>>>(m->nc, A, flags);
  else k_el_mark_active<3><<<gcells, block, 0, m->stream>>>(m->nc, A, flags);
  if (nbc > 0) k_el_mark_bc<<<dim3((unsigned)phx_div_up(nbc, 256)), block, 0, m->stream>>>(nbc, D, m->nv, dbcv, flags, bc);
  int32_t n = 0;
  PHX_CHECK(scan_flags(m, flags, scan, nent, &n));
  s->n = n; s->nu = n;
  PHX_REQUIRE(n > 0, PHX_ERR_VALUE, "no active DoF");
  PHX_HIP(phx_malloc(&s->dof_of_vertex_u, sizeof(int32_t) * (size_t)nent));
  PHX_HIP(phx_malloc(&s->dof_of_vertex_p, sizeof(int32_t) * 4));
  PHX_HIP(phx_malloc(&s->full_of_active, sizeof(int64_t) * (size_t)n));
  k_el_numbering<<<dim3((unsigned)phx_div_up(nent, 256)), block, 0, m->stream>>>(nent, flags, scan, s->dof_of_vertex_u, s->full_of_active);
  PHX_HIP(hipStreamSynchronize(m->stream));
  PHX_HIP(phx_free(flags)); PHX_HIP(phx_free(scan));
  A.dofmap = s->dof_of_vertex_u;
  int32_t *l_cut = nullptr, *l_f3 = nullptr, *l_f4 = nullptr;
  int64_t n_cut = 0, n_f3 = 0, n_f4 = 0;
  PHX_CHECK(build_list(m, m->nc, SelCut{m->cell_tags}, &l_cut, &n_cut));
  // per-row slot capacities (a uniform W = 512 would need 450 GB for the 256^3 box of BASELINE configs[3])
  Slots sl;
  sl.W = W;
  int64_t total_slots = 0;
  {
    uint8_t *cutv = nullptr, *wlog = nullptr;
    int64_t *cap = nullptr, *off = nullptr;
    PHX_HIP(phx_malloc(&cutv, (size_t)m->nv));
    PHX_HIP(hipMemsetAsync(cutv, 0, (size_t)m->nv, m->stream));
    if (n_cut > 0) {
      const dim3 g((unsigned)phx_div_up(n_cut, 256));
      if (D == 2) k_mark_cells<3><<<g, block, 0, m->stream>>>(n_cut, l_cut, m->cells, cutv);
      else k_mark_cells<4><<<g, block, 0, m->stream>>>(n_cut, l_cut, m->cells, cutv);
    }
    int wbig = 0;
    while ((1 << wbig) < W) ++wbig;
    PHX_HIP(phx_malloc(&wlog, (size_t)n));
    PHX_HIP(phx_malloc(&cap, sizeof(int64_t) * (size_t)(n + 1)));
    PHX_HIP(phx_malloc(&off, sizeof(int64_t) * (size_t)(n + 1)));
    k_el_row_caps<<<dim3((unsigned)phx_div_up((int64_t)n + 1, 256)), block, 0, m->stream>>>(
        n, s->full_of_active, m->nv, cutv, wbig, wlog, cap);
    PHX_CHECK(exclusive_sum<int64_t>(m, cap, off, (int64_t)n + 1));
    PHX_HIP(hipMemcpy(&total_slots, off + n, sizeof(int64_t), hipMemcpyDeviceToHost));
    PHX_HIP(phx_free(cutv)); PHX_HIP(phx_free(cap));
    sl.off = off;
    sl.wlog = wlog;
  }
  PHX_HIP(phx_malloc(&sl.cols, sizeof(int32_t) * (size_t)total_slots));
  PHX_HIP(phx_malloc(&sl.vals, sizeof(double) * (size_t)total_slots));
  PHX_HIP(phx_malloc(&sl.overflow, sizeof(int)));
  PHX_HIP(hipMemsetAsync(sl.cols, 0xff, sizeof(int32_t) * (size_t)total_slots, m->stream));
  PHX_HIP(hipMemsetAsync(sl.vals, 0, sizeof(double) * (size_t)total_slots, m->stream));
  PHX_HIP(hipMemsetAsync(sl.overflow, 0, sizeof(int), m->stream));
  PHX_HIP(phx_malloc(&s->rhs, sizeof(double) * (size_t)n));
  PHX_HIP(hipMemsetAsync(s->rhs, 0, sizeof(double) * (size_t)n, m->stream));
  A.rhs = s->rhs; A.slots = sl;
  PHX_CHECK(build_list(m, m->nf, SelFacetTagInterior{m->facet_tags, m->f2c, 3}, &l_f3, &n_f3));
  PHX_CHECK(build_list(m, m->nf, SelFacetTagInterior{m->facet_tags, m->f2c, 4}, &l_f4, &n_f4));
  PHX_CHECK(phx_collect_entities(m));
  uint8_t *touched = nullptr;
  if (m->is_box) {
    // rows of vertices no scattering kernel reaches are written dense and sorted by the gather kernel
    PHX_HIP(phx_malloc(&touched, (size_t)m->nv));
    PHX_HIP(phx_malloc(&sl.clean, (size_t)n));
    PHX_HIP(hipMemsetAsync(touched, 0, (size_t)m->nv, m->stream));
    PHX_HIP(hipMemsetAsync(sl.clean, 0, (size_t)n, m->stream));
    const dim3 gb(256);
#define PHX_MARK(NN)                                                                                             \
    do {                                                                                                          \
      if (n_cut) k_mark_cells<NN><<<dim3((unsigned)phx_div_up(n_cut, 256)), gb, 0, m->stream>>>(n_cut, l_cut, m->cells, touched); \
      if (n_f3) k_mark_facet_cells<NN><<<dim3((unsigned)phx_div_up(n_f3, 256)), gb, 0, m->stream>>>(n_f3, l_f3, m->f2c, m->cells, touched); \
      if (n_f4) k_mark_facet_cells<NN><<<dim3((unsigned)phx_div_up(n_f4, 256)), gb, 0, m->stream>>>(n_f4, l_f4, m->f2c, m->cells, touched); \
      for (int sd = 0; sd < 2; ++sd)                                                                              \
        if (m->ent_count[sd])                                                                                     \
          k_mark_entity_cells<NN><<<dim3((unsigned)phx_div_up(m->ent_count[sd], 256)), gb, 0, m->stream>>>(       \
              m->ent_count[sd], m->ent_buf[sd], m->cells, touched);                                               \
    } while (0)
    if (D == 2) PHX_MARK(3); else PHX_MARK(4);
#undef PHX_MARK
  }
  if (!m->is_box) PHX_REQUIRE_GRID(m->nc * 256, "elasticity bulk assembly");
  PHX_REQUIRE_GRID(n_cut * 256, "elasticity cut-cell assembly");
  // PHX_OPT_DETERMINISTIC: every kernel that adds into the slots or the right-hand side runs twice (Slots)
  bool det = false;
  PHX_CHECK(det_alloc(m, sl, total_slots, n, &det));
  for (int pass = det ? 1 : 0; pass <= (det ? 2 : 0); ++pass) {
    sl.pass = pass;
    A.slots = sl;
    if (m->is_box) {
      const BoxDims bd{{m->box_n[0], m->box_n[1], m->box_n[2]}, {m->box_h[0], m->box_h[1], m->box_h[2]}};
      const int64_t nthreads = 2 * (int64_t)D * m->nv;
      const dim3 g((unsigned)phx_div_up(nthreads, 256));
      if (D == 2) k_el_bulk_box<2><<<g, block, 0, m->stream>>>(nthreads, bd, touched, A);
      else k_el_bulk_box<3><<<g, block, 0, m->stream>>>(nthreads, bd, touched, A);
    }
    if (D == 2) {
      if (!m->is_box) k_el_bulk<2><<<dim3((unsigned)m->nc), block, 0, m->stream>>>(m->nc, A);
      if (n_cut) k_el_cut<2><<<dim3((unsigned)n_cut), block, 0, m->stream>>>(n_cut, l_cut, A);
      for (int sd = 0; sd < 2; ++sd)
        if (m->ent_count[sd]) k_el_ds<2><<<dim3((unsigned)m->ent_count[sd]), block, 0, m->stream>>>(m->ent_count[sd], m->ent_buf[sd], sd, A);
      if (n_f3) k_el_facets<2><<<dim3((unsigned)n_f3), block, 0, m->stream>>>(n_f3, l_f3, 0, A);
      if (n_f4) k_el_facets<2><<<dim3((unsigned)n_f4), block, 0, m->stream>>>(n_f4, l_f4, 1, A);
    } else {
      if (!m->is_box) k_el_bulk<3><<<dim3((unsigned)m->nc), block, 0, m->stream>>>(m->nc, A);
      if (n_cut) k_el_cut<3><<<dim3((unsigned)n_cut), block, 0, m->stream>>>(n_cut, l_cut, A);
      for (int sd = 0; sd < 2; ++sd)
        if (m->ent_count[sd]) k_el_ds<3><<<dim3((unsigned)m->ent_count[sd]), block, 0, m->stream>>>(m->ent_count[sd], m->ent_buf[sd], sd, A);
      if (n_f3) k_el_facets<3><<<dim3((unsigned)n_f3), block, 0, m->stream>>>(n_f3, l_f3, 0, A);
      if (n_f4) k_el_facets<3><<<dim3((unsigned)n_f4), block, 0, m->stream>>>(n_f4, l_f4, 1, A);
    }
    PHX_HIP(hipGetLastError());
    if (nbc > 0) k_el_bc_rows<<<dim3((unsigned)phx_div_up(nbc * D, 256)), block, 0, m->stream>>>(nbc, D, A, dbcv);
    PHX_HIP(hipGetLastError());
  }
  PHX_CHECK(det_finish(m, sl, total_slots, n, s->rhs));
  PHX_HIP(hipStreamSynchronize(m->stream));
  PHX_HIP(phx_free(l_cut)); PHX_HIP(phx_free(l_f3)); PHX_HIP(phx_free(l_f4)); PHX_HIP(phx_free(bc));
  PHX_HIP(phx_free(touched));
  const int rc = phx_finish_system(s, sl, (int32_t)nent);
  if (rc != PHX_OK) { phx_system_destroy(s); return rc; }
  *out = s;
  return PHX_OK;
}

extern "C" int phx_assemble_elasticity_if(phx_mesh *m, const double *params, const double *phi_h,
                                          const double *f_h, const double *u_D,
                                          const int32_t *bc_vertices, int64_t nbc, int loc,
                                          phx_system **out) {
  PHX_HIP(hipSetDevice(m->device));
  PHX_REQUIRE(m->cell_type == PHX_TRIANGLE || m->cell_type == PHX_TETRAHEDRON,
              PHX_ERR_NOT_IMPLEMENTED, "assembly supports simplices (triangle, tetrahedron) only");
  PHX_REQUIRE(m->have_cell_tags && m->have_facet_tags, PHX_ERR_VALUE,
              "cell and facet tags must be computed before assembly");
  PHX_REQUIRE(!m->is_submesh, PHX_ERR_NOT_IMPLEMENTED, "interface elasticity runs in box mode");
  const int D = m->gdim;
  const double *dphi, *df, *dud;
  double *o1, *o2, *o3;
  PHX_CHECK(to_device(m, phi_h, loc, m->nv, &dphi, &o1));
  PHX_CHECK(to_device(m, f_h, loc, (int64_t)D * m->nv, &df, &o2));
  PHX_CHECK(to_device(m, u_D, loc, (int64_t)D * m->nv, &dud, &o3));
  int32_t *dbcv = nullptr;
  if (nbc > 0) {
    if (loc == PHX_DEVICE) dbcv = (int32_t *)bc_vertices;
    else {
      PHX_HIP(phx_malloc(&dbcv, sizeof(int32_t) * (size_t)nbc));
      PHX_HIP(hipMemcpyAsync(dbcv, bc_vertices, sizeof(int32_t) * (size_t)nbc, hipMemcpyHostToDevice, m->stream));
    }
  }
  PHX_CHECK(phx_begin_timing(m));
  int W = 256;  // capacity of the rows of cut-cell vertices (the others take 64); doubled on overflow
  int rc = assemble_el_with_capacity(m, params, dphi, df, dud, dbcv, nbc, W, out);
  if (rc == PHX_ERR_CAPACITY) { W = 512; rc = assemble_el_with_capacity(m, params, dphi, df, dud, dbcv, nbc, W, out); }
  if (rc == PHX_ERR_CAPACITY && W < 1024) rc = assemble_el_with_capacity(m, params, dphi, df, dud, dbcv, nbc, 2 * W, out);
  if (rc == PHX_OK) rc = phx_end_timing(m, 2);
  if (o1) (void)phx_free(o1);
  if (o2) (void)phx_free(o2);
  if (o3) (void)phx_free(o3);
  if (dbcv && loc != PHX_DEVICE) (void)phx_free(dbcv);
  return rc;
}
