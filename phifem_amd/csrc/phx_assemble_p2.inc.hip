// P2 x P2 weak-Dirichlet phi-FEM Poisson (BASELINE configs[2]): included by phx_assemble.hip (same
// translation unit: the kernels share slot_add / Geo / load_cell).
// Forms: demo/weak-dirichlet/flower/main.py:112-151 with primal_degree = 2; the div(grad(.)) terms
// of :123-128 and :150 are live here.  Element integrals use Stroud conical (Gauss-Jacobi) rules
// that are exact for the polynomial degree of each term -- what FFCx generates [3P] -- with the
// P2 basis evaluated analytically from the barycentric coordinates of the quadrature point.
// DoFs: vertex v -> v, edge e -> nv + e (local edge order of basix), p block shifted by nv + ne.

// ---------------------------------------------------------------------------------------------
// host: Gauss-Jacobi nodes by Golub-Welsch (n <= 8), conical product rules in barycentric form
// ---------------------------------------------------------------------------------------------
static void sym_eig_jacobi(int n, double *A, double *V) {
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) V[i * n + j] = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < n; ++p) for (int q = p + 1; q < n; ++q) off += A[p * n + q] * A[p * n + q];
    if (off < 1e-300) break;
    for (int p = 0; p < n; ++p)
      for (int q = p + 1; q < n; ++q) {
        if (fabs(A[p * n + q]) < 1e-320) continue;
        const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * A[p * n + q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
        for (int k = 0; k < n; ++k) {
          const double akp = A[k * n + p], akq = A[k * n + q];
          A[k * n + p] = c * akp - sn * akq;
          A[k * n + q] = sn * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {
          const double apk = A[p * n + k], aqk = A[q * n + k];
          A[p * n + k] = c * apk - sn * aqk;
          A[q * n + k] = sn * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {
          const double vkp = V[k * n + p], vkq = V[k * n + q];
          V[k * n + p] = c * vkp - sn * vkq;
          V[k * n + q] = sn * vkp + c * vkq;
        }
      }
  }
}

// nodes t in (0,1) and weights for  int_0^1 (1-t)^alpha f(t) dt
static void gauss_jacobi01(int n, int alpha, std::vector<double> &t, std::vector<double> &w) {
  std::vector<double> A((size_t)n * n, 0.0), V((size_t)n * n);
  const double a = alpha;
  for (int k = 0; k < n; ++k) {
    A[k * n + k] = (k == 0) ? -a / (a + 2.0) : -a * a / ((2.0 * k + a) * (2.0 * k + a + 2.0));
    if (k > 0) {
      const double b = 2.0 * k * (k + a) / ((2.0 * k + a) * sqrt((2.0 * k + a - 1.0) * (2.0 * k + a + 1.0)));
      A[k * n + k - 1] = A[(k - 1) * n + k] = b;
    }
  }
  sym_eig_jacobi(n, A.data(), V.data());
  const double mu0 = pow(2.0, a + 1.0) / (a + 1.0);
  t.resize(n); w.resize(n);
  for (int i = 0; i < n; ++i) {
    t[i] = 0.5 * (A[i * n + i] + 1.0);
    w[i] = mu0 * V[0 * n + i] * V[0 * n + i] / pow(2.0, a + 1.0);
  }
}

// barycentric points (nq x (d+1)) and weights summing to one
static void conical_rule(int d, int degree, std::vector<double> &lam, std::vector<double> &w) {
  const int n = degree / 2 + 1;
  std::vector<std::vector<double>> ts(d), ws(d);
  for (int k = 0; k < d; ++k) gauss_jacobi01(n, d - 1 - k, ts[k], ws[k]);
  int nq = 1;
  for (int k = 0; k < d; ++k) nq *= n;
  lam.assign((size_t)nq * (d + 1), 0.0);
  w.assign(nq, 0.0);
  double wsum = 0.0;
  for (int q = 0; q < nq; ++q) {
    int idx = q;
    double rem = 1.0, wq = 1.0, xs = 0.0;
    for (int k = 0; k < d; ++k) {
      const int i = idx % n; idx /= n;
      const double x = ts[k][i] * rem;
      rem *= (1.0 - ts[k][i]);
      wq *= ws[k][i];
      lam[(size_t)q * (d + 1) + k + 1] = x;
      xs += x;
    }
    lam[(size_t)q * (d + 1)] = 1.0 - xs;
    w[q] = wq;
    wsum += wq;
  }
  for (int q = 0; q < nq; ++q) w[q] /= wsum;
}

struct DevRule {
  int nq;
  const double *lam;  // [nq][d+1] (cell rules) or [nq][d] (facet rules)
  const double *w;
};

static int upload_rule(phx_mesh *m, int d, int degree, DevRule *r, std::vector<void *> &keep) {
  std::vector<double> lam, w;
  conical_rule(d, degree, lam, w);
  double *dl = nullptr, *dw = nullptr;
  PHX_HIP(phx_malloc(&dl, sizeof(double) * lam.size()));
  PHX_HIP(phx_malloc(&dw, sizeof(double) * w.size()));
  PHX_HIP(hipMemcpyAsync(dl, lam.data(), sizeof(double) * lam.size(), hipMemcpyHostToDevice, m->stream));
  PHX_HIP(hipMemcpyAsync(dw, w.data(), sizeof(double) * w.size(), hipMemcpyHostToDevice, m->stream));
  PHX_HIP(hipStreamSynchronize(m->stream));
  keep.push_back(dl); keep.push_back(dw);
  r->nq = (int)w.size(); r->lam = dl; r->w = dw;
  return PHX_OK;
}

// ---------------------------------------------------------------------------------------------
// device: P2 basis on a simplex from barycentric coordinates
// ---------------------------------------------------------------------------------------------
__constant__ int c_tet_edge_a[6][2] = {{2, 3}, {1, 3}, {1, 2}, {0, 3}, {0, 2}, {0, 1}};
__constant__ int c_facet_verts3[4][3] = {{1, 2, 3}, {0, 2, 3}, {0, 1, 3}, {0, 1, 2}};
__constant__ int c_facet_verts2[3][2] = {{1, 2}, {0, 2}, {0, 1}};

template <int D>
struct P2B {
  static constexpr int N = D + 1, NE = D == 3 ? 6 : 3, NB = N + NE;
  __device__ static __forceinline__ void edge(int k, int &a, int &b) {
    if (D == 3) { a = c_tet_edge_a[k][0]; b = c_tet_edge_a[k][1]; }
    else { a = k == 0 ? 1 : 0; b = k == 2 ? 1 : 2; }
  }
  __device__ static __forceinline__ double val(int r, const double *lam) {
    if (r < N) return lam[r] * (2.0 * lam[r] - 1.0);
    int a, b; edge(r - N, a, b);
    return 4.0 * lam[a] * lam[b];
  }
  // grad N_r = sum_m c[m] g_m
  __device__ static __forceinline__ void gradc(int r, const double *lam, double *c) {
    for (int m = 0; m < N; ++m) c[m] = 0.0;
    if (r < N) { c[r] = 4.0 * lam[r] - 1.0; return; }
    int a, b; edge(r - N, a, b);
    c[a] = 4.0 * lam[b];
    c[b] = 4.0 * lam[a];
  }
  // Laplacian (constant on an affine cell), GG[m][n] = g_m . g_n
  __device__ static __forceinline__ double lapl(int r, const double (*GG)[N]) {
    if (r < N) return 4.0 * GG[r][r];
    int a, b; edge(r - N, a, b);
    return 8.0 * GG[a][b];
  }
  // nodal interpolant of degree kdeg (1 or 2) at lam
  __device__ static __forceinline__ double interp(int kdeg, const double *lam, const double *nod) {
    double v = 0.0;
    if (kdeg == 1) { for (int i = 0; i < N; ++i) v += lam[i] * nod[i]; return v; }
    for (int r = 0; r < NB; ++r) v += val(r, lam) * nod[r];
    return v;
  }
};

struct P2Args {
  AsmArgs A;              // du/dp are indexed by ENTITY (vertex, or nv + edge); A.nv holds nv + ne
  const int32_t *c2e;
  int32_t nvert;          // nv
  int kphi;               // degree of phi_h (1: values at vertices, 2: at vertices then edges)
  int pneg;               // structured systems: the column key of p DoF e is -2 - e (phx_slot_view::pneg)
  DevRule cell, cut, facet;
};

template <int D>
__device__ __forceinline__ void p2_cell_dofs(const P2Args &P, int64_t c, const int32_t *v, int32_t *dof) {
  using B = P2B<D>;
  for (int i = 0; i < B::N; ++i) dof[i] = v[i];
  for (int k = 0; k < B::NE; ++k) dof[B::N + k] = P.nvert + P.c2e[c * B::NE + k];
}

// structured systems: rows the solver applies from a stencil own no slots (their offset is a shared dummy slot)
__device__ __forceinline__ bool p2_row_skipped(const P2Args &P, int32_t row) {
  return row < 0 || (P.A.c0 && P.A.c0[row]);
}

template <int D>
__device__ __forceinline__ void gram(const Geo<D> &G, double (*GG)[D + 1]) {
  for (int m = 0; m <= D; ++m)
    for (int n = 0; n <= D; ++n) {
      double s = 0.0;
      for (int d = 0; d < D; ++d) s += G.g[m][d] * G.g[n][d];
      GG[m][n] = s;
    }
}

template <int D>
__global__ void k_p2_mark_active(int64_t nc, P2Args P, uint8_t *__restrict__ fu, uint8_t *__restrict__ fp) {
  using B = P2B<D>;
  const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (c >= nc) return;
  const int t = P.A.ctags[c] & PHX_TAG_MASK;
  if (t != 1 && t != 2) return;
  int32_t v[B::N], dof[B::NB];
  for (int i = 0; i < B::N; ++i) v[i] = P.A.cells[c * B::N + i];
  p2_cell_dofs<D>(P, c, v, dof);
  for (int r = 0; r < B::NB; ++r) { fu[dof[r]] = 1; if (t == 2) fp[dof[r]] = 1; }
}

// --- dx((1,2)): main.py:113 stiffness and :143 source; GS lanes per cell, lane = (r, s) ----------
template <int D, int GS>
__global__ void __launch_bounds__(256) k_p2_cells(int64_t nlist, const int32_t *__restrict__ list, P2Args P) {
  using B = P2B<D>;
  const int64_t gid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t e = gid / GS;
  const int l = (int)(gid % GS);
  if (e >= nlist || l >= B::NB * B::NB) return;
  const int r = l / B::NB, s = l % B::NB;
  const int64_t c = list[e];
  int32_t v[B::N], dof[B::NB];
  double X[B::N][D];
  load_cell<D>(P.A.cells, P.A.x, c, v, X);
  p2_cell_dofs<D>(P, c, v, dof);
  if (p2_row_skipped(P, P.A.du[dof[r]])) return;
  Geo<D> G;
  simplex_geometry<D>(X, G);
  double GG[B::N][B::N];
  gram<D>(G, GG);
  double acc = 0.0, rhs = 0.0;
  double fn[B::NB];
  if (s == 0) for (int b = 0; b < B::NB; ++b) fn[b] = P.A.f[dof[b]];
  for (int q = 0; q < P.cell.nq; ++q) {
    const double *lam = P.cell.lam + (int64_t)q * B::N;
    double cr[B::N], cs[B::N];
    B::gradc(r, lam, cr);
    B::gradc(s, lam, cs);
    double k = 0.0;
    for (int m = 0; m < B::N; ++m)
      for (int n = 0; n < B::N; ++n) k += cr[m] * cs[n] * GG[m][n];
    acc += P.cell.w[q] * k;
    if (s == 0) rhs += P.cell.w[q] * B::interp(2, lam, fn) * B::val(r, lam);
  }
  const int32_t row = P.A.du[dof[r]];
  if (p2_row_skipped(P, row)) return;
  slot_add(P.A.slots, row, dof[s], acc * G.vol);
  if (s == 0) slot_rhs_add(P.A.slots, P.A.rhs, row, rhs * G.vol);
}

// --- dx(2): penalisation main.py:115-122,144-149 and div(grad) terms :123-128,150 -----------------
// one 256-thread block per cut cell, thread -> entries of the (2 NB)^2 mixed tensor
template <int D>
__global__ void __launch_bounds__(256) k_p2_cut(int64_t nlist, const int32_t *__restrict__ list, P2Args P) {
  using B = P2B<D>;
  const int64_t e = blockIdx.x;
  if (e >= nlist) return;
  const int64_t c = list[e];
  int32_t v[B::N], dof[B::NB];
  double X[B::N][D];
  load_cell<D>(P.A.cells, P.A.x, c, v, X);
  p2_cell_dofs<D>(P, c, v, dof);
  Geo<D> G;
  simplex_geometry<D>(X, G);
  double GG[B::N][B::N];
  gram<D>(G, GG);
  double ph[B::NB];
  const int nphi = P.kphi == 1 ? B::N : B::NB;
  for (int b = 0; b < nphi; ++b) ph[b] = P.A.phi[dof[b]];
  const double h1 = 1.0 / G.h;
  const double gam = P.A.gamma * G.vol;
  constexpr int M = 2 * B::NB;
  for (int idx = threadIdx.x; idx < M * M; idx += blockDim.x) {
    const int a = idx / M, b = idx % M;
    const int r = a % B::NB, s = b % B::NB;
    const bool ap = a >= B::NB, bp = b >= B::NB;
    const int e_phi = (ap ? 1 : 0) + (bp ? 1 : 0);
    double acc = 0.0;
    for (int q = 0; q < P.cut.nq; ++q) {
      const double *lam = P.cut.lam + (int64_t)q * B::N;
      double wv = P.cut.w[q] * B::val(r, lam) * B::val(s, lam);
      if (e_phi) {
        const double pq = B::interp(P.kphi, lam, ph);
        wv *= e_phi == 2 ? pq * pq : pq;
      }
      acc += wv;
    }
    double val;
    if (e_phi == 0) val = gam * h1 * h1 * acc + P.A.sigma * G.h * G.h * G.vol * B::lapl(r, GG) * B::lapl(s, GG);
    else if (e_phi == 1) val = -gam * h1 * h1 * h1 * acc;
    else val = gam * h1 * h1 * h1 * h1 * acc;
    const int32_t row = ap ? P.A.dp[dof[r]] : P.A.du[dof[r]];
    if (p2_row_skipped(P, row)) continue;
    if (P.pneg) slot_add<true>(P.A.slots, row, bp ? -2 - dof[s] : dof[s], val);
    else slot_add(P.A.slots, row, (bp ? P.A.nv : 0) + dof[s], val);
  }
  // right-hand side: one thread per row of the mixed tensor
  if (threadIdx.x < M) {
    const int a = threadIdx.x, r = a % B::NB;
    const bool ap = a >= B::NB;
    double udn[B::NB], fn[B::NB];
    for (int b = 0; b < B::NB; ++b) { udn[b] = P.A.ud[dof[b]]; fn[b] = P.A.f[dof[b]]; }
    double acc = 0.0, fbar = 0.0;
    for (int q = 0; q < P.cut.nq; ++q) {
      const double *lam = P.cut.lam + (int64_t)q * B::N;
      const double uq = B::interp(2, lam, udn);
      double wv = P.cut.w[q] * uq * B::val(r, lam);
      if (ap) wv *= B::interp(P.kphi, lam, ph);
      acc += wv;
      fbar += P.cut.w[q] * B::interp(2, lam, fn);
    }
    double rv;
    if (!ap) rv = gam * h1 * h1 * acc - P.A.sigma * G.h * G.h * G.vol * fbar * B::lapl(r, GG);   // :147 (v), :150
    else rv = -gam * h1 * h1 * h1 * acc;                                                          // :147 (q)
    const int32_t rrow = ap ? P.A.dp[dof[r]] : P.A.du[dof[r]];
    if (!p2_row_skipped(P, rrow)) slot_rhs_add(P.A.slots, P.A.rhs, rrow, rv);
  }
}

template <int D>
__device__ __forceinline__ void facet_embed(int lf, const double *mu, double *lam) {
  for (int m = 0; m <= D; ++m) lam[m] = 0.0;
  for (int j = 0; j < D; ++j) lam[D == 3 ? c_facet_verts3[lf][j] : c_facet_verts2[lf][j]] = mu[j];
}

// --- ds(100): main.py:114  -int_F (grad u . n) v ;  GS lanes per (cell, local facet) --------------
template <int D, int GS>
__global__ void __launch_bounds__(256) k_p2_ds(int64_t nent, const int64_t *__restrict__ ent_packed,
                                               const int32_t *__restrict__ ent_pairs, P2Args P) {
  using B = P2B<D>;
  const int64_t gid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t e = gid / GS;
  const int l = (int)(gid % GS);
  if (e >= nent || l >= B::NB * B::NB) return;
  const int r = l / B::NB, s = l % B::NB;
  int64_t c;
  int lf;
  if (ent_packed) { c = ent_packed[2 * e + 1] >> 8; lf = (int)(ent_packed[2 * e + 1] & 0xff); }
  else { c = ent_pairs[2 * e]; lf = ent_pairs[2 * e + 1]; }
  int32_t v[B::N], dof[B::NB];
  double X[B::N][D];
  load_cell<D>(P.A.cells, P.A.x, c, v, X);
  p2_cell_dofs<D>(P, c, v, dof);
  Geo<D> G;
  simplex_geometry<D>(X, G);
  double gn = 0.0;
  for (int d = 0; d < D; ++d) gn += G.g[lf][d] * G.g[lf][d];
  gn = sqrt(gn);
  const double area = D * G.vol * gn;
  double gdotn[B::N];  // g_m . n with n = -g_lf/|g_lf|
  for (int m = 0; m < B::N; ++m) {
    double t = 0.0;
    for (int d = 0; d < D; ++d) t += G.g[m][d] * G.g[lf][d];
    gdotn[m] = -t / gn;
  }
  double acc = 0.0;
  for (int q = 0; q < P.facet.nq; ++q) {
    double lam[B::N], cs[B::N];
    facet_embed<D>(lf, P.facet.lam + (int64_t)q * D, lam);
    B::gradc(s, lam, cs);
    double dn = 0.0;
    for (int m = 0; m < B::N; ++m) dn += cs[m] * gdotn[m];
    acc += P.facet.w[q] * B::val(r, lam) * dn;
  }
  if (!p2_row_skipped(P, P.A.du[dof[r]])) slot_add(P.A.slots, P.A.du[dof[r]], dof[s], -area * acc);
}

// --- dS((2,3)): main.py:129-134  sigma avg(h) int_F [grad u . n][grad v . n] ----------------------
// one block per facet; the quadrature points are placed through the "+" cell and located in the
// "-" cell by its barycentric coordinates
template <int D>
__global__ void __launch_bounds__(256) k_p2_facets(int64_t nlist, const int32_t *__restrict__ list, P2Args P) {
  using B = P2B<D>;
  const int64_t e = blockIdx.x;
  if (e >= nlist) return;
  const int64_t f = list[e];
  int32_t dofs[2 * B::NB];
  double Xc[2][B::N][D];
  Geo<D> G[2];
  int lfs[2];
  double gdotn[2][B::N], hsum = 0.0, area = 0.0;
  for (int side = 0; side < 2; ++side) {
    const int64_t c = P.A.f2c[2 * f + side];
    int32_t v[B::N];
    load_cell<D>(P.A.cells, P.A.x, c, v, Xc[side]);
    p2_cell_dofs<D>(P, c, v, dofs + side * B::NB);
    simplex_geometry<D>(Xc[side], G[side]);
    int lf = 0;
    for (int k = 0; k < B::N; ++k)
      if (P.A.c2f[c * B::N + k] == (int32_t)f) lf = k;
    lfs[side] = lf;
    double gn = 0.0;
    for (int d = 0; d < D; ++d) gn += G[side].g[lf][d] * G[side].g[lf][d];
    gn = sqrt(gn);
    if (side == 0) area = D * G[0].vol * gn;
    hsum += G[side].h;
    for (int m = 0; m < B::N; ++m) {
      double t = 0.0;
      for (int d = 0; d < D; ++d) t += G[side].g[m][d] * G[side].g[lf][d];
      gdotn[side][m] = -t / gn;
    }
  }
  const double wgt = P.A.sigma * 0.5 * hsum * area;
  constexpr int M = 2 * B::NB;
  for (int idx = threadIdx.x; idx < M * M; idx += blockDim.x) {
    const int a = idx / M, b = idx % M;
    double acc = 0.0;
    for (int q = 0; q < P.facet.nq; ++q) {
      const double *mu = P.facet.lam + (int64_t)q * D;
      double lamp[B::N], lamm[B::N], xq[D];
      facet_embed<D>(lfs[0], mu, lamp);
      for (int d = 0; d < D; ++d) {
        double t = 0.0;
        for (int m = 0; m < B::N; ++m) t += lamp[m] * Xc[0][m][d];
        xq[d] = t;
      }
      for (int m = 0; m < B::N; ++m) {
        double t = m == 0 ? 1.0 : 0.0;
        for (int d = 0; d < D; ++d) t += G[1].g[m][d] * (xq[d] - Xc[1][0][d]);
        lamm[m] = t;
      }
      double J[2];
      for (int w = 0; w < 2; ++w) {
        const int id = w == 0 ? a : b;
        const int side = id / B::NB;
        double cc[B::N];
        B::gradc(id % B::NB, side == 0 ? lamp : lamm, cc);
        double t = 0.0;
        for (int m = 0; m < B::N; ++m) t += cc[m] * gdotn[side][m];
        J[w] = t;
      }
      acc += P.facet.w[q] * J[0] * J[1];
    }
    if (!p2_row_skipped(P, P.A.du[dofs[a]])) slot_add(P.A.slots, P.A.du[dofs[a]], dofs[b], wgt * acc);
  }
}

#include "phx_assemble_p2s.inc.hip"

static int assemble_p2_with_capacity(phx_mesh *m, double pen_coef, double stab_coef, int kphi,
                                     const double *dphi, const double *df, const double *dud,
                                     int W, phx_system **out) {
  const int D = m->gdim;
  const int64_t nent = m->nv + m->ne;
  // 3-D Kuhn boxes: structured system -- the interior rows are applied from stencils, only the band around Gamma_h is
  // assembled and stored (phx_assemble_p2s.inc.hip).  Not with PHX_OPT_EXPORT_CSR (the export wants every row).
  const bool structured = D == 3 && m->is_box && !m->is_submesh && m->structured != 0 && !m->export_csr;
  PHX_REQUIRE(nent < INT32_MAX - 2 && (structured || 2 * nent < INT32_MAX), PHX_ERR_VALUE,
              "too many P2 DoFs for 32-bit column keys");
  phx_system *s = new phx_system();
  s->mesh = m; s->device = m->device; s->nfull = 2 * nent; s->slot_cap = W; s->nent = nent;
  s->u_p2_block = true;
  const dim3 block(256);
  std::vector<void *> keep;
  P2Args P;
  memset(&P, 0, sizeof(P));
  PHX_CHECK(upload_rule(m, D, 4, &P.cell, keep));
  PHX_CHECK(upload_rule(m, D, 4 + 2 * kphi, &P.cut, keep));
  PHX_CHECK(upload_rule(m, D - 1, 3, &P.facet, keep));
  // facet rules carry D barycentric coordinates per point
  uint8_t *fu = nullptr, *fp = nullptr;
  int32_t *su = nullptr, *sp = nullptr;
  PHX_HIP(phx_malloc(&fu, (size_t)nent)); PHX_HIP(phx_malloc(&fp, (size_t)nent));
  PHX_HIP(phx_malloc(&su, sizeof(int32_t) * (size_t)nent)); PHX_HIP(phx_malloc(&sp, sizeof(int32_t) * (size_t)nent));
  PHX_HIP(hipMemsetAsync(fu, 0, (size_t)nent, m->stream));
  PHX_HIP(hipMemsetAsync(fp, 0, (size_t)nent, m->stream));
  P.A.cells = m->cells; P.A.x = m->x; P.A.ctags = m->cell_tags; P.A.ftags = m->facet_tags;
  P.A.c2f = m->c2f; P.A.f2c = m->f2c; P.A.phi = dphi; P.A.f = df; P.A.ud = dud;
  P.A.gamma = pen_coef; P.A.sigma = stab_coef; P.A.nv = (int32_t)nent;
  P.c2e = m->c2e; P.nvert = (int32_t)m->nv; P.kphi = kphi;
  const dim3 gcells((unsigned)phx_div_up(m->nc, 256));
  if (D == 2) k_p2_mark_active<2><<<gcells, block, 0, m->stream>>>(m->nc, P, fu, fp);
  else k_p2_mark_active<3><<<gcells, block, 0, m->stream>>>(m->nc, P, fu, fp);
  int32_t nu = 0, np = 0;
  PHX_CHECK(scan_flags(m, fu, su, nent, &nu));
  PHX_CHECK(scan_flags(m, fp, sp, nent, &np));
  s->nu = nu; s->n = (int64_t)nu + np;
  PHX_REQUIRE(s->n > 0, PHX_ERR_VALUE, "no active DoF: no cell is tagged 1 or 2");
  PHX_HIP(phx_malloc(&s->dof_of_vertex_u, sizeof(int32_t) * (size_t)nent));
  PHX_HIP(phx_malloc(&s->dof_of_vertex_p, sizeof(int32_t) * (size_t)nent));
  PHX_HIP(phx_malloc(&s->full_of_active, sizeof(int64_t) * (size_t)s->n));
  k_finish_numbering<<<dim3((unsigned)phx_div_up(nent, 256)), block, 0, m->stream>>>(
      nent, fu, fp, su, sp, nu, s->dof_of_vertex_u, s->dof_of_vertex_p, s->full_of_active);
  PHX_HIP(hipStreamSynchronize(m->stream));
  PHX_HIP(phx_free(fu)); PHX_HIP(phx_free(fp)); PHX_HIP(phx_free(su)); PHX_HIP(phx_free(sp));
  Slots sl;
  sl.W = W;
  sl.cols = nullptr; sl.vals = nullptr; sl.overflow = nullptr;
  PHX_HIP(phx_malloc(&s->rhs, sizeof(double) * (size_t)s->n));
  PHX_HIP(hipMemsetAsync(s->rhs, 0, sizeof(double) * (size_t)s->n, m->stream));
  P.A.du = s->dof_of_vertex_u; P.A.dp = s->dof_of_vertex_p; P.A.rhs = s->rhs;
  int32_t *l_om = nullptr, *l_cut = nullptr, *l_fac = nullptr;
  int64_t n_om = 0, n_cut = 0, n_fac = 0;
  PHX_CHECK(build_list(m, m->nc, SelCut{m->cell_tags}, &l_cut, &n_cut));
  PHX_CHECK(build_list(m, m->nf, SelGhostFacet{m->facet_tags, m->f2c}, &l_fac, &n_fac));
  P2SPrep prep;
  if (structured) {
    s->structured = true;
    s->u_unscaled = true;
    P.pneg = 1;
    const int rcp = p2s_prepare(s, P, sl, l_fac, n_fac, W, &prep);
    if (rcp != PHX_OK) { (void)phx_free(l_cut); (void)phx_free(l_fac); phx_system_destroy(s); return rcp; }
    l_om = prep.l_cells; n_om = prep.n_cells;
  } else {
    PHX_HIP(phx_malloc(&sl.cols, sizeof(int32_t) * (size_t)s->n * W));
    PHX_HIP(phx_malloc(&sl.vals, sizeof(double) * (size_t)s->n * W));
    PHX_HIP(phx_malloc(&sl.overflow, sizeof(int)));
    PHX_HIP(hipMemsetAsync(sl.cols, 0xff, sizeof(int32_t) * (size_t)s->n * W, m->stream));
    PHX_HIP(hipMemsetAsync(sl.vals, 0, sizeof(double) * (size_t)s->n * W, m->stream));
    PHX_HIP(hipMemsetAsync(sl.overflow, 0, sizeof(int), m->stream));
    PHX_CHECK(build_list(m, m->nc, SelOmega{m->cell_tags}, &l_om, &n_om));
  }
  // PHX_OPT_DETERMINISTIC: the element kernels run twice (exponent pass, exact accumulation pass; Slots)
  const int64_t nslots = structured ? prep.slot_rows * (int64_t)W + 64 : s->n * (int64_t)W;
  bool det = false;
  PHX_CHECK(det_alloc(m, sl, nslots, s->n, &det));
  const int64_t nds = m->is_submesh ? m->nbf : (phx_collect_entities(m) == PHX_OK ? m->ent_count[0] : -1);
  PHX_REQUIRE(nds >= 0, PHX_ERR_VALUE, "integration entities unavailable");
  for (int pass = det ? 1 : 0; pass <= (det ? 2 : 0); ++pass) {
    sl.pass = pass;
    P.A.slots = sl;
    if (n_om > 0) {
      PHX_REQUIRE_GRID(n_om * (D == 2 ? 64 : 128), "P2 cell assembly");
      if (D == 2) k_p2_cells<2, 64><<<dim3((unsigned)phx_div_up(n_om * 64, 256)), block, 0, m->stream>>>(n_om, l_om, P);
      else k_p2_cells<3, 128><<<dim3((unsigned)phx_div_up(n_om * 128, 256)), block, 0, m->stream>>>(n_om, l_om, P);
    }
    if (n_cut > 0) {
      if (D == 2) k_p2_cut<2><<<dim3((unsigned)n_cut), block, 0, m->stream>>>(n_cut, l_cut, P);
      else k_p2_cut<3><<<dim3((unsigned)n_cut), block, 0, m->stream>>>(n_cut, l_cut, P);
    }
    PHX_HIP(hipGetLastError());
    if (nds > 0) {
      const int64_t *pk = m->is_submesh ? nullptr : m->ent_buf[0];
      const int32_t *pr = m->is_submesh ? m->bfacets : nullptr;
      if (D == 2) k_p2_ds<2, 64><<<dim3((unsigned)phx_div_up(nds * 64, 256)), block, 0, m->stream>>>(nds, pk, pr, P);
      else k_p2_ds<3, 128><<<dim3((unsigned)phx_div_up(nds * 128, 256)), block, 0, m->stream>>>(nds, pk, pr, P);
    }
    if (n_fac > 0) {
      if (D == 2) k_p2_facets<2><<<dim3((unsigned)n_fac), block, 0, m->stream>>>(n_fac, l_fac, P);
      else k_p2_facets<3><<<dim3((unsigned)n_fac), block, 0, m->stream>>>(n_fac, l_fac, P);
    }
    PHX_HIP(hipGetLastError());
  }
  PHX_CHECK(det_finish(m, sl, nslots, s->n, s->rhs));
  PHX_HIP(hipGetLastError());
  PHX_HIP(hipStreamSynchronize(m->stream));
  PHX_HIP(phx_free(l_om)); PHX_HIP(phx_free(l_cut)); PHX_HIP(phx_free(l_fac));
  for (void *p : keep) PHX_HIP(phx_free(p));
  if (structured) {
    int rc = check_overflow(m, sl);
    if (rc == PHX_OK) {
      const phx_slot_view sv{sl.cols, sl.vals, sl.W, sl.clean, sl.off, sl.wlog, true};
      rc = phx_system_build_structured_p2(s, sv, (int32_t)nent, prep.latc0, prep.latc0i);
      (void)free_slots(sl);
    }
    (void)phx_free(prep.latc0); (void)phx_free(prep.latc0i); (void)phx_free(prep.coefM);
    if (rc != PHX_OK) { phx_system_destroy(s); return rc; }
    *out = s;
    return PHX_OK;
  }
  const int rc = phx_finish_system(s, sl, (int32_t)nent);
  if (rc != PHX_OK) { phx_system_destroy(s); return rc; }
  *out = s;
  return PHX_OK;
}

extern "C" int phx_assemble_poisson_wd_p2(phx_mesh *m, double pen_coef, double stab_coef,
                                          const double *phi_h, int phi_degree, const double *f_h,
                                          const double *u_D, int loc, phx_system **out) {
  PHX_HIP(hipSetDevice(m->device));
  PHX_REQUIRE(m->cell_type == PHX_TRIANGLE || m->cell_type == PHX_TETRAHEDRON,
              PHX_ERR_NOT_IMPLEMENTED, "assembly supports simplices (triangle, tetrahedron) only");
  PHX_REQUIRE(m->have_cell_tags && m->have_facet_tags, PHX_ERR_VALUE,
              "cell and facet tags must be computed before assembly");
  PHX_REQUIRE(phi_degree == 1 || phi_degree == 2, PHX_ERR_VALUE, "phi_degree must be 1 or 2");
  PHX_CHECK(phx_mesh_build_edges(m));
  const int64_t nent = m->nv + m->ne;
  const double *dphi, *df, *dud;
  double *o1, *o2, *o3;
  PHX_CHECK(to_device(m, phi_h, loc, phi_degree == 1 ? m->nv : nent, &dphi, &o1));
  PHX_CHECK(to_device(m, f_h, loc, nent, &df, &o2));
  PHX_CHECK(to_device(m, u_D, loc, nent, &dud, &o3));
  PHX_CHECK(phx_begin_timing(m));
  int W = m->gdim == 3 ? 256 : 128;
  int rc = assemble_p2_with_capacity(m, pen_coef, stab_coef, phi_degree, dphi, df, dud, W, out);
  if (rc == PHX_ERR_CAPACITY) rc = assemble_p2_with_capacity(m, pen_coef, stab_coef, phi_degree, dphi, df, dud, 2 * W, out);
  if (rc == PHX_OK) rc = phx_end_timing(m, 2);
  if (o1) (void)phx_free(o1);
  if (o2) (void)phx_free(o2);
  if (o3) (void)phx_free(o3);
  return rc;
}
