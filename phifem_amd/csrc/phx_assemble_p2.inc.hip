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

// quadrature points of the cut-cell rule the tables of k_p2_cut are sized for: degree 4 + 2 kphi <= 8 -> 5^D
#define P2_CUT_NQMAX(D) ((D) == 3 ? 125 : 25)

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

// local edge k of a simplex joins vertices (P2_EA, P2_EB): basix order, as c_tet_edge_a / P2B::edge -- compile-time copies
// for the kernels that select by lane instead of indexing register arrays
template <int D> struct P2E {
  static constexpr int A[6] = {D == 3 ? 2 : 1, D == 3 ? 1 : 0, D == 3 ? 1 : 0, 0, 0, 0};
  static constexpr int Bv[6] = {D == 3 ? 3 : 2, D == 3 ? 3 : 2, D == 3 ? 2 : 1, 3, 2, 1};
};
// coefficient of g_m in grad N_r at the barycentric point lam (lam: memory, any index)
template <int D>
__device__ __forceinline__ double p2_gradc_m(int r, int m, const double *lam) {
  using B = P2B<D>;
  if (r < B::N) return m == r ? 4.0 * lam[r] - 1.0 : 0.0;
  int a, b; B::edge(r - B::N, a, b);
  return m == a ? 4.0 * lam[b] : (m == b ? 4.0 * lam[a] : 0.0);
}

// --- dx((1,2)): main.py:113 stiffness and :143 source ---------------------------------------------
// One wavefront per cell, lane = entry (r, s).  On an affine cell  K[r][s] = |K| sum_{m,n} (g_m . g_n) C[r][s][m][n]  with
// C = sum_q w_q c_r,m(q) c_s,n(q)  (grad N_r = sum_m c_r,m g_m) a property of the RULE: the block tabulates C (and the
// reference mass matrix for the source term, f being a P2 function) once in LDS and an entry costs (D+1)^2 multiply-adds
// instead of a pass over the quadrature points.
template <int D>
__global__ void __launch_bounds__(256) k_p2_cells(int64_t nlist, const int32_t *__restrict__ list, P2Args P) {
  using B = P2B<D>;
  constexpr int NB = B::NB, N = B::N, NN = N * N, NE2 = NB * NB;
  __shared__ double CT[NN][NE2 | 1], Ms[NB][NB + 1];
  __shared__ int32_t wdof[4][NB], wrow[4][NB];
  __shared__ double wfn[4][NB];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  for (int i = threadIdx.x; i < NN * NE2; i += 256) {
    const int mn = i / NE2, rs = i - mn * NE2;
    const int m = mn / N, n = mn - m * N, r = rs / NB, t = rs - r * NB;
    double acc = 0.0;
    for (int q = 0; q < P.cell.nq; ++q) {
      const double *lam = P.cell.lam + (int64_t)q * N;
      acc += P.cell.w[q] * p2_gradc_m<D>(r, m, lam) * p2_gradc_m<D>(t, n, lam);
    }
    CT[mn][rs] = acc;
  }
  for (int i = threadIdx.x; i < NE2; i += 256) {
    const int r = i / NB, t = i - r * NB;
    double acc = 0.0;
    for (int q = 0; q < P.cell.nq; ++q) {
      const double *lam = P.cell.lam + (int64_t)q * N;
      acc += P.cell.w[q] * B::val(r, lam) * B::val(t, lam);
    }
    Ms[r][t] = acc;
  }
  __syncthreads();
  for (int64_t e = blockIdx.x * (int64_t)4 + wave; e < nlist; e += gridDim.x * (int64_t)4) {
    const int64_t c = list[e];
    int32_t v[N];
    double X[N][D];
    load_cell<D>(P.A.cells, P.A.x, c, v, X);
    Geo<D> G;
    simplex_geometry<D>(X, G);
    double GG[N][N];
    gram<D>(G, GG);
    if (lane < NB) {
      const int32_t dl = lane < N ? P.A.cells[c * N + lane] : P.nvert + P.c2e[c * B::NE + (lane - N)];
      wdof[wave][lane] = dl;
      wrow[wave][lane] = P.A.du[dl];
      wfn[wave][lane] = P.A.f[dl];
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int rs = lane; rs < NE2; rs += 64) {
      const int r = rs / NB, t = rs - r * NB;
      const int32_t row = wrow[wave][r];
      if (p2_row_skipped(P, row)) continue;
      double acc = 0.0;
#pragma unroll
      for (int m = 0; m < N; ++m)
#pragma unroll
        for (int n = 0; n < N; ++n) acc = __builtin_fma(GG[m][n], CT[m * N + n][rs], acc);
      slot_add(P.A.slots, row, wdof[wave][t], acc * G.vol);
      if (t == 0) {
        double rhs = 0.0;
        for (int b = 0; b < NB; ++b) rhs = __builtin_fma(wfn[wave][b], Ms[r][b], rhs);
        slot_rhs_add(P.A.slots, P.A.rhs, row, rhs * G.vol);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// --- dx(2): penalisation main.py:115-122,144-149 and div(grad) terms :123-128,150 -----------------
// One WAVEFRONT per cut cell (four per block, a grid-stride loop over the list).  The 2 NB x 2 NB mixed tensor has three
// distinct NB x NB blocks, each symmetric:  T_e[r][s] = sum_q w_q N_r N_s phi_q^e, e = 0 (u,u), 1 (u,p) = (p,u), 2 (p,p).
// The basis values at the quadrature points belong to the RULE, not to the cell: the block tabulates them once in LDS
// (N_r(q) and w_q N_r(q)), T_0 with them; per cell a lane evaluates phi_h at its quadrature points (phi_q -> LDS), then
// lane = pair (r <= s) runs over q for T_1 and T_2 (two table reads, one broadcast read, three multiply-adds per point),
// and the 4 NB^2 slot updates read the tables.  The right-hand sides int u_D N_r phi^e and the cell mean of f follow
// from T_0, T_1 (u_D, f are P2 functions: int u_D N_r phi^e = sum_b u_D[b] T_e[r][b], sum_s N_s = 1) -- no second pass
// over the points.  Round 3 gave every tensor entry a thread that re-evaluated the ten basis functions and phi_h at
// each of the 125 points: 396 ms for the 7e5 cut cells of the 256^3 box, a quarter of the P2 step.
template <int D>
__global__ void __launch_bounds__(256) k_p2_cut(int64_t nlist, const int32_t *__restrict__ list, P2Args P) {
  using B = P2B<D>;
  constexpr int NB = B::NB, N = B::N, NP = NB * (NB + 1) / 2, M = 2 * NB;
  constexpr int NQS = P2_CUT_NQMAX(D);               // odd: rows r of the tables start in distinct banks
  __shared__ double NqT[NB][NQS], NwT[NB][NQS], lamT[N][NQS];
  __shared__ double T0s[NP];
  __shared__ double phq[4][NQS + 1], Tw[4][2][NP + 1];
  const int nq = P.cut.nq;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  for (int i = threadIdx.x; i < NB * nq; i += 256) {
    const int r = i / nq, q = i - r * nq;
    const double v = B::val(r, P.cut.lam + (int64_t)q * N);
    NqT[r][q] = v;
    NwT[r][q] = P.cut.w[q] * v;
  }
  for (int i = threadIdx.x; i < N * nq; i += 256) {
    const int m = i / nq, q = i - m * nq;
    lamT[m][q] = P.cut.lam[(int64_t)q * N + m];
  }
  __syncthreads();
  // this lane's pair (r <= s) of the symmetric blocks
  int pr = 0, ps = 0;
  {
    int rem = lane < NP ? lane : 0;
    while (rem >= NB - pr) { rem -= NB - pr; ++pr; }
    ps = pr + rem;
  }
  if (threadIdx.x < NP) {
    double acc = 0.0;
    for (int q = 0; q < nq; ++q) acc = __builtin_fma(NwT[pr][q], NqT[ps][q], acc);
    T0s[threadIdx.x] = acc;
  }
  __syncthreads();
  auto pair_of = [](int r, int s) { const int a = r < s ? r : s, b = r < s ? s : r; return a * NB - (a * (a - 1)) / 2 + (b - a); };
  // per-wave cell data: lane b < NB owns DoF b of the cell
  __shared__ int32_t wdof[4][NB], wru[4][NB], wrp[4][NB];
  __shared__ double wph[4][NB], wud[4][NB], wfn[4][NB], wlap[4][NB];
  for (int64_t e = blockIdx.x * (int64_t)4 + wave; e < nlist; e += gridDim.x * (int64_t)4) {
    const int64_t c = list[e];
    int32_t v[N];
    double X[N][D];
    load_cell<D>(P.A.cells, P.A.x, c, v, X);
    Geo<D> G;
    simplex_geometry<D>(X, G);
    if (lane < NB) {
      const int32_t dl = lane < N ? P.A.cells[c * N + lane] : P.nvert + P.c2e[c * B::NE + (lane - N)];
      wdof[wave][lane] = dl;
      wru[wave][lane] = P.A.du[dl];
      wrp[wave][lane] = P.A.dp[dl];
      wph[wave][lane] = (P.kphi == 2 || lane < N) ? P.A.phi[dl] : 0.0;
      wud[wave][lane] = P.A.ud[dl];
      wfn[wave][lane] = P.A.f[dl];
      // Laplacian of N_lane (constant on the cell): 4 g_r.g_r (vertex), 8 g_a.g_b (edge)
      double lp = 0.0;
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const int ia = k < N ? k : P2E<D>::A[k - N], ib = k < N ? k : P2E<D>::Bv[k - N];
        double t = 0.0;
#pragma unroll
        for (int d = 0; d < D; ++d) t = __builtin_fma(G.g[ia][d], G.g[ib][d], t);
        lp = lane == k ? (k < N ? 4.0 : 8.0) * t : lp;
      }
      wlap[wave][lane] = lp;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // phi_h at the quadrature points
    for (int q = lane; q < nq; q += 64) {
      double t = 0.0;
      if (P.kphi == 1) {
#pragma unroll
        for (int i = 0; i < N; ++i) t = __builtin_fma(lamT[i][q], wph[wave][i], t);
      } else {
#pragma unroll
        for (int b = 0; b < NB; ++b) t = __builtin_fma(NqT[b][q], wph[wave][b], t);
      }
      phq[wave][q] = t;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    {
      double t1 = 0.0, t2 = 0.0;
      const double *nw = NwT[pr], *nn = NqT[ps], *pq = phq[wave];
#pragma unroll 5
      for (int q = 0; q < nq; ++q) {
        const double a = nw[q] * nn[q], f = pq[q], af = a * f;
        t1 += af;
        t2 = __builtin_fma(af, f, t2);
      }
      if (lane < NP) { Tw[wave][0][lane] = t1; Tw[wave][1][lane] = t2; }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const double h1 = 1.0 / G.h;
    const double gam = P.A.gamma * G.vol;
    const double shv = P.A.sigma * G.h * G.h * G.vol;
    for (int idx = lane; idx < M * M; idx += 64) {
      const int a = idx / M, b = idx - a * M;
      const bool ap = a >= NB, bp = b >= NB;
      const int r = ap ? a - NB : a, s = bp ? b - NB : b;
      const int e_phi = (ap ? 1 : 0) + (bp ? 1 : 0);
      const int32_t row = ap ? wrp[wave][r] : wru[wave][r];
      if (p2_row_skipped(P, row)) continue;
      const int pi = pair_of(r, s);
      double val;
      if (e_phi == 0) val = gam * h1 * h1 * T0s[pi] + shv * wlap[wave][r] * wlap[wave][s];
      else if (e_phi == 1) val = -gam * h1 * h1 * h1 * Tw[wave][0][pi];
      else val = gam * h1 * h1 * h1 * h1 * Tw[wave][1][pi];
      const int32_t ds = wdof[wave][s];
      if (P.pneg) slot_add<true>(P.A.slots, row, bp ? -2 - ds : ds, val);
      else slot_add(P.A.slots, row, (bp ? P.A.nv : 0) + ds, val);
    }
    // right-hand side: one lane per row of the mixed tensor
    if (lane < M) {
      const int a = lane;
      const bool ap = a >= NB;
      const int r = ap ? a - NB : a;
      double acc = 0.0, fbar = 0.0;
      for (int b = 0; b < NB; ++b) {
        const int pi = pair_of(r, b);
        acc = __builtin_fma(wud[wave][b], ap ? Tw[wave][0][pi] : T0s[pi], acc);
      }
      if (!ap)
        for (int b = 0; b < NB; ++b) {
          double mb = 0.0;                                   // int N_b / |K|
          for (int s2 = 0; s2 < NB; ++s2) mb += T0s[pair_of(b, s2)];
          fbar = __builtin_fma(wfn[wave][b], mb, fbar);
        }
      double rv;
      if (!ap) rv = gam * h1 * h1 * acc - shv * fbar * wlap[wave][r];   // :147 (v), :150
      else rv = -gam * h1 * h1 * h1 * acc;                               // :147 (q)
      const int32_t rrow = ap ? wrp[wave][r] : wru[wave][r];
      if (!p2_row_skipped(P, rrow)) slot_rhs_add(P.A.slots, P.A.rhs, rrow, rv);
    }
    // the next cell overwrites the per-wave tables: every lane is past its reads
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
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
// One wavefront per facet.  The quadrature points are placed through the "+" cell and located in the "-" cell by its
// barycentric coordinates; lane (id, q) evaluates the normal derivative J[id][q] of basis function id (0 .. NB-1 on "+",
// NB .. 2 NB - 1 on "-", with the sign of its side's outward normal) ONCE into LDS, then lane = entry (a, b) sums
// w_q J[a][q] J[b][q].  (Round 3: every entry's thread recomputed both cells' geometry and its two derivatives per point.)
#define P2_FACET_NQMAX 4
template <int D>
__global__ void __launch_bounds__(256) k_p2_facets(int64_t nlist, const int32_t *__restrict__ list, P2Args P) {
  using B = P2B<D>;
  constexpr int NB = B::NB, N = B::N, M = 2 * NB;
  __shared__ double Jw[4][M][P2_FACET_NQMAX + 1];
  __shared__ int32_t wdof[4][M], wrow[4][M];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int nq = P.facet.nq;
  for (int64_t e = blockIdx.x * (int64_t)4 + wave; e < nlist; e += gridDim.x * (int64_t)4) {
    const int64_t f = list[e];
    double Xc[2][N][D];
    Geo<D> G[2];
    int lf0 = 0;
    int64_t cs[2];
    double gdotn[2][N], hsum = 0.0, area = 0.0;
#pragma unroll
    for (int side = 0; side < 2; ++side) {
      const int64_t c = P.A.f2c[2 * f + side];
      cs[side] = c;
      int32_t v[N];
      load_cell<D>(P.A.cells, P.A.x, c, v, Xc[side]);
      simplex_geometry<D>(Xc[side], G[side]);
      int lf = 0;
#pragma unroll
      for (int k = 0; k < N; ++k)
        if (P.A.c2f[c * N + k] == (int32_t)f) lf = k;
      if (side == 0) lf0 = lf;
      // g_lf by selection (lf is wave-uniform but not a compile-time index)
      double gl[D];
#pragma unroll
      for (int d = 0; d < D; ++d) {
        double t = G[side].g[0][d];
#pragma unroll
        for (int k = 1; k < N; ++k) t = lf == k ? G[side].g[k][d] : t;
        gl[d] = t;
      }
      double gn = 0.0;
#pragma unroll
      for (int d = 0; d < D; ++d) gn += gl[d] * gl[d];
      gn = sqrt(gn);
      if (side == 0) area = D * G[0].vol * gn;
      hsum += G[side].h;
#pragma unroll
      for (int m = 0; m < N; ++m) {
        double t = 0.0;
#pragma unroll
        for (int d = 0; d < D; ++d) t += G[side].g[m][d] * gl[d];
        gdotn[side][m] = -t / gn;
      }
    }
    const double wgt = P.A.sigma * 0.5 * hsum * area;
    if (lane < M) {
      const int side = lane >= NB ? 1 : 0, r = lane - side * NB;
      const int64_t c = cs[side];
      const int32_t dl = r < N ? P.A.cells[c * N + r] : P.nvert + P.c2e[c * B::NE + (r - N)];
      wdof[wave][lane] = dl;
      wrow[wave][lane] = P.A.du[dl];
    }
    for (int i = lane; i < M * nq; i += 64) {
      const int id = i / nq, q = i - id * nq;
      const int side = id >= NB ? 1 : 0, r = id - side * NB;
      const double *mu = P.facet.lam + (int64_t)q * D;
      // the point in the "+" cell: the facet's vertices are the cell's vertices but lf0, in order
      double lamp[N], xq[D], lam[N], gd[N];
#pragma unroll
      for (int m = 0; m < N; ++m) lamp[m] = m == lf0 ? 0.0 : mu[m < lf0 ? m : m - 1];
#pragma unroll
      for (int d = 0; d < D; ++d) {
        double t = 0.0;
#pragma unroll
        for (int m = 0; m < N; ++m) t += lamp[m] * Xc[0][m][d];
        xq[d] = t;
      }
#pragma unroll
      for (int m = 0; m < N; ++m) {
        double t = m == 0 ? 1.0 : 0.0;
#pragma unroll
        for (int d = 0; d < D; ++d) t += G[1].g[m][d] * (xq[d] - Xc[1][0][d]);
        lam[m] = side ? t : lamp[m];
        gd[m] = side ? gdotn[1][m] : gdotn[0][m];
      }
      // grad N_r . n = sum_m c_m (g_m . n)
      double Jv = 0.0;
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const int ia = k < N ? k : P2E<D>::A[k - N], ib = k < N ? k : P2E<D>::Bv[k - N];
        const double t = k < N ? (4.0 * lam[k] - 1.0) * gd[k] : 4.0 * (lam[ib] * gd[ia] + lam[ia] * gd[ib]);
        Jv = r == k ? t : Jv;
      }
      Jw[wave][id][q] = Jv;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int idx = lane; idx < M * M; idx += 64) {
      const int a = idx / M, b = idx - a * M;
      const int32_t row = wrow[wave][a];
      if (p2_row_skipped(P, row)) continue;
      double acc = 0.0;
      for (int q = 0; q < nq; ++q) acc += P.facet.w[q] * Jw[wave][a][q] * Jw[wave][b][q];
      slot_add(P.A.slots, row, wdof[wave][b], wgt * acc);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
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
  PHX_REQUIRE(P.cut.nq <= P2_CUT_NQMAX(D), PHX_ERR_VALUE, "cut-cell rule larger than the tables of k_p2_cut");
  PHX_REQUIRE(P.facet.nq <= P2_FACET_NQMAX, PHX_ERR_VALUE, "facet rule larger than the tables of k_p2_facets");
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
      const dim3 gom((unsigned)std::min<int64_t>(phx_div_up(n_om, 4), 4096));
      if (D == 2) k_p2_cells<2><<<gom, block, 0, m->stream>>>(n_om, l_om, P);
      else k_p2_cells<3><<<gom, block, 0, m->stream>>>(n_om, l_om, P);
    }
    if (n_cut > 0) {
      const dim3 gcut((unsigned)std::min<int64_t>(phx_div_up(n_cut, 4), 4096));
      if (D == 2) k_p2_cut<2><<<gcut, block, 0, m->stream>>>(n_cut, l_cut, P);
      else k_p2_cut<3><<<gcut, block, 0, m->stream>>>(n_cut, l_cut, P);
    }
    PHX_HIP(hipGetLastError());
    if (nds > 0) {
      const int64_t *pk = m->is_submesh ? nullptr : m->ent_buf[0];
      const int32_t *pr = m->is_submesh ? m->bfacets : nullptr;
      if (D == 2) k_p2_ds<2, 64><<<dim3((unsigned)phx_div_up(nds * 64, 256)), block, 0, m->stream>>>(nds, pk, pr, P);
      else k_p2_ds<3, 128><<<dim3((unsigned)phx_div_up(nds * 128, 256)), block, 0, m->stream>>>(nds, pk, pr, P);
    }
    if (n_fac > 0) {
      const dim3 gfac((unsigned)std::min<int64_t>(phx_div_up(n_fac, 4), 4096));
      if (D == 2) k_p2_facets<2><<<gfac, block, 0, m->stream>>>(n_fac, l_fac, P);
      else k_p2_facets<3><<<gfac, block, 0, m->stream>>>(n_fac, l_fac, P);
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
