// Cell-wise discretisation errors: included by phx_assemble.hip (shares the conical rules,
// Geo / load_cell / gram and the Lagrange bases).  Follows demo/interface-elasticity/main.py:327-383: the
// exact solution and u_h are interpolated into the Lagrange space of degree primal_degree + 2 (:329-335),
// e = I(u_ex) - I(u_h), and the DG0-tested forms  int_K grad e : grad e  (:347-356) and  int_K e . e  (:369-377)
// give one number per cell; the norms of I(u_ex) (:339-345, :363-367) normalise the global errors.
// The reference space is basix's default Lagrange variant (GLL-warped [3P]): on an edge the two interior
// nodes of degree 3 sit at (1 -+ 1/sqrt 5)/2, the face node at the centroid.  I(u_h) = u_h (P1, P2 in P3).

#define PHX_ERR_MAXNB 20  // P3 on a tetrahedron

struct RefSpace3 {
  int d, nb;
  double bary[PHX_ERR_MAXNB][4];   // nodes
  int alpha[PHX_ERR_MAXNB][4];     // homogeneous cubic monomials lambda^alpha
  double C[PHX_ERR_MAXNB][PHX_ERR_MAXNB];  // N_i = sum_a C[a][i] lambda^alpha_a
};

static int ref_space3_build(int d, RefSpace3 *R) {
  R->d = d;
  const int n = d + 1;
  int nb = 0;
  for (int i = 0; i < n; ++i) { for (int m = 0; m < 4; ++m) R->bary[nb][m] = m == i ? 1.0 : 0.0; ++nb; }
  const double g1 = 0.5 * (1.0 - 1.0 / sqrt(5.0)), g2 = 0.5 * (1.0 + 1.0 / sqrt(5.0));
  static const int e2[3][2] = {{1, 2}, {0, 2}, {0, 1}};
  static const int e3[6][2] = {{2, 3}, {1, 3}, {1, 2}, {0, 3}, {0, 2}, {0, 1}};
  const int ne = d == 2 ? 3 : 6;
  for (int k = 0; k < ne; ++k) {
    const int a = d == 2 ? e2[k][0] : e3[k][0], b = d == 2 ? e2[k][1] : e3[k][1];
    for (int s = 0; s < 2; ++s) {
      const double t = s == 0 ? g1 : g2;
      for (int m = 0; m < 4; ++m) R->bary[nb][m] = 0.0;
      R->bary[nb][a] = 1.0 - t; R->bary[nb][b] = t;
      ++nb;
    }
  }
  if (d == 2) {
    for (int m = 0; m < 4; ++m) R->bary[nb][m] = m < 3 ? 1.0 / 3.0 : 0.0;
    ++nb;
  } else {
    for (int f = 0; f < 4; ++f) {   // face f is opposite vertex f
      for (int m = 0; m < 4; ++m) R->bary[nb][m] = m == f ? 0.0 : 1.0 / 3.0;
      ++nb;
    }
  }
  R->nb = nb;
  int na = 0;
  for (int a0 = 3; a0 >= 0; --a0)
    for (int a1 = 3 - a0; a1 >= 0; --a1)
      for (int a2 = 3 - a0 - a1; a2 >= 0; --a2) {
        const int a3 = 3 - a0 - a1 - a2;
        if (d == 2 && a3 != 0) continue;
        R->alpha[na][0] = a0; R->alpha[na][1] = a1; R->alpha[na][2] = a2; R->alpha[na][3] = a3;
        ++na;
      }
  PHX_REQUIRE(na == nb, PHX_ERR_VALUE, "P3 space: %d monomials for %d nodes", na, nb);
  // V[j][a] = lambda_j^alpha_a;  C = V^-1 by Gauss-Jordan with partial pivoting
  double V[PHX_ERR_MAXNB][2 * PHX_ERR_MAXNB];
  for (int j = 0; j < nb; ++j)
    for (int a = 0; a < nb; ++a) {
      double v = 1.0;
      for (int m = 0; m < n; ++m) for (int p = 0; p < R->alpha[a][m]; ++p) v *= R->bary[j][m];
      V[j][a] = v;
      V[j][nb + a] = j == a ? 1.0 : 0.0;
    }
  for (int c = 0; c < nb; ++c) {
    int piv = c;
    for (int r = c + 1; r < nb; ++r) if (fabs(V[r][c]) > fabs(V[piv][c])) piv = r;
    PHX_REQUIRE(fabs(V[piv][c]) > 1e-14, PHX_ERR_VALUE, "singular Vandermonde matrix");
    if (piv != c) for (int k = 0; k < 2 * nb; ++k) std::swap(V[c][k], V[piv][k]);
    const double ip = 1.0 / V[c][c];
    for (int k = 0; k < 2 * nb; ++k) V[c][k] *= ip;
    for (int r = 0; r < nb; ++r) {
      if (r == c) continue;
      const double fct = V[r][c];
      if (fct != 0.0) for (int k = 0; k < 2 * nb; ++k) V[r][k] -= fct * V[c][k];
    }
  }
  for (int a = 0; a < nb; ++a) for (int i = 0; i < nb; ++i) R->C[a][i] = V[a][nb + i];
  return PHX_OK;
}

static double mono_val(const RefSpace3 &R, int a, const double *lam, int skip) {
  double v = 1.0;
  for (int m = 0; m <= R.d; ++m) {
    const int p = R.alpha[a][m] - (m == skip ? 1 : 0);
    for (int k = 0; k < p; ++k) v *= lam[m];
  }
  return v;
}

extern "C" int phx_reference_nodes(int gdim, int degree, double *bary, int *n_nodes) {
  PHX_REQUIRE(gdim == 2 || gdim == 3, PHX_ERR_VALUE, "gdim must be 2 or 3");
  PHX_REQUIRE(degree == 3, PHX_ERR_NOT_IMPLEMENTED, "the reference space of the error evaluation has degree 3");
  RefSpace3 R;
  PHX_CHECK(ref_space3_build(gdim, &R));
  if (n_nodes) *n_nodes = R.nb;
  if (bary)
    for (int j = 0; j < R.nb; ++j) for (int m = 0; m <= gdim; ++m) bary[j * (gdim + 1) + m] = R.bary[j][m];
  return PHX_OK;
}

struct ErrTab {
  int nq, nb, nbh;
  const double *w;    // [nq]
  const double *N;    // [nq][nb]
  const double *dN;   // [nq][nb][D+1]  coefficients of g_m in grad N_j
  const double *Nh;   // [nb][nbh]      basis of u_h at the reference nodes
};

#define ERR_THREADS 128
template <int D>
__global__ void __launch_bounds__(ERR_THREADS)
k_cell_errors(int64_t ncells, const int32_t *__restrict__ list, const int32_t *__restrict__ cells,
              const double *__restrict__ x, const int32_t *__restrict__ c2e, int32_t nvert, int ncomp,
              int degh, int64_t ndh, const double *__restrict__ uh, const double *__restrict__ uref, ErrTab T,
              double *__restrict__ l2, double *__restrict__ h10, double *__restrict__ partial) {
  constexpr int N = D + 1, NE = D == 3 ? 6 : 3, NB = D == 3 ? 20 : 10;
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  double s[4] = {0.0, 0.0, 0.0, 0.0};   // l2, h10, |I u_ex|^2, |grad I u_ex|^2 of this cell
  if (i < ncells) {
    const int64_t c = list ? list[i] : i;
    int32_t v[N], dof[N + NE];
    double X[N][D];
    load_cell<D>(cells, x, c, v, X);
    for (int k = 0; k < N; ++k) dof[k] = v[k];
    if (degh == 2) for (int k = 0; k < NE; ++k) dof[N + k] = nvert + c2e[c * NE + k];
    Geo<D> G;
    simplex_geometry<D>(X, G);
    double GG[N][N];
    gram<D>(G, GG);
    for (int cp = 0; cp < ncomp; ++cp) {
      double e[NB], r[NB], un[N + NE];
      for (int b = 0; b < T.nbh; ++b) un[b] = uh[(int64_t)cp * ndh + dof[b]];
      for (int j = 0; j < NB; ++j) {
        double uj = 0.0;
        for (int b = 0; b < T.nbh; ++b) uj += T.Nh[j * T.nbh + b] * un[b];
        r[j] = uref[((int64_t)i * NB + j) * ncomp + cp];
        e[j] = r[j] - uj;
      }
      for (int q = 0; q < T.nq; ++q) {
        const double *Nq = T.N + (int64_t)q * NB, *dq = T.dN + (int64_t)q * NB * N;
        double se = 0.0, sr = 0.0, ge[N], gr[N];
        for (int m = 0; m < N; ++m) { ge[m] = 0.0; gr[m] = 0.0; }
        for (int j = 0; j < NB; ++j) {
          se += e[j] * Nq[j];
          sr += r[j] * Nq[j];
          for (int m = 0; m < N; ++m) { ge[m] += e[j] * dq[j * N + m]; gr[m] += r[j] * dq[j * N + m]; }
        }
        double he = 0.0, hr = 0.0;
        for (int m = 0; m < N; ++m)
          for (int n = 0; n < N; ++n) { he += ge[m] * ge[n] * GG[m][n]; hr += gr[m] * gr[n] * GG[m][n]; }
        const double w = T.w[q] * G.vol;
        s[0] += w * se * se; s[1] += w * he; s[2] += w * sr * sr; s[3] += w * hr;
      }
    }
    l2[i] = s[0];
    h10[i] = s[1];
  }
  // fixed-order block sums (deterministic): lane tree inside the wave, then wave 0 adds the waves in order
  __shared__ double sh[ERR_THREADS / 64][4];
  for (int k = 0; k < 4; ++k) {
    double t = s[k];
    for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][k] = t;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    double t = 0.0;
    for (int wv = 0; wv < ERR_THREADS / 64; ++wv) t += sh[wv][threadIdx.x];
    partial[(int64_t)blockIdx.x * 4 + threadIdx.x] = t;
  }
}

__global__ void __launch_bounds__(256) k_sum_partials(int64_t nblocks, const double *__restrict__ partial,
                                                      double *__restrict__ out) {
  // one block, fixed summation order: thread t adds blocks t, t + 256, ...; then a tree over the 256 threads
  __shared__ double sh[256];
  for (int k = 0; k < 4; ++k) {
    double t = 0.0;
    for (int64_t b = threadIdx.x; b < nblocks; b += 256) t += partial[b * 4 + k];
    sh[threadIdx.x] = t;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
      __syncthreads();
    }
    if (threadIdx.x == 0) out[k] = sh[0];
    __syncthreads();
  }
}

template <int D>
static int cell_errors_impl(phx_mesh *m, int ncomp, int degh, const double *duh, const double *duref,
                            int64_t ncells, const int32_t *dlist, double *dl2, double *dh10, double *norms_host) {
  constexpr int N = D + 1, NB = D == 3 ? 20 : 10;
  RefSpace3 R;
  PHX_CHECK(ref_space3_build(D, &R));
  std::vector<double> lam, w;
  conical_rule(D, 6, lam, w);
  const int nq = (int)w.size(), nbh = degh == 1 ? N : (D == 3 ? 10 : 6);
  std::vector<double> tab((size_t)nq + (size_t)nq * NB + (size_t)nq * NB * N + (size_t)NB * nbh);
  double *tw = tab.data(), *tN = tw + nq, *tdN = tN + (size_t)nq * NB, *tNh = tdN + (size_t)nq * NB * N;
  for (int q = 0; q < nq; ++q) {
    tw[q] = w[q];
    const double *lq = &lam[(size_t)q * N];
    for (int j = 0; j < NB; ++j) {
      double v = 0.0, dv[N];
      for (int mm = 0; mm < N; ++mm) dv[mm] = 0.0;
      for (int a = 0; a < NB; ++a) {
        const double cf = R.C[a][j];
        v += cf * mono_val(R, a, lq, -1);
        for (int mm = 0; mm < N; ++mm)
          if (R.alpha[a][mm] > 0) dv[mm] += cf * R.alpha[a][mm] * mono_val(R, a, lq, mm);
      }
      tN[(size_t)q * NB + j] = v;
      for (int mm = 0; mm < N; ++mm) tdN[((size_t)q * NB + j) * N + mm] = dv[mm];
    }
  }
  static const int e2[3][2] = {{1, 2}, {0, 2}, {0, 1}};
  static const int e3[6][2] = {{2, 3}, {1, 3}, {1, 2}, {0, 3}, {0, 2}, {0, 1}};
  for (int j = 0; j < NB; ++j) {
    const double *lj = R.bary[j];
    for (int b = 0; b < nbh; ++b) {
      double v;
      if (degh == 1) v = lj[b];
      else if (b < N) v = lj[b] * (2.0 * lj[b] - 1.0);
      else {
        const int k = b - N, a0 = D == 2 ? e2[k][0] : e3[k][0], a1 = D == 2 ? e2[k][1] : e3[k][1];
        v = 4.0 * lj[a0] * lj[a1];
      }
      tNh[(size_t)j * nbh + b] = v;
    }
  }
  hipStream_t st = m->stream;
  double *dtab = nullptr, *partial = nullptr, *dsum = nullptr;
  const int64_t nblocks = phx_div_up(ncells, ERR_THREADS);
  PHX_HIP(phx_malloc(&dtab, sizeof(double) * tab.size()));
  PHX_HIP(phx_malloc(&partial, sizeof(double) * (size_t)nblocks * 4));
  PHX_HIP(phx_malloc(&dsum, sizeof(double) * 4));
  PHX_HIP(hipMemcpyAsync(dtab, tab.data(), sizeof(double) * tab.size(), hipMemcpyHostToDevice, st));
  ErrTab T;
  T.nq = nq; T.nb = NB; T.nbh = nbh;
  T.w = dtab; T.N = dtab + nq; T.dN = T.N + (size_t)nq * NB; T.Nh = T.dN + (size_t)nq * NB * N;
  PHX_REQUIRE_GRID(nblocks * ERR_THREADS, "cell errors");
  k_cell_errors<D><<<dim3((unsigned)nblocks), dim3(ERR_THREADS), 0, st>>>(
      ncells, dlist, m->cells, m->x, m->c2e, (int32_t)m->nv, ncomp, degh, degh == 1 ? m->nv : m->nv + m->ne,
      duh, duref, T, dl2, dh10, partial);
  k_sum_partials<<<dim3(1), dim3(256), 0, st>>>(nblocks, partial, dsum);
  PHX_HIP(hipGetLastError());
  PHX_HIP(hipMemcpyAsync(norms_host, dsum, sizeof(double) * 4, hipMemcpyDeviceToHost, st));
  PHX_HIP(hipStreamSynchronize(st));
  PHX_HIP(phx_free(dtab)); PHX_HIP(phx_free(partial)); PHX_HIP(phx_free(dsum));
  return PHX_OK;
}

extern "C" int phx_cell_errors(phx_mesh *m, int ncomp, int degree_h, const double *u_h, const double *u_ref,
                               int64_t ncells, const int32_t *cell_list, int loc, double *l2_local,
                               double *h10_local, double *norms) {
  PHX_HIP(hipSetDevice(m->device));
  PHX_REQUIRE(m->cell_type == PHX_TRIANGLE || m->cell_type == PHX_TETRAHEDRON, PHX_ERR_NOT_IMPLEMENTED,
              "error evaluation supports simplices (triangle, tetrahedron) only");
  PHX_REQUIRE(ncomp >= 1 && ncomp <= 3, PHX_ERR_VALUE, "ncomp must be 1, 2 or 3");
  PHX_REQUIRE(degree_h == 1 || degree_h == 2, PHX_ERR_VALUE, "degree_h must be 1 or 2");
  PHX_REQUIRE(ncells >= 0 && (cell_list != nullptr || ncells == m->nc), PHX_ERR_VALUE,
              "without a cell list the evaluation covers all %lld cells", (long long)m->nc);
  PHX_REQUIRE(u_h && u_ref && l2_local && h10_local && norms, PHX_ERR_VALUE, "NULL array");
  if (degree_h == 2) PHX_CHECK(phx_mesh_build_edges(m));
  if (ncells == 0) { for (int k = 0; k < 4; ++k) norms[k] = 0.0; return PHX_OK; }
  const int D = m->gdim, NB = D == 3 ? 20 : 10;
  const int64_t ndh = degree_h == 1 ? m->nv : m->nv + m->ne;
  const double *duh, *duref;
  double *o1, *o2, *dl2 = l2_local, *dh10 = h10_local;
  int32_t *dlist = nullptr;
  PHX_CHECK(to_device(m, u_h, loc, ndh * ncomp, &duh, &o1));
  PHX_CHECK(to_device(m, u_ref, loc, ncells * NB * ncomp, &duref, &o2));
  if (loc != PHX_DEVICE) {
    PHX_HIP(phx_malloc(&dl2, sizeof(double) * (size_t)ncells));
    PHX_HIP(phx_malloc(&dh10, sizeof(double) * (size_t)ncells));
    if (cell_list) {
      for (int64_t i = 0; i < ncells; ++i)
        PHX_REQUIRE(cell_list[i] >= 0 && cell_list[i] < m->nc, PHX_ERR_VALUE, "cell index out of range");
      PHX_HIP(phx_malloc(&dlist, sizeof(int32_t) * (size_t)ncells));
      PHX_HIP(hipMemcpyAsync(dlist, cell_list, sizeof(int32_t) * (size_t)ncells, hipMemcpyHostToDevice, m->stream));
    }
  }
  const int32_t *lst = loc == PHX_DEVICE ? cell_list : dlist;
  int rc = D == 2 ? cell_errors_impl<2>(m, ncomp, degree_h, duh, duref, ncells, lst, dl2, dh10, norms)
                  : cell_errors_impl<3>(m, ncomp, degree_h, duh, duref, ncells, lst, dl2, dh10, norms);
  if (rc == PHX_OK && loc != PHX_DEVICE) {
    PHX_HIP(hipMemcpy(l2_local, dl2, sizeof(double) * (size_t)ncells, hipMemcpyDeviceToHost));
    PHX_HIP(hipMemcpy(h10_local, dh10, sizeof(double) * (size_t)ncells, hipMemcpyDeviceToHost));
  }
  if (loc != PHX_DEVICE) { (void)phx_free(dl2); (void)phx_free(dh10); (void)phx_free(dlist); }
  if (o1) (void)phx_free(o1);
  if (o2) (void)phx_free(o2);
  return rc;
}
