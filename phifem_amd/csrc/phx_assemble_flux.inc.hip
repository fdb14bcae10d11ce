// Neumann / Robin phi-FEM Poisson, mixed (u, y, p) in P1 x P1^d x DG0 with a P2 level-set: included by
// phx_assemble.hip after the P2 / strong-Dirichlet includes (shares P2B, the conical rules, slot_add, Geo).
// Forms: demo/robin/square/main.py:112-168 (robin_coef = 0 and the facet term on dS(3) give the formulation
// of demo/neumann/square/main.py:113-158 on simplices; that demo's quadrilateral cells are not covered).
//   B(u,y,p) = y . grad phi - |grad phi| kappa u + h^-1 p phi,   boundary condition  du/dn + kappa u = g.
// DoFs: u at vertex v -> v, y_k at vertex v -> (1 + k) nv + v, p on cell c -> (1 + D) nv + c.
// The cut-cell integrals (|grad phi_h| is not polynomial) use the conical rule of degree `qdeg` (10: UFL's
// estimate for the Robin integrand); everything else is closed form.

struct FxArgs {
  const int32_t *cells, *c2f, *f2c, *c2e;
  const double *x, *phi, *f, *g;
  const int8_t *ctags;
  const int32_t *dofmap;
  int64_t nv;
  double gamma, sigma, kappa;
  double *rhs;
  Slots slots;
  DevRule cut;
};

template <int D>
__global__ void k_fx_mark_active(int64_t nc, FxArgs A, uint8_t *__restrict__ flags) {
  const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (c >= nc) return;
  const int t = A.ctags[c] & PHX_TAG_MASK;
  if (t != 1 && t != 2) return;
  for (int i = 0; i <= D; ++i) {
    const int64_t v = A.cells[c * (D + 1) + i];
    flags[v] = 1;
    if (t == 2) for (int k = 0; k < D; ++k) flags[(1 + k) * A.nv + v] = 1;
  }
  if (t == 2) flags[(1 + D) * A.nv + c] = 1;
}

// --- dx((1,2)): robin main.py:115 (grad u . grad v + u v) and :151 (f v); 16 lanes per cell ----------
template <int D>
__global__ void __launch_bounds__(256) k_fx_bulk(int64_t nlist, const int32_t *__restrict__ list, FxArgs A) {
  constexpr int N = D + 1;
  const int64_t gid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t e = gid / 16;
  const int l = (int)(gid % 16);
  if (e >= nlist || l >= N * N) return;
  const int i = l / N, j = l % N;
  const int64_t c = list[e];
  int32_t v[N];
  double X[N][D];
  load_cell<D>(A.cells, A.x, c, v, X);
  Geo<D> G;
  simplex_geometry<D>(X, G);
  double gg = 0.0;
  for (int d = 0; d < D; ++d) gg += G.g[i][d] * G.g[j][d];
  const double mass = (i == j ? 2.0 : 1.0) / ((D + 1) * (D + 2));
  slot_add(A.slots, A.dofmap[v[i]], v[j], G.vol * (gg + mass));
  if (j == 0) {
    double s = 0.0;
    for (int k = 0; k < N; ++k) s += A.f[v[k]] * (i == k ? 2.0 : 1.0);
    unsafeAtomicAdd(&A.rhs[A.dofmap[v[i]]], G.vol * s / ((D + 1) * (D + 2)));
  }
}

// --- dx(2): main.py:118-133 and :152-165; one block per cut cell, threads walk the M x M tensor ---------
template <int D>
struct FxLocal {   // functionals of local DoF a at one quadrature point
  double U, DY, B, T1[D];
};
template <int D>
__device__ __forceinline__ void fx_eval(int a, const double *lam, const Geo<D> &G, double phq, const double *gphi,
                                        double ngp, double h1, double kappa, FxLocal<D> &o) {
  constexpr int N = D + 1;
  o.U = 0.0; o.DY = 0.0; o.B = 0.0;
  for (int d = 0; d < D; ++d) o.T1[d] = 0.0;
  if (a < N) {                       // u_i
    o.U = lam[a];
    for (int d = 0; d < D; ++d) o.T1[d] = G.g[a][d];
    o.B = -kappa * ngp * lam[a];
  } else if (a < N * (1 + D)) {      // y_{k,i}
    const int k = (a - N) / N, i = (a - N) % N;
    o.T1[k] = lam[i];
    o.DY = G.g[i][k];
    o.B = lam[i] * gphi[k];
  } else {                           // p
    o.B = phq * h1;
  }
}

template <int D>
__global__ void __launch_bounds__(256) k_fx_cut(int64_t nlist, const int32_t *__restrict__ list, FxArgs A) {
  using B2 = P2B<D>;
  constexpr int N = D + 1, M = N * (1 + D) + 1;
  const int64_t e = blockIdx.x;
  if (e >= nlist) return;
  const int64_t c = list[e];
  int32_t v[N];
  double X[N][D];
  load_cell<D>(A.cells, A.x, c, v, X);
  Geo<D> G;
  simplex_geometry<D>(X, G);
  double ph[B2::NB];
  for (int i = 0; i < N; ++i) ph[i] = A.phi[v[i]];
  for (int k = 0; k < B2::NE; ++k) ph[N + k] = A.phi[A.nv + A.c2e[c * B2::NE + k]];
  const double h1 = 1.0 / G.h;
  auto full = [&](int a) -> int32_t {
    if (a < N) return v[a];
    if (a < N * (1 + D)) return (int32_t)((1 + (a - N) / N) * A.nv + v[(a - N) % N]);
    return (int32_t)((1 + D) * A.nv + c);
  };
  for (int idx = threadIdx.x; idx < M * M + M; idx += blockDim.x) {
    const bool is_rhs = idx >= M * M;
    const int a = is_rhs ? idx - M * M : idx / M, b = is_rhs ? 0 : idx % M;
    double fn[N], gn[N];
    if (is_rhs) for (int i = 0; i < N; ++i) { fn[i] = A.f[v[i]]; gn[i] = A.g[v[i]]; }
    double acc = 0.0;
    for (int q = 0; q < A.cut.nq; ++q) {
      const double *lam = A.cut.lam + (int64_t)q * N;
      PhiAt<D> pq;
      phi_eval<D>(2, lam, ph, pq);
      double gphi[D], n2 = 0.0;
      for (int d = 0; d < D; ++d) {
        double t = 0.0;
        for (int m = 0; m < N; ++m) t += pq.c[m] * G.g[m][d];
        gphi[d] = t;
        n2 += t * t;
      }
      const double ngp = sqrt(n2);
      FxLocal<D> la;
      fx_eval<D>(a, lam, G, pq.v, gphi, ngp, h1, A.kappa, la);
      if (is_rhs) {
        double fq = 0.0, gq = 0.0;
        for (int i = 0; i < N; ++i) { fq += lam[i] * fn[i]; gq += lam[i] * gn[i]; }
        acc += A.cut.w[q] * (-h1 * h1 * gq * ngp * la.B + fq * (la.DY + la.U));
      } else {
        FxLocal<D> lb;
        fx_eval<D>(b, lam, G, pq.v, gphi, ngp, h1, A.kappa, lb);
        double t1 = 0.0;
        for (int d = 0; d < D; ++d) t1 += la.T1[d] * lb.T1[d];
        acc += A.cut.w[q] * (t1 + (la.DY + la.U) * (lb.DY + lb.U) + h1 * h1 * la.B * lb.B);
      }
    }
    const double val = A.gamma * G.vol * acc;
    const int32_t row = A.dofmap[full(a)];
    if (is_rhs) unsafeAtomicAdd(&A.rhs[row], val);
    else slot_add(A.slots, row, full(b), val);
  }
}

// --- ds: main.py:116  int_F (y . n) v; 64 lanes per (cell, local facet): lane = (i, j, k) ----------------
template <int D>
__global__ void __launch_bounds__(256) k_fx_ds(int64_t nent, const int64_t *__restrict__ ent_packed,
                                               const int32_t *__restrict__ ent_pairs, FxArgs A) {
  constexpr int N = D + 1;
  const int64_t gid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t e = gid / 64;
  const int l = (int)(gid % 64);
  if (e >= nent || l >= N * N * D) return;
  const int k = l / (N * N), i = (l / N) % N, j = l % N;
  int64_t c;
  int lf;
  if (ent_packed) { c = ent_packed[2 * e + 1] >> 8; lf = (int)(ent_packed[2 * e + 1] & 0xff); }
  else { c = ent_pairs[2 * e]; lf = ent_pairs[2 * e + 1]; }
  int32_t v[N];
  double X[N][D];
  load_cell<D>(A.cells, A.x, c, v, X);
  Geo<D> G;
  simplex_geometry<D>(X, G);
  double gn = 0.0;
  for (int d = 0; d < D; ++d) gn += G.g[lf][d] * G.g[lf][d];
  gn = sqrt(gn);
  const double area = D * G.vol * gn;
  // int_F lam_i lam_j = area (1 + delta_ij) / (D (D + 1)) for i, j on the facet, 0 when either is lf
  const double m = (i == lf || j == lf) ? 0.0 : (i == j ? 2.0 : 1.0) / (D * (D + 1));
  const int32_t col = (int32_t)((1 + k) * A.nv + v[j]);
  if (A.dofmap[col] < 0) return;  // y lives on cut cells only (a facet of an inside cell on the box boundary)
  slot_add(A.slots, A.dofmap[v[i]], col, area * m * (-G.g[lf][k] / gn));
}

// --- dS(tag): main.py:135-143  sigma avg(h) int_F [grad u . n][grad v . n]; 64 lanes per facet ------------
template <int D>
__global__ void __launch_bounds__(256) k_fx_facets(int64_t nlist, const int32_t *__restrict__ list, FxArgs A) {
  constexpr int N = D + 1;
  const int64_t gid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t e = gid / 64;
  const int l = (int)(gid % 64);
  if (e >= nlist || l >= 4 * N * N) return;
  const int a = l / (2 * N), b = l % (2 * N);
  const int64_t f = list[e];
  int32_t dofs[2 * N];
  double J[2 * N], hsum = 0.0, area = 0.0;
  for (int side = 0; side < 2; ++side) {
    const int64_t c = A.f2c[2 * f + side];
    int32_t v[N];
    double X[N][D];
    load_cell<D>(A.cells, A.x, c, v, X);
    Geo<D> G;
    simplex_geometry<D>(X, G);
    int lf = 0;
    for (int k = 0; k < N; ++k)
      if (A.c2f[c * N + k] == (int32_t)f) lf = k;
    double gn = 0.0;
    for (int d = 0; d < D; ++d) gn += G.g[lf][d] * G.g[lf][d];
    gn = sqrt(gn);
    if (side == 0) area = D * G.vol * gn;
    hsum += G.h;
    for (int i = 0; i < N; ++i) {
      double t = 0.0;
      for (int d = 0; d < D; ++d) t += G.g[i][d] * G.g[lf][d];
      J[side * N + i] = -t / gn;
      dofs[side * N + i] = v[i];
    }
  }
  slot_add(A.slots, A.dofmap[dofs[a]], dofs[b], A.sigma * 0.5 * hsum * area * J[a] * J[b]);
}

static int assemble_flux_quad_with_capacity(phx_mesh *m, const double *params, int facet_tag, int nq,
                                            const double *dphi, const double *df, const double *dg, int W,
                                            phx_system **out);

static int assemble_flux_with_capacity(phx_mesh *m, const double *params, int facet_tag, int qdeg,
                                       const double *dphi, const double *df, const double *dg, int W,
                                       phx_system **out) {
  const int D = m->gdim;
  const int64_t nent = (int64_t)(1 + D) * m->nv + m->nc;
  PHX_REQUIRE(nent < INT32_MAX, PHX_ERR_VALUE, "too many DoFs for 32-bit column keys");
  phx_system *s = new phx_system();
  s->mesh = m; s->device = m->device; s->nfull = nent; s->slot_cap = W; s->nent = nent;
  const dim3 block(256);
  std::vector<void *> keep;
  FxArgs A;
  memset(&A, 0, sizeof(A));
  A.cells = m->cells; A.x = m->x; A.ctags = m->cell_tags; A.c2f = m->c2f; A.f2c = m->f2c; A.c2e = m->c2e;
  A.phi = dphi; A.f = df; A.g = dg; A.nv = m->nv;
  A.gamma = params[0]; A.sigma = params[1]; A.kappa = params[2];
  PHX_CHECK(upload_rule(m, D, qdeg, &A.cut, keep));
  uint8_t *flags = nullptr;
  int32_t *scan = nullptr;
  PHX_HIP(phx_malloc(&flags, (size_t)nent));
  PHX_HIP(phx_malloc(&scan, sizeof(int32_t) * (size_t)nent));
  PHX_HIP(hipMemsetAsync(flags, 0, (size_t)nent, m->stream));
  const dim3 gcells((unsigned)phx_div_up(m->nc, 256));
  if (D == 2) k_fx_mark_active<2><<<gcells, block, 0, m->stream>>>(m->nc, A, flags);
  else k_fx_mark_active<3><<<gcells, block, 0, m->stream>>>(m->nc, A, flags);
  int32_t n = 0;
  PHX_CHECK(scan_flags(m, flags, scan, nent, &n));
  PHX_REQUIRE(n > 0, PHX_ERR_VALUE, "no active DoF: no cell is tagged 1 or 2");
  // Jacobi only: the lattice preconditioner on the u rows (which come first) makes BiCGStab worse here -- the
  // y / p penalty blocks dominate (Robin demo, 200^2: 10072 iterations instead of 2800; 400^2 diverges)
  s->n = n; s->nu = n;
  PHX_HIP(phx_malloc(&s->dof_of_vertex_u, sizeof(int32_t) * (size_t)nent));
  PHX_HIP(phx_malloc(&s->dof_of_vertex_p, sizeof(int32_t) * 4));
  PHX_HIP(phx_malloc(&s->full_of_active, sizeof(int64_t) * (size_t)n));
  k_el_numbering<<<dim3((unsigned)phx_div_up(nent, 256)), block, 0, m->stream>>>(nent, flags, scan, s->dof_of_vertex_u, s->full_of_active);
  PHX_HIP(hipStreamSynchronize(m->stream));
  PHX_HIP(phx_free(flags)); PHX_HIP(phx_free(scan));
  A.dofmap = s->dof_of_vertex_u;
  Slots sl;
  sl.W = W;
  PHX_HIP(phx_malloc(&sl.cols, sizeof(int32_t) * (size_t)n * W));
  PHX_HIP(phx_malloc(&sl.vals, sizeof(double) * (size_t)n * W));
  PHX_HIP(phx_malloc(&sl.overflow, sizeof(int)));
  PHX_HIP(hipMemsetAsync(sl.cols, 0xff, sizeof(int32_t) * (size_t)n * W, m->stream));
  PHX_HIP(hipMemsetAsync(sl.vals, 0, sizeof(double) * (size_t)n * W, m->stream));
  PHX_HIP(hipMemsetAsync(sl.overflow, 0, sizeof(int), m->stream));
  PHX_HIP(phx_malloc(&s->rhs, sizeof(double) * (size_t)n));
  PHX_HIP(hipMemsetAsync(s->rhs, 0, sizeof(double) * (size_t)n, m->stream));
  A.rhs = s->rhs; A.slots = sl;
  int32_t *l_om = nullptr, *l_cut = nullptr, *l_fac = nullptr;
  int64_t n_om = 0, n_cut = 0, n_fac = 0;
  PHX_CHECK(build_list(m, m->nc, SelOmega{m->cell_tags}, &l_om, &n_om));
  PHX_CHECK(build_list(m, m->nc, SelCut{m->cell_tags}, &l_cut, &n_cut));
  PHX_CHECK(build_list(m, m->nf, SelFacetTagInterior{m->facet_tags, m->f2c, facet_tag}, &l_fac, &n_fac));
  const int64_t nds = m->is_submesh ? m->nbf : (phx_collect_entities(m) == PHX_OK ? m->ent_count[0] : -1);
  PHX_REQUIRE(nds >= 0, PHX_ERR_VALUE, "integration entities unavailable");
  const int64_t *pk = m->is_submesh ? nullptr : m->ent_buf[0];
  const int32_t *pr = m->is_submesh ? m->bfacets : nullptr;
  PHX_REQUIRE_GRID(n_om * 16, "Neumann / Robin cell assembly");
  PHX_REQUIRE_GRID(n_fac * 64, "Neumann / Robin facet assembly");
  if (D == 2) {
    if (n_om) k_fx_bulk<2><<<dim3((unsigned)phx_div_up(n_om * 16, 256)), block, 0, m->stream>>>(n_om, l_om, A);
    if (n_cut) k_fx_cut<2><<<dim3((unsigned)n_cut), dim3(128), 0, m->stream>>>(n_cut, l_cut, A);
    if (nds) k_fx_ds<2><<<dim3((unsigned)phx_div_up(nds * 64, 256)), block, 0, m->stream>>>(nds, pk, pr, A);
    if (n_fac) k_fx_facets<2><<<dim3((unsigned)phx_div_up(n_fac * 64, 256)), block, 0, m->stream>>>(n_fac, l_fac, A);
  } else {
    if (n_om) k_fx_bulk<3><<<dim3((unsigned)phx_div_up(n_om * 16, 256)), block, 0, m->stream>>>(n_om, l_om, A);
    if (n_cut) k_fx_cut<3><<<dim3((unsigned)n_cut), block, 0, m->stream>>>(n_cut, l_cut, A);
    if (nds) k_fx_ds<3><<<dim3((unsigned)phx_div_up(nds * 64, 256)), block, 0, m->stream>>>(nds, pk, pr, A);
    if (n_fac) k_fx_facets<3><<<dim3((unsigned)phx_div_up(n_fac * 64, 256)), block, 0, m->stream>>>(n_fac, l_fac, A);
  }
  PHX_HIP(hipGetLastError());
  PHX_HIP(hipStreamSynchronize(m->stream));
  PHX_HIP(phx_free(l_om)); PHX_HIP(phx_free(l_cut)); PHX_HIP(phx_free(l_fac));
  for (void *p : keep) PHX_HIP(phx_free(p));
  const int rc = phx_finish_system(s, sl, (int32_t)nent);
  if (rc != PHX_OK) { phx_system_destroy(s); return rc; }
  *out = s;
  return PHX_OK;
}

extern "C" int phx_assemble_poisson_flux(phx_mesh *m, const double *params, int facet_tag, int quadrature_degree,
                                         const double *phi_h, const double *f_h, const double *g_h, int loc,
                                         phx_system **out) {
  PHX_HIP(hipSetDevice(m->device));
  PHX_REQUIRE(m->have_cell_tags && m->have_facet_tags, PHX_ERR_VALUE,
              "cell and facet tags must be computed before assembly");
  PHX_REQUIRE(facet_tag >= 1 && facet_tag <= 6, PHX_ERR_VALUE, "facet_tag must be one of the facet tags 1..6");
  PHX_REQUIRE(quadrature_degree >= 2 && quadrature_degree <= 15, PHX_ERR_VALUE, "quadrature_degree must be in 2..15");
  if (m->cell_type == PHX_QUADRILATERAL) {
    // Q1 x Q1^2 x DG0 with a Q2 level-set [nv + nf + nc] (phx_assemble_flux_quad.inc.hip); a Gauss rule of
    // quadrature_degree / 2 + 1 points per direction integrates that degree exactly
    const double *qphi, *qf, *qg;
    double *q1, *q2, *q3;
    PHX_CHECK(to_device(m, phi_h, loc, m->nv + m->nf + m->nc, &qphi, &q1));
    PHX_CHECK(to_device(m, f_h, loc, m->nv, &qf, &q2));
    PHX_CHECK(to_device(m, g_h, loc, m->nv, &qg, &q3));
    PHX_CHECK(phx_begin_timing(m));
    const int nq = std::min(8, quadrature_degree / 2 + 1);
    int rcq = assemble_flux_quad_with_capacity(m, params, facet_tag, nq, qphi, qf, qg, 64, out);
    if (rcq == PHX_ERR_CAPACITY) rcq = assemble_flux_quad_with_capacity(m, params, facet_tag, nq, qphi, qf, qg, 128, out);
    if (rcq == PHX_OK) rcq = phx_end_timing(m, 2);
    if (q1) (void)phx_free(q1);
    if (q2) (void)phx_free(q2);
    if (q3) (void)phx_free(q3);
    return rcq;
  }
  PHX_REQUIRE(m->cell_type == PHX_TRIANGLE || m->cell_type == PHX_TETRAHEDRON,
              PHX_ERR_NOT_IMPLEMENTED, "unknown cell type");
  PHX_CHECK(phx_mesh_build_edges(m));
  const double *dphi, *df, *dg;
  double *o1, *o2, *o3;
  PHX_CHECK(to_device(m, phi_h, loc, m->nv + m->ne, &dphi, &o1));
  PHX_CHECK(to_device(m, f_h, loc, m->nv, &df, &o2));
  PHX_CHECK(to_device(m, g_h, loc, m->nv, &dg, &o3));
  PHX_CHECK(phx_begin_timing(m));
  int W = m->gdim == 3 ? 256 : 64;
  int rc = assemble_flux_with_capacity(m, params, facet_tag, quadrature_degree, dphi, df, dg, W, out);
  if (rc == PHX_ERR_CAPACITY) rc = assemble_flux_with_capacity(m, params, facet_tag, quadrature_degree, dphi, df, dg, 2 * W, out);
  if (rc == PHX_OK) rc = phx_end_timing(m, 2);
  if (o1) (void)phx_free(o1);
  if (o2) (void)phx_free(o2);
  if (o3) (void)phx_free(o3);
  return rc;
}
