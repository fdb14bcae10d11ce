// Strong-Dirichlet ("direct") phi-FEM Poisson, u_h = phi_h w_h: included by phx_assemble.hip after
// phx_assemble_p2.inc.hip (shares P2B, the conical rules, slot_add, Geo, load_cell).
// Forms: demo/strong-dirichlet/flower/main.py:104-127 -- one scalar field w of Lagrange degree K
// (1 or 2), level-set phi_h of degree 1 or 2.  The trial/test "basis" is psi_r = phi_h N_r:
//   grad psi_r = phi grad N_r + N_r grad phi,   lap psi_r = phi lap N_r + 2 grad phi . grad N_r + N_r lap phi.
// DoFs: vertex v -> v, edge e -> nv + e (K = 2).  phi_h: vertex values, then edge values (degree 2).

template <int D, int K>
struct LB {  // Lagrange basis of degree K on a D-simplex, barycentric form
  using B2 = P2B<D>;
  static constexpr int N = D + 1, NB = K == 1 ? N : B2::NB;
  __device__ static __forceinline__ double val(int r, const double *lam) {
    if (K == 1) return lam[r];
    return B2::val(r, lam);
  }
  __device__ static __forceinline__ void gradc(int r, const double *lam, double *c) {
    if (K == 1) { for (int m = 0; m < N; ++m) c[m] = m == r ? 1.0 : 0.0; return; }
    B2::gradc(r, lam, c);
  }
  __device__ static __forceinline__ double lapl(int r, const double (*GG)[N]) {
    if (K == 1) return 0.0;
    return B2::lapl(r, GG);
  }
};

struct SdArgs {
  AsmArgs A;          // du indexed by entity; A.nv = number of entities (columns keys < A.nv)
  const int32_t *c2e;
  int32_t nvert;
  int kphi;
  DevRule cell, facet;
};

template <int D, int K>
__device__ __forceinline__ void sd_cell_dofs(const SdArgs &P, int64_t c, const int32_t *v, int32_t *dof,
                                             int32_t *dphi) {
  using B = P2B<D>;
  for (int i = 0; i < B::N; ++i) { dphi[i] = v[i]; if (i < LB<D, K>::NB) dof[i] = v[i]; }
  if (K == 2 || P.kphi == 2)
    for (int k = 0; k < B::NE; ++k) {
      const int32_t e = P.nvert + P.c2e[c * B::NE + k];
      dphi[B::N + k] = e;
      if (K == 2) dof[B::N + k] = e;
    }
}

// phi_h at a point: value, barycentric gradient coefficients, (constant) Laplacian
template <int D>
struct PhiAt {
  double v, c[D + 1];
};
template <int D>
__device__ __forceinline__ void phi_eval(int kphi, const double *lam, const double *ph, PhiAt<D> &o) {
  using B = P2B<D>;
  if (kphi == 1) {
    double v = 0.0;
    for (int m = 0; m < B::N; ++m) { v += lam[m] * ph[m]; o.c[m] = ph[m]; }
    o.v = v;
    return;
  }
  double v = 0.0;
  for (int m = 0; m < B::N; ++m) o.c[m] = 0.0;
  for (int b = 0; b < B::NB; ++b) {
    double cb[B::N];
    B::gradc(b, lam, cb);
    v += B::val(b, lam) * ph[b];
    for (int m = 0; m < B::N; ++m) o.c[m] += cb[m] * ph[b];
  }
  o.v = v;
}
template <int D>
__device__ __forceinline__ double phi_lapl(int kphi, const double *ph, const double (*GG)[D + 1]) {
  if (kphi == 1) return 0.0;
  using B = P2B<D>;
  double s = 0.0;
  for (int b = 0; b < B::NB; ++b) s += B::lapl(b, GG) * ph[b];
  return s;
}

// psi_r = phi N_r at lam: value, gradient coefficients on g_m, Laplacian
template <int D, int K>
__device__ __forceinline__ void psi_eval(int r, const double *lam, const PhiAt<D> &ph, double lphi,
                                         const double (*GG)[D + 1], double &val, double *c, double &lap) {
  using B = LB<D, K>;
  double cn[B::N];
  const double nr = B::val(r, lam);
  B::gradc(r, lam, cn);
  val = ph.v * nr;
  double cross = 0.0;
  for (int m = 0; m < B::N; ++m) {
    c[m] = ph.v * cn[m] + nr * ph.c[m];
    for (int n = 0; n < B::N; ++n) cross += ph.c[m] * cn[n] * GG[m][n];
  }
  lap = ph.v * B::lapl(r, GG) + 2.0 * cross + nr * lphi;
}

template <int D, int K>
__global__ void k_sd_mark_active(int64_t nc, SdArgs P, uint8_t *__restrict__ fu) {
  using B = LB<D, K>;
  const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (c >= nc) return;
  const int t = P.A.ctags[c] & PHX_TAG_MASK;
  if (t != 1 && t != 2) return;
  int32_t v[D + 1], dof[B::NB], dphi[P2B<D>::NB];
  for (int i = 0; i <= D; ++i) v[i] = P.A.cells[c * (D + 1) + i];
  sd_cell_dofs<D, K>(P, c, v, dof, dphi);
  for (int r = 0; r < B::NB; ++r) fu[dof[r]] = 1;
}

// --- dx((1,2)) main.py:104,125 and, on cut cells, dx(2) main.py:106-111,125-127: GS lanes per cell,
// lane = entry (r, s) of the element matrix; the lanes with s == 0 also build the load vector ------
template <int D, int K, int GS>
__global__ void __launch_bounds__(256) k_sd_cells(int64_t nlist, const int32_t *__restrict__ list, SdArgs P) {
  using B = LB<D, K>;
  const int64_t gid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t e = gid / GS;
  const int l = (int)(gid % GS);
  if (e >= nlist || l >= B::NB * B::NB) return;
  const int r = l / B::NB, s = l % B::NB;
  const int64_t c = list[e];
  int32_t v[B::N], dof[B::NB], dphi[P2B<D>::NB];
  double X[B::N][D];
  load_cell<D>(P.A.cells, P.A.x, c, v, X);
  sd_cell_dofs<D, K>(P, c, v, dof, dphi);
  Geo<D> G;
  simplex_geometry<D>(X, G);
  double GG[B::N][B::N];
  gram<D>(G, GG);
  double ph[P2B<D>::NB], fn[B::NB];
  const int nphi = P.kphi == 1 ? B::N : P2B<D>::NB;
  for (int b = 0; b < nphi; ++b) ph[b] = P.A.phi[dphi[b]];
  if (s == 0) for (int b = 0; b < B::NB; ++b) fn[b] = P.A.f[dof[b]];
  const double lphi = phi_lapl<D>(P.kphi, ph, GG);
  const bool cut = (P.A.ctags[c] & PHX_TAG_MASK) == 2;
  double acc = 0.0, accl = 0.0, rhs = 0.0, rhsl = 0.0;
  for (int q = 0; q < P.cell.nq; ++q) {
    const double *lam = P.cell.lam + (int64_t)q * B::N;
    PhiAt<D> pq;
    phi_eval<D>(P.kphi, lam, ph, pq);
    double vr, vs, cr[B::N], cs[B::N], lr, ls;
    psi_eval<D, K>(r, lam, pq, lphi, GG, vr, cr, lr);
    psi_eval<D, K>(s, lam, pq, lphi, GG, vs, cs, ls);
    double k = 0.0;
    for (int m = 0; m < B::N; ++m)
      for (int n = 0; n < B::N; ++n) k += cr[m] * cs[n] * GG[m][n];
    const double w = P.cell.w[q];
    acc += w * k;
    accl += w * lr * ls;
    if (s == 0) {
      double fq = 0.0;
      for (int b = 0; b < B::NB; ++b) fq += B::val(b, lam) * fn[b];
      rhs += w * fq * vr;
      rhsl += w * fq * lr;
    }
  }
  const double sc = cut ? P.A.sigma * G.h * G.h : 0.0;
  const int32_t row = P.A.du[dof[r]];
  slot_add(P.A.slots, row, dof[s], G.vol * (acc + sc * accl));
  if (s == 0 && row >= 0) unsafeAtomicAdd(&P.A.rhs[row], G.vol * (rhs - sc * rhsl));
}

// --- ds: main.py:105  -int_F (grad(phi w) . n) phi v ------------------------------------------------
template <int D, int K, int GS>
__global__ void __launch_bounds__(256) k_sd_ds(int64_t nent, const int64_t *__restrict__ ent_packed,
                                               const int32_t *__restrict__ ent_pairs, SdArgs P) {
  using B = LB<D, K>;
  const int64_t gid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t e = gid / GS;
  const int l = (int)(gid % GS);
  if (e >= nent || l >= B::NB * B::NB) return;
  const int r = l / B::NB, s = l % B::NB;
  int64_t c;
  int lf;
  if (ent_packed) { c = ent_packed[2 * e + 1] >> 8; lf = (int)(ent_packed[2 * e + 1] & 0xff); }
  else { c = ent_pairs[2 * e]; lf = ent_pairs[2 * e + 1]; }
  int32_t v[B::N], dof[B::NB], dphi[P2B<D>::NB];
  double X[B::N][D];
  load_cell<D>(P.A.cells, P.A.x, c, v, X);
  sd_cell_dofs<D, K>(P, c, v, dof, dphi);
  Geo<D> G;
  simplex_geometry<D>(X, G);
  double GG[B::N][B::N];
  gram<D>(G, GG);
  double ph[P2B<D>::NB];
  const int nphi = P.kphi == 1 ? B::N : P2B<D>::NB;
  for (int b = 0; b < nphi; ++b) ph[b] = P.A.phi[dphi[b]];
  const double gn = sqrt(GG[lf][lf]);
  const double area = D * G.vol * gn;
  double acc = 0.0;
  for (int q = 0; q < P.facet.nq; ++q) {
    double lam[B::N];
    facet_embed<D>(lf, P.facet.lam + (int64_t)q * D, lam);
    PhiAt<D> pq;
    phi_eval<D>(P.kphi, lam, ph, pq);
    double vr, vs, cr[B::N], cs[B::N], lr, ls;
    psi_eval<D, K>(r, lam, pq, 0.0, GG, vr, cr, lr);
    psi_eval<D, K>(s, lam, pq, 0.0, GG, vs, cs, ls);
    double dn = 0.0;  // grad psi_s . n, n = -g_lf / |g_lf|
    for (int m = 0; m < B::N; ++m) dn -= cs[m] * GG[m][lf];
    acc += P.facet.w[q] * vr * dn / gn;
  }
  slot_add(P.A.slots, P.A.du[dof[r]], dof[s], -area * acc);
}

// --- dS((2,3)): main.py:112-117  sigma avg(h) int_F [grad(phi w).n][grad(phi v).n] -------------------
// one block per facet; quadrature points placed through the "+" cell, located in the "-" cell by
// barycentric coordinates; phi_h is evaluated on each side with that side's nodal values
template <int D, int K>
__global__ void __launch_bounds__(256) k_sd_facets(int64_t nlist, const int32_t *__restrict__ list, SdArgs P) {
  using B = LB<D, K>;
  const int64_t e = blockIdx.x;
  if (e >= nlist) return;
  const int64_t f = list[e];
  int32_t dofs[2 * B::NB];
  double Xc[2][B::N][D], GG[2][B::N][B::N], ph[2][P2B<D>::NB];
  Geo<D> G[2];
  int lfs[2];
  double gnn[2], hsum = 0.0, area = 0.0;
  const int nphi = P.kphi == 1 ? B::N : P2B<D>::NB;
  for (int side = 0; side < 2; ++side) {
    const int64_t c = P.A.f2c[2 * f + side];
    int32_t v[B::N], dphi[P2B<D>::NB];
    load_cell<D>(P.A.cells, P.A.x, c, v, Xc[side]);
    sd_cell_dofs<D, K>(P, c, v, dofs + side * B::NB, dphi);
    for (int b = 0; b < nphi; ++b) ph[side][b] = P.A.phi[dphi[b]];
    simplex_geometry<D>(Xc[side], G[side]);
    gram<D>(G[side], GG[side]);
    int lf = 0;
    for (int k = 0; k < B::N; ++k)
      if (P.A.c2f[c * B::N + k] == (int32_t)f) lf = k;
    lfs[side] = lf;
    gnn[side] = sqrt(GG[side][lf][lf]);
    if (side == 0) area = D * G[0].vol * gnn[0];
    hsum += G[side].h;
  }
  const double wgt = P.A.sigma * 0.5 * hsum * area;
  constexpr int M = 2 * B::NB;
  for (int idx = threadIdx.x; idx < M * M; idx += blockDim.x) {
    const int a = idx / M, b = idx % M;
    double acc = 0.0;
    for (int q = 0; q < P.facet.nq; ++q) {
      const double *mu = P.facet.lam + (int64_t)q * D;
      double lamq[2][B::N], xq[D];
      facet_embed<D>(lfs[0], mu, lamq[0]);
      for (int d = 0; d < D; ++d) {
        double t = 0.0;
        for (int m = 0; m < B::N; ++m) t += lamq[0][m] * Xc[0][m][d];
        xq[d] = t;
      }
      for (int m = 0; m < B::N; ++m) {
        double t = m == 0 ? 1.0 : 0.0;
        for (int d = 0; d < D; ++d) t += G[1].g[m][d] * (xq[d] - Xc[1][0][d]);
        lamq[1][m] = t;
      }
      double J[2];
      for (int w = 0; w < 2; ++w) {
        const int id = w == 0 ? a : b;
        const int side = id / B::NB;
        PhiAt<D> pq;
        phi_eval<D>(P.kphi, lamq[side], ph[side], pq);
        double vv, cc[B::N], ll;
        psi_eval<D, K>(id % B::NB, lamq[side], pq, 0.0, GG[side], vv, cc, ll);
        double t = 0.0;
        for (int m = 0; m < B::N; ++m) t -= cc[m] * GG[side][m][lfs[side]];
        J[w] = t / gnn[side];
      }
      acc += P.facet.w[q] * J[0] * J[1];
    }
    slot_add(P.A.slots, P.A.du[dofs[a]], dofs[b], wgt * acc);
  }
}

template <int D, int K>
static int assemble_sd_impl(phx_mesh *m, double stab_coef, int kphi, const double *dphi, const double *df,
                            int W, phx_system **out) {
  using B = LB<D, K>;
  const int64_t nent = K == 1 ? m->nv : m->nv + m->ne;
  PHX_REQUIRE(nent < INT32_MAX, PHX_ERR_VALUE, "too many DoFs for 32-bit column keys");
  phx_system *s = new phx_system();
  s->mesh = m; s->device = m->device; s->nfull = nent; s->slot_cap = W; s->nent = nent;
  s->u_vertex_block = K == 1;
  // K = 1: lattice preconditioner with the nodal weight of u = phi w folded in (phx_precond.inc.hip).  K = 2
  // stays with Jacobi: the weight estimated from diag A is off by the vertex / edge difference of the P2
  // diagonal and the refined-lattice solve does not pay (2-D 128^2: 424 vs 408 iterations).
  s->u_weighted = true;
  const dim3 block(256);
  std::vector<void *> keep;
  SdArgs P;
  memset(&P, 0, sizeof(P));
  PHX_CHECK(upload_rule(m, D, 2 * K + kphi, &P.cell, keep));
  PHX_CHECK(upload_rule(m, D - 1, 2 * (K + kphi) - 1, &P.facet, keep));
  uint8_t *fu = nullptr, *fp = nullptr;
  int32_t *su = nullptr, *sp = nullptr;
  PHX_HIP(phx_malloc(&fu, (size_t)nent)); PHX_HIP(phx_malloc(&fp, (size_t)nent));
  PHX_HIP(phx_malloc(&su, sizeof(int32_t) * (size_t)nent)); PHX_HIP(phx_malloc(&sp, sizeof(int32_t) * (size_t)nent));
  PHX_HIP(hipMemsetAsync(fu, 0, (size_t)nent, m->stream));
  PHX_HIP(hipMemsetAsync(fp, 0, (size_t)nent, m->stream));
  PHX_HIP(hipMemsetAsync(sp, 0, sizeof(int32_t) * (size_t)nent, m->stream));
  P.A.cells = m->cells; P.A.x = m->x; P.A.ctags = m->cell_tags; P.A.ftags = m->facet_tags;
  P.A.c2f = m->c2f; P.A.f2c = m->f2c; P.A.phi = dphi; P.A.f = df; P.A.ud = nullptr;
  P.A.gamma = 0.0; P.A.sigma = stab_coef; P.A.nv = (int32_t)nent;
  P.c2e = m->c2e; P.nvert = (int32_t)m->nv; P.kphi = kphi;
  k_sd_mark_active<D, K><<<dim3((unsigned)phx_div_up(m->nc, 256)), block, 0, m->stream>>>(m->nc, P, fu);
  int32_t nu = 0;
  PHX_CHECK(scan_flags(m, fu, su, nent, &nu));
  s->nu = nu; s->n = nu;
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
  PHX_HIP(phx_malloc(&sl.cols, sizeof(int32_t) * (size_t)s->n * W));
  PHX_HIP(phx_malloc(&sl.vals, sizeof(double) * (size_t)s->n * W));
  PHX_HIP(phx_malloc(&sl.overflow, sizeof(int)));
  PHX_HIP(hipMemsetAsync(sl.cols, 0xff, sizeof(int32_t) * (size_t)s->n * W, m->stream));
  PHX_HIP(hipMemsetAsync(sl.vals, 0, sizeof(double) * (size_t)s->n * W, m->stream));
  PHX_HIP(hipMemsetAsync(sl.overflow, 0, sizeof(int), m->stream));
  PHX_HIP(phx_malloc(&s->rhs, sizeof(double) * (size_t)s->n));
  PHX_HIP(hipMemsetAsync(s->rhs, 0, sizeof(double) * (size_t)s->n, m->stream));
  P.A.du = s->dof_of_vertex_u; P.A.dp = s->dof_of_vertex_p; P.A.rhs = s->rhs; P.A.slots = sl;
  int32_t *l_om = nullptr, *l_fac = nullptr;
  int64_t n_om = 0, n_fac = 0;
  PHX_CHECK(build_list(m, m->nc, SelOmega{m->cell_tags}, &l_om, &n_om));
  PHX_CHECK(build_list(m, m->nf, SelGhostFacet{m->facet_tags, m->f2c}, &l_fac, &n_fac));
  constexpr int GS = B::NB * B::NB <= 16 ? 16 : (B::NB * B::NB <= 64 ? 64 : 128);
  if (n_om > 0) {
    PHX_REQUIRE_GRID(n_om * GS, "strong-Dirichlet cell assembly");
    k_sd_cells<D, K, GS><<<dim3((unsigned)phx_div_up(n_om * GS, 256)), block, 0, m->stream>>>(n_om, l_om, P);
  }
  PHX_HIP(hipGetLastError());
  const int64_t nds = m->is_submesh ? m->nbf : (phx_collect_entities(m) == PHX_OK ? m->ent_count[0] : -1);
  PHX_REQUIRE(nds >= 0, PHX_ERR_VALUE, "integration entities unavailable");
  if (nds > 0) {
    const int64_t *pk = m->is_submesh ? nullptr : m->ent_buf[0];
    const int32_t *pr = m->is_submesh ? m->bfacets : nullptr;
    k_sd_ds<D, K, GS><<<dim3((unsigned)phx_div_up(nds * GS, 256)), block, 0, m->stream>>>(nds, pk, pr, P);
  }
  if (n_fac > 0)
    k_sd_facets<D, K><<<dim3((unsigned)n_fac), dim3(K == 1 ? 64 : 256), 0, m->stream>>>(n_fac, l_fac, P);
  PHX_HIP(hipGetLastError());
  PHX_HIP(hipStreamSynchronize(m->stream));
  PHX_HIP(phx_free(l_om)); PHX_HIP(phx_free(l_fac));
  for (void *p : keep) PHX_HIP(phx_free(p));
  const int rc = phx_finish_system(s, sl, (int32_t)nent);
  if (rc != PHX_OK) { phx_system_destroy(s); return rc; }
  *out = s;
  return PHX_OK;
}

static int assemble_sd_with_capacity(phx_mesh *m, double stab_coef, int degree, int kphi, const double *dphi,
                                     const double *df, int W, phx_system **out) {
  if (m->gdim == 2) return degree == 1 ? assemble_sd_impl<2, 1>(m, stab_coef, kphi, dphi, df, W, out)
                                       : assemble_sd_impl<2, 2>(m, stab_coef, kphi, dphi, df, W, out);
  return degree == 1 ? assemble_sd_impl<3, 1>(m, stab_coef, kphi, dphi, df, W, out)
                     : assemble_sd_impl<3, 2>(m, stab_coef, kphi, dphi, df, W, out);
}

extern "C" int phx_assemble_poisson_sd(phx_mesh *m, double stab_coef, int degree, const double *phi_h,
                                       int phi_degree, const double *f_h, int loc, phx_system **out) {
  PHX_HIP(hipSetDevice(m->device));
  PHX_REQUIRE(m->cell_type == PHX_TRIANGLE || m->cell_type == PHX_TETRAHEDRON,
              PHX_ERR_NOT_IMPLEMENTED, "assembly supports simplices (triangle, tetrahedron) only");
  PHX_REQUIRE(m->have_cell_tags && m->have_facet_tags, PHX_ERR_VALUE,
              "cell and facet tags must be computed before assembly");
  PHX_REQUIRE(degree == 1 || degree == 2, PHX_ERR_VALUE, "degree must be 1 or 2");
  PHX_REQUIRE(phi_degree == 1 || phi_degree == 2, PHX_ERR_VALUE, "phi_degree must be 1 or 2");
  if (degree == 2 || phi_degree == 2) PHX_CHECK(phx_mesh_build_edges(m));
  const int64_t nent = m->nv + m->ne;
  const double *dphi, *df;
  double *o1, *o2;
  PHX_CHECK(to_device(m, phi_h, loc, phi_degree == 1 ? m->nv : nent, &dphi, &o1));
  PHX_CHECK(to_device(m, f_h, loc, degree == 1 ? m->nv : nent, &df, &o2));
  PHX_CHECK(phx_begin_timing(m));
  int W = degree == 1 ? (m->gdim == 3 ? 64 : 32) : (m->gdim == 3 ? 256 : 128);
  int rc = assemble_sd_with_capacity(m, stab_coef, degree, phi_degree, dphi, df, W, out);
  if (rc == PHX_ERR_CAPACITY) rc = assemble_sd_with_capacity(m, stab_coef, degree, phi_degree, dphi, df, 2 * W, out);
  if (rc == PHX_OK) rc = phx_end_timing(m, 2);
  if (o1) (void)phx_free(o1);
  if (o2) (void)phx_free(o2);
  return rc;
}
