// Structured P2 systems on 3-D Kuhn boxes (BASELINE configs[2] at its stated size): included by
// phx_assemble_p2.inc.hip.  The P2 DoFs of a Kuhn box are the points of the lattice of spacing h / 2; a DoF whose whole
// support is tagged inside and which no cut-cell / ghost-penalty / boundary term touches ("C0") has one of EIGHT
// translation-invariant rows, by the parity class (a, b, c) of its fine point: vertex, three axis edges, three face
// diagonals, body diagonal.  Rows whose 5 x 5 x 5 fine neighbourhood is all C0 ("c0i") are never assembled and never
// stored: the solver applies them from the stencils (phx_solve.hip, k_spmv_p2s), their right-hand side is the P2 mass
// stencil applied to f_h.  Everything else -- the band around Gamma_h -- goes through the row slots as before, but only
// THOSE rows own slots: at 512^3 the slots of all 1.7e8 rows would take 0.5 TB, the band takes 74 GB.
// Forms: demo/weak-dirichlet/flower/main.py:112-151 with primal_degree = 2 (:38).

// ---- host: the 8 stiffness and mass stencils from the element matrices of a 2 x 2 x 2 patch of Kuhn cubes ------------
static void p2s_box_stencils(const double h[3], std::vector<double> &K, std::vector<double> &M) {
  K.assign(8 * 125, 0.0);
  M.assign(8 * 125, 0.0);
  std::vector<double> lam, w;
  conical_rule(3, 4, lam, w);     // exact for the mass integrand (degree 4); the stiffness integrand has degree 2
  const int nq = (int)w.size();
  const int perm[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
  const int te[6][2] = {{2, 3}, {1, 3}, {1, 2}, {0, 3}, {0, 2}, {0, 1}};
  for (int cube = 0; cube < 8; ++cube)
    for (int t = 0; t < 6; ++t) {
      int wv[4][3];
      wv[0][0] = cube & 1; wv[0][1] = (cube >> 1) & 1; wv[0][2] = (cube >> 2) & 1;
      for (int s = 0; s < 3; ++s) {
        for (int a = 0; a < 3; ++a) wv[s + 1][a] = wv[s][a];
        wv[s + 1][perm[t][s]] += 1;
      }
      double e[3][3];
      for (int k = 0; k < 3; ++k)
        for (int d = 0; d < 3; ++d) e[k][d] = (wv[k + 1][d] - wv[0][d]) * h[d];
      double cr[3][3];
      cr[0][0] = e[1][1] * e[2][2] - e[1][2] * e[2][1]; cr[0][1] = e[1][2] * e[2][0] - e[1][0] * e[2][2]; cr[0][2] = e[1][0] * e[2][1] - e[1][1] * e[2][0];
      cr[1][0] = e[2][1] * e[0][2] - e[2][2] * e[0][1]; cr[1][1] = e[2][2] * e[0][0] - e[2][0] * e[0][2]; cr[1][2] = e[2][0] * e[0][1] - e[2][1] * e[0][0];
      cr[2][0] = e[0][1] * e[1][2] - e[0][2] * e[1][1]; cr[2][1] = e[0][2] * e[1][0] - e[0][0] * e[1][2]; cr[2][2] = e[0][0] * e[1][1] - e[0][1] * e[1][0];
      const double det = e[0][0] * cr[0][0] + e[0][1] * cr[0][1] + e[0][2] * cr[0][2];
      double g[4][3];
      for (int k = 0; k < 3; ++k)
        for (int d = 0; d < 3; ++d) g[k + 1][d] = cr[k][d] / det;
      for (int d = 0; d < 3; ++d) g[0][d] = -(g[1][d] + g[2][d] + g[3][d]);
      const double vol = fabs(det) / 6.0;
      int fine[10][3];
      for (int i = 0; i < 4; ++i)
        for (int d = 0; d < 3; ++d) fine[i][d] = 2 * wv[i][d];
      for (int k = 0; k < 6; ++k)
        for (int d = 0; d < 3; ++d) fine[4 + k][d] = wv[te[k][0]][d] + wv[te[k][1]][d];
      double Kl[10][10] = {{0.0}}, Ml[10][10] = {{0.0}};
      for (int q = 0; q < nq; ++q) {
        const double *l = &lam[(size_t)q * 4];
        double N[10], G[10][3];
        for (int r = 0; r < 10; ++r) {
          double c[4] = {0.0, 0.0, 0.0, 0.0};
          if (r < 4) { N[r] = l[r] * (2.0 * l[r] - 1.0); c[r] = 4.0 * l[r] - 1.0; }
          else { const int a = te[r - 4][0], b = te[r - 4][1]; N[r] = 4.0 * l[a] * l[b]; c[a] = 4.0 * l[b]; c[b] = 4.0 * l[a]; }
          for (int d = 0; d < 3; ++d) G[r][d] = c[0] * g[0][d] + c[1] * g[1][d] + c[2] * g[2][d] + c[3] * g[3][d];
        }
        for (int r = 0; r < 10; ++r)
          for (int s = 0; s < 10; ++s) {
            Kl[r][s] += w[q] * vol * (G[r][0] * G[s][0] + G[r][1] * G[s][1] + G[r][2] * G[s][2]);
            Ml[r][s] += w[q] * vol * N[r] * N[s];
          }
      }
      for (int r = 0; r < 10; ++r) {
        const int *p = fine[r];
        if (p[0] < 2 || p[0] > 3 || p[1] < 2 || p[1] > 3 || p[2] < 2 || p[2] > 3) continue;   // class representatives
        const int cls = (p[0] - 2) + 2 * (p[1] - 2) + 4 * (p[2] - 2);
        for (int s = 0; s < 10; ++s) {
          const int o = (fine[s][0] - p[0] + 2) + 5 * (fine[s][1] - p[1] + 2) + 25 * (fine[s][2] - p[2] + 2);
          K[(size_t)cls * 125 + o] += Kl[r][s];
          M[(size_t)cls * 125 + o] += Ml[r][s];
        }
      }
    }
}

// ---- device ------------------------------------------------------------------------------------------------------------
// every DoF of a cell that is NOT tagged inside: its rows are not translation invariant
__global__ void k_p2s_mark_bad_cells(int64_t nc, P2Args P, uint8_t *__restrict__ bad) {
  const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (c >= nc || (P.A.ctags[c] & PHX_TAG_MASK) == 1) return;
  for (int i = 0; i < 4; ++i) bad[P.A.cells[c * 4 + i]] = 1;
  for (int k = 0; k < 6; ++k) bad[P.nvert + P.c2e[c * 6 + k]] = 1;
}
__device__ __forceinline__ void p2s_mark_cell(const P2Args &P, int64_t c, uint8_t *bad) {
  for (int i = 0; i < 4; ++i) bad[P.A.cells[c * 4 + i]] = 1;
  for (int k = 0; k < 6; ++k) bad[P.nvert + P.c2e[c * 6 + k]] = 1;
}
// ... and of the two cells of every ghost-penalty facet, and of the cells that carry a one-sided boundary term
__global__ void k_p2s_mark_bad_facets(int64_t n, const int32_t *__restrict__ list, P2Args P, uint8_t *__restrict__ bad) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t f = list[i];
  for (int side = 0; side < 2; ++side) {
    const int64_t c = P.A.f2c[2 * f + side];
    if (c >= 0) p2s_mark_cell(P, c, bad);
  }
}
__global__ void k_p2s_mark_bad_ents(int64_t n, const int64_t *__restrict__ ent_packed, P2Args P, uint8_t *__restrict__ bad) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) p2s_mark_cell(P, ent_packed[2 * i + 1] >> 8, bad);
}

// lattice flags of the C0 rows: active, untouched, strictly inside the box
__global__ void k_p2s_lat_c0(int64_t nent, phx_p2_lattice L, const int32_t *__restrict__ du,
                             const uint8_t *__restrict__ bad, uint8_t *__restrict__ latc0) {
  const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= nent || du[e] < 0 || bad[e]) return;
  int64_t q[3];
  phx_p2_fine_of_entity(L, e, q);
  for (int a = 0; a < 3; ++a)
    if (q[a] < 1 || q[a] > L.F[a] - 2) return;
  latc0[q[0] + L.F[0] * (q[1] + L.F[1] * q[2])] = 1;
}

// out = AND of `in` over the five points p - 2 .. p + 2 along one axis (outside the lattice: 0)
__global__ void k_p2s_erode(int64_t nf, int64_t F0, int64_t F1, int64_t F2, int axis, const uint8_t *__restrict__ in,
                            uint8_t *__restrict__ out) {
  const int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (p >= nf) return;
  const int64_t idx[3] = {p % F0, (p / F0) % F1, p / (F0 * F1)};
  const int64_t Fa = axis == 0 ? F0 : (axis == 1 ? F1 : F2), st = axis == 0 ? 1 : (axis == 1 ? F0 : F0 * F1);
  bool ok = idx[axis] >= 2 && idx[axis] <= Fa - 3;
  if (ok)
    for (int k = -2; k <= 2; ++k) ok = ok && in[p + k * st] != 0;
  out[p] = ok ? 1 : 0;
}

// c0[row] (active numbering) = the stencils apply this u row
__global__ void k_p2s_act_flags(int64_t nent, phx_p2_lattice L, const int32_t *__restrict__ du,
                                const uint8_t *__restrict__ latc0i, uint8_t *__restrict__ c0) {
  const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= nent) return;
  const int32_t row = du[e];
  if (row < 0) return;
  int64_t q[3];
  phx_p2_fine_of_entity(L, e, q);
  c0[row] = latc0i[q[0] + L.F[0] * (q[1] + L.F[1] * q[2])];
}

// slots only for the rows that are stored; a c0i row points at ONE shared empty slot of capacity 1
__global__ void k_p2s_slot_offsets(int64_t n, int W, int wl, const uint8_t *__restrict__ c0, const int32_t *__restrict__ rank,
                                   int64_t dummy, int64_t *__restrict__ off, uint8_t *__restrict__ wlog) {
  const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r >= n) return;
  off[r] = c0[r] ? dummy : (int64_t)rank[r] * W;
  wlog[r] = c0[r] ? 0 : (uint8_t)wl;
}

// right-hand side (P2 mass stencil applied to f_h, main.py:143 over dx((1,2))) and diagonal of the c0i rows
__global__ void __launch_bounds__(256)
k_p2s_rows(int64_t nf, phx_p2_lattice L, const uint8_t *__restrict__ latc0i, const int32_t *__restrict__ du,
           const double *__restrict__ coefK, const double *__restrict__ coefM, const double *__restrict__ f,
           double *__restrict__ rhs, double *__restrict__ diag) {
  const int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (p >= nf || !latc0i[p]) return;
  const int64_t I = p % L.F[0], J = (p / L.F[0]) % L.F[1], K = p / (L.F[0] * L.F[1]);
  const int cls = (int)(I & 1) + 2 * (int)(J & 1) + 4 * (int)(K & 1);
  const double *cm = coefM + cls * 125;
  double acc = 0.0;
  for (int dz = -2; dz <= 2; ++dz)
    for (int dy = -2; dy <= 2; ++dy)
      for (int dx = -2; dx <= 2; ++dx) {
        const double m = cm[(dx + 2) + 5 * (dy + 2) + 25 * (dz + 2)];
        if (m != 0.0) acc += m * f[phx_p2_entity_of_fine(L, I + dx, J + dy, K + dz)];
      }
  const int32_t row = du[phx_p2_entity_of_fine(L, I, J, K)];
  rhs[row] = acc;
  diag[row] = coefK[cls * 125 + 62];
}

// cells of Omega_h (tags 1, 2) with at least one DoF whose row is stored: the work list of k_p2_cells
struct SelP2StoredCell {
  const int8_t *t; const int32_t *cells, *c2e; int32_t nvert; const int32_t *du; const uint8_t *c0;
  __device__ bool operator()(const int32_t &c) const {
    const int v = t[c] & PHX_TAG_MASK;
    if (v == 2) return true;
    if (v != 1) return false;
    bool any = false;
    for (int i = 0; i < 4; ++i) { const int32_t r = du[cells[(int64_t)c * 4 + i]]; any = any || (r >= 0 && !c0[r]); }
    for (int k = 0; k < 6; ++k) { const int32_t r = du[nvert + c2e[(int64_t)c * 6 + k]]; any = any || (r >= 0 && !c0[r]); }
    return any;
  }
};

static phx_p2_lattice p2s_lattice(const phx_mesh *m) {
  phx_p2_lattice L;
  memset(&L, 0, sizeof(L));
  for (int a = 0; a < 3; ++a) { L.n[a] = m->box_n[a]; L.F[a] = 2 * m->box_n[a] + 1; }
  L.nv = m->nv;
  const bool mv[7][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {1, 1, 0}, {1, 0, 1}, {0, 1, 1}, {1, 1, 1}};   // as phx_mesh_build_edges
  for (int cls = 0; cls < 7; ++cls)
    for (int a = 0; a < 3; ++a) L.ext[cls][a] = mv[cls][a] ? L.n[a] : L.n[a] + 1;
  L.base[0] = 0;
  for (int cls = 0; cls < 7; ++cls) L.base[cls + 1] = L.base[cls] + L.ext[cls][0] * L.ext[cls][1] * L.ext[cls][2];
  return L;
}

// Everything between the active numbering and the element kernels of a structured P2 assembly: C0 / c0i flags, the
// stencil tables, slots for the stored rows only, right-hand side and diagonal of the c0i rows, the cell work list.
struct P2SPrep {
  uint8_t *latc0 = nullptr, *latc0i = nullptr;   // [F0 F1 F2]
  int32_t *l_cells = nullptr;
  int64_t n_cells = 0;
  double *coefM = nullptr;
  int64_t slot_rows = 0;
};

static int p2s_prepare(phx_system *s, P2Args &P, Slots &sl, const int32_t *l_fac, int64_t n_fac, int W, P2SPrep *out) {
  phx_mesh *m = s->mesh;
  hipStream_t st = m->stream;
  const dim3 block(256);
  const int64_t nent = s->nent;
  phx_p2_struct *ps = new phx_p2_struct();
  s->p2s = ps;
  ps->lat = p2s_lattice(m);
  const phx_p2_lattice &L = ps->lat;
  PHX_REQUIRE(L.nv + L.base[7] == nent, PHX_ERR_VALUE, "structured P2: the box's edge numbering is not the closed form");
  const int64_t NF = L.F[0] * L.F[1] * L.F[2];
  PHX_REQUIRE(NF < INT32_MAX, PHX_ERR_VALUE, "structured P2: fine lattice too large for 32-bit positions");
  // ---- stencil tables
  std::vector<double> K, M;
  const double h[3] = {m->box_h[0], m->box_h[1], m->box_h[2]};
  p2s_box_stencils(h, K, M);
  unsigned long long mask[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
  for (int cls = 0; cls < 8; ++cls)
    for (int o = 0; o < 125; ++o)
      if (K[(size_t)cls * 125 + o] != 0.0) mask[cls >> 1][o >> 6] |= 1ull << (o & 63);
  PHX_HIP(phx_malloc(&ps->coef, sizeof(double) * 1000));
  PHX_HIP(phx_malloc(&out->coefM, sizeof(double) * 1000));
  PHX_HIP(phx_malloc(&ps->mask, sizeof(mask)));
  PHX_HIP(hipMemcpyAsync(ps->coef, K.data(), sizeof(double) * 1000, hipMemcpyHostToDevice, st));
  PHX_HIP(hipMemcpyAsync(out->coefM, M.data(), sizeof(double) * 1000, hipMemcpyHostToDevice, st));
  PHX_HIP(hipMemcpyAsync(ps->mask, mask, sizeof(mask), hipMemcpyHostToDevice, st));
  std::vector<double> tabE(500), tabO(500);
  unsigned linemask[4] = {0, 0, 0, 0};
  for (int bc = 0; bc < 4; ++bc)
    for (int l = 0; l < 25; ++l)
      for (int dx = 0; dx < 5; ++dx) {
        const int o = dx + 5 * l;
        const double ce = K[(size_t)(2 * bc) * 125 + o], co = K[(size_t)(2 * bc + 1) * 125 + o];
        tabE[(size_t)(bc * 25 + l) * 5 + dx] = ce;
        tabO[(size_t)(bc * 25 + l) * 5 + dx] = co;
        if (ce != 0.0 || co != 0.0) linemask[bc] |= 1u << l;
      }
  PHX_HIP(phx_malloc(&ps->tabE, sizeof(double) * 500));
  PHX_HIP(phx_malloc(&ps->tabO, sizeof(double) * 500));
  PHX_HIP(phx_malloc(&ps->linemask, sizeof(linemask)));
  PHX_HIP(hipMemcpyAsync(ps->tabE, tabE.data(), sizeof(double) * 500, hipMemcpyHostToDevice, st));
  PHX_HIP(hipMemcpyAsync(ps->tabO, tabO.data(), sizeof(double) * 500, hipMemcpyHostToDevice, st));
  PHX_HIP(hipMemcpyAsync(ps->linemask, linemask, sizeof(linemask), hipMemcpyHostToDevice, st));
  PHX_HIP(hipStreamSynchronize(st));   // K, M, mask, tab are host temporaries
  // ---- C0 on the fine lattice, c0i by erosion with the 5 x 5 x 5 box
  uint8_t *bad = nullptr, *tmp = nullptr;
  PHX_HIP(phx_malloc(&bad, (size_t)nent));
  PHX_HIP(hipMemsetAsync(bad, 0, (size_t)nent, st));
  k_p2s_mark_bad_cells<<<dim3((unsigned)phx_div_up(m->nc, 256)), block, 0, st>>>(m->nc, P, bad);
  if (n_fac > 0) k_p2s_mark_bad_facets<<<dim3((unsigned)phx_div_up(n_fac, 256)), block, 0, st>>>(n_fac, l_fac, P, bad);
  PHX_CHECK(phx_collect_entities(m));
  if (m->ent_count[0] > 0)
    k_p2s_mark_bad_ents<<<dim3((unsigned)phx_div_up(m->ent_count[0], 256)), block, 0, st>>>(m->ent_count[0], m->ent_buf[0], P, bad);
  PHX_HIP(phx_malloc(&out->latc0, (size_t)NF));
  PHX_HIP(phx_malloc(&out->latc0i, (size_t)NF));
  PHX_HIP(phx_malloc(&tmp, (size_t)NF));
  PHX_HIP(hipMemsetAsync(out->latc0, 0, (size_t)NF, st));
  const dim3 gent((unsigned)phx_div_up(nent, 256)), gfine((unsigned)phx_div_up(NF, 256));
  k_p2s_lat_c0<<<gent, block, 0, st>>>(nent, L, s->dof_of_vertex_u, bad, out->latc0);
  k_p2s_erode<<<gfine, block, 0, st>>>(NF, L.F[0], L.F[1], L.F[2], 0, out->latc0, out->latc0i);
  k_p2s_erode<<<gfine, block, 0, st>>>(NF, L.F[0], L.F[1], L.F[2], 1, out->latc0i, tmp);
  k_p2s_erode<<<gfine, block, 0, st>>>(NF, L.F[0], L.F[1], L.F[2], 2, tmp, out->latc0i);
  PHX_HIP(phx_malloc(&s->c0, (size_t)s->n));
  PHX_HIP(hipMemsetAsync(s->c0, 0, (size_t)s->n, st));
  k_p2s_act_flags<<<gent, block, 0, st>>>(nent, L, s->dof_of_vertex_u, out->latc0i, s->c0);
  PHX_HIP(hipGetLastError());
  // ---- slots of the stored rows
  int32_t *rank = nullptr, nstored = 0;
  uint8_t *notc0 = nullptr;
  PHX_HIP(phx_malloc(&rank, sizeof(int32_t) * (size_t)s->n));
  PHX_HIP(phx_malloc(&notc0, (size_t)s->n));
  const dim3 gn((unsigned)phx_div_up(s->n, 256));
  k_not_flags<<<gn, block, 0, st>>>(s->n, s->c0, notc0);
  PHX_CHECK(scan_flags(m, notc0, rank, s->n, &nstored));
  out->slot_rows = nstored;
  int64_t *off = nullptr;
  uint8_t *wl = nullptr;
  PHX_HIP(phx_malloc(&off, sizeof(int64_t) * (size_t)s->n));
  PHX_HIP(phx_malloc(&wl, (size_t)s->n));
  int lg = 0;
  while ((1 << lg) < W) ++lg;
  k_p2s_slot_offsets<<<gn, block, 0, st>>>(s->n, W, lg, s->c0, rank, (int64_t)nstored * W, off, wl);
  sl.off = off; sl.wlog = wl;
  const size_t ns = (size_t)nstored * W + 64;
  PHX_HIP(phx_malloc(&sl.cols, sizeof(int32_t) * ns));
  PHX_HIP(phx_malloc(&sl.vals, sizeof(double) * ns));
  PHX_HIP(phx_malloc(&sl.overflow, sizeof(int)));
  PHX_HIP(hipMemsetAsync(sl.cols, 0xff, sizeof(int32_t) * ns, st));
  PHX_HIP(hipMemsetAsync(sl.vals, 0, sizeof(double) * ns, st));
  PHX_HIP(hipMemsetAsync(sl.overflow, 0, sizeof(int), st));
  // ---- right-hand side and diagonal of the rows the stencils apply
  PHX_HIP(phx_malloc(&s->diag, sizeof(double) * (size_t)s->n));
  PHX_HIP(hipMemsetAsync(s->diag, 0, sizeof(double) * (size_t)s->n, st));
  k_p2s_rows<<<gfine, block, 0, st>>>(NF, L, out->latc0i, s->dof_of_vertex_u, ps->coef, out->coefM, P.A.f, s->rhs, s->diag);
  PHX_HIP(hipGetLastError());
  PHX_HIP(hipStreamSynchronize(st));
  PHX_HIP(phx_free(bad)); PHX_HIP(phx_free(tmp)); PHX_HIP(phx_free(rank)); PHX_HIP(phx_free(notc0));
  // ---- cells that touch a stored row
  P.A.c0 = s->c0;
  PHX_CHECK(build_list(m, m->nc, SelP2StoredCell{m->cell_tags, m->cells, m->c2e, (int32_t)m->nv, s->dof_of_vertex_u, s->c0},
                       &out->l_cells, &out->n_cells));
  return PHX_OK;
}
