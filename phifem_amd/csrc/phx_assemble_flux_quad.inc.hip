// Neumann / Robin phi-FEM Poisson on QUADRILATERALS: mixed (u, y, p) in Q1 x Q1^2 x DG0 with a Q2 level-set --
// the cell type of demo/neumann/square/main.py:49-50 with its forms :113-158 (and those of demo/robin/square/
// main.py:112-168 with robin_coef != 0).  Included by phx_assemble.hip after phx_assemble_flux.inc.hip (shares
// FxArgs' conventions, slot_add, the work-list builders).
// Cells: axis-parallel rectangles in tensor-product vertex order v0 (0,0), v1 (1,0), v2 (0,1), v3 (1,1) -- what
// dolfinx.mesh.create_rectangle builds; local facets f0 (v0,v1), f1 (v0,v2), f2 (v1,v3), f3 (v2,v3) [3P basix].
// h_T = the diagonal.  Q2 level-set nodal layout: [vertices (nv), edge midpoints by facet id (nf), cell centres (nc)].
// DoFs: u at vertex v -> v, y_k at vertex v -> (1 + k) nv + v, p on cell c -> 3 nv + c.
// Cell integrals of the cut cells: tensor Gauss rule, nq points per direction (|grad phi_h| is not polynomial:
// agreement with FFCx's Gauss-Jacobi rule to quadrature accuracy); bulk cells and edges: closed form / 3-point Gauss.

struct FxqArgs {
  const int32_t *cells, *c2f, *f2c;
  const double *x, *phi, *f, *g;
  const int8_t *ctags;
  const int32_t *dofmap;
  int64_t nv, nf;
  double gamma, sigma, kappa;
  double *rhs;
  Slots slots;
  int nq;
  double gx[8], gw[8];   // Gauss points / weights on [0, 1]
  int *bad;              // set when a cell is not an axis-parallel rectangle
};

__global__ void k_fxq_mark_active(int64_t nc, FxqArgs A, uint8_t *__restrict__ flags) {
  const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (c >= nc) return;
  const int t = A.ctags[c] & PHX_TAG_MASK;
  if (t != 1 && t != 2) return;
  for (int i = 0; i < 4; ++i) {
    const int64_t v = A.cells[c * 4 + i];
    flags[v] = 1;
    if (t == 2) { flags[A.nv + v] = 1; flags[2 * A.nv + v] = 1; }
  }
  if (t == 2) flags[3 * A.nv + c] = 1;
}

struct RectGeo { double hx, hy, h; int32_t v[4]; };
__device__ __forceinline__ bool rect_load(const FxqArgs &A, int64_t c, RectGeo &R) {
  double X[4][2];
  for (int i = 0; i < 4; ++i) {
    R.v[i] = A.cells[c * 4 + i];
    X[i][0] = A.x[2 * (int64_t)R.v[i]];
    X[i][1] = A.x[2 * (int64_t)R.v[i] + 1];
  }
  R.hx = X[1][0] - X[0][0];
  R.hy = X[2][1] - X[0][1];
  R.h = sqrt(R.hx * R.hx + R.hy * R.hy);
  const double tx = 1e-12 * fabs(R.hx), ty = 1e-12 * fabs(R.hy);
  return R.hx > 0.0 && R.hy > 0.0 && fabs(X[1][1] - X[0][1]) <= tx && fabs(X[2][0] - X[0][0]) <= ty &&
         fabs(X[3][0] - X[1][0]) <= tx && fabs(X[3][1] - X[2][1]) <= ty;
}

// 1-D linear element matrices on [0, 1]: stiffness A1 = [[1,-1],[-1,1]], mass M1 = [[2,1],[1,2]] / 6
__device__ __forceinline__ double a1(int i, int j) { return i == j ? 1.0 : -1.0; }
__device__ __forceinline__ double m1(int i, int j) { return (i == j ? 2.0 : 1.0) / 6.0; }

// --- dx((1,2)): neumann main.py:114 (grad u . grad v + u v) and :144 (f v); 16 lanes per cell, closed form ----
__global__ void __launch_bounds__(256) k_fxq_bulk(int64_t nlist, const int32_t *__restrict__ list, FxqArgs A) {
  const int64_t gid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t e = gid / 16;
  const int l = (int)(gid % 16);
  if (e >= nlist) return;
  const int i = l / 4, j = l % 4;
  RectGeo R;
  if (!rect_load(A, list[e], R)) { *A.bad = 1; return; }
  const int ix = i & 1, iy = i >> 1, jx = j & 1, jy = j >> 1;
  const double K = (R.hy / R.hx) * a1(ix, jx) * m1(iy, jy) + (R.hx / R.hy) * m1(ix, jx) * a1(iy, jy);
  const double Mm = R.hx * R.hy * m1(ix, jx) * m1(iy, jy);
  slot_add(A.slots, A.dofmap[R.v[i]], R.v[j], K + Mm);
  if (j == 0) {
    double s = 0.0;
    for (int k = 0; k < 4; ++k) s += m1(ix, k & 1) * m1(iy, k >> 1) * A.f[R.v[k]];
    unsafeAtomicAdd(&A.rhs[A.dofmap[R.v[i]]], R.hx * R.hy * s);
  }
}

// Q1 value / physical gradient of vertex function i at reference (xi, eta)
__device__ __forceinline__ void q1_at(int i, double xi, double eta, const RectGeo &R, double *val, double *gx, double *gy) {
  const double lx = (i & 1) ? xi : 1.0 - xi, ly = (i >> 1) ? eta : 1.0 - eta;
  const double dx = (i & 1) ? 1.0 : -1.0, dy = (i >> 1) ? 1.0 : -1.0;
  *val = lx * ly;
  *gx = dx * ly / R.hx;
  *gy = lx * dy / R.hy;
}
__device__ __forceinline__ void l3_at(double t, double *L, double *dL) {
  L[0] = 2.0 * (t - 0.5) * (t - 1.0); L[1] = 4.0 * t * (1.0 - t); L[2] = 2.0 * t * (t - 0.5);
  dL[0] = 4.0 * t - 3.0; dL[1] = 4.0 - 8.0 * t; dL[2] = 4.0 * t - 1.0;
}
// Q2 level-set and its physical gradient at (xi, eta) from the 9 nodal values (vertices, facet midpoints, centre)
__device__ __forceinline__ void q2_phi_at(const double *ph, double xi, double eta, const RectGeo &R, double *val, double *gx,
                                          double *gy) {
  constexpr int IX[9] = {0, 2, 0, 2, 1, 0, 2, 1, 1}, IY[9] = {0, 0, 2, 2, 0, 1, 1, 2, 1};
  double Lx[3], dLx[3], Ly[3], dLy[3];
  l3_at(xi, Lx, dLx);
  l3_at(eta, Ly, dLy);
  double v = 0.0, a = 0.0, b = 0.0;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    v += ph[k] * Lx[IX[k]] * Ly[IY[k]];
    a += ph[k] * dLx[IX[k]] * Ly[IY[k]];
    b += ph[k] * Lx[IX[k]] * dLy[IY[k]];
  }
  *val = v; *gx = a / R.hx; *gy = b / R.hy;
}

// functionals of local DoF a (0..3 u_i, 4..11 y_{k,i}, 12 p) at one quadrature point
struct FxqLocal { double U, DY, B, T1[2]; };
__device__ __forceinline__ void fxq_eval(int a, double xi, double eta, const RectGeo &R, double phq, double gpx, double gpy,
                                         double ngp, double kappa, FxqLocal &o) {
  o.U = 0.0; o.DY = 0.0; o.B = 0.0; o.T1[0] = 0.0; o.T1[1] = 0.0;
  double val, gx, gy;
  if (a < 4) {
    q1_at(a, xi, eta, R, &val, &gx, &gy);
    o.U = val; o.T1[0] = gx; o.T1[1] = gy;
    o.B = -kappa * ngp * val;
  } else if (a < 12) {
    const int k = (a - 4) / 4, i = (a - 4) % 4;
    q1_at(i, xi, eta, R, &val, &gx, &gy);
    o.T1[k] = val;
    o.DY = k == 0 ? gx : gy;
    o.B = val * (k == 0 ? gpx : gpy);
  } else {
    o.B = phq / R.h;
  }
}

// --- dx(2): main.py:117-130 and :145-156; one 64-thread block per cut cell walks the 13 x 13 tensor + rhs ----
__global__ void __launch_bounds__(64) k_fxq_cut(int64_t nlist, const int32_t *__restrict__ list, FxqArgs A) {
  constexpr int M = 13;
  const int64_t e = blockIdx.x;
  if (e >= nlist) return;
  const int64_t c = list[e];
  RectGeo R;
  if (!rect_load(A, c, R)) { *A.bad = 1; return; }
  double ph[9];
  for (int i = 0; i < 4; ++i) ph[i] = A.phi[R.v[i]];
  for (int k = 0; k < 4; ++k) ph[4 + k] = A.phi[A.nv + A.c2f[c * 4 + k]];
  ph[8] = A.phi[A.nv + A.nf + c];
  const double det = R.hx * R.hy, h2 = 1.0 / (R.h * R.h);
  auto full = [&](int a) -> int32_t {
    if (a < 4) return R.v[a];
    if (a < 12) return (int32_t)((1 + (a - 4) / 4) * A.nv + R.v[(a - 4) % 4]);
    return (int32_t)(3 * A.nv + c);
  };
  for (int idx = threadIdx.x; idx < M * M + M; idx += blockDim.x) {
    const bool is_rhs = idx >= M * M;
    const int a = is_rhs ? idx - M * M : idx / M, b = is_rhs ? 0 : idx % M;
    double acc = 0.0;
    for (int qx = 0; qx < A.nq; ++qx)
      for (int qy = 0; qy < A.nq; ++qy) {
        const double xi = A.gx[qx], eta = A.gx[qy], w = A.gw[qx] * A.gw[qy];
        double phq, gpx, gpy;
        q2_phi_at(ph, xi, eta, R, &phq, &gpx, &gpy);
        const double ngp = sqrt(gpx * gpx + gpy * gpy);
        FxqLocal la;
        fxq_eval(a, xi, eta, R, phq, gpx, gpy, ngp, A.kappa, la);
        if (is_rhs) {
          double fq = 0.0, gq = 0.0;
          for (int i = 0; i < 4; ++i) {
            double val, t0, t1;
            q1_at(i, xi, eta, R, &val, &t0, &t1);
            fq += val * A.f[R.v[i]];
            gq += val * A.g[R.v[i]];
          }
          acc += w * (-h2 * gq * ngp * la.B + fq * (la.DY + la.U));
        } else {
          FxqLocal lb;
          fxq_eval(b, xi, eta, R, phq, gpx, gpy, ngp, A.kappa, lb);
          acc += w * (la.T1[0] * lb.T1[0] + la.T1[1] * lb.T1[1] + (la.DY + la.U) * (lb.DY + lb.U) + h2 * la.B * lb.B);
        }
      }
    const double val = A.gamma * det * acc;
    const int32_t row = A.dofmap[full(a)];
    if (is_rhs) unsafeAtomicAdd(&A.rhs[row], val);
    else slot_add(A.slots, row, full(b), val);
  }
}

// local facet lf of a rectangle: its two vertices, the axis of its outward normal and the sign
__device__ __forceinline__ void quad_facet(int lf, int *va, int *vb, int *axis, double *sign) {
  constexpr int FA[4] = {0, 0, 1, 2}, FB[4] = {1, 2, 3, 3}, AX[4] = {1, 0, 0, 1};
  constexpr double SG[4] = {-1.0, -1.0, 1.0, 1.0};
  *va = FA[lf]; *vb = FB[lf]; *axis = AX[lf]; *sign = SG[lf];
}

// --- ds: main.py:115  int_F (y . n) v; 4 lanes per (cell, local facet): (i, j) over the facet's two vertices ----
__global__ void __launch_bounds__(256) k_fxq_ds(int64_t nent, const int64_t *__restrict__ ent_packed,
                                                const int32_t *__restrict__ ent_pairs, FxqArgs A) {
  const int64_t gid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t e = gid / 4;
  const int l = (int)(gid % 4);
  if (e >= nent) return;
  int64_t c;
  int lf;
  if (ent_packed) { c = ent_packed[2 * e + 1] >> 8; lf = (int)(ent_packed[2 * e + 1] & 0xff); }
  else { c = ent_pairs[2 * e]; lf = ent_pairs[2 * e + 1]; }
  RectGeo R;
  if (!rect_load(A, c, R)) { *A.bad = 1; return; }
  int va, vb, axis;
  double sign;
  quad_facet(lf, &va, &vb, &axis, &sign);
  const int ends[2] = {va, vb};
  const int i = l / 2, j = l % 2;
  const double len = axis == 0 ? R.hy : R.hx;   // normal along x: the facet runs along y
  const int32_t col = (int32_t)((1 + axis) * A.nv + R.v[ends[j]]);
  if (A.dofmap[col] < 0) return;  // y lives on cut cells only
  slot_add(A.slots, A.dofmap[R.v[ends[i]]], col, len * m1(i, j) * sign);
}

// --- dS(tag): main.py:132-135  sigma avg(h) int_F [grad u . n][grad v . n]; 64 lanes per facet -----------------
__global__ void __launch_bounds__(256) k_fxq_facets(int64_t nlist, const int32_t *__restrict__ list, FxqArgs A) {
  const int64_t gid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t e = gid / 64;
  const int l = (int)(gid % 64);
  if (e >= nlist) return;
  const int a = l / 8, b = l % 8;
  const int64_t f = list[e];
  // 3-point Gauss on [0, 1]
  const double s15 = 0.7745966692414834 * 0.5;
  const double tq[3] = {0.5 - s15, 0.5, 0.5 + s15}, wq[3] = {5.0 / 18.0, 8.0 / 18.0, 5.0 / 18.0};
  int32_t dofs[8];
  double J[3][8], hsum = 0.0, len = 0.0;
  for (int side = 0; side < 2; ++side) {
    const int64_t c = A.f2c[2 * f + side];
    RectGeo R;
    if (!rect_load(A, c, R)) { *A.bad = 1; return; }
    int lf = 0;
    for (int k = 0; k < 4; ++k)
      if (A.c2f[c * 4 + k] == (int32_t)f) lf = k;
    int va, vb, axis;
    double sign;
    quad_facet(lf, &va, &vb, &axis, &sign);
    if (side == 0) len = axis == 0 ? R.hy : R.hx;
    hsum += R.h;
    const double fixed = sign > 0.0 ? 1.0 : 0.0;
    for (int q = 0; q < 3; ++q) {
      const double xi = axis == 0 ? fixed : tq[q], eta = axis == 1 ? fixed : tq[q];
      for (int i = 0; i < 4; ++i) {
        double val, gx, gy;
        q1_at(i, xi, eta, R, &val, &gx, &gy);
        J[q][side * 4 + i] = sign * (axis == 0 ? gx : gy);
      }
    }
    for (int i = 0; i < 4; ++i) dofs[side * 4 + i] = R.v[i];
  }
  double acc = 0.0;
  for (int q = 0; q < 3; ++q) acc += wq[q] * J[q][a] * J[q][b];
  slot_add(A.slots, A.dofmap[dofs[a]], dofs[b], A.sigma * 0.5 * hsum * len * acc);
}

static int assemble_flux_quad_with_capacity(phx_mesh *m, const double *params, int facet_tag, int nq,
                                            const double *dphi, const double *df, const double *dg, int W,
                                            phx_system **out) {
  const int64_t nent = 3 * m->nv + m->nc;
  PHX_REQUIRE(nent < INT32_MAX, PHX_ERR_VALUE, "too many DoFs for 32-bit column keys");
  phx_system *s = new phx_system();
  s->mesh = m; s->device = m->device; s->nfull = nent; s->slot_cap = W; s->nent = nent;
  const dim3 block(256);
  FxqArgs A;
  memset(&A, 0, sizeof(A));
  A.cells = m->cells; A.x = m->x; A.ctags = m->cell_tags; A.c2f = m->c2f; A.f2c = m->f2c;
  A.phi = dphi; A.f = df; A.g = dg; A.nv = m->nv; A.nf = m->nf;
  A.gamma = params[0]; A.sigma = params[1]; A.kappa = params[2];
  A.nq = nq;
  {  // Gauss-Legendre points on [0, 1] by Newton iteration on the Legendre polynomial
    for (int i = 0; i < nq; ++i) {
      double z = cos(3.14159265358979323846 * (i + 0.75) / (nq + 0.5)), pp = 1.0;
      for (int it = 0; it < 100; ++it) {
        double p1 = 1.0, p2 = 0.0;
        for (int j = 0; j < nq; ++j) { const double p3 = p2; p2 = p1; p1 = ((2.0 * j + 1.0) * z * p2 - j * p3) / (j + 1.0); }
        pp = nq * (z * p1 - p2) / (z * z - 1.0);
        const double z1 = z;
        z = z1 - p1 / pp;
        if (fabs(z - z1) < 1e-15) break;
      }
      A.gx[i] = 0.5 * (1.0 - z);
      A.gw[i] = 1.0 / ((1.0 - z * z) * pp * pp);
    }
  }
  int *bad = nullptr;
  PHX_HIP(phx_malloc(&bad, sizeof(int)));
  PHX_HIP(hipMemsetAsync(bad, 0, sizeof(int), m->stream));
  A.bad = bad;
  uint8_t *flags = nullptr;
  int32_t *scan = nullptr;
  PHX_HIP(phx_malloc(&flags, (size_t)nent));
  PHX_HIP(phx_malloc(&scan, sizeof(int32_t) * (size_t)nent));
  PHX_HIP(hipMemsetAsync(flags, 0, (size_t)nent, m->stream));
  k_fxq_mark_active<<<dim3((unsigned)phx_div_up(m->nc, 256)), block, 0, m->stream>>>(m->nc, A, flags);
  int32_t n = 0;
  PHX_CHECK(scan_flags(m, flags, scan, nent, &n));
  PHX_REQUIRE(n > 0, PHX_ERR_VALUE, "no active DoF: no cell is tagged 1 or 2");
  s->n = n; s->nu = n;   // Jacobi only, as on simplices (phx_assemble_flux.inc.hip)
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
  if (n_om) k_fxq_bulk<<<dim3((unsigned)phx_div_up(n_om * 16, 256)), block, 0, m->stream>>>(n_om, l_om, A);
  if (n_cut) k_fxq_cut<<<dim3((unsigned)n_cut), dim3(64), 0, m->stream>>>(n_cut, l_cut, A);
  if (nds) k_fxq_ds<<<dim3((unsigned)phx_div_up(nds * 4, 256)), block, 0, m->stream>>>(nds, pk, pr, A);
  if (n_fac) k_fxq_facets<<<dim3((unsigned)phx_div_up(n_fac * 64, 256)), block, 0, m->stream>>>(n_fac, l_fac, A);
  PHX_HIP(hipGetLastError());
  int hbad = 0;
  PHX_HIP(hipMemcpyAsync(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost, m->stream));
  PHX_HIP(hipStreamSynchronize(m->stream));
  PHX_HIP(phx_free(l_om)); PHX_HIP(phx_free(l_cut)); PHX_HIP(phx_free(l_fac)); PHX_HIP(phx_free(bad));
  if (hbad) {
    PHX_HIP(phx_free(sl.cols)); PHX_HIP(phx_free(sl.vals)); PHX_HIP(phx_free(sl.overflow));
    phx_system_destroy(s);
    phx_set_error("quadrilateral assembly covers axis-parallel rectangles in tensor-product vertex order");
    return PHX_ERR_NOT_IMPLEMENTED;
  }
  const int rc = phx_finish_system(s, sl, (int32_t)nent);
  if (rc != PHX_OK) { phx_system_destroy(s); return rc; }
  *out = s;
  return PHX_OK;
}
