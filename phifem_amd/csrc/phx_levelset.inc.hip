// Device-side preparation of PHX_PHI_POINTS level-sets (include of phx_tag.hip: -ffp-contract=off):
//  * a degree-2 nodal level-set (P2 on simplices: vertex + edge values; Q2 on quadrilaterals: vertex + facet + cell
//    values) evaluated at the detection points of every cell and of every background-boundary facet -- what the host
//    shim computed with numpy in round 1 (phifem_amd/mesh_scripts.py `_evaluate_p2`, kept as the test reference);
//  * the PHYSICAL detection points themselves, so that a caller's expression ("UFL expression" leg of
//    /root/reference/tests/test_compute_meshtags.py:159-161) can be evaluated on device tensors.
// Layout of both outputs: cells first ([nc][npts_cell]), then the boundary facets in ascending facet id
// ([nbf][npts_facet]) -- the layout phx_tag_cells / phx_tag_facets read.

// value = sum_d nodal[dof_d] tab[(lf npts + q) ndof + d], d ascending (fixed order)
__global__ void __launch_bounds__(256)
k_eval_nodal_points(int64_t nent, int npts, int ndof, const double *__restrict__ tab,
                    const int32_t *__restrict__ cells, int nvpc, const int32_t *__restrict__ c2x, int nx,
                    int64_t nv, int64_t nsecond, int has_centre, const int32_t *__restrict__ ent,
                    const double *__restrict__ nodal, double *__restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= nent * npts) return;
  const int64_t e = i / npts;
  const int q = (int)(i - e * npts);
  const int64_t c = ent ? ent[2 * e] : e;
  const int lf = ent ? ent[2 * e + 1] : 0;
  const double *t = tab + ((int64_t)lf * npts + q) * ndof;
  double acc = 0.0;
  for (int d = 0; d < ndof; ++d) {
    int64_t dof;
    if (d < nvpc) dof = cells[c * nvpc + d];
    else if (d < nvpc + nx) dof = nv + c2x[c * nx + (d - nvpc)];
    else dof = nv + nsecond + c;
    (void)has_centre;
    acc = acc + nodal[dof] * t[d];
  }
  out[i] = acc;
}

// physical point = N_0 x_0 + N_1 x_1 + ... (the order of the host shim's `_push`)
__global__ void __launch_bounds__(256)
k_physical_points(int64_t nent, int npts, int nfun, int gdim, const double *__restrict__ tab,
                  const int32_t *__restrict__ cells, int nvpc, const int32_t *__restrict__ ent, FacetVerts fvs,
                  const double *__restrict__ x, double *__restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= nent * npts) return;
  const int64_t e = i / npts;
  const int q = (int)(i - e * npts);
  const int64_t c = ent ? ent[2 * e] : e;
  const int lf = ent ? ent[2 * e + 1] : 0;
  for (int a = 0; a < gdim; ++a) {
    double acc = 0.0;
    for (int j = 0; j < nfun; ++j) {
      const int lv = ent ? fvs.fv[lf][j] : j;
      const double xv = x[(int64_t)cells[c * nvpc + lv] * gdim + a];
      acc = j == 0 ? tab[q * nfun] * xv : acc + tab[q * nfun + j] * xv;
    }
    out[i * gdim + a] = acc;
  }
}

static void p2_basis_row(int nvpc, const double *lam, double *N) {
  static const int ev3[3][2] = {{1, 2}, {0, 2}, {0, 1}};
  static const int ev4[6][2] = {{2, 3}, {1, 3}, {1, 2}, {0, 3}, {0, 2}, {0, 1}};
  for (int i = 0; i < nvpc; ++i) N[i] = lam[i] * (2.0 * lam[i] - 1.0);
  const int ne = nvpc == 3 ? 3 : 6;
  for (int k = 0; k < ne; ++k) {
    const int a = nvpc == 3 ? ev3[k][0] : ev4[k][0], b = nvpc == 3 ? ev3[k][1] : ev4[k][1];
    N[nvpc + k] = 4.0 * lam[a] * lam[b];
  }
}
static void q2_basis_row(double px, double py, double *N) {
  static const int idx[9][2] = {{0, 0}, {2, 0}, {0, 2}, {2, 2}, {1, 0}, {0, 1}, {2, 1}, {1, 2}, {1, 1}};
  const double lx[3] = {2.0 * (px - 0.5) * (px - 1.0), 4.0 * px * (1.0 - px), 2.0 * px * (px - 0.5)};
  const double ly[3] = {2.0 * (py - 0.5) * (py - 1.0), 4.0 * py * (1.0 - py), 2.0 * py * (py - 0.5)};
  for (int d = 0; d < 9; ++d) N[d] = lx[idx[d][0]] * ly[idx[d][1]];
}

static int levelset_counts(phx_mesh *m, int degree, int *nptc, int *nptf) {
  int64_t n0 = 0, n1 = 0;
  PHX_CHECK(phx_detection_points(m->cell_type, degree, 0, nullptr, &n0));
  PHX_CHECK(phx_detection_points(m->cell_type, degree, 1, nullptr, &n1));
  *nptc = (int)n0; *nptf = (int)n1;
  return PHX_OK;
}

extern "C" int phx_levelset_points_count(phx_mesh *m, int detection_degree, int64_t *count) {
  int nptc, nptf;
  PHX_CHECK(levelset_counts(m, detection_degree, &nptc, &nptf));
  *count = m->nc * (int64_t)nptc + m->nbf * (int64_t)nptf;
  return PHX_OK;
}

extern "C" int phx_levelset_eval_points(phx_mesh *m, int detection_degree, const double *nodal, int loc,
                                        double *out_device) {
  PHX_HIP(hipSetDevice(m->device));
  const bool quad = m->cell_type == PHX_QUADRILATERAL;
  PHX_REQUIRE(quad || m->cell_type == PHX_TRIANGLE || m->cell_type == PHX_TETRAHEDRON, PHX_ERR_NOT_IMPLEMENTED,
              "degree-2 level-sets are implemented on simplices and quadrilaterals");
  const int nvpc = m->ci.nvpc, tdim = m->ci.tdim;
  int nptc, nptf;
  PHX_CHECK(levelset_counts(m, detection_degree, &nptc, &nptf));
  std::vector<double> pc((size_t)nptc * tdim), pf((size_t)nptf * (tdim - 1));
  int64_t n = nptc;
  PHX_CHECK(phx_detection_points(m->cell_type, detection_degree, 0, pc.data(), &n));
  n = nptf;
  PHX_CHECK(phx_detection_points(m->cell_type, detection_degree, 1, pf.data(), &n));
  int nx, ndof;
  const int32_t *c2x;
  int64_t nsecond, nnodal;
  if (quad) {
    nx = 4; ndof = 9; c2x = m->c2f; nsecond = m->nf; nnodal = m->nv + m->nf + m->nc;
  } else {
    PHX_CHECK(phx_mesh_build_edges(m));
    nx = nvpc == 3 ? 3 : 6; ndof = nvpc + nx; c2x = m->c2e; nsecond = m->ne; nnodal = m->nv + m->ne;
  }
  // tables: cells [nptc][ndof]; facets [nfpc][nptf][ndof]
  const int nfpc = m->ci.nfpc, nvpf = m->ci.nvpf;
  std::vector<double> tc((size_t)nptc * ndof), tf((size_t)nfpc * nptf * ndof);
  for (int q = 0; q < nptc; ++q) {
    if (quad) q2_basis_row(pc[2 * q], pc[2 * q + 1], &tc[(size_t)q * ndof]);
    else {
      double lam[4] = {1.0, 0.0, 0.0, 0.0};
      for (int a = 0; a < tdim; ++a) lam[a + 1] = pc[(size_t)q * tdim + a];
      lam[0] = tdim == 2 ? (1.0 - lam[1]) - lam[2] : ((1.0 - lam[1]) - lam[2]) - lam[3];   // as the P1 shape table
      p2_basis_row(nvpc, lam, &tc[(size_t)q * ndof]);
    }
  }
  for (int lf = 0; lf < nfpc; ++lf)
    for (int q = 0; q < nptf; ++q) {
      double *row = &tf[((size_t)lf * nptf + q) * ndof];
      if (quad) {
        // facet lf runs from its first to its second vertex (tensor-product vertex order of the unit square)
        static const double vx[4] = {0.0, 1.0, 0.0, 1.0}, vy[4] = {0.0, 0.0, 1.0, 1.0};
        const int a = m->ci.fv[lf][0], b = m->ci.fv[lf][1];
        const double s = pf[q];
        const double px = vx[a] == vx[b] ? vx[a] : s, py = vy[a] == vy[b] ? vy[a] : s;
        q2_basis_row(px, py, row);
      } else {
        double mu[3] = {0.0, 0.0, 0.0};
        if (nvpf == 2) { mu[0] = 1.0 - pf[q]; mu[1] = pf[q]; }
        else { mu[1] = pf[2 * q]; mu[2] = pf[2 * q + 1]; mu[0] = (1.0 - mu[1]) - mu[2]; }
        double lam[4] = {0.0, 0.0, 0.0, 0.0};
        for (int j = 0; j < nvpf; ++j) lam[m->ci.fv[lf][j]] = mu[j];
        p2_basis_row(nvpc, lam, row);
      }
    }
  double *dtc = nullptr, *dtf = nullptr, *dn = nullptr;
  PHX_HIP(phx_malloc(&dtc, sizeof(double) * tc.size()));
  PHX_HIP(phx_malloc(&dtf, sizeof(double) * tf.size()));
  PHX_HIP(hipMemcpyAsync(dtc, tc.data(), sizeof(double) * tc.size(), hipMemcpyHostToDevice, m->stream));
  PHX_HIP(hipMemcpyAsync(dtf, tf.data(), sizeof(double) * tf.size(), hipMemcpyHostToDevice, m->stream));
  const double *dnodal = nodal;
  if (loc != PHX_DEVICE) {
    PHX_HIP(phx_malloc(&dn, sizeof(double) * (size_t)nnodal));
    PHX_HIP(hipMemcpyAsync(dn, nodal, sizeof(double) * (size_t)nnodal, hipMemcpyHostToDevice, m->stream));
    dnodal = dn;
  }
  const dim3 block(256);
  if (m->nc > 0)
    k_eval_nodal_points<<<dim3((unsigned)phx_div_up(m->nc * nptc, 256)), block, 0, m->stream>>>(
        m->nc, nptc, ndof, dtc, m->cells, nvpc, c2x, nx, m->nv, nsecond, quad ? 1 : 0, nullptr, dnodal, out_device);
  if (m->nbf > 0)
    k_eval_nodal_points<<<dim3((unsigned)phx_div_up(m->nbf * nptf, 256)), block, 0, m->stream>>>(
        m->nbf, nptf, ndof, dtf, m->cells, nvpc, c2x, nx, m->nv, nsecond, quad ? 1 : 0, m->bfacets, dnodal,
        out_device + m->nc * (int64_t)nptc);
  PHX_HIP(hipGetLastError());
  PHX_HIP(hipStreamSynchronize(m->stream));   // the host tables and the staged copy go out of scope
  PHX_HIP(phx_free(dtc)); PHX_HIP(phx_free(dtf));
  if (dn) PHX_HIP(phx_free(dn));
  return PHX_OK;
}

extern "C" int phx_detection_points_physical(phx_mesh *m, int detection_degree, double *out_device) {
  PHX_HIP(hipSetDevice(m->device));
  DetTab tabc, tabf;
  PHX_CHECK(make_tab(m, detection_degree, 0, &tabc));
  PHX_CHECK(make_tab(m, detection_degree, 1, &tabf));
  FacetVerts fvs;
  fvs.nfpc = m->ci.nfpc; fvs.nvpf = m->ci.nvpf;
  for (int f = 0; f < 4; ++f) for (int k = 0; k < 3; ++k) fvs.fv[f][k] = m->ci.fv[f][k];
  double *dtc = nullptr, *dtf = nullptr;
  PHX_HIP(phx_malloc(&dtc, sizeof(double) * (size_t)(tabc.npts * tabc.nfun)));
  PHX_HIP(phx_malloc(&dtf, sizeof(double) * (size_t)(tabf.npts * tabf.nfun)));
  PHX_HIP(hipMemcpyAsync(dtc, tabc.N, sizeof(double) * (size_t)(tabc.npts * tabc.nfun), hipMemcpyHostToDevice, m->stream));
  PHX_HIP(hipMemcpyAsync(dtf, tabf.N, sizeof(double) * (size_t)(tabf.npts * tabf.nfun), hipMemcpyHostToDevice, m->stream));
  const dim3 block(256);
  if (m->nc > 0)
    k_physical_points<<<dim3((unsigned)phx_div_up(m->nc * tabc.npts, 256)), block, 0, m->stream>>>(
        m->nc, tabc.npts, tabc.nfun, m->gdim, dtc, m->cells, m->ci.nvpc, nullptr, fvs, m->x, out_device);
  if (m->nbf > 0)
    k_physical_points<<<dim3((unsigned)phx_div_up(m->nbf * tabf.npts, 256)), block, 0, m->stream>>>(
        m->nbf, tabf.npts, tabf.nfun, m->gdim, dtf, m->cells, m->ci.nvpc, m->bfacets, fvs, m->x,
        out_device + m->nc * (int64_t)tabc.npts * m->gdim);
  PHX_HIP(hipGetLastError());
  PHX_HIP(hipStreamSynchronize(m->stream));
  PHX_HIP(phx_free(dtc)); PHX_HIP(phx_free(dtf));
  return PHX_OK;
}
