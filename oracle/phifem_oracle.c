/* phifem_oracle.c -- C/OpenMP restatement of the hot path for the CPU BASELINE of bench.py.
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): called from tests/ and from the
 * cpu_baseline leg of bench.py, never from phifem_amd/.
 *
 * One call runs the whole pipeline on the synthetic BASELINE problem (3-D weak-Dirichlet Poisson,
 * unit sphere in [-1.5,1.5]^3, n^3 Kuhn cubes, P1 x P1, gamma = sigma = 1, detection degree 1,
 * single-layer cut, box mode):
 *   tag cells / facets   src/phifem/mesh_scripts.py:95-134, 284-390, 393-558, 137-192
 *   assemble             demo/weak-dirichlet/flower/main.py:112-154 (closed-form P1 integrals)
 *   solve                main.py:162-182 replaced by right-preconditioned BiCGStab on the active set (Jacobi, or the
 *                        box sine-transform preconditioner the GPU path uses)
 * Mesh, facet numbering and DoF layout follow the contracts of oracle/meshgen.py /
 * include/phifem_hip.h (closed-form Kuhn topology), so results can be compared index by index
 * with the numpy oracle and with the HIP library.
 * PARITY: tagging pinned through the numpy oracle (tests/test_c_oracle.py compares them);
 * assembly / solve PARITY UNPINNED against the reference.
 *
 * build: gcc -O2 -fopenmp -fPIC -shared -ffp-contract=off -o _build/libphifem_oracle.so phifem_oracle.c -lm
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static const int PERM3[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
static int perm_index3(int p0, int p1, int p2) { return p0 * 2 + (p1 > p2 ? 1 : 0); }

typedef struct {
  int64_t n, nv, nc, nf;
  int64_t base[13], ext[12][3];
  double *x;       /* nv*3 */
  int32_t *cells;  /* nc*4 */
  int32_t *c2f;    /* nc*4 */
  int32_t *f2c;    /* nf*2 */
} Mesh;

static void mesh_free(Mesh *m) { free(m->x); free(m->cells); free(m->c2f); free(m->f2c); }

static void mesh_build(Mesh *m, int64_t n) {
  m->n = n;
  m->nv = (n + 1) * (n + 1) * (n + 1);
  m->nc = 6 * n * n * n;
  for (int t = 0; t < 12; ++t) {
    for (int a = 0; a < 3; ++a) m->ext[t][a] = n;
    if (t < 6) m->ext[t][t / 2] += 1;
  }
  m->base[0] = 0;
  for (int t = 0; t < 12; ++t) m->base[t + 1] = m->base[t] + m->ext[t][0] * m->ext[t][1] * m->ext[t][2];
  m->nf = m->base[12];
  m->x = (double *)malloc(sizeof(double) * 3 * m->nv);
  m->cells = (int32_t *)malloc(sizeof(int32_t) * 4 * m->nc);
  m->c2f = (int32_t *)malloc(sizeof(int32_t) * 4 * m->nc);
  m->f2c = (int32_t *)malloc(sizeof(int32_t) * 2 * m->nf);
  const int64_t n1 = n + 1;
#pragma omp parallel for
  for (int64_t v = 0; v < m->nv; ++v) {
    const int64_t idx[3] = {v % n1, (v / n1) % n1, v / (n1 * n1)};
    for (int a = 0; a < 3; ++a) m->x[3 * v + a] = -1.5 + (1.5 - (-1.5)) * ((double)idx[a] / (double)n);
  }
  const int64_t stride[3] = {1, n1, n1 * n1};
#pragma omp parallel for
  for (int64_t c = 0; c < m->nc; ++c) {
    const int t = (int)(c % 6);
    const int64_t cube = c / 6;
    const int64_t o[3] = {cube % n, (cube / n) % n, cube / (n * n)};
    const int *p = PERM3[t];
    int64_t v = o[0] * stride[0] + o[1] * stride[1] + o[2] * stride[2];
    m->cells[4 * c] = (int32_t)v;
    for (int s = 0; s < 3; ++s) { v += stride[p[s]]; m->cells[4 * c + s + 1] = (int32_t)v; }
    for (int lf = 0; lf < 4; ++lf) {
      int type;
      int64_t an[3] = {o[0], o[1], o[2]};
      if (lf == 0) { type = p[0] * 2 + (p[1] > p[2] ? 1 : 0); an[p[0]] += 1; }
      else if (lf == 3) type = p[2] * 2 + (p[0] > p[1] ? 1 : 0);
      else if (lf == 1) type = 6 + p[2];
      else type = 9 + p[0];
      m->c2f[4 * c + lf] = (int32_t)(m->base[type] + an[0] + m->ext[type][0] * (an[1] + m->ext[type][1] * an[2]));
    }
  }
#pragma omp parallel for
  for (int64_t f = 0; f < m->nf; ++f) {
    int type = 0;
    while (type + 1 < 12 && f >= m->base[type + 1]) ++type;
    const int64_t r = f - m->base[type];
    int64_t o[3] = {r % m->ext[type][0], (r / m->ext[type][0]) % m->ext[type][1],
                    r / (m->ext[type][0] * m->ext[type][1])};
    int64_t c0 = -1, c1 = -1;
#define CUBE(q) ((q)[0] + n * ((q)[1] + n * (q)[2]))
    if (type < 6) {
      const int a = type / 2, s = type % 2;
      const int r0 = a == 0 ? 1 : 0, r1 = a == 2 ? 1 : 2;
      const int s0 = s ? r1 : r0, s1 = s ? r0 : r1;
      if (o[a] > 0) { int64_t q[3] = {o[0], o[1], o[2]}; q[a] -= 1; c0 = CUBE(q) * 6 + perm_index3(a, s0, s1); }
      if (o[a] < n) { const int64_t cc = CUBE(o) * 6 + perm_index3(s0, s1, a); if (c0 < 0) c0 = cc; else c1 = cc; }
    } else if (type < 9) {
      const int c = type - 6, r0 = c == 0 ? 1 : 0, r1 = c == 2 ? 1 : 2;
      c0 = CUBE(o) * 6 + perm_index3(r0, r1, c); c1 = CUBE(o) * 6 + perm_index3(r1, r0, c);
    } else {
      const int a = type - 9, r0 = a == 0 ? 1 : 0, r1 = a == 2 ? 1 : 2;
      c0 = CUBE(o) * 6 + perm_index3(a, r0, r1); c1 = CUBE(o) * 6 + perm_index3(a, r1, r0);
    }
#undef CUBE
    m->f2c[2 * f] = (int32_t)c0;
    m->f2c[2 * f + 1] = (int32_t)c1;
  }
}

/* ---- tagging (detection degree 1, nodal P1 level-set) ------------------------------------- */
static int detect(const double *ph, int k, double scale) {
  /* sum_q phi_q / sum_q |phi_q| over the k vertices, sequential (oracle/tagging.py:_ratio); samples of both signs: every
     term times `scale` (|det J| of the cell, the factor FFCx gives the terms of a dx sum) */
  double num = 0.0, den = 0.0;
  int pos = 0, neg = 0;
  for (int q = 0; q < k; ++q) { num = num + ph[q]; den = den + fabs(ph[q]); pos |= ph[q] > 0.0; neg |= ph[q] < 0.0; }
  if (pos && neg) {
    num = 0.0; den = 0.0;
    for (int q = 0; q < k; ++q) { const double t = ph[q] * scale; num = num + t; den = den + fabs(t); }
  }
  const double d = den > 0.0 ? num / den : 0.5;
  if (d == -1.0) return 1;
  if (d == 1.0) return 3;
  return 2;
}

static void tag_cells(const Mesh *m, const double *phi, int single_layer, int8_t *ct) {
#pragma omp parallel for
  for (int64_t c = 0; c < m->nc; ++c) {
    double ph[4], e[3][3];
    for (int i = 0; i < 4; ++i) ph[i] = phi[m->cells[4 * c + i]];
    /* |det J|: edge vectors from vertex 0, cofactor expansion along the first row (oracle/tagging.py:cell_scale) */
    const double *x0 = m->x + 3 * (int64_t)m->cells[4 * c];
    for (int a = 0; a < 3; ++a) {
      const double *xa = m->x + 3 * (int64_t)m->cells[4 * c + a + 1];
      for (int d = 0; d < 3; ++d) e[a][d] = xa[d] - x0[d];
    }
    const double c0 = e[1][1] * e[2][2] - e[1][2] * e[2][1];
    const double c1 = e[1][0] * e[2][2] - e[1][2] * e[2][0];
    const double c2 = e[1][0] * e[2][1] - e[1][1] * e[2][0];
    ct[c] = (int8_t)detect(ph, 4, fabs((e[0][0] * c0 - e[0][1] * c1) + e[0][2] * c2));
  }
  if (!single_layer) return;
  uint8_t *touched = (uint8_t *)calloc(m->nv, 1);
#pragma omp parallel for
  for (int64_t c = 0; c < m->nc; ++c)
    if (ct[c] == 1) for (int i = 0; i < 4; ++i) touched[m->cells[4 * c + i]] = 1;
#pragma omp parallel for
  for (int64_t c = 0; c < m->nc; ++c)
    if (ct[c] == 2) {
      int keep = 0;
      for (int i = 0; i < 4; ++i) keep |= touched[m->cells[4 * c + i]];
      if (!keep) ct[c] = 3;
    }
  free(touched);
}

static const int FV3[4][3] = {{1, 2, 3}, {0, 2, 3}, {0, 1, 3}, {0, 1, 2}};

static int64_t tag_facets(const Mesh *m, const double *phi, const int8_t *ct, int8_t *ft) {
  /* `ds` detection per boundary cell: one partial sum per boundary facet, local order */
  uint8_t *bcut = (uint8_t *)calloc(m->nc, 1);
  int64_t n3 = 0;
#pragma omp parallel for reduction(+ : n3)
  for (int64_t c = 0; c < m->nc; ++c) {
    if (ct[c] == 3) n3++;
    double num = 0.0, den = 0.0;
    int any = 0;
    for (int lf = 0; lf < 4; ++lf) {
      const int32_t f = m->c2f[4 * c + lf];
      if (m->f2c[2 * (int64_t)f + 1] >= 0) continue;
      any = 1;
      double pn = 0.0, pd = 0.0;
      for (int j = 0; j < 3; ++j) { const double p = phi[m->cells[4 * c + FV3[lf][j]]]; pn = pn + p; pd = pd + fabs(p); }
      num = num + pn; den = den + pd;
    }
    if (!any) continue;
    const double d = den > 0.0 ? num / den : 0.5;
    bcut[c] = (d > -1.0 && d < 1.0);
  }
  const int no_ext = n3 == 0;
  int64_t bad = 0;
#pragma omp parallel for reduction(+ : bad)
  for (int64_t f = 0; f < m->nf; ++f) {
    const int32_t c0 = m->f2c[2 * f], c1 = m->f2c[2 * f + 1];
    const int t0 = ct[c0], t1 = c1 >= 0 ? ct[c1] : 0;
    const int I = t0 == 1 || t1 == 1, C = t0 == 2 || t1 == 2, E = t0 == 3 || t1 == 3, B = c1 < 0;
    const int cc = bcut[c0];
    const int CB = B && cc, UB = B && !cc && !E && !I, IB = I && C;
    int BF = no_ext ? B : ((E && C) || UB);
    const int DI = E && I;
    const int cut = (C && !(BF || IB || DI || UB)) || CB;
    const int inte = I && !(IB || BF || DI), ext = E && !(IB || BF || DI);
    BF = BF && !cut;
    int t = 0;
    if (ext) t = 5;
    if (inte) t = 1;
    if (IB) t = 3;
    if (cut) t = 2;
    if (BF) t = 4;
    if (DI) t = 6;
    ft[f] = (int8_t)t;
    if (ext + inte + IB + cut + BF + DI != 1) bad++;
  }
  free(bcut);
  return bad;
}

/* ---- geometry + element integrals (closed forms of oracle/assembly.py) ---------------------- */
typedef struct { double g[4][3], vol, h; } Geo;
static void geometry(const Mesh *m, int64_t c, Geo *G) {
  const double *X[4];
  for (int i = 0; i < 4; ++i) X[i] = m->x + 3 * (int64_t)m->cells[4 * c + i];
  double e[3][3], cr[3][3];
  for (int k = 0; k < 3; ++k) for (int d = 0; d < 3; ++d) e[k][d] = X[k + 1][d] - X[0][d];
  cr[0][0] = e[1][1] * e[2][2] - e[1][2] * e[2][1]; cr[0][1] = e[1][2] * e[2][0] - e[1][0] * e[2][2]; cr[0][2] = e[1][0] * e[2][1] - e[1][1] * e[2][0];
  cr[1][0] = e[2][1] * e[0][2] - e[2][2] * e[0][1]; cr[1][1] = e[2][2] * e[0][0] - e[2][0] * e[0][2]; cr[1][2] = e[2][0] * e[0][1] - e[2][1] * e[0][0];
  cr[2][0] = e[0][1] * e[1][2] - e[0][2] * e[1][1]; cr[2][1] = e[0][2] * e[1][0] - e[0][0] * e[1][2]; cr[2][2] = e[0][0] * e[1][1] - e[0][1] * e[1][0];
  const double det = e[0][0] * cr[0][0] + e[0][1] * cr[0][1] + e[0][2] * cr[0][2];
  for (int k = 0; k < 3; ++k) for (int d = 0; d < 3; ++d) G->g[k + 1][d] = cr[k][d] / det;
  for (int d = 0; d < 3; ++d) G->g[0][d] = -(G->g[1][d] + G->g[2][d] + G->g[3][d]);
  G->vol = fabs(det) / 6.0;
  double h2 = 0.0;
  for (int i = 0; i < 4; ++i) for (int j = i + 1; j < 4; ++j) {
    double s = 0.0;
    for (int d = 0; d < 3; ++d) { const double t = X[i][d] - X[j][d]; s += t * t; }
    if (s > h2) h2 = s;
  }
  G->h = sqrt(h2);
}
static double mult4(int i, int j, int k, int l) {
  int cnt[4] = {0, 0, 0, 0};
  cnt[i]++; cnt[j]++; cnt[k]++; cnt[l]++;
  double r = 1.0;
  for (int a = 0; a < 4; ++a) r *= cnt[a] == 2 ? 2.0 : (cnt[a] == 3 ? 6.0 : (cnt[a] == 4 ? 24.0 : 1.0));
  return r;
}

typedef struct { int32_t r, c; double v; } Trip;
typedef struct { Trip *t; int64_t n, cap; } TripBuf;
static void push(TripBuf *b, int32_t r, int32_t c, double v) {
  if (b->n == b->cap) { b->cap = b->cap ? 2 * b->cap : 1 << 16; b->t = (Trip *)realloc(b->t, sizeof(Trip) * b->cap); }
  b->t[b->n].r = r; b->t[b->n].c = c; b->t[b->n].v = v; b->n++;
}
static int cmp_col(const void *a, const void *b) {
  const Trip *x = (const Trip *)a, *y = (const Trip *)b;
  return x->c < y->c ? -1 : (x->c > y->c ? 1 : 0);
}

typedef struct {
  int64_t n, nu, nnz;
  int64_t *rowptr;
  int32_t *col;
  double *val, *rhs;
  int64_t *full_of_active;
} Csr;

static void csr_free(Csr *A) { free(A->rowptr); free(A->col); free(A->val); free(A->rhs); free(A->full_of_active); }

static void assemble(const Mesh *m, const int8_t *ct, const int8_t *ft, const double *phi,
                     const double *f, const double *ud, double gam, double sig, Csr *A) {
  const int64_t nv = m->nv;
  int32_t *du = (int32_t *)malloc(sizeof(int32_t) * nv), *dp = (int32_t *)malloc(sizeof(int32_t) * nv);
  uint8_t *fu = (uint8_t *)calloc(nv, 1), *fp = (uint8_t *)calloc(nv, 1);
#pragma omp parallel for
  for (int64_t c = 0; c < m->nc; ++c)
    if (ct[c] == 1 || ct[c] == 2)
      for (int i = 0; i < 4; ++i) { fu[m->cells[4 * c + i]] = 1; if (ct[c] == 2) fp[m->cells[4 * c + i]] = 1; }
  int64_t nu = 0, np = 0;
  for (int64_t v = 0; v < nv; ++v) du[v] = fu[v] ? (int32_t)nu++ : -1;
  for (int64_t v = 0; v < nv; ++v) dp[v] = fp[v] ? (int32_t)(nu + np++) : -1;
  const int64_t n = nu + np;
  A->n = n; A->nu = nu;
  A->full_of_active = (int64_t *)malloc(sizeof(int64_t) * n);
  for (int64_t v = 0; v < nv; ++v) { if (fu[v]) A->full_of_active[du[v]] = v; if (fp[v]) A->full_of_active[dp[v]] = nv + v; }
  A->rhs = (double *)calloc(n, sizeof(double));
  const int nt = omp_get_max_threads();
  TripBuf *bufs = (TripBuf *)calloc(nt, sizeof(TripBuf));
  double **rhs_t = (double **)malloc(sizeof(double *) * nt);
  for (int t = 0; t < nt; ++t) rhs_t[t] = (double *)calloc(n, sizeof(double));
  const double c2 = 1.0 / 20.0, c3 = 1.0 / 120.0, c4 = 1.0 / 840.0;
#pragma omp parallel
  {
    TripBuf *B = &bufs[omp_get_thread_num()];
    double *rh = rhs_t[omp_get_thread_num()];
#pragma omp for schedule(dynamic, 4096)
    for (int64_t c = 0; c < m->nc; ++c) {
      const int t = ct[c];
      if (t != 1 && t != 2) continue;
      Geo G;
      geometry(m, c, &G);
      const int32_t *v = m->cells + 4 * c;
      double sf = 0.0;
      for (int i = 0; i < 4; ++i) sf += f[v[i]];
      const double pen = t == 2 ? gam * G.vol / (G.h * G.h) : 0.0;
      for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 4; ++j) {
          double k = 0.0;
          for (int d = 0; d < 3; ++d) k += G.g[i][d] * G.g[j][d];
          push(B, du[v[i]], du[v[j]], k * G.vol + pen * c2 * (i == j ? 2.0 : 1.0));
        }
        rh[du[v[i]]] += G.vol * c2 * (sf + f[v[i]]);
      }
      /* one-sided boundary term: facets tagged 4 seen from this cell (tags 1,2) */
      for (int lf = 0; lf < 4; ++lf) {
        if (ft[m->c2f[4 * c + lf]] != 4) continue;
        for (int j = 0; j < 4; ++j) {
          double k = 0.0;
          for (int d = 0; d < 3; ++d) k += G.g[j][d] * G.g[lf][d];
          for (int i = 0; i < 4; ++i) if (i != lf) push(B, du[v[i]], du[v[j]], k * G.vol);
        }
      }
      if (t != 2) continue;
      double ph[4], u[4], sp = 0.0, su = 0.0;
      for (int i = 0; i < 4; ++i) { ph[i] = phi[v[i]]; u[i] = ud[v[i]]; sp += ph[i]; su += u[i]; }
      const double h1 = 1.0 / G.h, w3 = -gam * G.vol * h1 * h1 * h1 * c3, w4 = gam * G.vol * h1 * h1 * h1 * h1 * c4;
      for (int i = 0; i < 4; ++i) {
        double bq = 0.0;
        for (int j = 0; j < 4; ++j) {
          const double m3 = (i == j ? 2.0 : 1.0) * (sp + ph[i] + ph[j]);
          push(B, du[v[i]], dp[v[j]], w3 * m3);
          push(B, dp[v[i]], du[v[j]], w3 * m3);
          double m4 = 0.0;
          for (int k = 0; k < 4; ++k) for (int l = 0; l < 4; ++l) m4 += mult4(i, j, k, l) * ph[k] * ph[l];
          push(B, dp[v[i]], dp[v[j]], w4 * m4);
          bq += u[j] * m3;
        }
        rh[du[v[i]]] += pen * c2 * (su + u[i]);
        rh[dp[v[i]]] += w3 * bq;
      }
    }
#pragma omp for schedule(dynamic, 4096)
    for (int64_t fct = 0; fct < m->nf; ++fct) {
      if ((ft[fct] != 2 && ft[fct] != 3) || m->f2c[2 * fct + 1] < 0) continue;
      int32_t dofs[8];
      double J[8], hsum = 0.0, area = 0.0;
      for (int side = 0; side < 2; ++side) {
        const int64_t c = m->f2c[2 * fct + side];
        Geo G;
        geometry(m, c, &G);
        int lf = 0;
        for (int k = 0; k < 4; ++k) if (m->c2f[4 * c + k] == (int32_t)fct) lf = k;
        double gn = 0.0;
        for (int d = 0; d < 3; ++d) gn += G.g[lf][d] * G.g[lf][d];
        gn = sqrt(gn);
        if (side == 0) area = 3.0 * G.vol * gn;
        hsum += G.h;
        for (int j = 0; j < 4; ++j) {
          double s = 0.0;
          for (int d = 0; d < 3; ++d) s += G.g[j][d] * G.g[lf][d];
          J[side * 4 + j] = -s / gn;
          dofs[side * 4 + j] = du[m->cells[4 * c + j]];
        }
      }
      const double w = sig * 0.5 * hsum * area;
      for (int a = 0; a < 8; ++a) for (int b = 0; b < 8; ++b) push(B, dofs[a], dofs[b], w * J[a] * J[b]);
    }
  }
  /* bucket the triplets by row (counting sort), then sort + merge every row */
  int64_t *cnt = (int64_t *)calloc(n + 1, sizeof(int64_t));
#pragma omp parallel
  {
    const TripBuf *B = &bufs[omp_get_thread_num()];
    for (int64_t i = 0; i < B->n; ++i) {
#pragma omp atomic
      cnt[B->t[i].r + 1]++;
    }
  }
  for (int64_t r = 0; r < n; ++r) cnt[r + 1] += cnt[r];
  const int64_t ntrip = cnt[n];
  Trip *all = (Trip *)malloc(sizeof(Trip) * (ntrip ? ntrip : 1));
  int64_t *cur = (int64_t *)malloc(sizeof(int64_t) * (n + 1));
  memcpy(cur, cnt, sizeof(int64_t) * (n + 1));
#pragma omp parallel
  {
    TripBuf *B = &bufs[omp_get_thread_num()];
    for (int64_t i = 0; i < B->n; ++i) {
      int64_t pos;
#pragma omp atomic capture
      pos = cur[B->t[i].r]++;
      all[pos] = B->t[i];
    }
    free(B->t);
  }
#pragma omp parallel for
  for (int64_t r = 0; r < n; ++r) {
    double acc = 0.0;
    for (int t = 0; t < nt; ++t) acc += rhs_t[t][r];
    A->rhs[r] = acc;
  }
  for (int t = 0; t < nt; ++t) free(rhs_t[t]);
  A->rowptr = (int64_t *)calloc(n + 1, sizeof(int64_t));
#pragma omp parallel for schedule(dynamic, 1024)
  for (int64_t r = 0; r < n; ++r) {
    Trip *b = all + cnt[r];
    const int64_t k = cnt[r + 1] - cnt[r];
    qsort(b, k, sizeof(Trip), cmp_col);
    int64_t u = 0;
    for (int64_t i = 0; i < k; ++i) {
      if (u > 0 && b[u - 1].c == b[i].c) b[u - 1].v += b[i].v; else b[u++] = b[i];
    }
    A->rowptr[r + 1] = u;
  }
  for (int64_t r = 0; r < n; ++r) A->rowptr[r + 1] += A->rowptr[r];
  A->nnz = A->rowptr[n];
  A->col = (int32_t *)malloc(sizeof(int32_t) * A->nnz);
  A->val = (double *)malloc(sizeof(double) * A->nnz);
#pragma omp parallel for schedule(dynamic, 1024)
  for (int64_t r = 0; r < n; ++r) {
    const Trip *b = all + cnt[r];
    for (int64_t i = 0; i < A->rowptr[r + 1] - A->rowptr[r]; ++i) { A->col[A->rowptr[r] + i] = b[i].c; A->val[A->rowptr[r] + i] = b[i].v; }
  }
  free(all); free(cnt); free(cur); free(bufs); free(rhs_t); free(du); free(dp); free(fu); free(fp);
}

/* ---- box preconditioner: lattice Laplacian of a box around the active u vertices, inverted by type-I sine
 * transforms (the CPU counterpart of phifem_amd/csrc/phx_precond.inc.hip; same box rule: margin 4, transform
 * lengths from {64,...,1024}).  DST-I of two real lines through one complex FFT of the odd extensions
 * (length 2L): W = FFT(ext(a) + i ext(b)),  F^a = -Im W / 2,  F^b = Re W / 2. ------------------------------ */
typedef struct { int n; double *wr, *wi; } Tw;          /* exp(-2 pi i j / n), j < n */
static void tw_init(Tw *t, int n) {
  t->n = n; t->wr = (double *)malloc(sizeof(double) * n); t->wi = (double *)malloc(sizeof(double) * n);
  for (int j = 0; j < n; ++j) { t->wr[j] = cos(-2.0 * M_PI * j / n); t->wi[j] = sin(-2.0 * M_PI * j / n); }
}
/* recursive decimation in time, radices 2 and 3; in[] read with stride s, out[] contiguous; tw for N, step = N / n */
static void fft_rec(int n, const double *ir, const double *ii, int s, double *or_, double *oi, const Tw *t) {
  if (n == 1) { or_[0] = ir[0]; oi[0] = ii[0]; return; }
  if (n == 2) {
    or_[0] = ir[0] + ir[s]; oi[0] = ii[0] + ii[s]; or_[1] = ir[0] - ir[s]; oi[1] = ii[0] - ii[s];
    return;
  }
  if (n == 4) {
    const double ar = ir[0] + ir[2 * s], ai = ii[0] + ii[2 * s], br = ir[0] - ir[2 * s], bi = ii[0] - ii[2 * s];
    const double cr = ir[s] + ir[3 * s], ci = ii[s] + ii[3 * s], dr = ir[s] - ir[3 * s], di = ii[s] - ii[3 * s];
    or_[0] = ar + cr; oi[0] = ai + ci; or_[2] = ar - cr; oi[2] = ai - ci;
    or_[1] = br + di; oi[1] = bi - dr; or_[3] = br - di; oi[3] = bi + dr;   /* (d) * (-i) = (di, -dr) */
    return;
  }
  const int step = t->n / n;
  if (n % 2 == 0) {
    const int h = n / 2;
    fft_rec(h, ir, ii, 2 * s, or_, oi, t);
    fft_rec(h, ir + s, ii + s, 2 * s, or_ + h, oi + h, t);
    for (int k = 0; k < h; ++k) {
      const double wr = t->wr[k * step], wi = t->wi[k * step];
      const double br = or_[h + k] * wr - oi[h + k] * wi, bi = or_[h + k] * wi + oi[h + k] * wr;
      const double ar = or_[k], ai = oi[k];
      or_[k] = ar + br; oi[k] = ai + bi; or_[h + k] = ar - br; oi[h + k] = ai - bi;
    }
  } else {  /* n % 3 == 0 */
    const int h = n / 3;
    for (int q = 0; q < 3; ++q) fft_rec(h, ir + q * s, ii + q * s, 3 * s, or_ + q * h, oi + q * h, t);
    const double c3 = -0.5, s3 = -0.86602540378443864676;  /* exp(-2 pi i / 3) */
    for (int k = 0; k < h; ++k) {
      const double w1r = t->wr[k * step], w1i = t->wi[k * step];
      const double w2r = t->wr[(2 * k * step) % t->n], w2i = t->wi[(2 * k * step) % t->n];
      const double ar = or_[k], ai = oi[k];
      const double br = or_[h + k] * w1r - oi[h + k] * w1i, bi = or_[h + k] * w1i + oi[h + k] * w1r;
      const double cr = or_[2 * h + k] * w2r - oi[2 * h + k] * w2i, ci = or_[2 * h + k] * w2i + oi[2 * h + k] * w2r;
      const double tr = br + cr, ti = bi + ci, dr = br - cr, di = bi - ci;
      or_[k] = ar + tr; oi[k] = ai + ti;
      const double mr = ar + c3 * tr, mi = ai + c3 * ti;
      /* (b - c) * (i s3):  i s3 (dr + i di) = -s3 di + i s3 dr */
      or_[h + k] = mr - s3 * di; oi[h + k] = mi + s3 * dr;
      or_[2 * h + k] = mr + s3 * di; oi[2 * h + k] = mi - s3 * dr;
    }
  }
}
typedef struct {
  int L[3], m[3], lo[3];
  int64_t n1;            /* vertices per axis of the mesh */
  double *lam[3], scale;
  Tw tw[3];
  double *G;             /* m0 * m1 * m2 lattice */
  int32_t *gmap;         /* lattice point -> active u index or -1 */
} BoxPc;
static int pick_length(int need) {
  const int cand[] = {64, 128, 192, 256, 384, 512, 768, 1024};
  for (int i = 0; i < 8; ++i) if (cand[i] >= need) return cand[i];
  return -1;
}
/* in-place DST-I of the two lines a, b (m = L - 1 values each, strides sa) */
static void dst_pair(int L, double *a, double *b, int64_t st, const Tw *t, double *buf) {
  const int N = 2 * L;
  double *ir = buf, *ii = buf + N, *or_ = buf + 2 * N, *oi = buf + 3 * N;
  ir[0] = ii[0] = ir[L] = ii[L] = 0.0;
  for (int j = 1; j < L; ++j) {
    const double va = a[(j - 1) * st], vb = b ? b[(j - 1) * st] : 0.0;
    ir[j] = va; ir[N - j] = -va; ii[j] = vb; ii[N - j] = -vb;
  }
  fft_rec(N, ir, ii, 1, or_, oi, t);
  for (int k = 1; k < L; ++k) { a[(k - 1) * st] = -0.5 * oi[k]; if (b) b[(k - 1) * st] = 0.5 * or_[k]; }
}
static void dst_axis(BoxPc *P, int axis) {
  const int64_t m0 = P->m[0], m1 = P->m[1], m2 = P->m[2];
  const int64_t st = axis == 0 ? 1 : (axis == 1 ? m0 : m0 * m1);
  const int64_t nl = m0 * m1 * m2 / P->m[axis];
  const int L = P->L[axis];
#pragma omp parallel
  {
    double *buf = (double *)malloc(sizeof(double) * 8 * L);
#pragma omp for schedule(static)
    for (int64_t lp = 0; lp < (nl + 1) / 2; ++lp) {
      double *ln[2] = {0, 0};
      for (int c = 0; c < 2; ++c) {
        const int64_t l = 2 * lp + c;
        if (l >= nl) break;
        int64_t off;
        if (axis == 0) off = l * m0;                       /* l = y + m1 z */
        else if (axis == 1) off = (l % m0) + (l / m0) * m0 * m1;  /* l = x + m0 z */
        else off = l;                                       /* l = x + m0 y */
        ln[c] = P->G + off;
      }
      dst_pair(L, ln[0], ln[1], st, &P->tw[axis], buf);
    }
    free(buf);
  }
}
static int boxpc_setup(BoxPc *P, const Csr *A, int64_t n_cubes) {
  const int64_t n1 = n_cubes + 1, nv = n1 * n1 * n1;
  int lo[3] = {1 << 30, 1 << 30, 1 << 30}, hi[3] = {-1, -1, -1};
  for (int64_t i = 0; i < A->nu; ++i) {
    const int64_t v = A->full_of_active[i];
    if (v >= nv) continue;
    const int idx[3] = {(int)(v % n1), (int)((v / n1) % n1), (int)(v / (n1 * n1))};
    for (int a = 0; a < 3; ++a) { if (idx[a] < lo[a]) lo[a] = idx[a]; if (idx[a] > hi[a]) hi[a] = idx[a]; }
  }
  if (hi[0] < 0) return -1;
  const double h = 3.0 / (double)n_cubes;
  P->n1 = n1;
  P->scale = 1.0;
  for (int a = 0; a < 3; ++a) {
    const int extent = hi[a] - lo[a] + 1;
    P->L[a] = pick_length(extent + 2 * 4 + 1);
    if (P->L[a] < 0) return -1;
    P->m[a] = P->L[a] - 1;
    P->lo[a] = lo[a] - 1 - (P->L[a] - 1 - extent) / 2;
    P->scale *= 2.0 / P->L[a];
    tw_init(&P->tw[a], 2 * P->L[a]);
    P->lam[a] = (double *)malloc(sizeof(double) * P->L[a]);
    for (int k = 0; k < P->L[a]; ++k) P->lam[a][k] = h * (2.0 - 2.0 * cos(M_PI * k / P->L[a]));  /* h_b h_c / h_a = h */
  }
  const int64_t tot = (int64_t)P->m[0] * P->m[1] * P->m[2];
  P->G = (double *)malloc(sizeof(double) * tot);
  P->gmap = (int32_t *)malloc(sizeof(int32_t) * tot);
  for (int64_t e = 0; e < tot; ++e) P->gmap[e] = -1;
  for (int64_t i = 0; i < A->nu; ++i) {
    const int64_t v = A->full_of_active[i];
    const int64_t x = v % n1 - P->lo[0] - 1, y = (v / n1) % n1 - P->lo[1] - 1, z = v / (n1 * n1) - P->lo[2] - 1;
    P->gmap[x + P->m[0] * (y + (int64_t)P->m[1] * z)] = (int32_t)i;
  }
  return 0;
}
static void boxpc_free(BoxPc *P) {
  for (int a = 0; a < 3; ++a) { free(P->lam[a]); free(P->tw[a].wr); free(P->tw[a].wi); }
  free(P->G); free(P->gmap);
}
/* out = P in:  u rows: D K_box^-1 (the iteration runs on A D^-1), other rows: identity */
static void boxpc_apply(BoxPc *P, const Csr *A, const double *dinv, const double *in, double *out) {
  const int64_t m0 = P->m[0], m1 = P->m[1], m2 = P->m[2], tot = m0 * m1 * m2;
#pragma omp parallel for
  for (int64_t e = 0; e < tot; ++e) P->G[e] = P->gmap[e] >= 0 ? in[P->gmap[e]] : 0.0;
  for (int a = 0; a < 3; ++a) dst_axis(P, a);
#pragma omp parallel for
  for (int64_t e = 0; e < tot; ++e) {
    const int64_t x = e % m0, y = (e / m0) % m1, z = e / (m0 * m1);
    P->G[e] *= P->scale / (P->lam[0][x + 1] + P->lam[1][y + 1] + P->lam[2][z + 1]);
  }
  for (int a = 2; a >= 0; --a) dst_axis(P, a);
#pragma omp parallel for
  for (int64_t i = A->nu; i < A->n; ++i) out[i] = in[i];
#pragma omp parallel for
  for (int64_t e = 0; e < tot; ++e) if (P->gmap[e] >= 0) out[P->gmap[e]] = P->G[e] / dinv[P->gmap[e]];
}

/* ---- right-preconditioned BiCGStab (precond = 0: Jacobi, 1: box sine transforms on u + Jacobi on p) ---- */
static void spmv_scaled(const Csr *A, const double *dinv, const double *x, double *y) {
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < A->n; ++r) {
    double s = 0.0;
    for (int64_t k = A->rowptr[r]; k < A->rowptr[r + 1]; ++k) s += A->val[k] * dinv[A->col[k]] * x[A->col[k]];
    y[r] = s;
  }
}
static double dot(int64_t n, const double *a, const double *b) {
  double s = 0.0;
#pragma omp parallel for reduction(+ : s)
  for (int64_t i = 0; i < n; ++i) s += a[i] * b[i];
  return s;
}
static int bicgstab(const Csr *A, double rtol, int64_t max_iter, double *xout, double *relres_out, BoxPc *pc) {
  const int64_t n = A->n;
  double *dinv = (double *)malloc(sizeof(double) * n), *w = (double *)calloc(9 * n, sizeof(double));
  double *r = w, *rh = w + n, *p = w + 2 * n, *v = w + 3 * n, *s = w + 4 * n, *t = w + 5 * n, *y = w + 6 * n;
  double *ph = pc ? w + 7 * n : p, *sh = pc ? w + 8 * n : s;
#pragma omp parallel for
  for (int64_t i = 0; i < n; ++i) {
    double d = 1.0;
    for (int64_t k = A->rowptr[i]; k < A->rowptr[i + 1]; ++k) if (A->col[k] == i) d = A->val[k];
    dinv[i] = 1.0 / d;
    r[i] = rh[i] = p[i] = A->rhs[i];
  }
  const double bb = dot(n, r, r);
  double rho = bb, relres = bb == 0.0 ? 0.0 : 1.0;
  int64_t it = 0;
  while (bb != 0.0 && it < max_iter) {
    if (pc) boxpc_apply(pc, A, dinv, p, ph);
    spmv_scaled(A, dinv, ph, v);
    const double alpha = rho / dot(n, rh, v);
#pragma omp parallel for
    for (int64_t i = 0; i < n; ++i) s[i] = r[i] - alpha * v[i];
    if (pc) boxpc_apply(pc, A, dinv, s, sh);
    spmv_scaled(A, dinv, sh, t);
    const double omega = dot(n, t, s) / dot(n, t, t);
#pragma omp parallel for
    for (int64_t i = 0; i < n; ++i) { y[i] += alpha * ph[i] + omega * sh[i]; r[i] = s[i] - omega * t[i]; }
    const double rho_new = dot(n, rh, r), rr = dot(n, r, r);
    ++it;
    relres = sqrt(rr / bb);
    if (relres <= rtol) break;
    const double beta = (rho_new / rho) * (alpha / omega);
#pragma omp parallel for
    for (int64_t i = 0; i < n; ++i) p[i] = r[i] + beta * (p[i] - omega * v[i]);
    rho = rho_new;
  }
#pragma omp parallel for
  for (int64_t i = 0; i < n; ++i) xout[i] = y[i] * dinv[i];
  *relres_out = relres;
  free(dinv); free(w);
  return (int)it;
}

/* ---- entry points --------------------------------------------------------------------------- */
/* stats[14] = {n_active, n_active_u, nnz, iterations, relres, t_tag, t_assemble, t_solve,
 *              threads, bad_facets, iterations_pc, relres_pc, t_solve_pc, pc_built}; the *_pc entries are the second
 * solve with the box preconditioner (precond != 0; stats 3, 4, 7 always describe the Jacobi solve unless
 * precond == 2, which skips it).  Optional outputs (may be NULL): cell_tags[nc], facet_tags[nf]
 * (int32), u_full[2*nv] (solution in full numbering, zeros elsewhere). */
int orc_poisson_sphere2(int n_cubes, int threads, double rtol, int64_t max_iter, int precond, double *stats,
                        int32_t *cell_tags, int32_t *facet_tags, double *u_full) {
  if (threads > 0) omp_set_num_threads(threads);
  Mesh m;
  mesh_build(&m, n_cubes);
  double *phi = (double *)malloc(sizeof(double) * m.nv), *f = (double *)malloc(sizeof(double) * m.nv),
         *ud = (double *)malloc(sizeof(double) * m.nv);
#pragma omp parallel for
  for (int64_t v = 0; v < m.nv; ++v) {
    const double *x = m.x + 3 * v;
    phi[v] = x[0] * x[0] + x[1] * x[1] + x[2] * x[2] - 1.0;
    ud[v] = sin(x[0]) * sin(x[1]) * sin(x[2]);
    f[v] = 3.0 * ud[v];
  }
  int8_t *ct = (int8_t *)malloc(m.nc), *ft = (int8_t *)malloc(m.nf);
  double t0 = omp_get_wtime();
  tag_cells(&m, phi, 1, ct);
  const int64_t bad = tag_facets(&m, phi, ct, ft);
  double t1 = omp_get_wtime();
  Csr A;
  memset(&A, 0, sizeof(A));
  assemble(&m, ct, ft, phi, f, ud, 1.0, 1.0, &A);
  double t2 = omp_get_wtime();
  double *xa = (double *)malloc(sizeof(double) * A.n), relres = 0.0;
  int it = 0;
  if (precond != 2) it = bicgstab(&A, rtol, max_iter, xa, &relres, NULL);
  double t3 = omp_get_wtime();
  stats[10] = stats[11] = stats[12] = stats[13] = 0.0;
  if (precond) {
    BoxPc P;
    memset(&P, 0, sizeof(P));
    const double t4 = omp_get_wtime();
    if (boxpc_setup(&P, &A, n_cubes) == 0) {
      double rr2 = 0.0;
      stats[10] = bicgstab(&A, rtol, max_iter, xa, &rr2, &P);   /* xa: the preconditioned solution wins */
      stats[11] = rr2;
      stats[13] = 1.0;
      boxpc_free(&P);
    }
    stats[12] = omp_get_wtime() - t4;
  }
  stats[0] = (double)A.n; stats[1] = (double)A.nu; stats[2] = (double)A.nnz; stats[3] = it;
  stats[4] = relres; stats[5] = t1 - t0; stats[6] = t2 - t1; stats[7] = t3 - t2;
  stats[8] = omp_get_max_threads(); stats[9] = (double)bad;
  if (cell_tags) for (int64_t c = 0; c < m.nc; ++c) cell_tags[c] = ct[c];
  if (facet_tags) for (int64_t fi = 0; fi < m.nf; ++fi) facet_tags[fi] = ft[fi];
  if (u_full) { memset(u_full, 0, sizeof(double) * 2 * m.nv); for (int64_t i = 0; i < A.n; ++i) u_full[A.full_of_active[i]] = xa[i]; }
  free(xa); csr_free(&A); free(ct); free(ft); free(phi); free(f); free(ud); mesh_free(&m);
  return 0;
}

int orc_poisson_sphere(int n_cubes, int threads, double rtol, int64_t max_iter, double *stats,
                       int32_t *cell_tags, int32_t *facet_tags, double *u_full) {
  double st[14];
  const int rc = orc_poisson_sphere2(n_cubes, threads, rtol, max_iter, 0, st, cell_tags, facet_tags, u_full);
  memcpy(stats, st, sizeof(double) * 10);
  return rc;
}
