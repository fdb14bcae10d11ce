// Vertex-block Jacobi for the 5-field interface-elasticity system (included by phx_solve.hip).
//
// The system couples up to 27 unknowns per vertex (u_in, u_out, y_in, y_out, p; 14 in 2-D) through the penalty terms
// on the cut cells (demo/interface-elasticity/main.py:188-203): blocks of O(1) entries that scalar Jacobi ignores.
// P = D B^-1 with B = the block diagonal of A over the active DoFs of each vertex (D: the diagonal the SELL copy is
// scaled with, so that A D^-1 P = A B^-1).  CPU prototype against the oracle matrices, E_out = 1e-3
// (tools/experiments/elasticity_lattice.py): 224 -> 90, 243 -> 98, 382 -> 129 iterations for n = 16, 24, 32; lattice
// Laplacians on the bulk displacement rows on top of it do NOT help (97 / 131 / 179): the band decides.
// Blocks: k x k per vertex (k = active DoFs of the vertex: 3 away from the interface), inverted by Gauss-Jordan with
// partial pivoting, one wavefront per vertex, the augmented block in LDS.
struct phx_blockjac {
  int64_t nvert = 0;            // mesh vertices
  int32_t *vptr = nullptr;      // [nvert + 1] first entry of a vertex in vpos / erow
  int32_t *vpos = nullptr;      // [n] solver position of entry e
  int32_t *evert = nullptr;     // [n] vertex of entry e
  int64_t *bptr = nullptr;      // [nvert + 1] first value of a vertex's k x k block in M
  double *M = nullptr;          // D B^-1, column-major per vertex
  int64_t nvals = 0;
};

static void blockjac_free(phx_blockjac *b) {
  if (!b) return;
  (void)phx_free(b->vptr); (void)phx_free(b->vpos); (void)phx_free(b->evert); (void)phx_free(b->bptr); (void)phx_free(b->M);
  delete b;
}

// active DoFs per vertex (dofmap: block-major full index -> active row or -1)
__global__ void k_bj_count(int64_t nvert, int nblk, const int32_t *__restrict__ dofmap, int32_t *__restrict__ cnt,
                           int64_t *__restrict__ cnt2) {
  const int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (v >= nvert) return;
  int k = 0;
  for (int b = 0; b < nblk; ++b) k += dofmap[(int64_t)b * nvert + v] >= 0;
  cnt[v] = k;
  cnt2[v] = (int64_t)k * k;
}

#define BJ_MAXK 27
// one wavefront per vertex: gather the k x k diagonal block from the CSR rows, invert, scale the rows by D
__global__ void __launch_bounds__(256)
k_bj_build(int64_t nvert, int nblk, const int32_t *__restrict__ dofmap, const int64_t *__restrict__ full_of_active,
           const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, const double *__restrict__ val,
           const double *__restrict__ diag, const int32_t *__restrict__ iperm, const int32_t *__restrict__ vptr,
           const int64_t *__restrict__ bptr, int32_t *__restrict__ vpos, int32_t *__restrict__ evert,
           double *__restrict__ M, int *__restrict__ singular) {
  __shared__ double aug_all[4][BJ_MAXK][2 * BJ_MAXK + 1];
  __shared__ int lidx_all[4][BJ_MAXK + 5], prow_all[4][BJ_MAXK + 5], rows_all[4][BJ_MAXK + 5];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t v = blockIdx.x * (int64_t)(blockDim.x >> 6) + w;
  if (v >= nvert) return;
  const int k = vptr[v + 1] - vptr[v];
  if (k == 0) return;
  double (*A)[2 * BJ_MAXK + 1] = aug_all[w];
  int *lidx = lidx_all[w], *prow = prow_all[w], *rows = rows_all[w];
  if (lane == 0) {
    int j = 0;
    for (int b = 0; b < nblk; ++b) {
      const int32_t r = dofmap[(int64_t)b * nvert + v];
      lidx[b] = r >= 0 ? j : -1;
      if (r >= 0) { rows[j] = r; prow[j] = j; ++j; }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (int e = lane; e < k * 2 * k; e += 64) { const int i = e / (2 * k), j = e % (2 * k); A[i][j] = j - k == i ? 1.0 : 0.0; }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // the 64 lanes walk each row of the vertex together (a lane per row chained 45-250 dependent look-ups: 288 ms at 256^3)
  for (int j = 0; j < k; ++j) {
    const int32_t r = rows[j];
    for (int64_t q = rowptr[r] + lane; q < rowptr[r + 1]; q += 64) {
      const int64_t f = full_of_active[col[q]];
      const int64_t blk = f / nvert;
      if (f - blk * nvert == v) A[j][lidx[blk]] = val[q];
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (int p = 0; p < k; ++p) {
    // partial pivoting among the logical rows p .. k-1
    double best = (lane >= p && lane < k) ? fabs(A[prow[lane]][p]) : -1.0;
    int bi = lane;
    for (int o = 32; o > 0; o >>= 1) {
      const double ob = __shfl_xor(best, o);
      const int oi = __shfl_xor(bi, o);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (!(best > 0.0)) { if (lane == 0) atomicOr(singular, 1); return; }
    if (lane == 0) { const int t = prow[p]; prow[p] = prow[bi]; prow[bi] = t; }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int R = prow[p];
    const double inv = 1.0 / A[R][p];
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < 2 * k) A[R][lane] *= inv;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < k && lane != R) {
      const double fct = A[lane][p];
      if (fct != 0.0)
        for (int j = 0; j < 2 * k; ++j) A[lane][j] -= fct * A[R][j];
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  // logical row i of the inverse sits in physical row prow[i]
  const int32_t e0 = vptr[v];
  double *Mv = M + bptr[v];
  if (lane < k) {
    const int32_t r = rows[lane];
    vpos[e0 + lane] = iperm[r];
    evert[e0 + lane] = (int32_t)v;
    const double d = diag[r];
    for (int j = 0; j < k; ++j) Mv[j * k + lane] = d * A[prow[lane]][k + j];   // column-major: the rows of a vertex read consecutive addresses
  }
}

// out = P in over the solver vector: one thread per entry (row of a vertex block).  The entries of a vertex are
// consecutive, so a block of 256 entries stages ITS inputs in LDS once (one coalesced-by-vpos gather per thread) and the
// rows read their k inputs from there; only the entries of a vertex that straddles the block boundary go to global
// memory.  (One dependent gather per term: 238 us per application at 96^3 against 353 us for the SpMV.)
__global__ void __launch_bounds__(256)
k_bj_apply(int64_t n, const int32_t *__restrict__ vptr, const int32_t *__restrict__ vpos, const int32_t *__restrict__ evert,
           const int64_t *__restrict__ bptr, const double *__restrict__ M, const uint8_t *__restrict__ own,
           const double *__restrict__ vin, double *__restrict__ vout) {
  __shared__ double xs[256];
  const int64_t b0 = blockIdx.x * (int64_t)blockDim.x, e = b0 + threadIdx.x;
  int32_t pos = 0;
  if (e < n) { pos = vpos[e]; xs[threadIdx.x] = vin[pos]; }
  __syncthreads();
  if (e >= n) return;
  const int32_t v = evert[e], e0 = vptr[v];
  const int k = vptr[v + 1] - e0, i = (int)(e - e0);
  const double *Mc = M + bptr[v] + i;
  double acc = 0.0;
  for (int j = 0; j < k; ++j) {
    const int64_t ej = (int64_t)e0 + j;
    const double xj = (ej >= b0 && ej < b0 + 256) ? xs[ej - b0] : vin[vpos[ej]];
    acc = __builtin_fma(Mc[(int64_t)j * k], xj, acc);
  }
  vout[pos] = (own && !own[pos]) ? 0.0 : acc;
}

static int blockjac_build(phx_system *s, int nblk, phx_blockjac **out) {
  phx_mesh *m = s->mesh;
  hipStream_t st = m->stream;
  const int64_t nvert = m->nv, n = s->n;
  PHX_REQUIRE(s->rowptr && s->nent == (int64_t)nblk * nvert && nblk <= BJ_MAXK, PHX_ERR_VALUE,
              "vertex-block Jacobi needs the CSR copy of a block-major system with at most %d blocks", BJ_MAXK);
  phx_blockjac *b = new phx_blockjac();
  b->nvert = nvert;
  int32_t *cnt = nullptr;
  int64_t *cnt2 = nullptr;
  int *sing = nullptr, hsing = 0;
  const dim3 block(256), gv((unsigned)phx_div_up(nvert, 256));
  auto fail = [&](int code) { blockjac_free(b); (void)phx_free(cnt); (void)phx_free(cnt2); (void)phx_free(sing); return code; };
  if (phx_malloc(&cnt, sizeof(int32_t) * (size_t)(nvert + 1)) != hipSuccess || phx_malloc(&cnt2, sizeof(int64_t) * (size_t)(nvert + 1)) != hipSuccess ||
      phx_malloc(&b->vptr, sizeof(int32_t) * (size_t)(nvert + 1)) != hipSuccess || phx_malloc(&b->bptr, sizeof(int64_t) * (size_t)(nvert + 1)) != hipSuccess ||
      phx_malloc(&b->vpos, sizeof(int32_t) * (size_t)std::max<int64_t>(n, 1)) != hipSuccess ||
      phx_malloc(&b->evert, sizeof(int32_t) * (size_t)std::max<int64_t>(n, 1)) != hipSuccess || phx_malloc(&sing, sizeof(int)) != hipSuccess)
    return fail(PHX_ERR_HIP);
  PHX_HIP(hipMemsetAsync(cnt + nvert, 0, sizeof(int32_t), st));
  PHX_HIP(hipMemsetAsync(cnt2 + nvert, 0, sizeof(int64_t), st));
  PHX_HIP(hipMemsetAsync(sing, 0, sizeof(int), st));
  k_bj_count<<<gv, block, 0, st>>>(nvert, nblk, s->dof_of_vertex_u, cnt, cnt2);
  {
    size_t b1 = 0, b2 = 0;
    PHX_HIP(phx_exclusive_sum(nullptr, b1, cnt, b->vptr, (size_t)(nvert + 1), st));
    PHX_HIP(phx_exclusive_sum(nullptr, b2, cnt2, b->bptr, (size_t)(nvert + 1), st));
    void *tmp = nullptr;
    PHX_HIP(phx_malloc(&tmp, std::max(b1, b2) ? std::max(b1, b2) : 16));
    PHX_HIP(phx_exclusive_sum(tmp, b1, cnt, b->vptr, (size_t)(nvert + 1), st));
    PHX_HIP(phx_exclusive_sum(tmp, b2, cnt2, b->bptr, (size_t)(nvert + 1), st));
    PHX_HIP(hipMemcpyAsync(&b->nvals, b->bptr + nvert, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    PHX_HIP(hipStreamSynchronize(st));
    PHX_HIP(phx_free(tmp));
  }
  if (phx_malloc(&b->M, sizeof(double) * (size_t)std::max<int64_t>(b->nvals, 1)) != hipSuccess) return fail(PHX_ERR_HIP);
  k_bj_build<<<dim3((unsigned)phx_div_up(nvert, 4)), block, 0, st>>>(nvert, nblk, s->dof_of_vertex_u, s->full_of_active, s->rowptr, s->col,
                                                                     s->val, s->diag, s->iperm, b->vptr, b->bptr, b->vpos, b->evert, b->M, sing);
  PHX_HIP(hipGetLastError());
  PHX_HIP(hipMemcpyAsync(&hsing, sing, sizeof(int), hipMemcpyDeviceToHost, st));
  PHX_HIP(hipStreamSynchronize(st));
  PHX_HIP(phx_free(cnt)); PHX_HIP(phx_free(cnt2)); PHX_HIP(phx_free(sing));
  cnt = nullptr; cnt2 = nullptr; sing = nullptr;
  if (hsing) { blockjac_free(b); *out = nullptr; return PHX_OK; }   // a singular vertex block: stay with scalar Jacobi
  *out = b;
  return PHX_OK;
}

static int blockjac_apply(phx_system *s, const phx_blockjac *b, const double *vin, double *vout) {
  if (s->n == 0) return PHX_OK;
  k_bj_apply<<<dim3((unsigned)phx_div_up(s->n, 256)), dim3(256), 0, s->mesh->stream>>>(s->n, b->vptr, b->vpos, b->evert, b->bptr, b->M,
                                                                                      s->own, vin, vout);
  PHX_HIP(hipGetLastError());
  return PHX_OK;
}
