// Structured P2 systems, solver side (included by phx_solve.hip): ordering, SELL over the stored rows, stencil runs,
// and the kernel that applies the interior rows.  See phx_assemble_p2s.inc.hip for what C0 / c0i mean.
//
// Solver order: the C0 rows in FINE-LATTICE order (x fastest), then the other u rows, then the p rows.  A c0i row has
// all 124 points of its 5 x 5 x 5 neighbourhood in C0, so along a run of consecutive c0i points of one x line the rows
// of the 25 neighbouring lines sit at CONSTANT position offsets: a run record is {first position, length, class bits,
// 25 line offsets}.  One wavefront walks one run, 64 rows per trip; lane parity alternates between the two classes
// (a = 0 / 1) of the line, whose coefficients are wave-uniform loads.

// position of every C0 row = exclusive scan of the lattice flags; other u rows behind them in active order
__global__ void k_p2s_perm(int64_t n, int64_t nu, phx_p2_lattice L, const int64_t *__restrict__ full,
                           const uint8_t *__restrict__ latc0, const int32_t *__restrict__ latpos,
                           const int32_t *__restrict__ rank_not, int32_t nc0all, int32_t *__restrict__ perm,
                           int32_t *__restrict__ iperm) {
  const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r >= n) return;
  int32_t pos = (int32_t)r;
  if (r < nu) {
    int64_t q[3];
    phx_p2_fine_of_entity(L, full[r], q);
    const int64_t p = q[0] + L.F[0] * (q[1] + L.F[1] * q[2]);
    pos = latc0[p] ? latpos[p] : nc0all + rank_not[r];
  }
  iperm[r] = pos;
  perm[pos] = (int32_t)r;
}

// notc0[r] = 1 for the u rows outside C0 (their rank orders them behind the lattice block)
__global__ void k_p2s_not_c0(int64_t nu, phx_p2_lattice L, const int64_t *__restrict__ full,
                             const uint8_t *__restrict__ latc0, uint8_t *__restrict__ notc0) {
  const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r >= nu) return;
  int64_t q[3];
  phx_p2_fine_of_entity(L, full[r], q);
  notc0[r] = latc0[q[0] + L.F[0] * (q[1] + L.F[1] * q[2])] ? 0 : 1;
}

// run starts along the fine x lines
__global__ void k_p2s_run_flags(int64_t nf, int64_t F0, const uint8_t *__restrict__ latc0i, uint8_t *__restrict__ fs) {
  const int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (p >= nf) return;
  fs[p] = latc0i[p] && (p % F0 == 0 || !latc0i[p - 1]);
}

__global__ void k_p2s_run_fill(int64_t nf, phx_p2_lattice L, const uint8_t *__restrict__ latc0i,
                               const uint8_t *__restrict__ fs, const int32_t *__restrict__ runid,
                               const int32_t *__restrict__ latpos, int32_t *__restrict__ runs) {
  const int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (p >= nf || !fs[p]) return;
  const int64_t F0 = L.F[0], F01 = L.F[0] * L.F[1];
  const int64_t I = p % F0, J = (p / F0) % L.F[1], K = p / F01;
  int len = 1;
  while (I + len < F0 && latc0i[p + len]) ++len;
  int32_t *rec = runs + (int64_t)PHX_P2S_REC * runid[p];
  rec[0] = latpos[p];
  rec[1] = len;
  rec[2] = (int)(J & 1) + 2 * (int)(K & 1) + 4 * (int)(I & 1);
  for (int dz = -2; dz <= 2; ++dz)
    for (int dy = -2; dy <= 2; ++dy) rec[3 + (dy + 2) + 5 * (dz + 2)] = latpos[p + dy * F0 + dz * F01] - latpos[p];
  for (int k = 28; k < PHX_P2S_REC; ++k) rec[k] = 0;
}

// one block of W threads per stored row: kept entries (non-zero, or the diagonal) sorted by SOLVER position go to
// consecutive k of the row's SELL lane
template <int W>
__global__ void __launch_bounds__(W)
k_sell_fill_slots_wide(int64_t ns, const int32_t *__restrict__ rows, phx_slot_view sv, int32_t nent,
                       const int32_t *__restrict__ du, const int32_t *__restrict__ dp, int64_t nu,
                       const double *__restrict__ diag, const int32_t *__restrict__ iperm,
                       const int64_t *__restrict__ slice_ptr, int32_t *__restrict__ scol, double *__restrict__ sval,
                       double *__restrict__ sraw) {
  __shared__ int32_t sc[W];
  __shared__ double sv_[W];
  __shared__ double sd[W];
  const int64_t i = blockIdx.x;
  if (i >= ns) return;
  const int t = threadIdx.x;
  const int32_t row = rows[i];
  int Wr;
  const int64_t base = sv_base(sv, row, &Wr);
  int32_t c = 0x7fffffff;
  double v = 0.0, dsc = 1.0;
  if (t < Wr) {
    const int32_t cc = sv.cols[base + t];
    if (cc != -1) {
      const int32_t col = sv_col(sv, cc, nent, du, dp);
      const double vv = sv.vals[base + t];
      if (vv != 0.0 || col == row) { c = iperm[col]; v = vv; dsc = col < nu ? 1.0 : 1.0 / diag[col]; }
    }
  }
  sc[t] = c; sv_[t] = v; sd[t] = dsc;
  __syncthreads();
  for (int k = 2; k <= W; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      const int p = t ^ j;
      if (p > t) {
        const bool up = ((t & k) == 0);
        const int32_t a = sc[t], b = sc[p];
        if ((a > b) == up) {
          sc[t] = b; sc[p] = a;
          double x = sv_[t]; sv_[t] = sv_[p]; sv_[p] = x;
          x = sd[t]; sd[t] = sd[p]; sd[p] = x;
        }
      }
      __syncthreads();
    }
  const int64_t sl = i / SELL_S, li = i % SELL_S;
  const int64_t sb = slice_ptr[sl];
  const int width = (int)((slice_ptr[sl + 1] - sb) / SELL_S);
  for (int k = t; k < width; k += W) {
    const int64_t o = sb + (int64_t)k * SELL_S + li;
    if (k < W && sc[k] != 0x7fffffff) { scol[o] = sc[k]; sraw[o] = sv_[k]; sval[o] = sv_[k] * sd[k]; }
    else { scol[o] = iperm[row]; sraw[o] = 0.0; sval[o] = 0.0; }
  }
}

// Sort key of a stored row: (tile of its fine-lattice point, x fastest) above (1023 - length).  Sorting by length alone
// (round 3) gives slices without padding whose 16 rows lie anywhere along the band: their ~50 columns each then share no
// x lines, and the gathers of the stored rows cost 530 of 911 us at 256^3 (the same stream without the gathers: 378 us).
// With the tile on top a slice's rows are neighbours (u and p rows of one tile^3 block of fine points, longest first
// inside the tile) and gather from one small neighbourhood.
__global__ void k_p2s_stored_keys(int64_t ns, const int32_t *__restrict__ list, const int32_t *__restrict__ len,
                                  uint32_t *__restrict__ keys, const int64_t *__restrict__ full, int64_t nent,
                                  phx_p2_lattice L, int tile, int t0, int t1, int lenq) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= ns) return;
  // ascending key = descending length, in steps of lenq entries (a slice is padded to a multiple of four anyway): rows
  // of one step keep their lattice order
  uint32_t key = (uint32_t)(1023 - min((len[i] + lenq - 1) / lenq * lenq, 1023));
  if (tile > 0) {
    int64_t e = full[list[i]];
    if (e >= nent) e -= nent;
    int64_t q[3];
    phx_p2_fine_of_entity(L, e, q);
    key |= (uint32_t)(((q[2] / tile) * t1 + q[1] / tile) * t0 + q[0] / tile) << 10;
  }
  keys[i] = key;
}

int phx_system_build_structured_p2(phx_system *s, const phx_slot_view &sv, int32_t nent, const uint8_t *latc0,
                                   const uint8_t *latc0i) {
  phx_mesh *m = s->mesh;
  hipStream_t st = m->stream;
  phx_p2_struct *ps = s->p2s;
  const phx_p2_lattice &L = ps->lat;
  const int64_t n = s->n, nu = s->nu, NF = L.F[0] * L.F[1] * L.F[2];
  const dim3 block(256), gn((unsigned)phx_div_up(n, 256)), gu((unsigned)phx_div_up(std::max<int64_t>(nu, 1), 256)),
      gfine((unsigned)phx_div_up(NF, 256));
  // ---- solver order
  int32_t *latpos = nullptr, *rank_not = nullptr, nc0all = 0, nnot = 0;
  uint8_t *notc0 = nullptr;
  PHX_HIP(phx_malloc(&latpos, sizeof(int32_t) * (size_t)NF));
  PHX_CHECK(scan_u8(st, latc0, latpos, NF, &nc0all));
  PHX_HIP(phx_malloc(&rank_not, sizeof(int32_t) * (size_t)std::max<int64_t>(nu, 1)));
  PHX_HIP(phx_malloc(&notc0, (size_t)std::max<int64_t>(nu, 1)));
  if (nu > 0) {
    k_p2s_not_c0<<<gu, block, 0, st>>>(nu, L, s->full_of_active, latc0, notc0);
    PHX_CHECK(scan_u8(st, notc0, rank_not, nu, &nnot));
  }
  PHX_REQUIRE((int64_t)nc0all + nnot == nu, PHX_ERR_HIP, "structured P2: %d C0 rows + %d others != %lld u rows",
              nc0all, nnot, (long long)nu);
  PHX_HIP(phx_malloc(&s->perm, sizeof(int32_t) * (size_t)n));
  PHX_HIP(phx_malloc(&s->iperm, sizeof(int32_t) * (size_t)n));
  k_p2s_perm<<<gn, block, 0, st>>>(n, nu, L, s->full_of_active, latc0, latpos, rank_not, nc0all, s->perm, s->iperm);
  PHX_HIP(hipGetLastError());
  // ---- stored rows (everything the stencils do not apply): kept entries, diagonal
  int32_t *list = nullptr;
  int64_t ns = 0;
  PHX_CHECK(phx_select_indices(st, n, SelStored{s->c0}, &list, &ns));   // synchronises
  PHX_HIP(phx_free(rank_not)); PHX_HIP(phx_free(notc0));
  int32_t *len = nullptr, *nstruct = nullptr;
  unsigned long long *dtot = nullptr, htot[2] = {0, 0};
  PHX_HIP(phx_malloc(&len, sizeof(int32_t) * (size_t)std::max<int64_t>(ns, 1)));
  PHX_HIP(phx_malloc(&nstruct, sizeof(int32_t) * (size_t)std::max<int64_t>(ns, 1)));
  PHX_HIP(phx_malloc(&dtot, sizeof(htot)));
  PHX_HIP(hipMemsetAsync(dtot, 0, sizeof(htot), st));
  if (ns > 0) {
    PHX_REQUIRE_GRID(ns * 64, "stored-row scan");
    k_slot_row_meta<<<dim3((unsigned)phx_div_up(ns * 64, 256)), block, 0, st>>>(
        ns, list, sv, nullptr, nent, s->dof_of_vertex_u, s->dof_of_vertex_p, m->gdim, len, nstruct, s->diag);
    k_sum2_i32<<<dim3((unsigned)std::min<int64_t>(phx_div_up(ns, 256), 512)), block, 0, st>>>(ns, nstruct, len, dtot);
  }
  PHX_HIP(hipMemcpyAsync(htot, dtot, sizeof(htot), hipMemcpyDeviceToHost, st));
  PHX_HIP(hipStreamSynchronize(st));
  PHX_HIP(phx_free(dtot)); PHX_HIP(phx_free(nstruct));
  s->n_sell_rows = ns;
  s->nc0 = n - ns;
  ps->nc0i = n - ns;
  // entries of the stencil rows: counted from the tables (structural = applied: the tables hold no explicit zeros)
  {
    int cnt[8];
    std::vector<double> K(1000);
    PHX_HIP(hipMemcpy(K.data(), ps->coef, sizeof(double) * 1000, hipMemcpyDeviceToHost));
    double avg = 0.0;
    for (int c = 0; c < 8; ++c) { cnt[c] = 0; for (int o = 0; o < 125; ++o) cnt[c] += K[(size_t)c * 125 + o] != 0.0; avg += cnt[c] / 8.0; }
    // the eight classes are equally frequent among interior fine points
    s->nnz = (int64_t)htot[0] + (int64_t)(avg * (double)s->nc0);
    s->sell_true_nnz = (int64_t)htot[1] + (int64_t)(avg * (double)s->nc0);
  }
  PHX_HIP(phx_malloc(&s->cscale, sizeof(double) * (size_t)n));
  k_cscale<<<gn, block, 0, st>>>(n, nu, s->perm, s->diag, s->cscale);
  // ---- SELL over the stored rows, longest first
  s->nslices = phx_div_up(ns, SELL_S);
  PHX_HIP(phx_malloc(&s->slice_ptr, sizeof(int64_t) * (size_t)(s->nslices + 1)));
  PHX_HIP(phx_malloc(&s->sell_rows, sizeof(int32_t) * (size_t)std::max<int64_t>(s->nslices * SELL_S, 1)));
  int32_t *rows_active = nullptr;
  PHX_HIP(phx_malloc(&rows_active, sizeof(int32_t) * (size_t)std::max<int64_t>(s->nslices * SELL_S, 1)));
  if (ns > 0) {
    uint32_t *keys = nullptr, *keys2 = nullptr;
    PHX_HIP(phx_malloc(&keys, sizeof(uint32_t) * (size_t)ns));
    PHX_HIP(phx_malloc(&keys2, sizeof(uint32_t) * (size_t)ns));
    const dim3 gs((unsigned)phx_div_up(ns, 256));
    static const int tile_env = getenv("PHX_P2S_TILE") ? atoi(getenv("PHX_P2S_TILE")) : 16;   // 0: by length alone
    int tile = tile_env, tn[3] = {1, 1, 1}, key_bits = 10;
    if (tile > 0) {
      for (int a = 0; a < 3; ++a) tn[a] = (int)(L.F[a] / tile + 1);
      while (tile > 0 && (int64_t)tn[0] * tn[1] * tn[2] > (1ll << 22)) {   // 22 + 10 key bits
        tile *= 2;
        for (int a = 0; a < 3; ++a) tn[a] = (int)(L.F[a] / tile + 1);
      }
      while ((1ll << (key_bits - 10)) < (int64_t)tn[0] * tn[1] * tn[2]) ++key_bits;
    }
    static const int lenq = getenv("PHX_P2S_LENQ") ? std::max(1, atoi(getenv("PHX_P2S_LENQ"))) : 4;
    k_p2s_stored_keys<<<gs, block, 0, st>>>(ns, list, len, keys, s->full_of_active, (int64_t)nent, L, tile, tn[0], tn[1], lenq);
    k_fill_i32<<<dim3((unsigned)phx_div_up(s->nslices * SELL_S, 256)), block, 0, st>>>(s->nslices * SELL_S, rows_active, -1, 0);
    size_t bytes = 0;
    PHX_HIP(phx_sort_pairs(nullptr, bytes, keys, keys2, list, rows_active, (size_t)ns, 0, key_bits, st));
    void *tmp = nullptr;
    PHX_HIP(phx_malloc(&tmp, bytes ? bytes : 16));
    PHX_HIP(phx_sort_pairs(tmp, bytes, keys, keys2, list, rows_active, (size_t)ns, 0, key_bits, st));
    PHX_HIP(hipMemcpyAsync(s->sell_rows, rows_active, sizeof(int32_t) * (size_t)(s->nslices * SELL_S), hipMemcpyDeviceToDevice, st));
    k_map_i32<<<dim3((unsigned)phx_div_up(s->nslices * SELL_S, 256)), block, 0, st>>>(s->nslices * SELL_S, s->iperm, s->sell_rows);
    int64_t *widths = nullptr;
    PHX_HIP(phx_malloc(&widths, sizeof(int64_t) * (size_t)(s->nslices + 1)));
    k_slice_widths16<<<dim3((unsigned)phx_div_up(s->nslices + 1, 256)), block, 0, st>>>(s->nslices, ns, keys2, widths);
    size_t b2 = 0;
    PHX_HIP(phx_exclusive_sum(nullptr, b2, widths, s->slice_ptr, (size_t)(s->nslices + 1), st));
    void *t2 = nullptr;
    PHX_HIP(phx_malloc(&t2, b2 ? b2 : 16));
    PHX_HIP(phx_exclusive_sum(t2, b2, widths, s->slice_ptr, (size_t)(s->nslices + 1), st));
    PHX_HIP(hipMemcpyAsync(&s->sell_nnz, s->slice_ptr + s->nslices, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    PHX_HIP(hipStreamSynchronize(st));
    PHX_HIP(phx_free(tmp)); PHX_HIP(phx_free(t2)); PHX_HIP(phx_free(widths)); PHX_HIP(phx_free(keys)); PHX_HIP(phx_free(keys2));
  } else {
    PHX_HIP(hipMemsetAsync(s->slice_ptr, 0, sizeof(int64_t), st));
    s->sell_nnz = 0;
  }
  const size_t ne = (size_t)std::max<int64_t>(s->sell_nnz, 1);
  PHX_HIP(phx_malloc(&s->sell_col, sizeof(int32_t) * ne));
  PHX_HIP(phx_malloc(&s->sell_val, sizeof(double) * ne));
  PHX_HIP(phx_malloc(&s->sell_val_raw, sizeof(double) * ne));
  PHX_HIP(phx_malloc(&s->sell_kind, (size_t)std::max<int64_t>(s->nslices, 1) * 2));
  s->sell_kind_raw = s->sell_kind + std::max<int64_t>(s->nslices, 1);
  PHX_HIP(hipMemsetAsync(s->sell_kind, 0, (size_t)std::max<int64_t>(s->nslices, 1) * 2, st));
  s->sell_stream_bytes = 0;
  if (ns > 0) {
    const dim3 gf((unsigned)ns);
    if (sv.W <= 128)
      k_sell_fill_slots_wide<128><<<gf, dim3(128), 0, st>>>(ns, rows_active, sv, nent, s->dof_of_vertex_u, s->dof_of_vertex_p, nu, s->diag, s->iperm, s->slice_ptr, s->sell_col, s->sell_val, s->sell_val_raw);
    else if (sv.W == 256)
      k_sell_fill_slots_wide<256><<<gf, dim3(256), 0, st>>>(ns, rows_active, sv, nent, s->dof_of_vertex_u, s->dof_of_vertex_p, nu, s->diag, s->iperm, s->slice_ptr, s->sell_col, s->sell_val, s->sell_val_raw);
    else if (sv.W == 512)
      k_sell_fill_slots_wide<512><<<gf, dim3(512), 0, st>>>(ns, rows_active, sv, nent, s->dof_of_vertex_u, s->dof_of_vertex_p, nu, s->diag, s->iperm, s->slice_ptr, s->sell_col, s->sell_val, s->sell_val_raw);
    else { phx_set_error("structured P2: unsupported slot capacity %d", sv.W); return PHX_ERR_VALUE; }
    if (ns < s->nslices * SELL_S)
      k_sell_pad_tail<<<1, 64, 0, st>>>(ns, s->nslices, s->slice_ptr, s->sell_col, s->sell_val, s->sell_val_raw);
    PHX_HIP(hipGetLastError());
    s->sell_stream_bytes = 12 * s->sell_nnz + 8 * s->nslices + 4 * s->nslices * SELL_S;
  }
  PHX_HIP(hipStreamSynchronize(st));
  PHX_HIP(phx_free(len)); PHX_HIP(phx_free(list)); PHX_HIP(phx_free(rows_active));
  // ---- stencil runs
  s->nseg = 0;
  s->nstencil_pos = nc0all;
  ps->nrun = 0;
  if (s->nc0 > 0) {
    uint8_t *fs = nullptr;
    int32_t *runid = nullptr, nrun = 0;
    PHX_HIP(phx_malloc(&fs, (size_t)NF));
    PHX_HIP(phx_malloc(&runid, sizeof(int32_t) * (size_t)NF));
    k_p2s_run_flags<<<gfine, block, 0, st>>>(NF, L.F[0], latc0i, fs);
    PHX_CHECK(scan_u8(st, fs, runid, NF, &nrun));
    ps->nrun = nrun;
    PHX_HIP(phx_malloc(&ps->runs, sizeof(int32_t) * PHX_P2S_REC * (size_t)std::max(nrun, 1)));
    k_p2s_run_fill<<<gfine, block, 0, st>>>(NF, L, latc0i, fs, runid, latpos, ps->runs);
    PHX_HIP(hipGetLastError());
    PHX_HIP(hipStreamSynchronize(st));
    PHX_HIP(phx_free(fs)); PHX_HIP(phx_free(runid));
    s->sell_stream_bytes += 4 * PHX_P2S_REC * (int64_t)nrun;
  }
  PHX_HIP(phx_free(latpos));
  // solver workspace: 9 vectors + scalars
  PHX_HIP(phx_malloc(&s->work, sizeof(double) * (size_t)n * 9));
  PHX_HIP(phx_malloc(&s->scal, sizeof(double) * PHX_SCAL_DOUBLES));
  PHX_CHECK(phx_mesh_pinned_scalars(s->mesh, &s->scal_h));
  return PHX_OK;
}

// y_r = sum_o coef[class(r)][o] x[r + off(o)] over the runs; DOTS as k_spmv_sell.
// One wavefront per run, 124 rows per trip: a lane owns TWO consecutive rows, one of each parity class of the line, so
// every coefficient is a wave-uniform SCALAR operand and each class is computed once (round 3 had every lane accumulate
// both classes of its row and keep one: twice the multiply-adds; a per-lane coefficient select had turned the
// coefficient into a dependent vector load per term before that).  The lane's two rows need the SIX entries
// 2 lane .. 2 lane + 5 of each neighbouring line.  Round 4, first version: the lines staged in LDS (coalesced 8-byte
// loads, three 16-byte LDS reads per line) -- 32 LDS bytes per row and line, and the kernel ran at 60 % of the LDS peak
// with its fetch traffic irrelevant.  Now NO LDS: the lane loads entries 2 lane, 2 lane + 1 as one
// 16-byte access (the wavefront reads 1 KB contiguously) and takes entries + 2 .. + 5 from lanes + 1 and + 2 with DPP
// wave shifts.  A trip covers 124 rows: lanes 62 and 63 only supply entries 124 .. 127 to their neighbours.
__device__ __forceinline__ double2 p2s_shl1(double2 v) {   // lane l <- lane l + 1 (lane 63 <- 0)
  int w[4] = {__double2loint(v.x), __double2hiint(v.x), __double2loint(v.y), __double2hiint(v.y)};
#pragma unroll
  for (int q = 0; q < 4; ++q) w[q] = __builtin_amdgcn_update_dpp(0, w[q], 0x130, 0xf, 0xf, true);   // wave_shl:1
  return make_double2(__hiloint2double(w[1], w[0]), __hiloint2double(w[3], w[2]));
}

template <int DOTS>
__global__ void __launch_bounds__(256)
k_spmv_p2s(int64_t nrun, const int32_t *__restrict__ runs, const double *__restrict__ tabE,
           const double *__restrict__ tabO, const unsigned *__restrict__ linemask, const double *__restrict__ x,
           double *__restrict__ y, const double *__restrict__ d0, double *__restrict__ out0, double *__restrict__ out1,
           DotPart part) {
  const int lane = threadIdx.x & 63;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  double p0 = 0.0, p1 = 0.0;
  const unsigned lm0 = linemask[0], lm1 = linemask[1], lm2 = linemask[2], lm3 = linemask[3];
  for (int64_t w = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6)); w < nrun; w += nwaves) {
    const int32_t *rec = runs + (int64_t)PHX_P2S_REC * w;
    const int first = rec[0], len = rec[1], bits = rec[2];
    const int bc = bits & 3, a0 = bits >> 2;
    const unsigned lm = bc == 0 ? lm0 : bc == 1 ? lm1 : bc == 2 ? lm2 : lm3;   // (no load behind the record's)
    // class of the lane's first / second row: the run starts with class a0 and a trip starts at an even offset
    const double *cA = (a0 ? tabO : tabE) + (size_t)bc * 125, *cB = (a0 ? tabE : tabO) + (size_t)bc * 125;
    for (int base = 0; base < len; base += 124) {
      const int i = base + 2 * lane;
      const bool onA = lane < 62 && i < len, onB = lane < 62 && i + 1 < len;
      const int64_t r0 = (int64_t)first + base;
      // entries i - 2 .. i + 2 of every neighbouring line exist for the rows of the run (their whole 5 x 5 x 5
      // neighbourhood is C0): entry e of a line = position base - 2 + e of the run's coordinates, e < len + 4 - base
      const bool ok0 = base - 2 + 2 * lane < len + 2, ok1 = base - 1 + 2 * lane < len + 2;
      double aA = 0.0, aB = 0.0;
      double2 qc[5], qn[5];
      auto load_plane = [&](int dz, double2 *q) {
        const unsigned pm = (lm >> (5 * dz)) & 31u;
#pragma unroll
        for (int dy = 0; dy < 5; ++dy) {
          q[dy] = make_double2(0.0, 0.0);
          if ((pm >> dy) & 1u) {
            const double *xl = x + (r0 + rec[3 + dy + 5 * dz] - 2);
            if (ok1) __builtin_memcpy(&q[dy], xl + 2 * lane, sizeof(double2));   // 8-byte aligned: one 16-byte access
            else if (ok0) q[dy].x = xl[2 * lane];
          }
        }
      };
      // planes with a coefficient, in order; the loads of the next one are in flight while this one is multiplied
      int dz = 0;
      while (dz < 5 && ((lm >> (5 * dz)) & 31u) == 0u) ++dz;
      if (dz < 5) load_plane(dz, qc);
#pragma unroll 1
      while (dz < 5) {
        const unsigned pm = (lm >> (5 * dz)) & 31u;
        int nz = dz + 1;
        while (nz < 5 && ((lm >> (5 * nz)) & 31u) == 0u) ++nz;
        if (nz < 5) load_plane(nz, qn);
#pragma unroll
        for (int dy = 0; dy < 5; ++dy) {
          if ((pm >> dy) & 1u) {
            const double *ca = cA + (5 * dz + dy) * 5, *cb = cB + (5 * dz + dy) * 5;
            const double2 q0 = qc[dy], q1 = p2s_shl1(q0), q2 = p2s_shl1(q1);
            aA = __builtin_fma(ca[0], q0.x, aA); aA = __builtin_fma(ca[1], q0.y, aA); aA = __builtin_fma(ca[2], q1.x, aA);
            aA = __builtin_fma(ca[3], q1.y, aA); aA = __builtin_fma(ca[4], q2.x, aA);
            aB = __builtin_fma(cb[0], q0.y, aB); aB = __builtin_fma(cb[1], q1.x, aB); aB = __builtin_fma(cb[2], q1.y, aB);
            aB = __builtin_fma(cb[3], q2.x, aB); aB = __builtin_fma(cb[4], q2.y, aB);
          }
        }
#pragma unroll
        for (int dy = 0; dy < 5; ++dy) qc[dy] = qn[dy];
        dz = nz;
      }
      const int64_t r = r0 + 2 * lane;
      if (onA) {
        y[r] = aA;
        if (DOTS > 0) { p0 = __builtin_fma(aA, d0[r], p0); if (DOTS > 1) p1 = __builtin_fma(aA, aA, p1); }
      }
      if (onB) {
        y[r + 1] = aB;
        if (DOTS > 0) { p0 = __builtin_fma(aB, d0[r + 1], p0); if (DOTS > 1) p1 = __builtin_fma(aB, aB, p1); }
      }
    }
  }
  if (DOTS > 0) {
    __shared__ double red[2][4];
    p0 = wave_sum(p0);
    if (DOTS > 1) p1 = wave_sum(p1);
    const int wv = threadIdx.x >> 6;
    if (lane == 0) { red[0][wv] = p0; red[1][wv] = p1; }
    __syncthreads();
    if (threadIdx.x == 0) {
      const double s0 = red[0][0] + red[0][1] + red[0][2] + red[0][3], s1 = red[1][0] + red[1][1] + red[1][2] + red[1][3];
      if (part.p0) {   // PHX_OPT_DETERMINISTIC
        part.p0[blockIdx.x] = s0;
        if (DOTS > 1) part.p1[blockIdx.x] = s1;
      } else {
        const int slot = (blockIdx.x & (NSLOT - 1)) * SLOT_STRIDE;
        unsafeAtomicAdd(out0 + slot, s0);
        if (DOTS > 1) unsafeAtomicAdd(out1 + slot, s1);
      }
    }
  }
}
