// In-place inverse of a dense n x n f64 matrix (row-major) on the device: blocked Gauss-Jordan elimination with partial
// pivoting -- included by phx_solve.hip in front of phx_coarse.inc.hip, whose Galerkin coarse matrix (<= ~10^4 rows) it
// inverts once per system.  Rounds 2-3 called rocSOLVER (getrf + getri through dlopen) for this; round 4 owns it (VERDICT
// r3 item 9): no vendor library on the product path, and nothing to degrade silently when one is missing.
//
// Block step over the columns K = [k0, k0 + B), B = 32:
//   1. pivots: the B pivot rows of the panel A[k0:, K] are chosen by tournament pivoting (blocks of 256 rows, one row per
//      thread in registers, eliminate among themselves; winners meet in the next round); the row swaps are then applied to
//      whole rows of A (column-parallel);
//   2. P = A[K, K] (after the swaps an unpivoted elimination of P is stable: its LU has multipliers <= 1) is inverted in LDS;
//   3. W = A[:, K] (saved), R = P^-1 A[K, :] with R[:, K] = P^-1;
//   4. every row i outside K:  A[i, j] = (j in K ? 0 : A[i, j]) - W[i, :] R[:, j];  rows K: A[K, :] = R
//      -- one pass over the matrix per block step: a rank-32 update, 4 flop per byte: bound by the 2 x 8 n^2 bytes it moves.
// At the end the columns are permuted by the inverse of the accumulated row permutation (composed on the host from the
// n pivot indices).  2 n^3 flop, n / 32 passes over the matrix.
// A zero pivot column (singular to working precision) is reported through *singular; the matrix is then garbage.

#define PHX_DINV_B 32

// ---- 1. pivot rows of the panel by TOURNAMENT pivoting (Grigori, Demmel, Xiang: CALU).  A sequential panel
// factorisation by one workgroup cost ~1 ms per panel of 10^4 rows (450 block barriers): 0.3 s of the 0.5 s of a 6000-row
// inverse.  Here every block of 256 candidate rows -- ONE ROW PER THREAD, its 32 panel entries in registers -- runs the
// elimination with partial pivoting among its own rows and names its 32 pivot rows; the winners of eight blocks meet in
// the next round, on their ORIGINAL rows, until one block is left (10^4 rows: 40 -> 5 -> 1 blocks, three launches).
// cand == nullptr: the candidates are the rows k0 + slot (first round).  winners[block * 32 + j]: row chosen j-th, -1: none.
__global__ void __launch_bounds__(256)
k_dinv_tournament(int n, int k0, int nb, const double *__restrict__ A, const int *__restrict__ cand, int ncand,
                  int *__restrict__ winners) {
  constexpr int B = PHX_DINV_B;
  __shared__ double prow[B];
  __shared__ double wmax[4];
  __shared__ int wtid[4];
  const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int slot = (int)blockIdx.x * 256 + tid;
  const int row = cand ? (slot < ncand ? cand[slot] : -1) : (slot < ncand ? k0 + slot : -1);
  double a[B];
#pragma unroll
  for (int c = 0; c < B; ++c) a[c] = (row >= 0 && c < nb) ? A[(int64_t)row * n + k0 + c] : 0.0;
  bool taken = row < 0;
  for (int j = 0; j < B; ++j) {
    if (j >= nb) { if (tid == 0) winners[(int64_t)blockIdx.x * B + j] = -1; continue; }
    double v = -1.0;
#pragma unroll
    for (int c = 0; c < B; ++c) if (c == j && !taken) v = fabs(a[c]);
    int bt = tid;
    for (int o = 32; o > 0; o >>= 1) {
      const double v2 = __shfl_xor(v, o);
      const int t2 = __shfl_xor(bt, o);
      if (v2 > v || (v2 == v && t2 < bt)) { v = v2; bt = t2; }
    }
    if (lane == 0) { wmax[wv] = v; wtid[wv] = bt; }
    __syncthreads();
    double bv = wmax[0];
    bt = wtid[0];
#pragma unroll
    for (int q = 1; q < 4; ++q) if (wmax[q] > bv || (wmax[q] == bv && wtid[q] < bt)) { bv = wmax[q]; bt = wtid[q]; }
    const bool none = !(bv > 0.0);   // no candidate with a non-zero entry left (block-uniform)
    if (tid == (none ? 0 : bt)) {
      winners[(int64_t)blockIdx.x * B + j] = none ? -1 : row;
      if (!none) {
        taken = true;
#pragma unroll
        for (int c = 0; c < B; ++c) prow[c] = a[c];
      }
    }
    __syncthreads();
    if (!none && !taken) {
      double aj = 0.0, pj = 1.0;
#pragma unroll
      for (int c = 0; c < B; ++c) if (c == j) { aj = a[c]; pj = prow[c]; }
      const double l = aj / pj;
#pragma unroll
      for (int c = 0; c < B; ++c) if (c > j) a[c] -= l * prow[c];
    }
    __syncthreads();
  }
}

// the winners of the last round, in pivot order, as the LAPACK-style sequence of row swaps piv[k0 + j].  ONE WAVEFRONT:
// lane t keeps one tracked (position, original row that sits there now) pair in registers and a lookup is a ballot (a
// single thread walking the 2 B pairs in private memory took 275 us per panel: 26 ms of a 96^3 elasticity step).
__global__ void __launch_bounds__(64)
k_dinv_swaps_from_winners(int k0, int nb, const int *__restrict__ winners, int *__restrict__ piv, int *__restrict__ singular) {
  const int lane = (int)threadIdx.x;
  int tp = -1, tr = -1, nt = 0;   // nt <= 2 B = 64 pairs, wave-uniform
  for (int j = 0; j < nb; ++j) {
    const int w = winners[j];
    if (w < 0) { if (lane == 0) { *singular = 1; piv[k0 + j] = k0 + j; } continue; }
    unsigned long long m = __ballot(lane < nt && tr == w);                 // where row w sits now
    const int p = m ? __shfl(tp, __ffsll((long long)m) - 1) : w;
    m = __ballot(lane < nt && tp == k0 + j);                               // the row at the pivot position
    const int ra = m ? __shfl(tr, __ffsll((long long)m) - 1) : k0 + j;
    if (m) { if (lane == __ffsll((long long)m) - 1) tr = w; }
    else { if (lane == nt) { tp = k0 + j; tr = w; } ++nt; }
    m = __ballot(lane < nt && tp == p);
    if (m) { if (lane == __ffsll((long long)m) - 1) tr = ra; }
    else { if (lane == nt) { tp = p; tr = ra; } ++nt; }
    if (lane == 0) piv[k0 + j] = p;
  }
}

// ---- the B row swaps of the block on whole rows (thread = column)
__global__ void k_dinv_swap_rows(int n, int k0, int nb, double *__restrict__ A, const int *__restrict__ piv) {
  const int c = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (c >= n) return;
  for (int j = 0; j < nb; ++j) {
    const int p = piv[k0 + j];
    if (p != k0 + j) {
      const double a = A[(int64_t)(k0 + j) * n + c], b = A[(int64_t)p * n + c];
      A[(int64_t)(k0 + j) * n + c] = b;
      A[(int64_t)p * n + c] = a;
    }
  }
}

// ---- 2. P^-1 of the nb x nb pivot block (unpivoted Gauss-Jordan in LDS, one workgroup of B x B threads)
__global__ void __launch_bounds__(PHX_DINV_B *PHX_DINV_B)
k_dinv_block_inverse(int n, int k0, int nb, const double *__restrict__ A, double *__restrict__ Pinv, int *__restrict__ singular) {
  constexpr int B = PHX_DINV_B;
  __shared__ double P[B][B + 1], Q[B][B + 1];
  const int r = (int)threadIdx.x / B, c = (int)threadIdx.x % B;
  const bool in = r < nb && c < nb;
  P[r][c] = in ? A[(int64_t)(k0 + r) * n + k0 + c] : (r == c ? 1.0 : 0.0);
  Q[r][c] = r == c ? 1.0 : 0.0;
  __syncthreads();
  for (int k = 0; k < nb; ++k) {
    const double d = P[k][k];
    if (d == 0.0) { if (threadIdx.x == 0) *singular = 1; break; }
    __syncthreads();
    const double f = r == k ? 0.0 : P[r][k] / d;
    const double pk = P[k][c], qk = Q[k][c];
    __syncthreads();
    if (r == k) { P[r][c] = pk / d; Q[r][c] = qk / d; }
    else { P[r][c] -= f * pk; Q[r][c] -= f * qk; }
    __syncthreads();
  }
  Pinv[r * B + c] = Q[r][c];
}

// ---- 3. W = A[:, K] (n x B, zero-padded columns) and R = P^-1 A[K, :] (B x n) with R[:, K] = P^-1
__global__ void k_dinv_save_w(int n, int k0, int nb, const double *__restrict__ A, double *__restrict__ W) {
  const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= (int64_t)n * PHX_DINV_B) return;
  const int i = (int)(e / PHX_DINV_B), t = (int)(e % PHX_DINV_B);
  W[e] = t < nb ? A[(int64_t)i * n + k0 + t] : 0.0;
}
__global__ void __launch_bounds__(256)
k_dinv_rowblock(int n, int k0, int nb, const double *__restrict__ A, const double *__restrict__ Pinv, double *__restrict__ R) {
  constexpr int B = PHX_DINV_B;
  __shared__ double Pi[B * B];
  for (int e = (int)threadIdx.x; e < B * B; e += 256) Pi[e] = Pinv[e];
  __syncthreads();
  const int j = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (j >= n) return;
  const bool inK = j >= k0 && j < k0 + nb;
  double a[B];
#pragma unroll
  for (int t = 0; t < B; ++t) a[t] = (!inK && t < nb) ? A[(int64_t)(k0 + t) * n + j] : 0.0;
  for (int c = 0; c < B; ++c) {
    double s = 0.0;
    if (inK) {
      s = c < nb ? Pi[c * B + (j - k0)] : 0.0;
    } else {
#pragma unroll
      for (int t = 0; t < B; ++t) s = __builtin_fma(Pi[c * B + t], a[t], s);
    }
    R[(int64_t)c * n + j] = s;
  }
}

// ---- 4. the pass over the matrix: 64 x 64 tiles, 256 threads, 4 x 4 entries per thread
__global__ void __launch_bounds__(256)
k_dinv_update(int n, int k0, int nb, double *__restrict__ A, const double *__restrict__ W, const double *__restrict__ R) {
  constexpr int B = PHX_DINV_B, T = 64;
  __shared__ double Ws[T][B + 1];
  __shared__ double Rs[B][T + 1];
  const int i0 = (int)blockIdx.y * T, j0 = (int)blockIdx.x * T;
  const int tid = (int)threadIdx.x;
  for (int e = tid; e < T * B; e += 256) {
    const int i = e / B, t = e % B;
    Ws[i][t] = i0 + i < n ? W[(int64_t)(i0 + i) * B + t] : 0.0;
  }
  for (int e = tid; e < B * T; e += 256) {
    const int t = e / T, j = e % T;
    Rs[t][j] = j0 + j < n ? R[(int64_t)t * n + j0 + j] : 0.0;
  }
  __syncthreads();
  const int ti = (tid / 16) * 4, tj = (tid % 16) * 4;
  double acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
#pragma unroll 8
  for (int t = 0; t < B; ++t) {
    double w[4], r[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) { w[a] = Ws[ti + a][t]; r[a] = Rs[t][tj + a]; }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_fma(w[a], r[b], acc[a][b]);
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int i = i0 + ti + a;
    if (i >= n) continue;
    const bool rowK = i >= k0 && i < k0 + nb;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int j = j0 + tj + b;
      if (j >= n) continue;
      double *p = A + (int64_t)i * n + j;
      if (rowK) *p = Rs[i - k0][tj + b];
      else *p = ((j >= k0 && j < k0 + nb) ? 0.0 : *p) - acc[a][b];
    }
  }
}

// ---- columns back: out[i][j] = in[i][src[j]], one workgroup per row, the row staged in LDS
__global__ void __launch_bounds__(256)
k_dinv_permute_cols(int n, double *__restrict__ A, const int *__restrict__ src) {
  extern __shared__ double drow[];
  double *a = A + (int64_t)blockIdx.x * n;
  for (int j = (int)threadIdx.x; j < n; j += 256) drow[j] = a[j];
  __syncthreads();
  for (int j = (int)threadIdx.x; j < n; j += 256) a[j] = drow[src[j]];
}

// A (device, n x n row-major) := A^-1.  *singular_out = 1: a pivot vanished.
static int dense_inverse_inplace(double *A, int n, hipStream_t st, int *singular_out) {
  constexpr int B = PHX_DINV_B;
  *singular_out = 0;
  if (n <= 0) return PHX_OK;
  PHX_REQUIRE((size_t)n * sizeof(double) <= (size_t)(160 * 1024), PHX_ERR_VALUE, "dense inverse: %d rows exceed the row staging of the column permutation", n);
  double *W = nullptr, *R = nullptr, *Pinv = nullptr;
  int *piv = nullptr, *flag = nullptr, *win[2] = {nullptr, nullptr};
  const size_t nwin = (size_t)phx_div_up(n, 256) * B;
  auto drop = [&]() { (void)phx_free(win[0]); (void)phx_free(win[1]); (void)phx_free(W); (void)phx_free(R); (void)phx_free(Pinv); (void)phx_free(piv); (void)phx_free(flag); };
  if (phx_malloc(&win[0], sizeof(int) * nwin) != hipSuccess || phx_malloc(&win[1], sizeof(int) * nwin) != hipSuccess ||
      phx_malloc(&W, sizeof(double) * (size_t)n * B) != hipSuccess ||
      phx_malloc(&R, sizeof(double) * (size_t)n * B) != hipSuccess || phx_malloc(&Pinv, sizeof(double) * B * B) != hipSuccess ||
      phx_malloc(&piv, sizeof(int) * (size_t)(n + B)) != hipSuccess || phx_malloc(&flag, sizeof(int)) != hipSuccess) {
    drop();
    return PHX_ERR_HIP;
  }
  (void)hipMemsetAsync(flag, 0, sizeof(int), st);
  static bool lds_ok = false;
  if (!lds_ok) {
    PHX_HIP(hipFuncSetAttribute((const void *)k_dinv_permute_cols, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    lds_ok = true;
  }
  const dim3 gcol((unsigned)phx_div_up(n, 256)), b256(256);
  const dim3 gt((unsigned)phx_div_up(n, 64), (unsigned)phx_div_up(n, 64));
  for (int k0 = 0; k0 < n; k0 += B) {
    const int nb = std::min(B, n - k0);
    {
      // tournament: rounds of 256-row blocks until one block has named the nb pivot rows
      int ncand = n - k0, lvl = 0;
      const int *cand = nullptr;
      for (;;) {
        const int nblk = (int)phx_div_up(ncand, 256);
        k_dinv_tournament<<<dim3((unsigned)nblk), b256, 0, st>>>(n, k0, nb, A, cand, ncand, win[lvl & 1]);
        cand = win[lvl & 1];
        ncand = nblk * B;
        ++lvl;
        if (nblk == 1) break;
      }
      k_dinv_swaps_from_winners<<<1, 64, 0, st>>>(k0, nb, cand, piv, flag);
    }
    k_dinv_swap_rows<<<gcol, b256, 0, st>>>(n, k0, nb, A, piv);
    k_dinv_block_inverse<<<1, B * B, 0, st>>>(n, k0, nb, A, Pinv, flag);
    k_dinv_save_w<<<dim3((unsigned)phx_div_up((int64_t)n * B, 256)), b256, 0, st>>>(n, k0, nb, A, W);
    k_dinv_rowblock<<<gcol, b256, 0, st>>>(n, k0, nb, A, Pinv, R);
    k_dinv_update<<<gt, b256, 0, st>>>(n, k0, nb, A, W, R);
  }
  PHX_HIP(hipGetLastError());
  // the accumulated row permutation: replayed on the host (n integers), its inverse applied to the columns
  std::vector<int> hp((size_t)n), pos((size_t)n), src((size_t)n);
  int hflag = 0;
  PHX_HIP(hipMemcpyAsync(hp.data(), piv, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, st));
  PHX_HIP(hipMemcpyAsync(&hflag, flag, sizeof(int), hipMemcpyDeviceToHost, st));
  PHX_HIP(hipStreamSynchronize(st));
  if (hflag) { drop(); *singular_out = 1; return PHX_OK; }
  // rows were swapped k <-> piv[k] for k = 0 .. n-1: the elimination ran on P A with (P A)[k] = A[pos[k]]; (P A)^-1 = A^-1 P^T,
  // so column j of the result is column pos^-1... of A^-1: A^-1[:, pos[k]] = result[:, k]
  for (int k = 0; k < n; ++k) pos[(size_t)k] = k;
  for (int k = 0; k < n; ++k) std::swap(pos[(size_t)k], pos[(size_t)hp[(size_t)k]]);
  for (int k = 0; k < n; ++k) src[(size_t)pos[(size_t)k]] = k;
  PHX_HIP(hipMemcpyAsync(piv, src.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, st));
  k_dinv_permute_cols<<<dim3((unsigned)n), b256, sizeof(double) * (size_t)n, st>>>(n, A, piv);
  PHX_HIP(hipGetLastError());
  PHX_HIP(hipStreamSynchronize(st));
  drop();
  return PHX_OK;
}

// Test / inspection entry: a_host (n x n, row-major) := its inverse, computed on `device`.
extern "C" int phx_dense_inverse(int device, int64_t n, double *a_host, int *singular) {
  PHX_HIP(hipSetDevice(device));
  PHX_REQUIRE(n > 0 && n <= 20000 && a_host != nullptr && singular != nullptr, PHX_ERR_VALUE, "phx_dense_inverse: bad arguments");
  double *A = nullptr;
  PHX_HIP(phx_malloc(&A, sizeof(double) * (size_t)n * (size_t)n));
  int rc = hipMemcpy(A, a_host, sizeof(double) * (size_t)n * (size_t)n, hipMemcpyHostToDevice) == hipSuccess ? PHX_OK : PHX_ERR_HIP;
  if (rc == PHX_OK) rc = dense_inverse_inplace(A, (int)n, nullptr, singular);
  if (rc == PHX_OK && hipMemcpy(a_host, A, sizeof(double) * (size_t)n * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) rc = PHX_ERR_HIP;
  (void)phx_free(A);
  return rc;
}
