// Sparse solve on the device (gfx950): SELL-64 SpMV + right-Jacobi BiCGStab on the active system.
// Replaces PETSc KSP(preonly)+PC(lu)+MUMPS, demo/weak-dirichlet/flower/main.py:162-182 [3P].
// The weak-Dirichlet matrix is NOT symmetric (main.py:114 has no transposed partner), so plain CG
// diverges on it (DESIGN.md "Solver"); BiCGStab uses the same SpMV/dot/axpy kernels.
//
// SpMV layout: rows are grouped in slices of 64 (one wavefront per slice, one lane per row);
// inside a slice entries are stored column-major, so lane r reads val[base + k*64 + r]:
// every wave-instruction is one contiguous 512 B (f64) / 256 B (i32) segment.  Rows are sorted
// by length with a STABLE sort (removes the padding, keeps equal-length rows in natural order), and
// the whole system is renumbered into that order so y is written coalesced.  Explicitly stored zeros
// of the assembled CSR are not carried into the SELL copy.
#include "phx_prim.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "phx_common.h"
#include "phx_select.h"

#define SELL_C 64
#define SELL_S 16   // rows per slice of the stored rows of a structured system (four lanes per row)
// Rows per length-sorting window.  Measured on the 256^3 system (2.9 M rows): no sort 180 us,
// 512 -> 113 us, 4096 -> 184 us, 32768 -> 122 us, one global window -> 81 us per SpMV; the stable
// global sort keeps the 7-point interior rows in natural order and only moves the few long
// (cut-region) rows, so the x-gather locality survives.  PHX_SELL_WINDOW overrides (tuning aid).
static int64_t g_sell_window = (int64_t)1 << 40;
// Dot products: every block adds its partial sum to one of NSLOT accumulators (one 64-byte line
// each, so the memory-side atomics of different blocks do not queue on one address); a one-wave
// kernel folds the slots into the scalar the next phase reads.
#ifdef PHX_SPMV_NT
#define NT_LOAD(p) __builtin_nontemporal_load(p)
#else
#define NT_LOAD(p) (*(p))
#endif
#define NSLOT 64
#define SLOT_STRIDE 8
#define P_OFF 16
#define PHX_SCAL_DOUBLES (P_OFF + 2 * 8 * NSLOT * SLOT_STRIDE)  // two parity sets of slots

// ---------------------------------------------------------------------------------------------
// SELL construction
// ---------------------------------------------------------------------------------------------
__global__ void k_row_lengths(int64_t n, const int64_t *__restrict__ rowptr,
                              const int32_t *__restrict__ col, const double *__restrict__ val,
                              const int32_t *__restrict__ row_nz, uint32_t *__restrict__ keys,
                              int32_t *__restrict__ rows, unsigned long long *__restrict__ total,
                              int64_t window) {
  const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int len = 0;
  if (r < n) {
    if (row_nz) {  // counted by the compaction kernel while the row sat in a wavefront
      len = row_nz[r];
    } else {
      for (int64_t k = rowptr[r]; k < rowptr[r + 1]; ++k)
        if (val[k] != 0.0 || col[k] == (int32_t)r) ++len;
    }
    if (len > 1023) len = 1023;
  }
  int wsum = len;
  for (int o = 32; o > 0; o >>= 1) wsum += __shfl_xor(wsum, o);
  if ((threadIdx.x & 63) == 0 && wsum) atomicAdd(total, (unsigned long long)wsum);
  if (r >= n) return;
  // ascending key = (window, 1023-len): descending length inside each window
  keys[r] = ((uint32_t)(r / window) << 10) | (uint32_t)(1023 - len);
  rows[r] = (int32_t)r;
}

__global__ void k_invert_perm(int64_t n, const int32_t *__restrict__ perm, int32_t *__restrict__ iperm) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) iperm[perm[i]] = (int32_t)i;
}

__global__ void k_slice_widths(int64_t nslices, int64_t n, const uint32_t *__restrict__ sorted_keys,
                               int64_t *__restrict__ widths) {
  const int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (s > nslices) return;
  if (s == nslices) { widths[s] = 0; return; }
  // rows of a slice may straddle two windows: take the max over the slice
  int w = 0;
  for (int r = 0; r < SELL_C; ++r) {
    const int64_t pos = s * SELL_C + r;
    if (pos < n) w = max(w, 1023 - (int)(sorted_keys[pos] & 0x3ff));
  }
  widths[s] = (int64_t)w * SELL_C;
}

__global__ void k_sell_fill(int64_t n, const int64_t *__restrict__ rowptr,
                            const int32_t *__restrict__ col, const double *__restrict__ val,
                            const double *__restrict__ diag, const int32_t *__restrict__ perm,
                            const int32_t *__restrict__ iperm, const int64_t *__restrict__ slice_ptr,
                            int32_t *__restrict__ scol, double *__restrict__ sval,
                            double *__restrict__ sraw) {
  const int64_t pos = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t s = pos / SELL_C;
  const int lane = (int)(pos % SELL_C);
  if (s * SELL_C >= n) return;
  const int64_t base = slice_ptr[s];
  const int width = (int)((slice_ptr[s + 1] - base) / SELL_C);
  int k = 0;
  if (pos < n) {
    const int32_t r = perm[pos];
    for (int64_t e = rowptr[r]; e < rowptr[r + 1]; ++e) {
      const double v = val[e];
      const int32_t c = col[e];
      if (v == 0.0 && c != r) continue;
      scol[base + (int64_t)k * SELL_C + lane] = iperm[c];
      sraw[base + (int64_t)k * SELL_C + lane] = v;
      sval[base + (int64_t)k * SELL_C + lane] = v / diag[c];  // A D^-1
      ++k;
    }
  }
  const int32_t self = pos < n ? (int32_t)pos : 0;
  for (; k < width; ++k) {
    scol[base + (int64_t)k * SELL_C + lane] = self;
    sraw[base + (int64_t)k * SELL_C + lane] = 0.0;
    sval[base + (int64_t)k * SELL_C + lane] = 0.0;
  }
}

// Value-indexed slices (CSR-VI idea on SELL).  phi-FEM matrices on a structured background mesh repeat
// a handful of doubles over the interior rows (7-point rows: ~6 distinct values per slice when the cell
// size is a dyadic rational, a few dozen otherwise), so a slice whose values take K <= 192 distinct bit
// patterns is rewritten IN PLACE, inside its own value region, as
//   [ceil(K/64) x 64-entry dictionary][one byte code per entry, 4 consecutive k packed per lane dword]:
// 5 instead of 12 bytes per entry for the SpMV to stream, and the products are bit-identical.
// kind[s] = K (0: the slice keeps raw doubles).  One wavefront per slice; `totals` = {indexed slices,
// bytes of value stream the SpMV reads, slices with K > 64}.
#define VI_MAX 192
typedef uint32_t __attribute__((may_alias)) u32_alias;
typedef unsigned long long __attribute__((may_alias)) u64_alias;

__global__ void __launch_bounds__(256)
k_sell_index(int64_t nslices, const int64_t *__restrict__ slice_ptr, double *sval,
             uint8_t *__restrict__ kind, unsigned long long *__restrict__ totals, int enable) {
  __shared__ unsigned long long sdict[4][VI_MAX];
  const int lane = threadIdx.x & 63;
  unsigned long long *dict = sdict[threadIdx.x >> 6];  // private to this wavefront
  const int64_t s = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (s >= nslices) return;
  const int64_t base = slice_ptr[s];
  const int width = (int)((slice_ptr[s + 1] - base) >> 6);
  u64_alias *v = reinterpret_cast<u64_alias *>(sval + base);
  int K = 0;
  bool ok = enable != 0;
  for (int k = 0; k < width && ok; ++k) {
    const unsigned long long val = v[(int64_t)k * SELL_C + lane];
    bool found = false;
    for (int j = 0; j < K; ++j) found |= (val == dict[j]);
    unsigned long long pending = __ballot(!found);
    while (pending) {
      if (K == VI_MAX) { ok = false; break; }
      const int src = __ffsll((long long)pending) - 1;
      const unsigned long long nv = (unsigned long long)__shfl((long long)val, src);
      if (lane == 0) dict[K] = nv;
      ++K;
      pending &= ~__ballot(val == nv);
    }
    __builtin_amdgcn_wave_barrier();
  }
  // the region (512 B per k) must hold dictionary chunks + codes; chunks <= 3 keeps the in-place
  // rewrite below from clobbering rows it has not read yet
  const int chunks = (K + SELL_C - 1) / SELL_C, code_rows = (width + 3) / 4;
  if (ok && 2 * chunks + code_rows > 2 * width) ok = false;
  if (!ok) {
    if (lane == 0) {
      kind[s] = 0;
      atomicAdd(&totals[1], (unsigned long long)width * SELL_C * 8ull);
    }
    return;
  }
  u32_alias *cw = reinterpret_cast<u32_alias *>(v + (int64_t)chunks * SELL_C);
  for (int k4 = 0; k4 < code_rows; ++k4) {
    unsigned long long q[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = 4 * k4 + j;
      q[j] = k < width ? v[(int64_t)k * SELL_C + lane] : 0ull;
    }
    uint32_t codes = 0;
    for (int e = 0; e < K; ++e) {
      const unsigned long long d = dict[e];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (4 * k4 + j < width && q[j] == d) codes |= (uint32_t)e << (8 * j);
    }
    // rows 4 k4 .. 4 k4 + 3 were read by every lane above; the dword lands in row
    // chunks + k4 / 2 <= 3 + k4 / 2 <= 4 k4 + 3, never in a row still to be read
    cw[(int64_t)k4 * SELL_C + lane] = codes;
  }
  for (int j = lane; j < chunks * SELL_C; j += SELL_C) v[j] = j < K ? dict[j] : 0ull;
  if (lane == 0) {
    kind[s] = (uint8_t)K;
    atomicAdd(&totals[0], 1ull);
    if (chunks > 1) atomicAdd(&totals[2], 1ull);
    atomicAdd(&totals[1], (unsigned long long)K * 8ull + (unsigned long long)code_rows * 256ull);
  }
}

int phx_system_build_sell(phx_system *s) {
  phx_mesh *m = s->mesh;
  const int64_t n = s->n;
  if (const char *e = getenv("PHX_SELL_WINDOW")) g_sell_window = atoll(e) > 0 ? atoll(e) : g_sell_window;
  PHX_REQUIRE(n / g_sell_window < (1 << 21), PHX_ERR_VALUE, "system too large for the SELL sort key");
  uint32_t *keys = nullptr, *keys2 = nullptr;
  int32_t *rows = nullptr;
  PHX_HIP(phx_malloc(&keys, sizeof(uint32_t) * (size_t)n));
  PHX_HIP(phx_malloc(&keys2, sizeof(uint32_t) * (size_t)n));
  PHX_HIP(phx_malloc(&rows, sizeof(int32_t) * (size_t)n));
  PHX_HIP(phx_malloc(&s->perm, sizeof(int32_t) * (size_t)n));
  PHX_HIP(phx_malloc(&s->iperm, sizeof(int32_t) * (size_t)n));
  const dim3 block(256), grid((unsigned)phx_div_up(n, 256));
  unsigned long long *dtotal = nullptr;
  PHX_HIP(phx_malloc(&dtotal, sizeof(unsigned long long)));
  PHX_HIP(hipMemsetAsync(dtotal, 0, sizeof(unsigned long long), m->stream));
  k_row_lengths<<<grid, block, 0, m->stream>>>(n, s->rowptr, s->col, s->val, s->row_nz, keys, rows, dtotal, g_sell_window);
  unsigned long long htotal = 0;
  PHX_HIP(hipMemcpyAsync(&htotal, dtotal, sizeof(htotal), hipMemcpyDeviceToHost, m->stream));
  PHX_HIP(hipStreamSynchronize(m->stream));
  PHX_HIP(phx_free(dtotal));
  s->sell_true_nnz = (int64_t)htotal;
  // only the bits that can differ are sorted: 10 length bits (+ the window index if windowed)
  int end_bit = 10;
  for (int64_t w = (n - 1) / g_sell_window; w > 0; w >>= 1) ++end_bit;
  size_t bytes = 0;
  PHX_HIP(phx_sort_pairs(nullptr, bytes, keys, keys2, rows, s->perm, (size_t)n, 0, end_bit, m->stream));
  void *tmp = nullptr;
  PHX_HIP(phx_malloc(&tmp, bytes ? bytes : 16));
  PHX_HIP(phx_sort_pairs(tmp, bytes, keys, keys2, rows, s->perm, (size_t)n, 0, end_bit, m->stream));
  k_invert_perm<<<grid, block, 0, m->stream>>>(n, s->perm, s->iperm);
  s->nslices = phx_div_up(n, SELL_C);
  int64_t *widths = nullptr;
  PHX_HIP(phx_malloc(&widths, sizeof(int64_t) * (size_t)(s->nslices + 1)));
  PHX_HIP(phx_malloc(&s->slice_ptr, sizeof(int64_t) * (size_t)(s->nslices + 1)));
  k_slice_widths<<<dim3((unsigned)phx_div_up(s->nslices + 1, 256)), block, 0, m->stream>>>(
      s->nslices, n, keys2, widths);
  {
    size_t b2 = 0;
    PHX_HIP(phx_exclusive_sum(nullptr, b2, widths, s->slice_ptr, (size_t)(s->nslices + 1), m->stream));
    void *t2 = nullptr;
    PHX_HIP(phx_malloc(&t2, b2 ? b2 : 16));
    PHX_HIP(phx_exclusive_sum(t2, b2, widths, s->slice_ptr, (size_t)(s->nslices + 1), m->stream));
    PHX_HIP(hipStreamSynchronize(m->stream));
    PHX_HIP(phx_free(t2));
  }
  PHX_HIP(hipMemcpy(&s->sell_nnz, s->slice_ptr + s->nslices, sizeof(int64_t), hipMemcpyDeviceToHost));
  PHX_HIP(phx_malloc(&s->sell_col, sizeof(int32_t) * (size_t)s->sell_nnz));
  PHX_HIP(phx_malloc(&s->sell_val, sizeof(double) * (size_t)s->sell_nnz));
  PHX_HIP(phx_malloc(&s->sell_val_raw, sizeof(double) * (size_t)s->sell_nnz));
  k_sell_fill<<<dim3((unsigned)phx_div_up(s->nslices * SELL_C, 256)), block, 0, m->stream>>>(
      n, s->rowptr, s->col, s->val, s->diag, s->perm, s->iperm, s->slice_ptr, s->sell_col,
      s->sell_val, s->sell_val_raw);
  PHX_HIP(hipGetLastError());
  {
    PHX_HIP(phx_malloc(&s->sell_kind, (size_t)s->nslices * 2));
    s->sell_kind_raw = s->sell_kind + s->nslices;
    unsigned long long *tot = nullptr, htot[8];
    PHX_HIP(phx_malloc(&tot, sizeof(htot)));
    PHX_HIP(hipMemsetAsync(tot, 0, sizeof(htot), m->stream));
    const dim3 gi((unsigned)phx_div_up(s->nslices, 4));
    k_sell_index<<<gi, block, 0, m->stream>>>(s->nslices, s->slice_ptr, s->sell_val, s->sell_kind, tot, m->spmv_value_index);
    // the unscaled copy only serves phx_spmv (inspection): it stays raw
    PHX_HIP(hipMemsetAsync(s->sell_kind_raw, 0, (size_t)s->nslices, m->stream));
    PHX_HIP(hipGetLastError());
    PHX_HIP(hipMemcpyAsync(htot, tot, sizeof(htot), hipMemcpyDeviceToHost, m->stream));
    PHX_HIP(hipStreamSynchronize(m->stream));
    PHX_HIP(phx_free(tot));
    s->sell_indexed_slices = (int64_t)htot[0];
    s->sell_indexed_large = (int64_t)htot[2];
    // what one SpMV of the solve streams from the matrix: columns + value stream + slice table
    s->sell_stream_bytes = 4 * s->sell_nnz + (int64_t)htot[1] + 9 * s->nslices;
  }
  PHX_HIP(hipStreamSynchronize(m->stream));
  PHX_HIP(phx_free(tmp)); PHX_HIP(phx_free(keys)); PHX_HIP(phx_free(keys2)); PHX_HIP(phx_free(rows));
  PHX_HIP(phx_free(widths));
  // solver workspace: 9 vectors + scalars
  PHX_HIP(phx_malloc(&s->work, sizeof(double) * (size_t)n * 9));
  PHX_HIP(phx_malloc(&s->scal, sizeof(double) * PHX_SCAL_DOUBLES));
  PHX_CHECK(phx_mesh_pinned_scalars(s->mesh, &s->scal_h));
  return PHX_OK;
}

// ---------------------------------------------------------------------------------------------
// Structured systems (phx_common.h): stencil segments over the rows of C0, SELL over the stored rows, built
// straight from the row slots of the assembly.  Solver order = active order (perm = identity).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int64_t sv_base(const phx_slot_view &sv, int64_t row, int *W) {
  if (sv.off) { *W = 1 << sv.wlog[row]; return sv.off[row]; }
  *W = sv.W;
  return row * sv.W;
}

// active row of the column key `cc` of a slot
__device__ __forceinline__ int32_t sv_col(const phx_slot_view &sv, int32_t cc, int32_t nent, const int32_t *du, const int32_t *dp) {
  if (sv.pneg) return cc < -1 ? dp[-2 - cc] : du[cc];
  return cc < nent ? du[cc] : dp[cc - nent];
}

// per stored row (one wavefront each, over the compacted list): entries the SELL copy keeps (non-zero, or the
// diagonal), the diagonal itself, the structural count.  A C0 row in the list (C0 rows next to a non-C0 row are
// stored, not applied) has no slots unless the CSR is exported: its row IS the stencil row.
__global__ void __launch_bounds__(256)
k_slot_row_meta(int64_t ns, const int32_t *__restrict__ list, phx_slot_view sv, const uint8_t *__restrict__ c0,
                int32_t nent, const int32_t *__restrict__ du, const int32_t *__restrict__ dp, int gdim,
                int32_t *__restrict__ len, int32_t *__restrict__ nstruct, double *__restrict__ diag) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  if (i >= ns) return;
  const int32_t row = list[i];
  if (c0 && c0[row]) {
    if (lane == 0) { len[i] = gdim == 3 ? 7 : 5; nstruct[i] = gdim == 3 ? 15 : 7; }
    return;
  }
  int W;
  const int64_t base = sv_base(sv, row, &W);
  const int nclean = sv.clean ? sv.clean[row] : 0;
  const int limit = nclean ? nclean : W;
  int st = 0, keep = 0;
  for (int k = lane; k < limit; k += 64) {
    const int32_t cc = sv.cols[base + k];
    if (cc == -1) continue;
    const int32_t col = sv_col(sv, cc, nent, du, dp);
    const double v = sv.vals[base + k];
    ++st;
    if (col == row) { diag[row] = v; ++keep; }
    else if (v != 0.0) ++keep;
  }
  for (int o = 32; o > 0; o >>= 1) { st += __shfl_xor(st, o); keep += __shfl_xor(keep, o); }
  if (lane == 0) { len[i] = keep; nstruct[i] = st; }
}

// totals[0] += sum a, totals[1] += sum b (one atomic pair per block; per-row atomics on a few addresses cost
// 2.7 ms for 6e5 rows)
__global__ void __launch_bounds__(256)
k_sum2_i32(int64_t n, const int32_t *__restrict__ a, const int32_t *__restrict__ b,
           unsigned long long *__restrict__ totals) {
  __shared__ unsigned long long red[2][4];
  unsigned long long sa = 0, sb = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    sa += (unsigned long long)a[i];
    sb += (unsigned long long)b[i];
  }
  for (int o = 32; o > 0; o >>= 1) { sa += __shfl_xor(sa, o); sb += __shfl_xor(sb, o); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sa; red[1][threadIdx.x >> 6] = sb; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&totals[0], red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    atomicAdd(&totals[1], red[1][0] + red[1][1] + red[1][2] + red[1][3]);
  }
}

struct SelStored { const uint8_t *c0; __host__ __device__ bool operator()(const int32_t &i) const { return c0[i] == 0; } };

// tile > 0: rows of one tile^3 block of lattice vertices (u and p rows alike) come first in the key: a slice then
// gathers x entries of ONE (tile + 4)^3 neighbourhood, which the L1 of its CU holds
__global__ void k_stored_keys(int64_t ns, const int32_t *__restrict__ list, const int32_t *__restrict__ len,
                              uint32_t *__restrict__ keys, const int64_t *__restrict__ full, int64_t nv, int64_t n0,
                              int64_t n01, int tile, int t0, int t1) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= ns) return;
  uint32_t key = (uint32_t)(1023 - min(len[i], 1023));   // ascending key = descending length
  if (tile > 0) {
    int64_t v = full[list[i]];
    if (v >= nv) v -= nv;
    const int x = (int)(v % n0) / tile, y = (int)((v % n01) / n0) / tile, z = (int)(v / n01) / tile;
    key |= (uint32_t)((z * t1 + y) * t0 + x) << 10;
  }
  keys[i] = key;
}

// Stable counting sort of (key, value) pairs with keys < 1024 (the stored rows by length: half a million items, a few
// dozen distinct keys).  rocPRIM sorts an input of this size with a merge sort: 23 launches of 5-8 us each, more than
// the SELL fill it feeds.  Here: one wavefront per segment of 256 consecutive items, keys in [klo, klo + nk).  Pass 1 leaves H[key][segment] =
// items of the segment with that key; an exclusive scan over H in (key, segment) order turns the counts into
// destinations; pass 2 sends every item to H[key][segment] + its rank among the equal keys of its segment before it
// (equal keys of one round of 64 items are peeled off with ballots, in lane order).  Four launches.
#define PHX_CSORT_KEYS 1024
#define PHX_CSORT_SEG 256
template <bool SCATTER>
__global__ void __launch_bounds__(256)
k_csort(int64_t n, const uint32_t *__restrict__ keys, const int32_t *__restrict__ vals, int64_t nseg, uint32_t klo, int nk,
        int32_t *__restrict__ H, uint32_t *__restrict__ keys_out, int32_t *__restrict__ vals_out) {
  __shared__ int32_t cnt[4][PHX_CSORT_KEYS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t seg = blockIdx.x * (int64_t)4 + wv;
  if (seg >= nseg) return;   // no block-wide synchronisation below: a wave works on its own LDS strip
  int32_t *c = cnt[wv];
  for (int i = lane; i < nk; i += 64) c[i] = SCATTER ? H[(int64_t)i * nseg + seg] : 0;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int64_t base = seg * PHX_CSORT_SEG;
  const unsigned long long below = (1ull << lane) - 1ull;
  for (int r = 0; r < PHX_CSORT_SEG / 64; ++r) {
    const int64_t i = base + r * 64 + lane;
    const bool ok = i < n;
    const uint32_t k = ok ? min(max(keys[i], klo) - klo, (uint32_t)(nk - 1)) : 0u;
    const int32_t v = SCATTER && ok ? vals[i] : 0;
    unsigned long long todo = __ballot(ok);
    while (todo) {
      const int first = __ffsll((long long)todo) - 1;
      const uint32_t kf = (uint32_t)__shfl((int)k, first);
      const bool mine = ok && k == kf;
      const unsigned long long same = __ballot(mine);
      if (SCATTER && mine) {
        const int64_t dst = (int64_t)c[kf] + __popcll(same & below);
        keys_out[dst] = keys[i];
        vals_out[dst] = v;
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (lane == first) c[kf] += __popcll(same);
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
      todo &= ~same;
    }
  }
  if (!SCATTER)
    for (int i = lane; i < nk; i += 64) H[(int64_t)i * nseg + seg] = c[i];
}

// keys_out / vals_out = the pairs sorted by key (< 1024), stable.  `later`: temporaries, freed by the caller behind its
// next synchronisation.
static int phx_counting_sort_pairs(hipStream_t st, const uint32_t *keys, uint32_t *keys_out, const int32_t *vals,
                                   int32_t *vals_out, int64_t n, uint32_t klo, int nk, std::vector<void *> *later) {
  if (n <= 0) return PHX_OK;
  PHX_REQUIRE(nk >= 1 && nk <= PHX_CSORT_KEYS, PHX_ERR_VALUE, "counting sort over %d keys", nk);
  const int64_t nseg = phx_div_up(n, (int64_t)PHX_CSORT_SEG);
  const size_t nh = (size_t)nk * (size_t)nseg;
  int32_t *H = nullptr, *H2 = nullptr;
  PHX_HIP(phx_malloc(&H, sizeof(int32_t) * nh));
  PHX_HIP(phx_malloc(&H2, sizeof(int32_t) * nh));
  const dim3 grid((unsigned)phx_div_up(nseg, 4)), block(256);
  k_csort<false><<<grid, block, 0, st>>>(n, keys, vals, nseg, klo, nk, H, nullptr, nullptr);
  size_t bytes = 0;
  PHX_HIP(phx_exclusive_sum(nullptr, bytes, H, H2, nh, st));
  void *tmp = nullptr;
  PHX_HIP(phx_malloc(&tmp, bytes ? bytes : 16));
  PHX_HIP(phx_exclusive_sum(tmp, bytes, H, H2, nh, st));
  k_csort<true><<<grid, block, 0, st>>>(n, keys, vals, nseg, klo, nk, H2, keys_out, vals_out);
  PHX_HIP(hipGetLastError());
  later->push_back(H); later->push_back(H2); later->push_back(tmp);
  return PHX_OK;
}

__global__ void k_fill_i32(int64_t n, int32_t *__restrict__ a, int32_t v, int iota) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) a[i] = iota ? (int32_t)i : v;
}

// one wavefront per stored row: the kept entries of its slots, sorted by column (lanes of a slice then gather
// neighbouring x entries at equal k), go to consecutive k of its SELL lane; columns become solver positions.
// A stored C0 row is written from the stencil coefficients.  Rows of up to 64 slots (P1).
__global__ void __launch_bounds__(256)
k_sell_fill_slots(int64_t ns, const int32_t *__restrict__ rows, phx_slot_view sv, const uint8_t *__restrict__ c0,
                  int32_t nent, const int32_t *__restrict__ du, const int32_t *__restrict__ dp, int64_t nu, int gdim,
                  const double *__restrict__ stencil, const double *__restrict__ diag,
                  const int32_t *__restrict__ iperm, const int64_t *__restrict__ slice_ptr,
                  int32_t *__restrict__ scol, double *__restrict__ sval, double *__restrict__ sraw,
                  const int64_t *__restrict__ full, int64_t n0, int64_t n01) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  if (i >= ns) return;
  const int32_t row = rows[i];
  const int64_t sl = i / SELL_S, li = i % SELL_S;
  const int64_t sb = slice_ptr[sl];
  const int width = (int)((slice_ptr[sl + 1] - sb) / SELL_S);
  int32_t c = 0x7fffffff;   // ACTIVE column index while sorting
  double v = 0.0;
  if (c0[row]) {
    // lattice Laplacian row: lanes 0..6 = self, -x, +x, -y, +y, -z, +z
    const int64_t vtx = full[row];
    const int nent7 = gdim == 3 ? 7 : 5;
    if (lane < nent7) {
      const int64_t off = lane == 0 ? 0 : (lane <= 2 ? 1 : (lane <= 4 ? n0 : n01));
      const int64_t w = (lane & 1) ? vtx - off : vtx + off;   // odd lanes: minus side (lane 0: off = 0)
      c = du[w];
      v = stencil[lane == 0 ? 0 : (lane + 1) / 2];
    }
  } else {
    int W;
    const int64_t base = sv_base(sv, row, &W);
    const int nclean = sv.clean ? sv.clean[row] : 0;
    const int limit = nclean ? nclean : W;
    if (lane < limit) {
      const int32_t cc = sv.cols[base + lane];
      if (cc != -1) {
        const int32_t col = cc < nent ? du[cc] : dp[cc - nent];
        const double vv = sv.vals[base + lane];
        if (vv != 0.0 || col == row) { c = col; v = vv; }
      }
    }
  }
  // bitonic sort by column across the wavefront (invalid lanes carry INT_MAX and sink to the end)
  for (int k = 2; k <= 64; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      const int32_t oc = __shfl_xor(c, j);
      const double ov = __shfl_xor(v, j);
      const bool up = ((lane & k) == 0);
      const bool lower = ((lane & j) == 0);
      const bool take = (lower == up) ? (oc < c) : (oc > c);
      if (take) { c = oc; v = ov; }
    }
  if (lane < width) {
    const int64_t o = sb + (int64_t)lane * SELL_S + li;
    if (c != 0x7fffffff) {
      scol[o] = iperm[c];
      sraw[o] = v;
      sval[o] = c < nu ? v : v / diag[c];   // u columns unscaled, p columns A D^-1
    } else {
      scol[o] = iperm[row]; sraw[o] = 0.0; sval[o] = 0.0;
    }
  }
}

// slots of the last slice that hold no row: padding entries pointing at row 0 with value 0
__global__ void k_sell_pad_tail(int64_t ns, int64_t nslices, const int64_t *__restrict__ slice_ptr,
                                int32_t *__restrict__ scol, double *__restrict__ sval, double *__restrict__ sraw) {
  const int64_t i = ns + threadIdx.x;
  if (i >= nslices * SELL_S) return;
  const int64_t sl = i / SELL_S, li = i % SELL_S, sb = slice_ptr[sl];
  const int width = (int)((slice_ptr[sl + 1] - sb) / SELL_S);
  for (int k = 0; k < width; ++k) {
    const int64_t o = sb + (int64_t)k * SELL_S + li;
    scol[o] = 0; sval[o] = 0.0; sraw[o] = 0.0;
  }
}

// widths of the 16-row slices of a structured system, rounded up to the four lanes a row is spread over
__global__ void k_slice_widths16(int64_t nslices, int64_t n, const uint32_t *__restrict__ sorted_keys,
                                 int64_t *__restrict__ widths) {
  const int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (s > nslices) return;
  if (s == nslices) { widths[s] = 0; return; }
  int w = 0;
  for (int r = 0; r < SELL_S; ++r) {
    const int64_t pos = s * SELL_S + r;
    if (pos < n) w = max(w, 1023 - (int)(sorted_keys[pos] & 0x3ff));
  }
  w = (w + 3) & ~3;
  widths[s] = (int64_t)w * SELL_S;
}

// C0 rows whose six (2-D: four) axis neighbours are C0 rows as well: the rows the stencil blocks apply
__global__ void k_c0_interior(int64_t nu, const uint8_t *__restrict__ c0, const int64_t *__restrict__ full,
                              const int32_t *__restrict__ du, int64_t n0, int64_t n01, int has_z,
                              uint8_t *__restrict__ c0i) {
  const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r >= nu) return;
  bool ok = c0[r] != 0;
  if (ok) {   // every neighbour of a C0 vertex is active: its whole star is tagged inside
    const int64_t v = full[r];
    ok = c0[du[v - 1]] && c0[du[v + 1]] && c0[du[v - n0]] && c0[du[v + n0]];
    if (ok && has_z) ok = c0[du[v - n01]] && c0[du[v + n01]];
  }
  c0i[r] = ok ? 1 : 0;
}

// solver order: the C0 rows in lattice order, then the other u rows, then the p rows (active order).  Keeping the
// rows near Gamma_h together makes the x entries their SELL rows gather a compact range that stays in L2.
__global__ void k_build_perm(int64_t n, int64_t nu, const uint8_t *__restrict__ c0, const int32_t *__restrict__ rank,
                             int32_t nc0all, int32_t *__restrict__ perm, int32_t *__restrict__ iperm) {
  const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r >= n) return;
  int32_t pos = (int32_t)r;
  if (r < nu) pos = c0[r] ? rank[r] : nc0all + ((int32_t)r - rank[r]);
  iperm[r] = pos;
  perm[pos] = (int32_t)r;
}

// run starts / ends of stencil rows along x lines, in SOLVER positions q < nc0all (C0 rows keep their lattice
// order there: two of them are x neighbours iff their vertices are consecutive)
__global__ void k_seg_flags(int64_t nq, const int32_t *__restrict__ perm, const uint8_t *__restrict__ c0i,
                            const int64_t *__restrict__ full, uint8_t *__restrict__ fs, uint8_t *__restrict__ fe) {
  const int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (q >= nq) return;
  const int32_t r = perm[q];
  const bool in = c0i[r] != 0;
  bool prev = false, next = false;
  if (in && q > 0) { const int32_t rp = perm[q - 1]; prev = c0i[rp] && full[rp] + 1 == full[r]; }
  if (in && q + 1 < nq) { const int32_t rn = perm[q + 1]; next = c0i[rn] && full[r] + 1 == full[rn]; }
  fs[q] = in && !prev;
  fe[q] = in && !next;
}

__global__ void k_seg_fill(int64_t nq, const int32_t *__restrict__ perm, const int32_t *__restrict__ iperm,
                           const uint8_t *__restrict__ fs, const uint8_t *__restrict__ fe,
                           const int32_t *__restrict__ is, const int32_t *__restrict__ ie,
                           const int64_t *__restrict__ full, const int32_t *__restrict__ du, int64_t n0, int64_t n01,
                           int has_z, int32_t *__restrict__ seg) {
  const int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (q >= nq) return;
  if (fs[q]) {
    int32_t *sg = seg + 6 * (int64_t)is[q];
    const int64_t v = full[perm[q]];
    sg[0] = (int32_t)q;
    sg[2] = iperm[du[v + n0]] - (int32_t)q;
    sg[3] = iperm[du[v - n0]] - (int32_t)q;
    sg[4] = has_z ? iperm[du[v + n01]] - (int32_t)q : 0;
    sg[5] = has_z ? iperm[du[v - n01]] - (int32_t)q : 0;
  }
  if (fe[q]) seg[6 * (int64_t)ie[q] + 1] = (int32_t)(q + 1);
}

// ---- stencil blocks of large lattice planes.  The C0 rows are numbered in lattice order, so the z neighbours of a row
// sit one PLANE of rows away.  Where three planes of x (z - 1, z, z + 1) exceed an XCD's 4 MiB L2 -- 1024 x 1024 x 128 slab:
// 3.7e5 rows = 2.9 MB per plane -- a walk over contiguous eighths of the rows fetched x FOUR times through the fabric
// (FETCH_SIZE of the stencil blocks alone: 1.5 GB against the 0.38 GB of x; 283 us).  Instead XCD k takes the k-th eighth of
// EVERY plane, plane after plane: its window is three eighth-planes.  zstart[z] = first position of plane z (zstart[nzp] = nq).
__global__ void k_plane_starts(int64_t nq, const int32_t *__restrict__ perm, const int64_t *__restrict__ full, int64_t n01,
                               int nzp, int32_t *__restrict__ zstart) {
  const int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (q >= nq) return;
  const int z = (int)(full[perm[q]] / n01);
  const int zp = q > 0 ? (int)(full[perm[q - 1]] / n01) : -1;
  for (int zz = zp + 1; zz <= z; ++zz) zstart[zz] = (int32_t)q;
  if (q == nq - 1) for (int zz = z + 1; zz <= nzp; ++zz) zstart[zz] = (int32_t)nq;
}
// slice boundary of part k of plane z
__device__ __forceinline__ int64_t stmap_bound(const int32_t *zstart, int nzp, int64_t nq, int z, int k) {
  if (z >= nzp) return (nq + 63) >> 6;
  const int64_t a = zstart[z], len = zstart[z + 1] - a;
  return (a + (len * k) / 8) >> 6;
}
// block k (one per XCD): entries {first slice, slices <= 16} of its blocks, plane after plane; unused entries {0, 0}
__global__ void __launch_bounds__(1024)
k_stmap_build(int nzp, int64_t nq, const int32_t *__restrict__ zstart, int64_t chunk, int32_t *__restrict__ map) {
  __shared__ int cnt[1024];
  __shared__ int base;
  const int k = (int)blockIdx.x, tid = (int)threadIdx.x;
  int32_t *mk = map + 2 * (int64_t)k * chunk;
  if (tid == 0) base = 0;
  __syncthreads();
  for (int z0 = 0; z0 < nzp; z0 += 1024) {
    const int z = z0 + tid;
    int64_t b0 = 0, b1 = 0;
    if (z < nzp) {
      b0 = stmap_bound(zstart, nzp, nq, z, k);
      b1 = k == 7 ? stmap_bound(zstart, nzp, nq, z + 1, 0) : stmap_bound(zstart, nzp, nq, z, k + 1);
    }
    const int c = (int)((b1 - b0 + 15) >> 4);
    cnt[tid] = c;
    __syncthreads();
    // exclusive scan of the 1024 counts (Hillis-Steele in shared memory)
    for (int o = 1; o < 1024; o <<= 1) {
      const int v = tid >= o ? cnt[tid - o] : 0;
      __syncthreads();
      cnt[tid] += v;
      __syncthreads();
    }
    const int off = base + cnt[tid] - c;
    for (int j = 0; j < c; ++j) {
      mk[2 * (int64_t)(off + j)] = (int32_t)(b0 + 16 * j);
      mk[2 * (int64_t)(off + j) + 1] = (int32_t)min((int64_t)16, b1 - b0 - 16 * j);
    }
    __syncthreads();
    if (tid == 1023) base += cnt[1023];
    __syncthreads();
  }
  for (int64_t j = base + tid; j < chunk; j += 1024) { mk[2 * j] = 0; mk[2 * j + 1] = 0; }
}

// slice record of 64 consecutive positions: {runs intersecting the slice (3 = more than two), index of the first,
// the first two runs inline}
__global__ void k_slice_seg(int64_t nw, int nseg, const int32_t *__restrict__ seg, int32_t *__restrict__ srec) {
  const int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (w >= nw) return;
  int lo = 0, hi = nseg;   // first run whose end lies beyond position 64 w
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if ((int64_t)seg[6 * (int64_t)mid + 1] > w * 64) hi = mid; else lo = mid + 1;
  }
  int32_t *rec = srec + 16 * w;
  for (int j = 0; j < 16; ++j) rec[j] = 0;
  int cnt = 0;
  for (int k = lo; k < nseg && (int64_t)seg[6 * (int64_t)k] < (w + 1) * 64 && cnt < 3; ++k) {
    if (cnt < 2)
      for (int j = 0; j < 6; ++j) rec[2 + 6 * cnt + j] = seg[6 * (int64_t)k + j];
    ++cnt;
  }
  rec[0] = cnt;
  rec[1] = lo;
}

__global__ void k_cscale(int64_t n, int64_t nu, const int32_t *__restrict__ perm, const double *__restrict__ diag,
                         double *__restrict__ cs) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) cs[i] = i < nu ? 1.0 : 1.0 / diag[perm[i]];
}

__global__ void k_map_i32(int64_t n, const int32_t *__restrict__ map, int32_t *__restrict__ a) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n && a[i] >= 0) a[i] = map[a[i]];
}

struct U8AsI32 { __host__ __device__ int32_t operator()(const uint8_t &a) const { return (int32_t)a; } };
static int scan_u8(hipStream_t st, const uint8_t *flags, int32_t *out, int64_t n, int32_t *total) {
  auto it = rocprim::make_transform_iterator(flags, U8AsI32());
  size_t bytes = 0;
  PHX_HIP(phx_exclusive_sum(nullptr, bytes, it, out, (size_t)(n), st));
  void *tmp = nullptr;
  PHX_HIP(phx_malloc(&tmp, bytes ? bytes : 16));
  PHX_HIP(phx_exclusive_sum(tmp, bytes, it, out, (size_t)(n), st));
  int32_t last = 0;
  uint8_t lastf = 0;
  const phx_rb_item rb[2] = {{out + (n - 1), 4, &last}, {flags + (n - 1), 1, &lastf}};
  PHX_CHECK(phx_read_back(st, rb, 2));
  PHX_HIP(phx_free(tmp));
  *total = last + (int32_t)lastf;
  return PHX_OK;
}

static int scan_u8_pair(hipStream_t st, const uint8_t *fa, int32_t *oa, int32_t *ta, const uint8_t *fb, int32_t *ob,
                        int32_t *tb, int64_t n) {
  auto ia = rocprim::make_transform_iterator(fa, U8AsI32());
  auto ib = rocprim::make_transform_iterator(fb, U8AsI32());
  size_t bytes = 0;
  PHX_HIP(phx_exclusive_sum(nullptr, bytes, ia, oa, (size_t)(n), st));
  void *tmp = nullptr;
  PHX_HIP(phx_malloc(&tmp, bytes ? bytes : 16));
  PHX_HIP(phx_exclusive_sum(tmp, bytes, ia, oa, (size_t)(n), st));
  PHX_HIP(phx_exclusive_sum(tmp, bytes, ib, ob, (size_t)(n), st));
  int32_t last[2] = {0, 0};
  uint8_t lastf[2] = {0, 0};
  const phx_rb_item rb[4] = {{oa + (n - 1), 4, &last[0]}, {fa + (n - 1), 1, &lastf[0]}, {ob + (n - 1), 4, &last[1]},
                             {fb + (n - 1), 1, &lastf[1]}};
  PHX_CHECK(phx_read_back(st, rb, 4));
  PHX_HIP(phx_free(tmp));
  *ta = last[0] + (int32_t)lastf[0];
  *tb = last[1] + (int32_t)lastf[1];
  return PHX_OK;
}

int phx_system_build_structured(phx_system *s, const phx_slot_view &sv, int32_t nent) {
  phx_mesh *m = s->mesh;
  hipStream_t st = m->stream;
  const int64_t n = s->n, nu = s->nu;
  const dim3 block(256), gn((unsigned)phx_div_up(n, 256)), gu((unsigned)phx_div_up(std::max<int64_t>(nu, 1), 256));
  const int64_t n0 = m->box_n[0] + 1, n01 = n0 * (m->box_n[1] + 1);
  const int has_z = m->gdim == 3 ? 1 : 0;
  // ---- solver order: C0 rows (lattice order), other u rows, p rows
  int32_t *rank = nullptr, nc0all = 0;
  uint8_t *c0i = nullptr;   // [n] rows the stencil applies: C0 rows whose axis neighbours are C0 rows too
  PHX_HIP(phx_malloc(&rank, sizeof(int32_t) * (size_t)std::max<int64_t>(nu, 1)));
  PHX_HIP(phx_malloc(&c0i, (size_t)n));
  PHX_HIP(hipMemsetAsync(c0i, 0, (size_t)n, st));
  if (nu > 0) {
    PHX_CHECK(scan_u8(st, s->c0, rank, nu, &nc0all));
    k_c0_interior<<<gu, block, 0, st>>>(nu, s->c0, s->full_of_active, s->dof_of_vertex_u, n0, n01, has_z, c0i);
  }
  PHX_HIP(phx_malloc(&s->perm, sizeof(int32_t) * (size_t)n));
  PHX_HIP(phx_malloc(&s->iperm, sizeof(int32_t) * (size_t)n));
  k_build_perm<<<gn, block, 0, st>>>(n, nu, s->c0, rank, nc0all, s->perm, s->iperm);
  // ---- stored rows (everything the stencil does not apply): kept entries, diagonal
  int32_t *list = nullptr;
  int64_t ns = 0;
  std::vector<void *> later;   // temporaries whose last kernel is only enqueued: freed behind the last synchronisation
  PHX_CHECK(phx_select_indices(st, n, SelStored{c0i}, &list, &ns, &later));   // one host round trip (the count)
  later.push_back(rank);
  int32_t *len = nullptr, *nstruct = nullptr;
  unsigned long long *dtot = nullptr, htot[2] = {0, 0};
  PHX_HIP(phx_malloc(&len, sizeof(int32_t) * (size_t)std::max<int64_t>(ns, 1)));
  PHX_HIP(phx_malloc(&nstruct, sizeof(int32_t) * (size_t)std::max<int64_t>(ns, 1)));
  PHX_HIP(phx_malloc(&dtot, sizeof(htot)));
  PHX_HIP(hipMemsetAsync(dtot, 0, sizeof(htot), st));
  if (ns > 0) {
    PHX_REQUIRE_GRID(ns * 64, "stored-row scan");
    k_slot_row_meta<<<dim3((unsigned)phx_div_up(ns * 64, 256)), block, 0, st>>>(
        ns, list, sv, s->c0, nent, s->dof_of_vertex_u, s->dof_of_vertex_p, m->gdim, len, nstruct, s->diag);
    k_sum2_i32<<<dim3((unsigned)std::min<int64_t>(phx_div_up(ns, 256), 512)), block, 0, st>>>(ns, nstruct, len, dtot);
  }
  PHX_HIP(hipMemcpyAsync(htot, dtot, sizeof(htot), hipMemcpyDeviceToHost, st));
  PHX_HIP(hipStreamSynchronize(st));
  PHX_HIP(phx_free(dtot)); PHX_HIP(phx_free(nstruct));
  s->n_sell_rows = ns;
  s->nc0 = n - ns;
  const int per_c0 = m->gdim == 3 ? 7 : 5, struct_c0 = m->gdim == 3 ? 15 : 7;
  if (!s->rowptr) s->nnz = (int64_t)htot[0] + (int64_t)struct_c0 * s->nc0;     // structural entries (dolfinx pattern)
  s->sell_true_nnz = (int64_t)htot[1] + (int64_t)per_c0 * s->nc0;              // non-zeros one SpMV applies
  PHX_HIP(phx_malloc(&s->cscale, sizeof(double) * (size_t)n));
  k_cscale<<<gn, block, 0, st>>>(n, nu, s->perm, s->diag, s->cscale);
  // ---- SELL over the stored rows, longest first (stable: equal lengths keep their lattice order)
  s->nslices = phx_div_up(ns, SELL_S);   // slices of 16 rows, four lanes per row (k_spmv_sell)
  PHX_HIP(phx_malloc(&s->slice_ptr, sizeof(int64_t) * (size_t)(s->nslices + 1)));
  PHX_HIP(phx_malloc(&s->sell_rows, sizeof(int32_t) * (size_t)std::max<int64_t>(s->nslices * SELL_S, 1)));
  int32_t *rows_active = nullptr;   // the same list in active numbering (the slots are indexed by it)
  PHX_HIP(phx_malloc(&rows_active, sizeof(int32_t) * (size_t)std::max<int64_t>(s->nslices * SELL_S, 1)));
  if (ns > 0) {
    uint32_t *keys = nullptr, *keys2 = nullptr;
    PHX_HIP(phx_malloc(&keys, sizeof(uint32_t) * (size_t)ns));
    PHX_HIP(phx_malloc(&keys2, sizeof(uint32_t) * (size_t)ns));
    const dim3 gs((unsigned)phx_div_up(ns, 256));
    int tile = 0, tn[3] = {1, 1, 1}, key_bits = 10;
    if (const char *e = getenv("PHX_SELL_TILE")) tile = atoi(e);
    if (tile > 0) {
      for (int a = 0; a < 3; ++a) tn[a] = (int)(m->box_n[a] / tile + 1);
      const int64_t nt = (int64_t)tn[0] * tn[1] * tn[2];
      if (nt >= (1 << 21)) tile = 0;
      else while ((1ll << (key_bits - 10)) < nt) ++key_bits;
    }
    k_stored_keys<<<gs, block, 0, st>>>(ns, list, len, keys, s->full_of_active, m->nv, n0, n01, tile, tn[0], tn[1]);
    k_fill_i32<<<dim3((unsigned)phx_div_up(s->nslices * SELL_S, 256)), block, 0, st>>>(s->nslices * SELL_S, rows_active, -1, 0);
    size_t bytes = 0;
    void *tmp = nullptr;
    static const bool sort_rocprim = getenv("PHX_SORT_ROCPRIM") && atoi(getenv("PHX_SORT_ROCPRIM")) != 0;   // A/B aid
    if (key_bits <= 10 && !sort_rocprim) {
      // keys = 1023 - row length, and a row holds at most slot-capacity entries
      const int wmax = std::min(std::max(sv.W, 1), 1023);
      PHX_CHECK(phx_counting_sort_pairs(st, keys, keys2, list, rows_active, ns, (uint32_t)(1023 - wmax), wmax + 1, &later));
    } else {
      PHX_HIP(phx_sort_pairs(nullptr, bytes, keys, keys2, list, rows_active, (size_t)ns, 0, key_bits, st));
      PHX_HIP(phx_malloc(&tmp, bytes ? bytes : 16));
      PHX_HIP(phx_sort_pairs(tmp, bytes, keys, keys2, list, rows_active, (size_t)ns, 0, key_bits, st));
    }
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
    const dim3 gf((unsigned)phx_div_up(ns * 64, 256));
    k_sell_fill_slots<<<gf, block, 0, st>>>(ns, rows_active, sv, s->c0, nent, s->dof_of_vertex_u, s->dof_of_vertex_p, nu,
                                            m->gdim, s->stencil, s->diag, s->iperm, s->slice_ptr, s->sell_col, s->sell_val,
                                            s->sell_val_raw, s->full_of_active, n0, n01);
    if (ns < s->nslices * SELL_S)
      k_sell_pad_tail<<<1, 64, 0, st>>>(ns, s->nslices, s->slice_ptr, s->sell_col, s->sell_val, s->sell_val_raw);
    PHX_HIP(hipGetLastError());
    // the stored rows (near Gamma_h) hold few repeated values: no dictionary coding here
    s->sell_stream_bytes = 12 * s->sell_nnz + 8 * s->nslices + 4 * s->nslices * SELL_S;
  }
  later.push_back(len); later.push_back(list); later.push_back(rows_active);
  // ---- stencil runs over the positions of the C0 rows
  s->nseg = 0;
  s->nstencil_pos = nc0all;
  if (s->nc0 > 0 && nc0all > 0) {
    const int64_t nq = nc0all;
    uint8_t *fs = nullptr, *fe = nullptr;
    int32_t *is = nullptr, *ie = nullptr;
    PHX_HIP(phx_malloc(&fs, (size_t)nq)); PHX_HIP(phx_malloc(&fe, (size_t)nq));
    PHX_HIP(phx_malloc(&is, sizeof(int32_t) * (size_t)nq)); PHX_HIP(phx_malloc(&ie, sizeof(int32_t) * (size_t)nq));
    const dim3 gq((unsigned)phx_div_up(nq, 256));
    k_seg_flags<<<gq, block, 0, st>>>(nq, s->perm, c0i, s->full_of_active, fs, fe);
    int32_t nstart = 0, nend = 0;
    PHX_CHECK(scan_u8_pair(st, fs, is, &nstart, fe, ie, &nend, nq));   // one host round trip for both totals
    PHX_REQUIRE(nstart == nend, PHX_ERR_HIP, "stencil runs: %d starts, %d ends", nstart, nend);
    s->nseg = nstart;
    PHX_HIP(phx_malloc(&s->seg, sizeof(int32_t) * 6 * (size_t)std::max(nstart, 1)));
    k_seg_fill<<<gq, block, 0, st>>>(nq, s->perm, s->iperm, fs, fe, is, ie, s->full_of_active, s->dof_of_vertex_u,
                                     n0, n01, has_z, s->seg);
    const int64_t nw = phx_div_up(nq, 64);
    PHX_HIP(phx_malloc(&s->slice_seg, sizeof(int32_t) * 16 * (size_t)nw));
    k_slice_seg<<<dim3((unsigned)phx_div_up(nw, 256)), block, 0, st>>>(nw, s->nseg, s->seg, s->slice_seg);
    PHX_HIP(hipGetLastError());
    later.push_back(fs); later.push_back(fe); later.push_back(is); later.push_back(ie);
    // planes too large for the L2 window of a contiguous walk: per-plane eighths (see k_stmap_build)
    const int nzp = has_z ? (int)(m->box_n[2] + 1) : 0;
    if (has_z && m->stencil_plane_rows > 0 && nq / nzp >= m->stencil_plane_rows) {
      int32_t *zstart = nullptr;
      PHX_HIP(phx_malloc(&zstart, sizeof(int32_t) * (size_t)(nzp + 1)));
      k_plane_starts<<<gq, block, 0, st>>>(nq, s->perm, s->full_of_active, n01, nzp, zstart);
      s->st_chunk = nw / 128 + 2 * (int64_t)nzp + 2;
      PHX_HIP(phx_malloc(&s->st_map, sizeof(int32_t) * 2 * 8 * (size_t)s->st_chunk));
      k_stmap_build<<<8, 1024, 0, st>>>(nzp, nq, zstart, s->st_chunk, s->st_map);
      PHX_HIP(hipGetLastError());
      later.push_back(zstart);
      s->sell_stream_bytes += 8 * 8 * s->st_chunk;
    }
    s->sell_stream_bytes += 24 * (int64_t)s->nseg + 64 * nw;
  }
  PHX_HIP(hipGetLastError());
  PHX_HIP(hipStreamSynchronize(st));
  for (void *q : later) PHX_HIP(phx_free(q));
  PHX_HIP(phx_free(c0i));
  // solver workspace: 9 vectors + scalars
  PHX_HIP(phx_malloc(&s->work, sizeof(double) * (size_t)n * 9));
  PHX_HIP(phx_malloc(&s->scal, sizeof(double) * PHX_SCAL_DOUBLES));
  PHX_CHECK(phx_mesh_pinned_scalars(s->mesh, &s->scal_h));
  return PHX_OK;
}

// Workspace of a system without any active DoF (a slab that does not touch the domain): it takes part in
// every collective of a multi-GPU solve with zero contributions.
int phx_system_build_empty(phx_system *s) {
  s->n = 0; s->nu = 0; s->nnz = 0; s->nslices = 0; s->sell_nnz = 0; s->sell_true_nnz = 0;
  PHX_HIP(phx_malloc(&s->work, sizeof(double) * 16));
  PHX_HIP(phx_malloc(&s->scal, sizeof(double) * PHX_SCAL_DOUBLES));
  PHX_HIP(hipMemsetAsync(s->scal, 0, sizeof(double) * PHX_SCAL_DOUBLES, s->mesh->stream));
  PHX_CHECK(phx_mesh_pinned_scalars(s->mesh, &s->scal_h));
  PHX_HIP(hipStreamSynchronize(s->mesh->stream));
  return PHX_OK;
}

// ---------------------------------------------------------------------------------------------
// SpMV: one wavefront per slice, 4 slices per 256-thread block.  DOTS > 0 fuses dot products of
// the result with up to two vectors into the same pass (block reduction + one f64 atomic each).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// PHX_OPT_DETERMINISTIC: where the blocks of a dot-product kernel leave their partial sums (p0 == nullptr: atomics)
struct DotPart { double *p0, *p1; };
// fixed-order sum of nb partials per quantity into slot 0 of its slot set (the other slots stay zero: no atomics ran)
__global__ void __launch_bounds__(1024) k_fold_partials(int64_t nb, DotPart part, double *out0, double *out1) {
  __shared__ double red[1024];
  for (int q = 0; q < 2; ++q) {
    const double *p = q == 0 ? part.p0 : part.p1;
    double *out = q == 0 ? out0 : out1;
    if (!p || !out) continue;
    double a = 0.0;
    for (int64_t i = threadIdx.x; i < nb; i += 1024) a += p[i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) *out = red[0];
    __syncthreads();
  }
}

struct StencilArgs {
  int64_t nu;             // solver positions the stencil slices cover (the C0 rows)
  const int32_t *seg;     // [nseg][6]
  int nseg;
  const int32_t *srec;    // [ceil(nu / 64)][16] slice records
  const double *st;       // {diag, x, y, z entries} of a C0 row
  int64_t chunk;          // stencil blocks per XCD
  const int32_t *map;     // nullable: [8][chunk][2] {first slice, slices} per block (planes beyond the L2 window)
};

template <int DOTS>
__global__ void __launch_bounds__(256)
k_spmv_sell(int64_t n, int64_t nslices, const int64_t *__restrict__ slice_ptr,
            const int32_t *__restrict__ scol, const double *__restrict__ sval,
            const double *__restrict__ x, double *__restrict__ y,
            const uint8_t *__restrict__ own, const double *__restrict__ d0,
            double *__restrict__ out0, double *__restrict__ out1, int xcd_group,
            const uint8_t *__restrict__ kind, const int32_t *__restrict__ rows, int64_t nb_sell, StencilArgs sa,
            const uint8_t *__restrict__ bnd, DotPart part) {
  // bnd (nullable, multi-GPU): rows flagged here reference halo entries that are still in flight; this launch
  // leaves them (no store, no dot-product share) to k_spmv_bnd, which runs after the halo has been unpacked
  __shared__ double vi_dict[4][VI_MAX];
  const int lane = threadIdx.x & 63;
  // Optional XCD-aware block -> slice map (PHX_OPT_SPMV_XCD_GROUP): blocks b and b+8 share an XCD and
  // its L2; inside every run of 8*G consecutive block ids XCD k is handed G CONSECUTIVE slice groups,
  // so the x entries its rows gather stay in one L2.  Speed only: any placement is correct.
  int64_t bid = blockIdx.x;
  if (xcd_group < 0) {
    // contiguous eighths (structured systems: nb_sell is a multiple of 8): XCD k walks the k-th eighth of the stored rows
    if (bid < nb_sell) bid = (bid & 7) * (nb_sell >> 3) + (bid >> 3);
  } else if (xcd_group > 0) {
    const int64_t super = 8 * (int64_t)xcd_group, sg = bid / super;
    if ((sg + 1) * super <= nb_sell) {
      const int64_t rem = bid - sg * super;
      bid = sg * super + (rem & 7) * xcd_group + (rem >> 3);
    }
  }
  const int64_t s = bid * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
  double acc = 0.0;
  int64_t row = -1;
  double p0 = 0.0, p1 = 0.0;   // this lane's share of (y, d0) and (y, y)
#define PHX_DOT_ACC(a_, r_) \
  do { if (DOTS > 0) { p0 = __builtin_fma((a_), d0[(r_)], p0); if (DOTS > 1) p1 = __builtin_fma((a_), (a_), p1); } } while (0)
  if (blockIdx.x >= nb_sell) {
    // ---- rows of C0 of a structured system (blocks behind the SELL blocks of the same launch):
    // y_i = d x_i + cx (x_{i-1} + x_{i+1}) + cy (x_{i+oy+} + x_{i+oy-}) + cz (x_{i+oz+} + x_{i+oz-}), the row of the
    // lattice Laplacian, over runs of consecutive rows of one x line (`seg`: {first, end, offsets of the +y, -y,
    // +z, -z neighbour rows}, constant along a run because active rows are numbered in lattice order).  A
    // wavefront takes FOUR slices of 64 consecutive u rows (one slice per wave is latency bound: 21.5 us for
    // 2.4e6 rows); a slice record carries its first two runs inline (one wave-uniform 64-byte load), further
    // runs are looked up in `seg`.  A lane whose row lies in no run belongs to a stored row and leaves it to the
    // SELL blocks.  Nothing of the matrix is read: seven coalesced x streams per slice.
    constexpr int U = 4;
    const int64_t bl = blockIdx.x - nb_sell;                       // nb_sell is a multiple of 8
    const int64_t sb_ = (bl & 7) * sa.chunk + (bl >> 3);           // XCD (bl % 8) takes the blocks of its eighth
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int64_t w0 = (sb_ * (int64_t)(blockDim.x >> 6) + wave) * U;
    int wn = U;                                                    // slices of this wave
    if (sa.map) {
      const int32_t *e = sa.map + 2 * sb_;                         // block-uniform
      w0 = (int64_t)e[0] + wave * U;
      wn = min(U, max(0, e[1] - wave * U));
    }
    bool live[U];
    int32_t oyp[U], oym[U], ozp[U], ozm[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t w = w0 + u, r = w * 64 + lane;
      live[u] = false;
      oyp[u] = oym[u] = ozp[u] = ozm[u] = 0;
      if (u < wn && w * 64 < sa.nu) {
        // {runs in this slice (3: more than two), first run, run A[6], run B[6]}: wave-uniform, scalar loads
        const int32_t *rec = sa.srec + 16 * w;
        const int cnt = rec[0];
        const bool inA = cnt >= 1 && r >= rec[2] && r < rec[3];
        const bool inB = cnt >= 2 && r >= rec[8] && r < rec[9];
        live[u] = inA || inB;
        oyp[u] = inA ? rec[4] : rec[10];
        oym[u] = inA ? rec[5] : rec[11];
        ozp[u] = inA ? rec[6] : rec[12];
        ozm[u] = inA ? rec[7] : rec[13];
        if (!live[u] && cnt >= 3) {
          int k = rec[1] + 2;
          while (k < sa.nseg && sa.seg[6 * (int64_t)k + 1] <= r) ++k;
          if (k < sa.nseg && r >= sa.seg[6 * (int64_t)k]) {
            const int32_t *sg = sa.seg + 6 * (int64_t)k;
            live[u] = true;
            oyp[u] = sg[2]; oym[u] = sg[3]; ozp[u] = sg[4]; ozm[u] = sg[5];
          }
        }
        if (r >= sa.nu) live[u] = false;
        if (bnd && live[u] && bnd[r]) live[u] = false;
      }
    }
    double xv[U][7];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = (w0 + u) * 64 + lane;
      if (live[u]) {
        xv[u][0] = x[r]; xv[u][1] = x[r - 1]; xv[u][2] = x[r + 1];
        xv[u][3] = x[r + oyp[u]]; xv[u][4] = x[r + oym[u]];
        const bool hz = ozp[u] != 0;
        xv[u][5] = hz ? x[r + ozp[u]] : 0.0; xv[u][6] = hz ? x[r + ozm[u]] : 0.0;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = (w0 + u) * 64 + lane;
      if (live[u]) {
        double a2 = __builtin_fma(sa.st[1], xv[u][1] + xv[u][2], sa.st[0] * xv[u][0]);
        a2 = __builtin_fma(sa.st[2], xv[u][3] + xv[u][4], a2);
        a2 = __builtin_fma(sa.st[3], xv[u][5] + xv[u][6], a2);
        if (own && !own[r]) a2 = 0.0;
        y[r] = a2;
        PHX_DOT_ACC(a2, r);
      }
    }
  } else if (rows) {
    // ---- stored rows of a structured system: SELL-16 with FOUR lanes per row.  Entry k of row r of a slice sits
    // at base + 16 k + r, so with lane = 16 (k mod 4) + r every wave-instruction still reads 64 consecutive
    // entries; the four partial sums of a row meet in two shuffles.  The stored rows are few and long (4.9e5 rows
    // of 24-60 entries at 256^3): one lane per row left 7600 wavefronts for the whole chip and the loop latency
    // bound (55 us); a quarter of the trip count and four times the waves.
    if (s < nslices) {
      const int64_t base = slice_ptr[s];
      const int trips = (int)((slice_ptr[s + 1] - base) >> 6);   // width / 4
      const int32_t *c = scol + base + lane;
      const double *v = sval + base + lane;
      // four trips (16 entries of a row) per round, their column / value loads in flight before the first gather.
      // Measured at 256^3 (SELL part alone): 4 trips 44 us; 16 trips with clamped indices 81 us -- the rows average
      // 6.5 trips and the kernel is bound by vector-memory instructions (gathers of 64 scattered doubles), not by
      // latency: redundant clamped loads cost their full price.
      int j = 0;
      for (; j + 4 <= trips; j += 4) {
        int32_t cc[4];
        double vv[4], xs[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { cc[q] = NT_LOAD(&c[(j + q) * 64]); vv[q] = NT_LOAD(&v[(j + q) * 64]); }
#pragma unroll
        for (int q = 0; q < 4; ++q) xs[q] = x[cc[q]];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_fma(vv[q], xs[q], acc);
      }
      for (; j < trips; ++j) acc = __builtin_fma(NT_LOAD(&v[j * 64]), x[NT_LOAD(&c[j * 64])], acc);
      acc += __shfl_xor(acc, 16);
      acc += __shfl_xor(acc, 32);
      const int32_t rr = rows[s * 16 + (lane & 15)];
      if (lane < 16 && rr >= 0 && !(bnd && bnd[rr])) {
        if (own && !own[rr]) acc = 0.0;
        y[rr] = acc;
        PHX_DOT_ACC(acc, rr);
      }
    }
  } else if (s < nslices) {
    const int64_t base = slice_ptr[s];
    const int width = (int)((slice_ptr[s + 1] - base) >> 6);
    const int32_t *c = scol + base + lane;
    const double *v = sval + base + lane;
    const int nd = kind[s];
    int k = 0;
    if (nd) {
      // value-indexed slice (k_sell_index): one code byte per entry; dictionaries of up to 64 doubles
      // live one per lane (looked up with a cross-lane read), larger ones in this wave's LDS strip
      const int chunks = (nd + SELL_C - 1) >> 6;
      const uint32_t *cw = reinterpret_cast<const uint32_t *>(sval + base + chunks * SELL_C) + lane;
      if (chunks == 1) {
        const double dreg = lane < nd ? v[0] : 0.0;
        // eight entries per trip (the 7-point interior rows in one): all column and code loads are
        // issued before the first gather; indices past the row end are clamped and their terms skipped
        const int klast = width - 1;
        for (; k < width; k += 8) {
          const uint32_t wa = NT_LOAD(&cw[(k >> 2) * SELL_C]);
          const uint32_t wb = NT_LOAD(&cw[(min(k + 4, klast) >> 2) * SELL_C]);
          int32_t cc[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) cc[j] = NT_LOAD(&c[min(k + j, klast) * SELL_C]);
          double xs[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) xs[j] = x[cc[j]];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const uint32_t code = ((j < 4 ? wa : wb) >> (8 * (j & 3))) & 255u;
            const double dv = __shfl(dreg, (int)code);
            acc = k + j < width ? __builtin_fma(dv, xs[j], acc) : acc;
          }
        }
      } else {
        double *sd = vi_dict[threadIdx.x >> 6];
        for (int j = lane; j < nd; j += SELL_C) sd[j] = sval[base + j];
        __builtin_amdgcn_wave_barrier();
        for (; k < width; k += 4) {
          uint32_t w4 = NT_LOAD(&cw[(k >> 2) * SELL_C]);
          const int ke = min(k + 4, width);
          for (int kk = k; kk < ke; ++kk, w4 >>= 8) acc = __builtin_fma(sd[w4 & 255u], x[NT_LOAD(&c[kk * SELL_C])], acc);
        }
        k = width;
      }
    }
    // raw slice.  The matrix is streamed once: non-temporal loads keep it from evicting x out of
    // L2 / MALL; eight entries per trip so that eight gathers are in flight per lane
    for (; k + 8 <= width; k += 8) {
      int32_t cc[8];
      double vv[8], xs[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { cc[j] = NT_LOAD(&c[(k + j) * SELL_C]); vv[j] = NT_LOAD(&v[(k + j) * SELL_C]); }
#pragma unroll
      for (int j = 0; j < 8; ++j) xs[j] = x[cc[j]];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc = __builtin_fma(vv[j], xs[j], acc);
    }
    for (; k + 4 <= width; k += 4) {
      const int32_t c0 = NT_LOAD(&c[(k + 0) * SELL_C]), c1 = NT_LOAD(&c[(k + 1) * SELL_C]);
      const int32_t c2 = NT_LOAD(&c[(k + 2) * SELL_C]), c3 = NT_LOAD(&c[(k + 3) * SELL_C]);
      const double v0 = NT_LOAD(&v[(k + 0) * SELL_C]), v1 = NT_LOAD(&v[(k + 1) * SELL_C]);
      const double v2 = NT_LOAD(&v[(k + 2) * SELL_C]), v3 = NT_LOAD(&v[(k + 3) * SELL_C]);
      acc = __builtin_fma(v0, x[c0], acc);
      acc = __builtin_fma(v1, x[c1], acc);
      acc = __builtin_fma(v2, x[c2], acc);
      acc = __builtin_fma(v3, x[c3], acc);
    }
    for (; k < width; ++k) acc = __builtin_fma(NT_LOAD(&v[k * SELL_C]), x[NT_LOAD(&c[k * SELL_C])], acc);
    // structured systems: the slices run over a list of stored rows (-1: padding of the last slice)
    row = s * SELL_C + lane;
    if (row < n && !(bnd && bnd[row])) {
      if (own && !own[row]) acc = 0.0;  // ghost rows stay zero; the halo exchange refreshes them
      y[row] = acc;
      PHX_DOT_ACC(acc, row);
    }
  }
#undef PHX_DOT_ACC
  if (DOTS > 0) {
    // DOTS == 1: out0 += (y, d0);  DOTS == 2: also out1 += (y, y)
    __shared__ double red[2][4];
    p0 = wave_sum(p0);
    if (DOTS > 1) p1 = wave_sum(p1);
    const int w = threadIdx.x >> 6;
    if (lane == 0) { red[0][w] = p0; red[1][w] = p1; }
    __syncthreads();
    if (threadIdx.x == 0) {
      const double s0 = red[0][0] + red[0][1] + red[0][2] + red[0][3], s1 = red[1][0] + red[1][1] + red[1][2] + red[1][3];
      if (part.p0) {   // PHX_OPT_DETERMINISTIC: one entry per block, summed in a fixed order by k_fold_partials
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


// ---------------------------------------------------------------------------------------------
// Multi-GPU overlap: rows whose columns include a halo entry (`isg[pos]` = 1: some neighbour sends this
// position).  k_spmv_flag walks the COLUMN structure of k_spmv_sell (stencil runs, SELL-16 row list, SELL-64
// slices -- the value streams are not touched) and writes bnd[row]; with `rec` it also appends one record per
// flagged row (order arbitrary: the dot products are summed through atomics anyway).  k_spmv_bnd applies the
// recorded rows, one lane per row, after the halo has been unpacked.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_spmv_flag(int64_t n, int64_t nslices, const int64_t *__restrict__ slice_ptr, const int32_t *__restrict__ scol,
            const uint8_t *__restrict__ isg, const uint8_t *__restrict__ own, const int32_t *__restrict__ rows,
            int64_t nb_sell, StencilArgs sa, uint8_t *__restrict__ bnd, int32_t *__restrict__ rec,
            unsigned long long *__restrict__ counter) {
  const int lane = threadIdx.x & 63;
  const int64_t s = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (blockIdx.x >= nb_sell) {
    const int64_t w = (blockIdx.x - nb_sell) * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t r = w * 64 + lane;
    if (w * 64 >= sa.nu || r >= sa.nu) return;
    const int32_t *rc = sa.srec + 16 * w;
    const int cnt = rc[0];
    const bool inA = cnt >= 1 && r >= rc[2] && r < rc[3];
    const bool inB = cnt >= 2 && r >= rc[8] && r < rc[9];
    bool live = inA || inB;
    int32_t o[4] = {inA ? rc[4] : rc[10], inA ? rc[5] : rc[11], inA ? rc[6] : rc[12], inA ? rc[7] : rc[13]};
    if (!live && cnt >= 3) {
      int k = rc[1] + 2;
      while (k < sa.nseg && sa.seg[6 * (int64_t)k + 1] <= r) ++k;
      if (k < sa.nseg && r >= sa.seg[6 * (int64_t)k]) {
        const int32_t *sg = sa.seg + 6 * (int64_t)k;
        live = true;
        o[0] = sg[2]; o[1] = sg[3]; o[2] = sg[4]; o[3] = sg[5];
      }
    }
    if (!live || (own && !own[r])) return;
    int g = isg[r - 1] | isg[r + 1] | isg[r + o[0]] | isg[r + o[1]];
    if (o[2] != 0) g |= isg[r + o[2]] | isg[r + o[3]];
    if (!g) return;
    bnd[r] = 1;
    const unsigned long long at = atomicAdd(counter, 1ull);
    if (rec) { int32_t *q = rec + 6 * at; q[0] = (int32_t)r; q[1] = 0; q[2] = o[0]; q[3] = o[1]; q[4] = o[2]; q[5] = o[3]; }
  } else if (rows) {
    if (s >= nslices) return;
    const int64_t base = slice_ptr[s];
    const int trips = (int)((slice_ptr[s + 1] - base) >> 6);
    const int32_t *c = scol + base + lane;
    int g = 0;
    for (int j = 0; j < trips; ++j) g |= isg[c[j * 64]];
    g |= __shfl_xor(g, 16);
    g |= __shfl_xor(g, 32);
    const int32_t rr = rows[s * 16 + (lane & 15)];
    if (lane < 16 && rr >= 0 && g && !(own && !own[rr])) {
      bnd[rr] = 1;
      const unsigned long long at = atomicAdd(counter, 1ull);
      if (rec) { int32_t *q = rec + 6 * at; q[0] = rr; q[1] = 1; q[2] = (int32_t)s; q[3] = lane; q[4] = 0; q[5] = 0; }
    }
  } else if (s < nslices) {
    const int64_t base = slice_ptr[s];
    const int width = (int)((slice_ptr[s + 1] - base) >> 6);
    const int32_t *c = scol + base + lane;
    int g = 0;
    for (int k = 0; k < width; ++k) g |= isg[c[k * SELL_C]];
    const int64_t row = s * SELL_C + lane;
    if (row < n && g && !(own && !own[row])) {
      bnd[row] = 1;
      const unsigned long long at = atomicAdd(counter, 1ull);
      if (rec) { int32_t *q = rec + 6 * at; q[0] = (int32_t)row; q[1] = 2; q[2] = (int32_t)s; q[3] = lane; q[4] = 0; q[5] = 0; }
    }
  }
}

template <int DOTS>
__global__ void __launch_bounds__(256)
k_spmv_bnd(int64_t nrec, const int32_t *__restrict__ rec, const int64_t *__restrict__ slice_ptr,
           const int32_t *__restrict__ scol, const double *__restrict__ sval, const uint8_t *__restrict__ kind,
           const double *__restrict__ st, const double *__restrict__ x, double *__restrict__ y,
           const double *__restrict__ d0, double *__restrict__ out0, double *__restrict__ out1, DotPart part) {
  double p0 = 0.0, p1 = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nrec; i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t *q = rec + 6 * i;
    const int64_t r = q[0];
    double acc = 0.0;
    if (q[1] == 0) {
      acc = __builtin_fma(st[1], x[r - 1] + x[r + 1], st[0] * x[r]);
      acc = __builtin_fma(st[2], x[r + q[2]] + x[r + q[3]], acc);
      if (q[4] != 0) acc = __builtin_fma(st[3], x[r + q[4]] + x[r + q[5]], acc);
    } else if (q[1] == 1) {
      const int64_t base = slice_ptr[q[2]];
      const int width = (int)((slice_ptr[q[2] + 1] - base) >> 4);
      // the four lanes of a row hold entries k = 4 j + (lane >> 4) at base + 64 j + lane: entry k at base + 16 k + (lane & 15)
      const int64_t b0 = base + (q[3] & 15);
      for (int k = 0; k < width; ++k) acc = __builtin_fma(sval[b0 + 16 * (int64_t)k], x[scol[b0 + 16 * (int64_t)k]], acc);
    } else {
      const int64_t base = slice_ptr[q[2]];
      const int width = (int)((slice_ptr[q[2] + 1] - base) >> 6);
      const int lane = q[3];
      const int nd = kind[q[2]];
      const int32_t *c = scol + base + lane;
      if (nd) {
        const int chunks = (nd + SELL_C - 1) >> 6;
        const uint32_t *cw = reinterpret_cast<const uint32_t *>(sval + base + chunks * SELL_C) + lane;
        for (int k = 0; k < width; ++k) {
          const uint32_t code = (cw[(k >> 2) * SELL_C] >> (8 * (k & 3))) & 255u;
          acc = __builtin_fma(sval[base + code], x[c[k * SELL_C]], acc);
        }
      } else {
        const double *v = sval + base + lane;
        for (int k = 0; k < width; ++k) acc = __builtin_fma(v[k * SELL_C], x[c[k * SELL_C]], acc);
      }
    }
    y[r] = acc;
    if (DOTS > 0) { p0 = __builtin_fma(acc, d0[r], p0); if (DOTS > 1) p1 = __builtin_fma(acc, acc, p1); }
  }
  if (DOTS > 0) {
    __shared__ double red[2][4];
    p0 = wave_sum(p0);
    if (DOTS > 1) p1 = wave_sum(p1);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][w] = p0; red[1][w] = p1; }
    __syncthreads();
    if (threadIdx.x == 0) {
      const double s0 = red[0][0] + red[0][1] + red[0][2] + red[0][3], s1 = red[1][0] + red[1][1] + red[1][2] + red[1][3];
      if (part.p0) {   // PHX_OPT_DETERMINISTIC: one entry per block, summed in a fixed order by k_fold_partials
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

#include "phx_spmv_p2s.inc.hip"

// ---------------------------------------------------------------------------------------------
// BiCGStab.  All scalars live on the device: S[0..7] is solver state, R = S+8 holds the dot
// products of the running iteration (local partial sums; a multi-GPU driver all-reduces R between
// phases).  Every kernel derives alpha / omega / beta from them, so an iteration is a fixed
// launch sequence with no host round trip.  `own` (nullable) masks the rows this rank owns:
// ghost rows are kept at zero and refreshed by the halo exchange of the driver.
// ---------------------------------------------------------------------------------------------
enum { S_RHO = 0, S_ALPHA = 1, S_OMEGA = 2, S_BB = 3, S_RR = 4, S_RESTARTS = 5, S_RHO_NEXT = 6,
       S_MODE = 7 };
enum { R_RV = 0, R_TS = 1, R_TT = 2, R_SS = 3, R_RHO = 4, R_RR = 5 };
#define R_OFF 8
#define S_RR0 14   // S[14 + parity]: (r, r) at the last restart (kr_restart)
// S_MODE = 1: every dot product is folded into R by k_reduce_slots (and all-reduced by a
// multi-GPU driver) before its consumer runs.  S_MODE = 0 (native single-GPU loop): consumers
// fold the 64 slots themselves -- no reduce / roll launches; the slot sets alternate with the
// iteration parity so that a set is cleared while nobody reads it.

__host__ __device__ __forceinline__ double *slot_base(double *S, int par, int q) {
  return S + P_OFF + ((par * 8 + q) * NSLOT) * SLOT_STRIDE;
}

__device__ __forceinline__ void block_atomic_sum(double v, double *out, double *part = nullptr) {
  __shared__ double red[4];
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    if (part) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];   // PHX_OPT_DETERMINISTIC
    else unsafeAtomicAdd(out + (blockIdx.x & (NSLOT - 1)) * SLOT_STRIDE, red[0] + red[1] + red[2] + red[3]);
  }
  __syncthreads();
}

// value of dot product q of the running iteration.  Native loop: every wavefront folds the 64 slots
// itself -- lane k loads slot k, a fixed butterfly adds them -- so all waves of all blocks obtain the
// same bits from one load per lane (a serial 64-term sum per thread cost ~4 us per kernel).
__device__ __forceinline__ double dotv(const double *S, int par, int q) {
  if (S[S_MODE] != 0.0) return S[R_OFF + q];
  const double *p = S + P_OFF + ((par * 8 + q) * NSLOT) * SLOT_STRIDE;
  return wave_sum(p[(threadIdx.x & (NSLOT - 1)) * SLOT_STRIDE]);
}

// fold the slots of quantities q0..q0+nq-1 into R; `clear` also zeroes them (one wave)
__global__ void k_reduce_slots(double *S, int par, int q0, int nq, int clear) {
  const int lane = threadIdx.x;
  for (int q = q0; q < q0 + nq; ++q) {
    double *p = slot_base(S, par, q) + lane * SLOT_STRIDE;
    double v = *p;
    if (clear) *p = 0.0;
    v = wave_sum(v);
    if (lane == 0) S[R_OFF + q] = v;
  }
}

#define GRID_STRIDE(i, n) \
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

// The box preconditioner is the identity on the rows outside the u block (p rows): the kernels that produce p and s
// write those entries of phat / shat themselves (`RestOut`), so an application needs no copy pass over them (a list
// copy of 5 us + a launch, twice per iteration).  perm == nullptr: the u rows come first in solver order (structured
// systems), otherwise row i belongs to the u block iff perm[i] < nu.
struct RestOut {
  double *hat;           // phat or shat; nullptr: nothing to do (no box preconditioner, or hat aliases the vector)
  const int32_t *perm;
  int64_t nu;
  __device__ __forceinline__ void put(int64_t i, double v) const {
    if (hat && (perm ? perm[i] >= nu : i >= nu)) hat[i] = v;
  }
};

// r = rhat = p = own ? b : 0; y = 0; R_RHO += (b,b)
__global__ void __launch_bounds__(256)
k_kr_begin(int64_t n, const int32_t *__restrict__ perm, const double *__restrict__ rhs,
           const uint8_t *__restrict__ own, double *__restrict__ b, double *__restrict__ r,
           double *__restrict__ rhat, double *__restrict__ p, double *__restrict__ y,
           double *__restrict__ S, RestOut ro, DotPart part) {
  double acc = 0.0;
  GRID_STRIDE(i, n) {
    const bool mine = !own || own[i];
    const double bi = mine ? rhs[perm[i]] : 0.0;
    b[i] = bi; r[i] = bi; rhat[i] = bi; p[i] = bi; y[i] = 0.0;
    ro.put(i, bi);
    acc += bi * bi;
  }
  block_atomic_sum(acc, slot_base(S, 0, R_RHO), part.p0);
}

// Multi-GPU: the preconditioner must be chosen by ALL ranks together.  Phase 0 leaves this rank's veto (1: the
// box preconditioner is configured out or cannot be built here, 0: built, or nothing to build because the rank
// owns no u DoF) in R[R_RR], the driver all-reduces it with (b, b), and every rank that reads a sum > 0 after
// phase 1 drops its preconditioner (phx_krylov_precond_disable).
__global__ void k_set_scalar(double *p, double v) { *p = v; }

// after the (optional) all-reduce of R: rho = bb = R_RHO
__global__ void k_kr_begin2(double *S, int mode) {
  S[S_MODE] = mode ? 1.0 : 0.0;
  S[S_RHO] = S[R_OFF + R_RHO];
  S[S_RHO_NEXT] = S[R_OFF + R_RHO];
  S[S_BB] = S[R_OFF + R_RHO];
  S[S_RR] = S[R_OFF + R_RHO];
  S[S_RR0] = S[S_RR0 + 1] = S[R_OFF + R_RHO];
}

// s = r - alpha v, alpha = rho/(rhat,v)
__global__ void __launch_bounds__(256)
k_update_s(int64_t n, int par, const uint8_t *__restrict__ own, const double *__restrict__ r,
           const double *__restrict__ v, double *__restrict__ sv, double *__restrict__ S, RestOut ro) {
  const double rho = S[S_RHO_NEXT];
  const double alpha = rho / dotv(S, par, R_RV);
  GRID_STRIDE(i, n) {
    const bool mine = !own || own[i];
    const double si = mine ? r[i] - alpha * v[i] : 0.0;
    sv[i] = si;
    ro.put(i, si);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { S[S_ALPHA] = alpha; S[S_RHO] = rho; }
}

// x += alpha phat + omega shat;  r = s - omega t;  accumulates rho_next = (rhat,r) and (r,r)
__global__ void __launch_bounds__(256)
k_update_xr(int64_t n, int par, const uint8_t *__restrict__ own, const double *__restrict__ phat,
            const double *__restrict__ shat, const double *__restrict__ sv, const double *__restrict__ t,
            const double *__restrict__ rhat, double *__restrict__ x, double *__restrict__ r,
            double *__restrict__ S, DotPart part) {
  const double alpha = S[S_ALPHA];
  const double omega = dotv(S, par, R_TS) / dotv(S, par, R_TT);
  double a0 = 0.0, a1 = 0.0;
  GRID_STRIDE(i, n) {
    const bool mine = !own || own[i];
    const double si = sv[i];
    double ri = 0.0;
    if (mine) {
      x[i] += alpha * phat[i] + omega * shat[i];
      ri = si - omega * t[i];
    }
    r[i] = ri;
    a0 += rhat[i] * ri;
    a1 += ri * ri;
  }
  block_atomic_sum(a0, slot_base(S, par, R_RHO), part.p0);
  block_atomic_sum(a1, slot_base(S, par, R_RR), part.p1);
  if (blockIdx.x == 0 && threadIdx.x == 0) S[S_OMEGA] = omega;
}

// Breakdown guard: when (rhat, r) has collapsed relative to (r, r) (or a scalar went non-finite)
// the iteration is RESTARTED from the current residual: rhat = p = r, rho = (r, r).  Every thread
// evaluates the same predicate from the same device scalars.
// drop2 > 0 (f32 lattice, native loop): ALSO restart whenever (r, r) has fallen by the factor drop2 since the last
// restart.  With right preconditioning r = b - A x holds for whatever phat / shat were used, so a restart from r is
// a step of iterative refinement in f64: the rounding noise of the f32 transforms (a slightly non-linear M^-1,
// which BiCGStab's short recurrences do not tolerate over many iterations) only ever acts over one short cycle.
// S[14 + par] holds (r, r) at the last restart for the iteration of parity par (read here, the other one written).
__device__ __forceinline__ bool kr_restart(const double *S, double rho_new, double rr, double drop2 = 0.0, int par = 0) {
  const double beta = (rho_new / S[S_RHO]) * (S[S_ALPHA] / S[S_OMEGA]);
  if (drop2 > 0.0 && S[S_MODE] == 0.0 && rr <= drop2 * S[S_RR0 + par]) return true;
  return !(fabs(beta) <= 1.0e300) || !(fabs(rho_new) > 1.0e-14 * rr);
}

// p = r + beta (p - omega v), beta = (rho_next/rho)(alpha/omega); rolls rho for the next
// iteration and clears the slot set the next iteration accumulates into
__global__ void __launch_bounds__(256)
k_update_p(int64_t n, int par, const uint8_t *__restrict__ own, const double *__restrict__ r,
           const double *__restrict__ v, double *__restrict__ p, double *__restrict__ rhat,
           double *__restrict__ S, double drop2, RestOut ro) {
  const double rho_new = dotv(S, par, R_RHO), rr = dotv(S, par, R_RR);
  const bool restart = kr_restart(S, rho_new, rr, drop2, par);
  if (blockIdx.x == 0) {
    if (threadIdx.x == 0) {
      S[S_RHO_NEXT] = restart ? rr : rho_new;
      S[S_RR] = rr;
      if (restart) S[S_RESTARTS] += 1.0;
      if (S[S_MODE] == 0.0) S[S_RR0 + (par ^ 1)] = restart ? rr : S[S_RR0 + par];
    }
    double *nxt = S + P_OFF + ((par ^ 1) * 8 * NSLOT) * SLOT_STRIDE;
    for (int k = threadIdx.x; k < 8 * NSLOT; k += blockDim.x) nxt[k * SLOT_STRIDE] = 0.0;
  }
  if (restart) {
    GRID_STRIDE(i, n) { const double ri = r[i]; p[i] = ri; rhat[i] = ri; ro.put(i, ri); }
    return;
  }
  const double beta = (rho_new / S[S_RHO]) * (S[S_ALPHA] / S[S_OMEGA]);
  const double omega = S[S_OMEGA];
  GRID_STRIDE(i, n) {
    const bool mine = !own || own[i];
    const double pi = mine ? r[i] + beta * (p[i] - omega * v[i]) : 0.0;
    p[i] = pi;
    ro.put(i, pi);
  }
}

__global__ void k_gather(int64_t n, const int32_t *__restrict__ perm, const double *__restrict__ in,
                         double *__restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[perm[i]];
}

// solution back to FULL numbering: x_full[full_of_active[perm[pos]]] = y[pos] / diag[perm[pos]]
__global__ void k_scatter_solution(int64_t n, const int32_t *__restrict__ perm,
                                   const int64_t *__restrict__ full_of_active,
                                   const double *__restrict__ diag, const double *__restrict__ cscale,
                                   const double *__restrict__ y, double *__restrict__ xfull,
                                   const int32_t *__restrict__ out_vertex, int64_t nent) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t r = perm[i];
  int64_t f = full_of_active[r];
  if (out_vertex) { const int64_t blk = f / nent; f = blk * nent + out_vertex[f - blk * nent]; }   // caller's numbering
  // x = S y: the column scaling folded into the stored values (1 / diag, or 1 for unscaled u columns)
  xfull[f] = cscale ? y[i] * cscale[i] : y[i] / diag[r];
}

static inline dim3 vec_grid(int64_t n) { return dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(phx_div_up(n, 256), 2048))); }

// PHX_OPT_DETERMINISTIC: `nb` entries per quantity for the blocks of the next dot-product launch, behind the ones handed
// out since the last fold (a product may take several launches: SELL + stencil blocks, interior + halo rows)
static int det_part(phx_system *s, int64_t nb, DotPart *out) {
  out->p0 = out->p1 = nullptr;
  if (!s->mesh->deterministic) return PHX_OK;
  if (!s->dpart) {
    s->dpart_cap = phx_div_up(s->nslices, 4) + phx_div_up(s->nstencil_pos, 1024) + 8 * s->st_chunk + 16384;
    PHX_HIP(phx_malloc(&s->dpart, sizeof(double) * 2 * (size_t)s->dpart_cap));
    s->dpart_used = 0;
  }
  PHX_REQUIRE(s->dpart_used + nb <= s->dpart_cap, PHX_ERR_HIP, "deterministic dot products: %lld partial sums exceed the buffer",
              (long long)(s->dpart_used + nb));
  out->p0 = s->dpart + s->dpart_used;
  out->p1 = s->dpart + s->dpart_cap + s->dpart_used;
  s->dpart_used += nb;
  return PHX_OK;
}
static int det_fold(phx_system *s, double *out0, double *out1) {
  if (!s->mesh->deterministic || s->dpart_used == 0) return PHX_OK;
  k_fold_partials<<<1, 1024, 0, s->mesh->stream>>>(s->dpart_used, DotPart{s->dpart, s->dpart + s->dpart_cap}, out0, out1);
  PHX_HIP(hipGetLastError());
  s->dpart_used = 0;
  return PHX_OK;
}

// part 0: every row.  Multi-GPU overlap (s->bnd set by phx_spmv_flag_rows): part 1 = every row that references no halo
// entry (runs while the halo is in flight), part 2 = the flagged rows (after the unpack); both add into the same slots.
static int launch_spmv(phx_system *s, const double *vals, const double *x, double *y, int dots,
                       const double *d0, double *o0, double *o1, int part_of = 0) {
  hipStream_t st = s->mesh->stream;
  if (s->n == 0) return PHX_OK;  // empty system (a slab outside the domain): the dot-product slots stay zero
  const dim3 block(256);
  const uint8_t *own = s->own;
  static const int xg_env = getenv("PHX_SPMV_XCD_GROUP") ? atoi(getenv("PHX_SPMV_XCD_GROUP")) : -1;   // tuning aid
  const int xg = xg_env >= 0 ? xg_env : s->mesh->spmv_xcd_group;
  const uint8_t *kinds = vals == s->sell_val ? s->sell_kind : s->sell_kind_raw;
  const int32_t *rows = s->structured ? s->sell_rows : nullptr;
  DotPart dp{nullptr, nullptr};
  if (part_of == 2) {
    if (s->nbnd > 0) {
      const dim3 gb((unsigned)std::min<int64_t>(phx_div_up(s->nbnd, 256), 2048));
      if (dots > 0) PHX_CHECK(det_part(s, gb.x, &dp));
      if (dots == 0) k_spmv_bnd<0><<<gb, block, 0, st>>>(s->nbnd, s->bnd_rec, s->slice_ptr, s->sell_col, vals, kinds, s->stencil, x, y, d0, o0, o1, dp);
      else if (dots == 1) k_spmv_bnd<1><<<gb, block, 0, st>>>(s->nbnd, s->bnd_rec, s->slice_ptr, s->sell_col, vals, kinds, s->stencil, x, y, d0, o0, o1, dp);
      else k_spmv_bnd<2><<<gb, block, 0, st>>>(s->nbnd, s->bnd_rec, s->slice_ptr, s->sell_col, vals, kinds, s->stencil, x, y, d0, o0, o1, dp);
      PHX_HIP(hipGetLastError());
    }
    if (dots > 0) PHX_CHECK(det_fold(s, o0, o1));   // ... together with the partial sums of the interior launch
    return PHX_OK;
  }
  const uint8_t *bnd = part_of == 1 ? s->bnd : nullptr;
  // structured systems: the stencil blocks (rows of C0; u columns are unscaled in both value streams) ride behind
  // the SELL blocks of the same launch
  StencilArgs sa{0, nullptr, 0, nullptr, nullptr, 0, nullptr};
  int64_t nb_sell = phx_div_up(s->nslices, 4);
  static const int part = getenv("PHX_SPMV_PART") ? atoi(getenv("PHX_SPMV_PART")) : 0;  // timing aid: 1 SELL only, 2 stencil only
  if (part == 2) nb_sell = 0;
  int64_t nb = nb_sell;
  static const int sell_xcd = getenv("PHX_SELL_XCD") ? atoi(getenv("PHX_SELL_XCD")) : 0;   // experiment: 1 = contiguous eighths
  // structured systems: a multiple of 8 SELL blocks (the surplus finds no slice), so that blockIdx % 8 is the XCD
  if (s->structured) nb_sell = (nb_sell + 7) & ~(int64_t)7;
  nb = nb_sell;
  if (s->structured && s->nseg > 0 && part != 1) {
    // the stencil blocks start at a multiple of 8 so that blockIdx % 8 (the XCD a block lands on) is theirs to
    // use: XCD k walks the k-th contiguous eighth of the rows, whose x entries then stay in ITS 4 MiB L2
    // (round-robin placement had every L2 fetch the whole vector: 3.5 x the bytes, 44 % hits)
    nb_sell = (nb_sell + 7) & ~(int64_t)7;
    const int64_t nbst = phx_div_up(phx_div_up(s->nstencil_pos, 64), 16);   // four waves per block, four slices per wave
    sa = StencilArgs{s->nstencil_pos, s->seg, s->nseg, s->slice_seg, s->stencil, (nbst + 7) / 8, nullptr};
    if (s->st_map) { sa.chunk = s->st_chunk; sa.map = s->st_map; }
    nb = nb_sell + 8 * sa.chunk;
  }
  const int xg2 = (sell_xcd == 1 && s->structured && nb_sell % 8 == 0 && nb_sell > 0) ? -1 : xg;
  if (nb == 0 && !s->p2s) return PHX_OK;
  const dim3 g2((unsigned)std::max<int64_t>(nb, 1));
  if (dots > 0 && nb > 0) PHX_CHECK(det_part(s, nb, &dp));
  if (nb == 0) {}
  else if (dots == 0)
    k_spmv_sell<0><<<g2, block, 0, st>>>(s->n, s->nslices, s->slice_ptr, s->sell_col, vals, x, y, own, d0, o0, o1, xg2, kinds, rows, nb_sell, sa, bnd, dp);
  else if (dots == 1)
    k_spmv_sell<1><<<g2, block, 0, st>>>(s->n, s->nslices, s->slice_ptr, s->sell_col, vals, x, y, own, d0, o0, o1, xg2, kinds, rows, nb_sell, sa, bnd, dp);
  else
    k_spmv_sell<2><<<g2, block, 0, st>>>(s->n, s->nslices, s->slice_ptr, s->sell_col, vals, x, y, own, d0, o0, o1, xg2, kinds, rows, nb_sell, sa, bnd, dp);
  PHX_HIP(hipGetLastError());
  if (s->p2s && s->p2s->nrun > 0 && part_of == 0 && part != 1) {
    // structured P2: the interior rows from the eight class stencils, one wavefront per run
    const phx_p2_struct *ps = s->p2s;
    const dim3 gp((unsigned)std::min<int64_t>(phx_div_up(ps->nrun, 4), 4096));
    DotPart dq{nullptr, nullptr};
    if (dots > 0) PHX_CHECK(det_part(s, gp.x, &dq));
    if (dots == 0) k_spmv_p2s<0><<<gp, block, 0, st>>>(ps->nrun, ps->runs, ps->tabE, ps->tabO, ps->linemask, x, y, d0, o0, o1, dq);
    else if (dots == 1) k_spmv_p2s<1><<<gp, block, 0, st>>>(ps->nrun, ps->runs, ps->tabE, ps->tabO, ps->linemask, x, y, d0, o0, o1, dq);
    else k_spmv_p2s<2><<<gp, block, 0, st>>>(ps->nrun, ps->runs, ps->tabE, ps->tabO, ps->linemask, x, y, d0, o0, o1, dq);
    PHX_HIP(hipGetLastError());
  }
  if (dots > 0 && part_of == 0) PHX_CHECK(det_fold(s, o0, o1));
  return PHX_OK;
}

// Flags the rows that read one of the `nidx` solver positions in idx[] (the entries the neighbours send) and lists
// them for k_spmv_bnd.  Two passes: count, then fill.
__global__ void k_mark_positions(int64_t n, const int64_t *__restrict__ idx, uint8_t *__restrict__ isg) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) isg[idx[i]] = 1;
}
static int phx_spmv_flag_rows(phx_system *s, int nlists, const int64_t *const *idx, const int64_t *counts) {
  hipStream_t st = s->mesh->stream;
  PHX_HIP(phx_free(s->bnd)); PHX_HIP(phx_free(s->bnd_rec));
  s->bnd = nullptr; s->bnd_rec = nullptr; s->nbnd = 0;
  if (s->n == 0) return PHX_OK;
  uint8_t *isg = nullptr;
  unsigned long long *cnt = nullptr, hcnt = 0;
  PHX_HIP(phx_malloc(&isg, (size_t)s->n));
  PHX_HIP(phx_malloc(&s->bnd, (size_t)s->n));
  PHX_HIP(phx_malloc(&cnt, sizeof(unsigned long long)));
  PHX_HIP(hipMemsetAsync(isg, 0, (size_t)s->n, st));
  for (int l = 0; l < nlists; ++l)
    if (counts[l] > 0)
      k_mark_positions<<<dim3((unsigned)phx_div_up(counts[l], 256)), dim3(256), 0, st>>>(counts[l], idx[l], isg);
  const int32_t *rows = s->structured ? s->sell_rows : nullptr;
  StencilArgs sa{0, nullptr, 0, nullptr, nullptr, 0, nullptr};
  const int64_t nb_sell = phx_div_up(s->nslices, 4);
  int64_t nb = nb_sell;
  if (s->structured && s->nseg > 0) {
    sa = StencilArgs{s->nstencil_pos, s->seg, s->nseg, s->slice_seg, s->stencil, 0, nullptr};
    nb = nb_sell + phx_div_up(phx_div_up(s->nstencil_pos, 64), 4);
  }
  for (int pass = 0; pass < 2 && nb > 0; ++pass) {
    PHX_HIP(hipMemsetAsync(s->bnd, 0, (size_t)s->n, st));
    PHX_HIP(hipMemsetAsync(cnt, 0, sizeof(unsigned long long), st));
    k_spmv_flag<<<dim3((unsigned)nb), dim3(256), 0, st>>>(s->n, s->nslices, s->slice_ptr, s->sell_col, isg, s->own, rows,
                                                          nb_sell, sa, s->bnd, pass ? s->bnd_rec : nullptr, cnt);
    PHX_HIP(hipGetLastError());
    if (pass == 0) {
      PHX_HIP(hipMemcpyAsync(&hcnt, cnt, sizeof(hcnt), hipMemcpyDeviceToHost, st));
      PHX_HIP(hipStreamSynchronize(st));
      s->nbnd = (int64_t)hcnt;
      if (s->nbnd == 0) break;
      PHX_HIP(phx_malloc(&s->bnd_rec, sizeof(int32_t) * 6 * (size_t)s->nbnd));
    }
  }
  PHX_HIP(hipStreamSynchronize(st));
  PHX_HIP(phx_free(isg)); PHX_HIP(phx_free(cnt));
  return PHX_OK;
}

// Launches inside the solve can be bracketed by HIP events on the launch stream (phx_set_option
// PHX_OPT_PROFILE_SPMV): the roofline figures of bench.py come from here.  Class 0: the SpMV launches,
// class 1: the y passes of the sine-transform preconditioner.  The option value is the sampling stride:
// 1 brackets every launch, k every k-th (an event pair costs ~2.5 us of stream time, 4 % of a solve when
// every launch carries one).
static inline bool prof_sampled(const phx_system *s, int c) {
  return s->mesh->profile_spmv > 0 && s->mesh->prof_used[c] < (int)s->mesh->prof_ev[c].size() / 2 &&
         s->mesh->prof_seen[c] % s->mesh->profile_spmv == 0;
}
static int prof_begin(phx_system *s, int c = 0) {
  if (prof_sampled(s, c)) PHX_HIP(hipEventRecord(s->mesh->prof_ev[c][2 * s->mesh->prof_used[c]], s->mesh->stream));
  return PHX_OK;
}
static int prof_end(phx_system *s, int c = 0) {
  if (prof_sampled(s, c)) {
    PHX_HIP(hipEventRecord(s->mesh->prof_ev[c][2 * s->mesh->prof_used[c] + 1], s->mesh->stream));
    s->mesh->prof_used[c]++;
  }
  s->mesh->prof_seen[c]++;
  return PHX_OK;
}
static int prof_reset(phx_system *s) {
  for (int c = 0; c < 2; ++c) {
    s->mesh->prof_used[c] = 0;
    s->mesh->prof_seen[c] = 0;
    if (s->mesh->profile_spmv && s->mesh->prof_ev[c].empty()) {
      s->mesh->prof_ev[c].resize(2 * 1024);
      for (auto &e : s->mesh->prof_ev[c]) PHX_HIP(hipEventCreate(&e));
    }
  }
  return PHX_OK;
}
static int prof_collect(phx_system *s, double *avg_s, int *count, int c = 0) {
  *avg_s = 0.0;
  *count = 0;
  if (!s->mesh->profile_spmv || s->mesh->prof_used[c] == 0) return PHX_OK;
  PHX_HIP(hipEventSynchronize(s->mesh->prof_ev[c][2 * s->mesh->prof_used[c] - 1]));
  double tot = 0.0;
  for (int i = 0; i < s->mesh->prof_used[c]; ++i) {
    float ms = 0.f;
    PHX_HIP(hipEventElapsedTime(&ms, s->mesh->prof_ev[c][2 * i], s->mesh->prof_ev[c][2 * i + 1]));
    tot += ms;
  }
  *avg_s = tot * 1e-3 / s->mesh->prof_used[c];
  *count = s->mesh->prof_used[c];
  if (c == 0) {
    s->mesh->timings[4] = *avg_s;
    s->mesh->timings[5] = (double)*count;
  }
  return PHX_OK;
}

#include "phx_precond.inc.hip"
#include "phx_blockjac.inc.hip"
#include "phx_dense.inc.hip"
#include "phx_coarse.inc.hip"
void phx_blockjac_destroy(phx_blockjac *b) { blockjac_free(b); }

// restart threshold of the native loop: ratio of |r| since the last restart below which BiCGStab restarts from r
// (PHX_RESTART_DROP, e.g. 1e-4; default 0 = only on breakdown).  Measured with f32 transforms (round 2): the 3-D
// problems converge as with f64 transforms with or without it; the 2-D flower problem stays erratic (100-600
// iterations against 36-66) for every threshold tried -- refinement restarts are not what that problem lacks.
static double kr_drop2(const phx_system *s) {
  static const double env = getenv("PHX_RESTART_DROP") ? atof(getenv("PHX_RESTART_DROP")) : 0.0;
  (void)s;
  return env > 0.0 ? env * env : 0.0;
}


// Structured systems without the box preconditioner (configured out, vetoed, box too long): their u columns are
// unscaled, so the Jacobi scaling of the u block is applied here: P = D_u^-1 on u rows, the identity elsewhere
// (diag is stored in active order: through perm).
__global__ void k_jacobi_u(int64_t n, int64_t nu, const int32_t *__restrict__ perm, const double *__restrict__ diag,
                           const double *__restrict__ vin, double *__restrict__ vout) {
  GRID_STRIDE(i, n) vout[i] = i < nu ? vin[i] / diag[perm[i]] : vin[i];
}

// phat / shat are the preconditioned directions P p, P s the two SpMVs act on; without a preconditioner
// beyond the Jacobi scaling folded into the SELL values they ARE p and s.
// r = own ? b - t : 0 (t = A y, the TRUE residual of the accumulated iterate); (r, r) into slot set 0
__global__ void __launch_bounds__(256)
k_true_residual(int64_t n, const uint8_t *__restrict__ own, const double *__restrict__ b,
                const double *__restrict__ t, double *__restrict__ r, double *__restrict__ S, DotPart part) {
  double acc = 0.0;
  GRID_STRIDE(i, n) {
    const double ri = (!own || own[i]) ? b[i] - t[i] : 0.0;
    r[i] = ri;
    acc += ri * ri;
  }
  block_atomic_sum(acc, slot_base(S, 0, R_RR), part.p0);
}

// restart of the recurrences from r (x keeps its value): rhat = p = r, rho = (r, r) = S[R_OFF + R_RR]
__global__ void __launch_bounds__(256)
k_restart_from_r(int64_t n, const double *__restrict__ r, double *__restrict__ rhat, double *__restrict__ p,
                 double *__restrict__ S, RestOut ro) {
  GRID_STRIDE(i, n) { const double ri = r[i]; rhat[i] = ri; p[i] = ri; ro.put(i, ri); }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const double rr = S[R_OFF + R_RR];
    S[S_RHO] = rr; S[S_RHO_NEXT] = rr; S[S_RR] = rr; S[S_RR0] = rr; S[S_RR0 + 1] = rr;
    S[S_ALPHA] = 1.0; S[S_OMEGA] = 1.0;
    S[S_RESTARTS] += 1.0;
  }
}

struct KrVecs {
  double *r, *rhat, *p, *v, *sv, *t, *y, *b, *phat, *shat;
};
static inline KrVecs kr_vecs(phx_system *s) {
  double *w = s->kr_work ? s->kr_work : s->work;
  const int64_t n = s->n;
  KrVecs V{w, w + n, w + 2 * n, w + 3 * n, w + 4 * n, w + 5 * n, w + 6 * n, w + 7 * n, w + 2 * n, w + 4 * n};
  if (s->precond_state == 1 || s->u_unscaled || s->bj) {
    // attached workspaces (multi-GPU drivers, 10 n doubles) carry the two extra vectors themselves
    V.phat = s->kr_work ? w + 8 * n : s->pvec;
    V.shat = s->kr_work ? w + 9 * n : s->pvec + n;
  }
  return V;
}
static inline double *kr_scal(phx_system *s) { return s->kr_scal ? s->kr_scal : s->scal; }

// phases: 0 begin (local (b,b) -> R_RHO), 1 begin2 (after all-reduce), 2 v = A p (+R_RV),
// 3 s-update, 4 t = A s (+R_TS, R_TT), 5 x/r-update (+R_RHO, R_RR), 6 p-update + roll.
// mode 1 (phase API, multi-GPU): every dot product is folded into R right after its producer so
// the driver can all-reduce it; mode 0 (native loop): consumers fold the slots, `par` alternates.
static int kr_phase(phx_system *s, int phase, int mode, int par) {
  phx_mesh *m = s->mesh;
  const int64_t n = s->n;
  hipStream_t st = m->stream;
  if (phase == 0) {
    // (re)build the preconditioner for this system and ownership mask
    if (s->precond_state == 1 && s->precond->own_ptr != s->own) {
      phx_box_precond_destroy(s->precond);
      s->precond = nullptr;
      s->precond_state = 0;
    }
    if (s->precond_state == 0) PHX_CHECK(box_precond_setup(s));
    if (s->el_nblk > 0 && !s->bj && !s->bj_tried && m->precond != 0 && s->rowptr && n > 0) {
      s->bj_tried = true;
      PHX_CHECK(blockjac_build(s, s->el_nblk, &s->bj));   // nullptr when a vertex block is singular: scalar Jacobi then
      // coarse correction on top of the vertex blocks (one rank; a partitioned box keeps the blocks alone)
      if (s->bj && !s->own && !s->kr_work && !s->cc_tried) { s->cc_tried = true; PHX_CHECK(coarse_build(s, s->el_nblk, &s->cc)); }
    }
    if ((s->precond_state == 1 || s->u_unscaled || s->bj) && n > 0 && !s->kr_work) {
      // phat / shat: rows the preconditioner never writes (u rows another rank owns) stay zero
      if (!s->pvec) PHX_HIP(phx_malloc(&s->pvec, sizeof(double) * (size_t)n * 2));
      PHX_HIP(hipMemsetAsync(s->pvec, 0, sizeof(double) * (size_t)n * 2, st));
    }
  }
  const KrVecs V = kr_vecs(s);
  double *S = kr_scal(s);
  const dim3 block(256);
  // identity part of the box preconditioner, written by the producers of p and s (RestOut)
  const bool rest_out = s->precond_state == 1 && V.phat != V.p;
  const int32_t *rperm = s->structured ? nullptr : s->perm;
  const RestOut rop{rest_out ? V.phat : nullptr, rperm, s->nu}, ros{rest_out ? V.shat : nullptr, rperm, s->nu};
  switch (phase) {
    case 0:
      PHX_HIP(hipMemsetAsync(S, 0, sizeof(double) * PHX_SCAL_DOUBLES, st));
      {
        DotPart dp{nullptr, nullptr};
        PHX_CHECK(det_part(s, vec_grid(n).x, &dp));
        k_kr_begin<<<vec_grid(n), block, 0, st>>>(n, s->perm, s->rhs, s->own, V.b, V.r, V.rhat, V.p, V.y, S, rop, dp);
        PHX_CHECK(det_fold(s, slot_base(S, 0, R_RHO), nullptr));
      }
      k_reduce_slots<<<1, 64, 0, st>>>(S, 0, R_RHO, 1, 1);
      if (mode) k_set_scalar<<<1, 1, 0, st>>>(S + R_OFF + R_RR, s->precond_veto ? 1.0 : 0.0);
      break;
    case 1:
      k_kr_begin2<<<1, 1, 0, st>>>(S, mode);
      break;
    case 2:
      PHX_CHECK(prof_begin(s));
      PHX_CHECK(launch_spmv(s, s->sell_val, V.phat, V.v, 1, V.rhat, slot_base(S, par, R_RV), nullptr));
      PHX_CHECK(prof_end(s));
      if (mode) k_reduce_slots<<<1, 64, 0, st>>>(S, par, R_RV, 1, 1);
      break;
    case 3:
      k_update_s<<<vec_grid(n), block, 0, st>>>(n, par, s->own, V.r, V.v, V.sv, S, ros);
      break;
    case 4:
      PHX_CHECK(prof_begin(s));
      PHX_CHECK(launch_spmv(s, s->sell_val, V.shat, V.t, 2, V.sv, slot_base(S, par, R_TS), slot_base(S, par, R_TT)));
      PHX_CHECK(prof_end(s));
      if (mode) k_reduce_slots<<<1, 64, 0, st>>>(S, par, R_TS, 2, 1);
      break;
    case 5:
      {
        DotPart dp{nullptr, nullptr};
        PHX_CHECK(det_part(s, vec_grid(n).x, &dp));
        k_update_xr<<<vec_grid(n), block, 0, st>>>(n, par, s->own, V.phat, V.shat, V.sv, V.t, V.rhat, V.y, V.r, S, dp);
        PHX_CHECK(det_fold(s, slot_base(S, par, R_RHO), slot_base(S, par, R_RR)));
      }
      if (mode) k_reduce_slots<<<1, 64, 0, st>>>(S, par, R_RHO, 2, 1);
      break;
    case 6:
      k_update_p<<<vec_grid(n), block, 0, st>>>(n, par, s->own, V.r, V.v, V.p, V.rhat, S, kr_drop2(s), rop);
      break;
    case 7:  // phat = P p   (before the halo exchange of phat and phase 2)
      if (s->precond_state == 1) PHX_CHECK(box_precond_apply(s, V.p, V.phat, s->precond->dist ? 1 : 0));
      else if (s->bj) { PHX_CHECK(blockjac_apply(s, s->bj, V.p, V.phat)); PHX_CHECK(coarse_apply_add(s, s->cc, V.p, V.phat)); }
      else if (s->u_unscaled && n > 0) k_jacobi_u<<<vec_grid(n), block, 0, st>>>(n, s->nu, s->perm, s->diag, V.p, V.phat);
      break;
    case 8:  // shat = P s   (before the halo exchange of shat and phase 4)
      if (s->precond_state == 1) PHX_CHECK(box_precond_apply(s, V.sv, V.shat, s->precond->dist ? 1 : 0));
      else if (s->bj) { PHX_CHECK(blockjac_apply(s, s->bj, V.sv, V.shat)); PHX_CHECK(coarse_apply_add(s, s->cc, V.sv, V.shat)); }
      else if (s->u_unscaled && n > 0) k_jacobi_u<<<vec_grid(n), block, 0, st>>>(n, s->nu, s->perm, s->diag, V.sv, V.shat);
      break;
    // --- coarse correction of a partitioned elasticity system: 30 / 32 restrict p / s (the driver all-reduces the coarse
    // vector), 31 / 33 add the prolonged correction to phat / shat
    case 30: if (s->cc) PHX_CHECK(coarse_restrict(s, s->cc, V.p)); break;
    case 31: if (s->cc) PHX_CHECK(coarse_apply_end(s, s->cc, V.phat)); break;
    case 32: if (s->cc) PHX_CHECK(coarse_restrict(s, s->cc, V.sv)); break;
    case 33: if (s->cc) PHX_CHECK(coarse_apply_end(s, s->cc, V.shat)); break;
    // --- multi-GPU overlap: phases 2 / 4 in two launches (rows that read no halo entry | the rows that do)
    case 20:
      PHX_CHECK(prof_begin(s));
      PHX_CHECK(launch_spmv(s, s->sell_val, V.phat, V.v, 1, V.rhat, slot_base(S, par, R_RV), nullptr, 1));
      PHX_CHECK(prof_end(s));
      break;
    case 21:
      PHX_CHECK(launch_spmv(s, s->sell_val, V.phat, V.v, 1, V.rhat, slot_base(S, par, R_RV), nullptr, 2));
      if (mode) k_reduce_slots<<<1, 64, 0, st>>>(S, par, R_RV, 1, 1);
      break;
    case 40:
      PHX_CHECK(prof_begin(s));
      PHX_CHECK(launch_spmv(s, s->sell_val, V.shat, V.t, 2, V.sv, slot_base(S, par, R_TS), slot_base(S, par, R_TT), 1));
      PHX_CHECK(prof_end(s));
      break;
    case 41:
      PHX_CHECK(launch_spmv(s, s->sell_val, V.shat, V.t, 2, V.sv, slot_base(S, par, R_TS), slot_base(S, par, R_TT), 2));
      if (mode) k_reduce_slots<<<1, 64, 0, st>>>(S, par, R_TS, 2, 1);
      break;
    // --- verification of the TRUE residual (as phx_solve): 11: t = A y (the driver has exchanged the halo of y),
    // 12: r = own ? b - t : 0 and (r, r) -> R[R_RR] (all-reduced by the driver), 13: restart of the recurrences from r
    case 11:
      PHX_CHECK(launch_spmv(s, s->sell_val, V.y, V.t, 0, nullptr, nullptr, nullptr));
      break;
    case 12:
      if (n > 0) {
        DotPart dp{nullptr, nullptr};
        PHX_CHECK(det_part(s, vec_grid(n).x, &dp));
        k_true_residual<<<vec_grid(n), block, 0, st>>>(n, s->own, V.b, V.t, V.r, S, dp);
        PHX_CHECK(det_fold(s, slot_base(S, 0, R_RR), nullptr));
      }
      k_reduce_slots<<<1, 64, 0, st>>>(S, 0, R_RR, 1, 1);
      break;
    case 13:
      k_restart_from_r<<<vec_grid(n), block, 0, st>>>(n, V.r, V.rhat, V.p, S, rop);
      break;
    case 9:   // slab-exact preconditioner: second half of phase 7, after the all-gather of the carries
      if (s->precond_state == 1 && s->precond->dist) PHX_CHECK(box_precond_apply(s, V.p, V.phat, 2));
      break;
    case 10:  // ... of phase 8
      if (s->precond_state == 1 && s->precond->dist) PHX_CHECK(box_precond_apply(s, V.sv, V.shat, 2));
      break;
    default:
      phx_set_error("unknown Krylov phase %d", phase);
      return PHX_ERR_VALUE;
  }
  PHX_HIP(hipGetLastError());
  return PHX_OK;
}

extern "C" int phx_krylov_phase(phx_system *s, int phase) {
  PHX_HIP(hipSetDevice(s->mesh->device));
  return kr_phase(s, phase, 1, 0);
}

// x = D^-1 y scattered to FULL numbering, inactive (and non-owned) DoFs = 0
extern "C" int phx_krylov_finish(phx_system *s, double *x_out, int loc) {
  phx_mesh *m = s->mesh;
  PHX_HIP(hipSetDevice(m->device));
  const int64_t n = s->n;
  hipStream_t st = m->stream;
  const KrVecs V = kr_vecs(s);
  double *xfull = x_out;
  double *owned = nullptr;
  if (loc != PHX_DEVICE) { PHX_HIP(phx_malloc(&owned, sizeof(double) * (size_t)s->nfull)); xfull = owned; }
  PHX_HIP(hipMemsetAsync(xfull, 0, sizeof(double) * (size_t)s->nfull, st));
  if (n > 0)
    k_scatter_solution<<<dim3((unsigned)phx_div_up(n, 256)), dim3(256), 0, st>>>(
        n, s->perm, s->full_of_active, s->diag, s->cscale, V.y, xfull, s->out_vertex, s->nent);
  PHX_HIP(hipGetLastError());
  PHX_HIP(hipStreamSynchronize(st));
  if (owned) {
    PHX_HIP(hipMemcpy(x_out, owned, sizeof(double) * (size_t)s->nfull, hipMemcpyDeviceToHost));
    PHX_HIP(phx_free(owned));
  }
  return PHX_OK;
}

// after phase 0: 1 when the system is preconditioned beyond Jacobi (phases 7 / 8 then fill phat / shat,
// the vectors a multi-GPU driver must halo-exchange instead of p / s)
extern "C" int phx_krylov_precond_active(const phx_system *s, int *active) {
  // 1: the SpMV inputs are phat / shat (box preconditioner, or the u-block Jacobi of a structured system)
  *active = (s->precond_state == 1 || s->u_unscaled || s->bj) ? 1 : 0;
  return PHX_OK;
}

// Collective decision (see k_set_scalar): drop this rank's preconditioner; phat / shat alias p / s again.
extern "C" int phx_krylov_precond_disable(phx_system *s) {
  if (s->precond_state == 1) {
    PHX_HIP(hipSetDevice(s->mesh->device));
    PHX_HIP(hipStreamSynchronize(s->mesh->stream));
    phx_box_precond_destroy(s->precond);
    s->precond = nullptr;
  }
  s->precond_state = -1;
  return PHX_OK;
}

extern "C" int phx_krylov_attach(phx_system *s, double *work, double *scal, const uint8_t *own) {
  s->kr_work = work;
  s->kr_scal = scal;
  s->own = own;
  return PHX_OK;
}

// Elapsed time of an EMPTY event pair on the mesh stream (mean of 32): what the bracketing itself adds
// to every figure PHX_OPT_PROFILE_SPMV reports; bench.py prints it next to the raw average.
extern "C" int phx_event_pair_overhead(phx_mesh *m, double *seconds) {
  PHX_HIP(hipSetDevice(m->device));
  const int reps = 32;
  hipEvent_t ev[2 * reps];
  for (auto &e : ev) PHX_HIP(hipEventCreate(&e));
  for (int i = 0; i < reps; ++i) {
    PHX_HIP(hipEventRecord(ev[2 * i], m->stream));
    PHX_HIP(hipEventRecord(ev[2 * i + 1], m->stream));
  }
  PHX_HIP(hipEventSynchronize(ev[2 * reps - 1]));
  double tot = 0.0;
  for (int i = 0; i < reps; ++i) {
    float ms = 0.f;
    PHX_HIP(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
    tot += ms;
  }
  for (auto &e : ev) PHX_HIP(hipEventDestroy(e));
  *seconds = tot * 1e-3 / reps;
  return PHX_OK;
}

extern "C" int phx_krylov_profile(phx_system *s, int reset, double *avg_seconds, int64_t *count) {
  if (reset) return prof_reset(s);
  int c = 0;
  double a = 0.0;
  PHX_CHECK(prof_collect(s, &a, &c));
  if (avg_seconds) *avg_seconds = a;
  if (count) *count = c;
  return PHX_OK;
}

// out[8] = {preconditioner (0 Jacobi, 1 lattice solve, 2 vertex-block Jacobi), transform lengths L0, L1, L2, stored lattice points of the box,
//           sampled average seconds of a y-pass launch, launches sampled, bytes per lattice value (4 / 8)}
extern "C" int phx_precond_info(phx_system *s, double *out) {
  for (int i = 0; i < 8; ++i) out[i] = 0.0;
  if (s->bj) out[0] = s->cc ? 3.0 : 2.0;   // dense vertex blocks (interface elasticity), 3: with the coarse correction
  if (s->cc) { out[1] = (double)s->cc->ratio; out[4] = (double)s->cc->nc; }
  if (s->precond_state != 1) return PHX_OK;
  const BoxGrid &g = s->precond->g;
  out[0] = 1.0;
  for (int a = 0; a < 3; ++a) out[1 + a] = (double)g.L[a];
  out[4] = (double)g.m[0] * (double)g.m[1] * (double)g.m[2];
  double avg = 0.0;
  int cnt = 0;
  PHX_CHECK(prof_collect(s, &avg, &cnt, 1));
  out[5] = avg;
  out[6] = (double)cnt;
  out[7] = s->precond->f32 ? 4.0 : 8.0;
  return PHX_OK;
}

extern "C" int phx_solve(phx_system *s, int method, double rtol, int64_t max_iter, double *x_out,
                         int loc, double *stats) {
  phx_mesh *m = s->mesh;
  PHX_HIP(hipSetDevice(m->device));
  PHX_REQUIRE(method == PHX_BICGSTAB_JACOBI, PHX_ERR_NOT_IMPLEMENTED, "unknown method %d", method);
  hipStream_t st = m->stream;
  double *S = kr_scal(s);
  PHX_CHECK(prof_reset(s));
  PHX_CHECK(phx_begin_timing(m));
  PHX_CHECK(kr_phase(s, 0, 0, 0));
  // The host looks at the residual every 8th iteration with Jacobi (cheap iterations).  With the box preconditioner
  // (few, expensive iterations) a check drains the pipeline for ~30 us, so the next one is scheduled from the
  // observed convergence rate: half of the predicted remaining iterations ahead (at most 12, at least 2) -- near the
  // end every 2nd iteration, so no iteration is wasted on a late check (28 -> ~9 drains over 56 iterations).
  const bool pc = s->precond_state == 1;
  PHX_CHECK(kr_phase(s, 1, 0, 0));
  PHX_HIP(hipMemcpyAsync(s->scal_h, S, sizeof(double) * 16, hipMemcpyDeviceToHost, st));
  PHX_HIP(hipStreamSynchronize(st));
  const double bb = s->scal_h[S_BB];
  int64_t it = 0, spmvs = 0, next_check = pc ? 2 : 8, last_check = 0;
  double relres = bb == 0.0 ? 0.0 : 1.0, last_relres = 1.0;
  int rc = PHX_OK;
  // Outer loop: when the recurrences report convergence, the TRUE residual b - A y is computed once (one SpMV) and,
  // should it not meet the tolerance -- after thousands of iterations the recursive residual drifts away from it
  // (3-D P2 systems, cond 1e8: observed 1e-5 instead of 1e-11) --, the iteration restarts from it.  The relres that
  // is returned, and stats[6], refer to the true residual.
  int verifications = 0;
  for (;;) {
    while (bb != 0.0 && it < max_iter) {
      const int par = (int)(it & 1);
      static const int seq[6] = {7, 2, 3, 8, 4, 5};
      for (int ph : seq) PHX_CHECK(kr_phase(s, ph, 0, par));
      spmvs += 2;
      ++it;
      if (it >= next_check || it == max_iter) {
        // fold (r,r) and (rhat,r) of this iteration for the host, without clearing the slots
        k_reduce_slots<<<1, 64, 0, st>>>(S, par, R_RHO, 2, 0);
        PHX_HIP(hipMemcpyAsync(s->scal_h, S, sizeof(double) * 16, hipMemcpyDeviceToHost, st));
        PHX_HIP(hipStreamSynchronize(st));
        const double rr = s->scal_h[R_OFF + R_RR];
        const double omega = s->scal_h[S_OMEGA];
        relres = sqrt(rr / bb);
        if (!(rr == rr) || !(fabs(rr) <= 1.0e300) || !(omega == omega)) {
          phx_set_error("BiCGStab breakdown at iteration %lld (omega=%g rr=%g)", (long long)it, omega, rr);
          rc = PHX_ERR_BREAKDOWN;
          break;
        }
        if (relres <= rtol) break;
        int64_t step = pc ? 2 : 8;
        if (pc && relres < last_relres && relres > 0.0) {
          // iterations still needed at the rate seen since the last check
          const double rate = log(last_relres / relres) / (double)(it - last_check);
          const double remaining = log(relres / rtol) / rate;
          step = std::max<int64_t>(2, std::min<int64_t>(12, (int64_t)(0.5 * remaining)));
          step &= ~(int64_t)1;   // checks stay on even iterations
        }
        last_check = it;
        last_relres = relres;
        next_check = it + step;
      }
      PHX_CHECK(kr_phase(s, 6, 0, par));
    }
    if (rc != PHX_OK || bb == 0.0 || !(relres <= rtol)) break;
    // verify
    const KrVecs V = kr_vecs(s);
    PHX_HIP(hipMemsetAsync(S + P_OFF, 0, sizeof(double) * (PHX_SCAL_DOUBLES - P_OFF), st));
    PHX_CHECK(launch_spmv(s, s->sell_val, V.y, V.t, 0, nullptr, nullptr, nullptr));
    {
      DotPart dp{nullptr, nullptr};
      PHX_CHECK(det_part(s, vec_grid(s->n).x, &dp));
      k_true_residual<<<vec_grid(s->n), dim3(256), 0, st>>>(s->n, s->own, V.b, V.t, V.r, S, dp);
      PHX_CHECK(det_fold(s, slot_base(S, 0, R_RR), nullptr));
    }
    k_reduce_slots<<<1, 64, 0, st>>>(S, 0, R_RR, 1, 1);
    PHX_HIP(hipMemcpyAsync(s->scal_h, S, sizeof(double) * 16, hipMemcpyDeviceToHost, st));
    PHX_HIP(hipStreamSynchronize(st));
    spmvs += 1;
    const double rr_true = s->scal_h[R_OFF + R_RR];
    if (!(rr_true == rr_true)) { phx_set_error("non-finite true residual"); rc = PHX_ERR_BREAKDOWN; break; }
    relres = sqrt(rr_true / bb);
    if (relres <= rtol || ++verifications > 8 || it >= max_iter) break;
    {
      const bool rest_out = s->precond_state == 1 && V.phat != V.p;
      const RestOut rop{rest_out ? V.phat : nullptr, s->structured ? nullptr : s->perm, s->nu};
      k_restart_from_r<<<vec_grid(s->n), dim3(256), 0, st>>>(s->n, V.r, V.rhat, V.p, S, rop);
    }
    last_relres = relres;
    last_check = it;
    next_check = it + 2;   // (every slot is clear now: the parity the loop derives from `it` needs no care)
  }
  // back to full numbering, x = D^-1 y, inactive DoFs = 0 (MUMPS ICNTL(24)=1 semantics)
  PHX_CHECK(phx_krylov_finish(s, x_out, loc));
  PHX_CHECK(phx_end_timing(m, 3));
  double pavg = 0.0;
  int pcount = 0;
  PHX_CHECK(prof_collect(s, &pavg, &pcount));
  if (stats) {
    stats[0] = (double)it;
    stats[1] = relres;
    stats[2] = m->timings[3];
    stats[3] = (double)spmvs;
    stats[4] = pavg;
    stats[5] = (double)pcount;
    stats[6] = relres <= rtol ? 1.0 : 0.0;  // converged: the caller decides what an unconverged iterate is worth
    stats[7] = s->scal_h[S_RESTARTS];
  }
  return rc;
}

#include "phx_dist.inc.hip"

// y = A x in ORIGINAL active numbering (for tests and externally driven iterations)
__global__ void k_scatter_perm(int64_t n, const int32_t *__restrict__ perm,
                               const double *__restrict__ in, double *__restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) out[perm[i]] = in[i];
}

extern "C" int phx_spmv(phx_system *s, const double *x, double *y, int loc) {
  phx_mesh *m = s->mesh;
  PHX_HIP(hipSetDevice(m->device));
  const int64_t n = s->n;
  hipStream_t st = m->stream;
  double *xs = s->work + 6 * n, *ys = s->work + 7 * n, *tmp = s->work + 8 * n;
  const dim3 block(256), gn((unsigned)phx_div_up(n, 256));
  const double *xd = x;
  if (loc != PHX_DEVICE) {
    PHX_HIP(hipMemcpyAsync(tmp, x, sizeof(double) * n, hipMemcpyHostToDevice, st));
    xd = tmp;
  }
  k_gather<<<gn, block, 0, st>>>(n, s->perm, xd, xs);
  PHX_CHECK(launch_spmv(s, s->sell_val_raw, xs, ys, 0, nullptr, nullptr, nullptr));
  if (loc == PHX_DEVICE) {
    k_scatter_perm<<<gn, block, 0, st>>>(n, s->perm, ys, y);
    PHX_HIP(hipStreamSynchronize(st));
  } else {
    k_scatter_perm<<<gn, block, 0, st>>>(n, s->perm, ys, tmp);
    PHX_HIP(hipMemcpyAsync(y, tmp, sizeof(double) * n, hipMemcpyDeviceToHost, st));
    PHX_HIP(hipStreamSynchronize(st));
  }
  return PHX_OK;
}

#include "phx_spmv_exp.inc.hip"

extern "C" int phx_spmv_bench(phx_system *s, int reps, double *out) {
  phx_mesh *m = s->mesh;
  PHX_HIP(hipSetDevice(m->device));
  const int64_t n = s->n;
  hipStream_t st = m->stream;
  double *xs = s->work + 6 * n, *ys = s->work + 7 * n;
  PHX_HIP(hipMemcpyAsync(xs, s->rhs, sizeof(double) * n, hipMemcpyDeviceToDevice, st));
  if (const char *e = getenv("PHX_SELL_EXP")) {
    // load-schedule experiments on the stored rows alone (phx_spmv_exp.inc.hip); out[1] = max |y_exp - y| over them
    PHX_REQUIRE(s->structured && s->n_sell_rows > 0, PHX_ERR_VALUE, "PHX_SELL_EXP needs a structured system");
    const int var = atoi(e);
    PHX_CHECK(launch_spmv(s, s->sell_val, xs, ys, 0, nullptr, nullptr, nullptr));
    double *ye = s->work + 5 * n;
    PHX_HIP(hipMemcpyAsync(ye, ys, sizeof(double) * n, hipMemcpyDeviceToDevice, st));
    for (int i = 0; i < 3; ++i) PHX_CHECK(launch_sell16_exp(s, var, s->sell_val, xs, ye));
    PHX_HIP(hipEventRecord(m->ev0, st));
    for (int i = 0; i < reps; ++i) PHX_CHECK(launch_sell16_exp(s, var, s->sell_val, xs, ye));
    PHX_HIP(hipEventRecord(m->ev1, st));
    PHX_HIP(hipEventSynchronize(m->ev1));
    float ms = 0.f;
    PHX_HIP(hipEventElapsedTime(&ms, m->ev0, m->ev1));
    std::vector<double> a((size_t)n), b((size_t)n);
    PHX_HIP(hipMemcpy(a.data(), ys, sizeof(double) * n, hipMemcpyDeviceToHost));
    PHX_HIP(hipMemcpy(b.data(), ye, sizeof(double) * n, hipMemcpyDeviceToHost));
    double d = 0.0;
    for (int64_t i = 0; i < n; ++i) d = std::max(d, fabs(a[i] - b[i]));
    out[0] = (double)ms / reps; out[1] = d; out[2] = 12.0 * (double)s->sell_nnz;
    return PHX_OK;
  }
  for (int i = 0; i < 3; ++i)
    PHX_CHECK(launch_spmv(s, s->sell_val, xs, ys, 0, nullptr, nullptr, nullptr));
  PHX_HIP(hipEventRecord(m->ev0, st));
  for (int i = 0; i < reps; ++i)
    PHX_CHECK(launch_spmv(s, s->sell_val, xs, ys, 0, nullptr, nullptr, nullptr));
  PHX_HIP(hipEventRecord(m->ev1, st));
  PHX_HIP(hipEventSynchronize(m->ev1));
  float ms = 0.f;
  PHX_HIP(hipEventElapsedTime(&ms, m->ev0, m->ev1));
  out[0] = (double)ms / reps;
  out[1] = 12.0 * (double)s->sell_true_nnz + 20.0 * (double)n;  // algorithmic
  out[2] = 12.0 * (double)s->sell_nnz + 20.0 * (double)n;       // incl. SELL padding
  m->timings[4] = out[0] * 1e-3;
  return PHX_OK;
}
