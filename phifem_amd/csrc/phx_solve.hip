// Sparse solve on the device (gfx950): SELL-64 SpMV + right-Jacobi BiCGStab on the active system.
// Replaces PETSc KSP(preonly)+PC(lu)+MUMPS, demo/weak-dirichlet/flower/main.py:162-182 [3P].
// The weak-Dirichlet matrix is NOT symmetric (main.py:114 has no transposed partner), so plain CG
// diverges on it (DESIGN.md "Solver"); BiCGStab uses the same SpMV/dot/axpy kernels.
//
// SpMV layout: rows are grouped in slices of 64 (one wavefront per slice, one lane per row);
// inside a slice entries are stored column-major, so lane r reads val[base + k*64 + r]:
// every wave-instruction is one contiguous 512 B (f64) / 256 B (i32) segment.  Rows are sorted
// by length inside windows of 4096 rows (keeps x-gather locality, removes most padding), and the
// whole system is renumbered into that order so y is written coalesced.  Explicitly stored zeros
// of the assembled CSR are not carried into the SELL copy.
#include <hipcub/hipcub.hpp>
#include <math.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "phx_common.h"

#define SELL_C 64
#define SELL_WINDOW 4096

// ---------------------------------------------------------------------------------------------
// SELL construction
// ---------------------------------------------------------------------------------------------
__global__ void k_row_lengths(int64_t n, const int64_t *__restrict__ rowptr,
                              const int32_t *__restrict__ col, const double *__restrict__ val,
                              uint32_t *__restrict__ keys, int32_t *__restrict__ rows,
                              unsigned long long *__restrict__ total) {
  const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r >= n) return;
  int len = 0;
  for (int64_t k = rowptr[r]; k < rowptr[r + 1]; ++k)
    if (val[k] != 0.0 || col[k] == (int32_t)r) ++len;
  if (len > 255) len = 255;
  atomicAdd(total, (unsigned long long)len);
  // ascending key = (window, 255-len): descending length inside each window
  keys[r] = ((uint32_t)(r / SELL_WINDOW) << 8) | (uint32_t)(255 - len);
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
    if (pos < n) w = max(w, 255 - (int)(sorted_keys[pos] & 0xff));
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

int phx_system_build_sell(phx_system *s) {
  phx_mesh *m = s->mesh;
  const int64_t n = s->n;
  PHX_REQUIRE(n / SELL_WINDOW < (1 << 23), PHX_ERR_VALUE, "system too large for the SELL sort key");
  uint32_t *keys = nullptr, *keys2 = nullptr;
  int32_t *rows = nullptr;
  PHX_HIP(hipMalloc(&keys, sizeof(uint32_t) * (size_t)n));
  PHX_HIP(hipMalloc(&keys2, sizeof(uint32_t) * (size_t)n));
  PHX_HIP(hipMalloc(&rows, sizeof(int32_t) * (size_t)n));
  PHX_HIP(hipMalloc(&s->perm, sizeof(int32_t) * (size_t)n));
  PHX_HIP(hipMalloc(&s->iperm, sizeof(int32_t) * (size_t)n));
  const dim3 block(256), grid((unsigned)phx_div_up(n, 256));
  unsigned long long *dtotal = nullptr;
  PHX_HIP(hipMalloc(&dtotal, sizeof(unsigned long long)));
  PHX_HIP(hipMemsetAsync(dtotal, 0, sizeof(unsigned long long), m->stream));
  k_row_lengths<<<grid, block, 0, m->stream>>>(n, s->rowptr, s->col, s->val, keys, rows, dtotal);
  unsigned long long htotal = 0;
  PHX_HIP(hipMemcpyAsync(&htotal, dtotal, sizeof(htotal), hipMemcpyDeviceToHost, m->stream));
  PHX_HIP(hipStreamSynchronize(m->stream));
  PHX_HIP(hipFree(dtotal));
  s->sell_true_nnz = (int64_t)htotal;
  size_t bytes = 0;
  PHX_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, keys, keys2, rows, s->perm, (int)n, 0, 32, m->stream));
  void *tmp = nullptr;
  PHX_HIP(hipMalloc(&tmp, bytes ? bytes : 16));
  PHX_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, bytes, keys, keys2, rows, s->perm, (int)n, 0, 32, m->stream));
  k_invert_perm<<<grid, block, 0, m->stream>>>(n, s->perm, s->iperm);
  s->nslices = phx_div_up(n, SELL_C);
  int64_t *widths = nullptr;
  PHX_HIP(hipMalloc(&widths, sizeof(int64_t) * (size_t)(s->nslices + 1)));
  PHX_HIP(hipMalloc(&s->slice_ptr, sizeof(int64_t) * (size_t)(s->nslices + 1)));
  k_slice_widths<<<dim3((unsigned)phx_div_up(s->nslices + 1, 256)), block, 0, m->stream>>>(
      s->nslices, n, keys2, widths);
  {
    size_t b2 = 0;
    PHX_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, b2, widths, s->slice_ptr, (int)(s->nslices + 1), m->stream));
    void *t2 = nullptr;
    PHX_HIP(hipMalloc(&t2, b2 ? b2 : 16));
    PHX_HIP(hipcub::DeviceScan::ExclusiveSum(t2, b2, widths, s->slice_ptr, (int)(s->nslices + 1), m->stream));
    PHX_HIP(hipStreamSynchronize(m->stream));
    PHX_HIP(hipFree(t2));
  }
  PHX_HIP(hipMemcpy(&s->sell_nnz, s->slice_ptr + s->nslices, sizeof(int64_t), hipMemcpyDeviceToHost));
  PHX_HIP(hipMalloc(&s->sell_col, sizeof(int32_t) * (size_t)s->sell_nnz));
  PHX_HIP(hipMalloc(&s->sell_val, sizeof(double) * (size_t)s->sell_nnz));
  PHX_HIP(hipMalloc(&s->sell_val_raw, sizeof(double) * (size_t)s->sell_nnz));
  k_sell_fill<<<dim3((unsigned)phx_div_up(s->nslices * SELL_C, 256)), block, 0, m->stream>>>(
      n, s->rowptr, s->col, s->val, s->diag, s->perm, s->iperm, s->slice_ptr, s->sell_col,
      s->sell_val, s->sell_val_raw);
  PHX_HIP(hipGetLastError());
  PHX_HIP(hipStreamSynchronize(m->stream));
  PHX_HIP(hipFree(tmp)); PHX_HIP(hipFree(keys)); PHX_HIP(hipFree(keys2)); PHX_HIP(hipFree(rows));
  PHX_HIP(hipFree(widths));
  // solver workspace: 9 vectors + scalars
  PHX_HIP(hipMalloc(&s->work, sizeof(double) * (size_t)n * 9));
  PHX_HIP(hipMalloc(&s->scal, sizeof(double) * 16));
  PHX_HIP(hipHostMalloc(&s->scal_h, sizeof(double) * 16));
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

template <int DOTS>
__global__ void __launch_bounds__(256)
k_spmv_sell(int64_t n, int64_t nslices, const int64_t *__restrict__ slice_ptr,
            const int32_t *__restrict__ scol, const double *__restrict__ sval,
            const double *__restrict__ x, double *__restrict__ y,
            const double *__restrict__ d0, double *__restrict__ out0,
            double *__restrict__ out1) {
  const int lane = threadIdx.x & 63;
  const int64_t s = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
  double acc = 0.0;
  int64_t row = -1;
  if (s < nslices) {
    const int64_t base = slice_ptr[s];
    const int width = (int)((slice_ptr[s + 1] - base) >> 6);
    const int32_t *c = scol + base + lane;
    const double *v = sval + base + lane;
    int k = 0;
    for (; k + 4 <= width; k += 4) {
      const int32_t c0 = c[(k + 0) * SELL_C], c1 = c[(k + 1) * SELL_C];
      const int32_t c2 = c[(k + 2) * SELL_C], c3 = c[(k + 3) * SELL_C];
      const double v0 = v[(k + 0) * SELL_C], v1 = v[(k + 1) * SELL_C];
      const double v2 = v[(k + 2) * SELL_C], v3 = v[(k + 3) * SELL_C];
      acc += v0 * x[c0];
      acc += v1 * x[c1];
      acc += v2 * x[c2];
      acc += v3 * x[c3];
    }
    for (; k < width; ++k) acc += v[k * SELL_C] * x[c[k * SELL_C]];
    row = s * SELL_C + lane;
    if (row < n) y[row] = acc; else row = -1;
  }
  if (DOTS > 0) {
    // DOTS == 1: out0 += (y, d0);  DOTS == 2: also out1 += (y, y)
    __shared__ double red[2][4];
    double p0 = 0.0, p1 = 0.0;
    if (row >= 0) {
      p0 = acc * d0[row];
      if (DOTS > 1) p1 = acc * acc;
    }
    p0 = wave_sum(p0);
    if (DOTS > 1) p1 = wave_sum(p1);
    const int w = threadIdx.x >> 6;
    if (lane == 0) { red[0][w] = p0; red[1][w] = p1; }
    __syncthreads();
    if (threadIdx.x == 0) {
      unsafeAtomicAdd(out0, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
      if (DOTS > 1) unsafeAtomicAdd(out1, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// BiCGStab vector kernels.  Scalars live on the device (S[...]); every kernel derives alpha /
// omega / beta from them, so an iteration is a fixed launch sequence without host round trips.
// ---------------------------------------------------------------------------------------------
enum { S_RHO0 = 0, S_RHO1 = 1, S_RV = 2, S_TS = 3, S_TT = 4, S_RR = 5, S_SS = 6, S_ALPHA = 7,
       S_OMEGA = 8, S_BB = 9, S_FLAG = 10 };

__device__ __forceinline__ void block_atomic_sum(double v, double *out) {
  __shared__ double red[4];
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(out, red[0] + red[1] + red[2] + red[3]);
  __syncthreads();
}

// s = r - alpha v, alpha = rho/(rhat,v); accumulates (s,s); zeroes the slots of the next products
__global__ void __launch_bounds__(256)
k_update_s(int64_t n, int it, const double *__restrict__ r, const double *__restrict__ v,
           double *__restrict__ sv, double *__restrict__ S) {
  const double alpha = S[it & 1] / S[S_RV];
  double acc = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double t = r[i] - alpha * v[i];
    sv[i] = t;
    acc += t * t;
  }
  block_atomic_sum(acc, &S[S_SS]);
  if (blockIdx.x == 0 && threadIdx.x == 0) S[S_ALPHA] = alpha;
}

// x += alpha p + omega s;  r = s - omega t;  accumulates rho_next = (rhat,r) and (r,r)
__global__ void __launch_bounds__(256)
k_update_xr(int64_t n, int it, const double *__restrict__ p, const double *__restrict__ sv,
            const double *__restrict__ t, const double *__restrict__ rhat, double *__restrict__ x,
            double *__restrict__ r, double *__restrict__ S) {
  const double alpha = S[S_ALPHA];
  const double omega = S[S_TS] / S[S_TT];
  double a0 = 0.0, a1 = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double si = sv[i];
    x[i] += alpha * p[i] + omega * si;
    const double ri = si - omega * t[i];
    r[i] = ri;
    a0 += rhat[i] * ri;
    a1 += ri * ri;
  }
  block_atomic_sum(a0, &S[(it + 1) & 1]);
  block_atomic_sum(a1, &S[S_RR]);
  if (blockIdx.x == 0 && threadIdx.x == 0) S[S_OMEGA] = omega;
}

// p = r + beta (p - omega v), beta = (rho_next/rho)(alpha/omega); clears the per-iteration sums
__global__ void __launch_bounds__(256)
k_update_p(int64_t n, int it, const double *__restrict__ r, const double *__restrict__ v,
           double *__restrict__ p, double *__restrict__ S, double *__restrict__ Snext) {
  const double beta = (S[(it + 1) & 1] / S[it & 1]) * (S[S_ALPHA] / S[S_OMEGA]);
  const double omega = S[S_OMEGA];
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    p[i] = r[i] + beta * (p[i] - omega * v[i]);
}

__global__ void k_clear_iter_scalars(int it, double *S) {
  // the sums the coming iteration accumulates into (rho of iteration it+1 lives in S[(it+1)&1])
  S[S_RV] = 0.0; S[S_TS] = 0.0; S[S_TT] = 0.0; S[S_RR] = 0.0; S[S_SS] = 0.0;
  S[(it + 1) & 1] = 0.0;
}

__global__ void k_dot2(int64_t n, const double *__restrict__ a, const double *__restrict__ b,
                       double *__restrict__ out) {
  double acc = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    acc += a[i] * b[i];
  block_atomic_sum(acc, out);
}

__global__ void k_gather(int64_t n, const int32_t *__restrict__ perm, const double *__restrict__ in,
                         double *__restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[perm[i]];
}

// solution back to FULL numbering: x_full[full_of_active[perm[pos]]] = y[pos] / diag[perm[pos]]
__global__ void k_scatter_solution(int64_t n, const int32_t *__restrict__ perm,
                                   const int64_t *__restrict__ full_of_active,
                                   const double *__restrict__ diag, const double *__restrict__ y,
                                   double *__restrict__ xfull) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t r = perm[i];
  xfull[full_of_active[r]] = y[i] / diag[r];
}

static inline dim3 vec_grid(int64_t n) { return dim3((unsigned)std::min<int64_t>(phx_div_up(n, 256), 2048)); }

static int launch_spmv(phx_system *s, const double *vals, const double *x, double *y, int dots,
                       const double *d0, double *o0, double *o1) {
  hipStream_t st = s->mesh->stream;
  const dim3 block(256), grid((unsigned)phx_div_up(s->nslices, 4));
  if (dots == 0)
    k_spmv_sell<0><<<grid, block, 0, st>>>(s->n, s->nslices, s->slice_ptr, s->sell_col, vals, x, y, d0, o0, o1);
  else if (dots == 1)
    k_spmv_sell<1><<<grid, block, 0, st>>>(s->n, s->nslices, s->slice_ptr, s->sell_col, vals, x, y, d0, o0, o1);
  else
    k_spmv_sell<2><<<grid, block, 0, st>>>(s->n, s->nslices, s->slice_ptr, s->sell_col, vals, x, y, d0, o0, o1);
  PHX_HIP(hipGetLastError());
  return PHX_OK;
}

extern "C" int phx_solve(phx_system *s, int method, double rtol, int64_t max_iter, double *x_out,
                         int loc, double *stats) {
  phx_mesh *m = s->mesh;
  PHX_HIP(hipSetDevice(m->device));
  PHX_REQUIRE(method == PHX_BICGSTAB_JACOBI, PHX_ERR_NOT_IMPLEMENTED, "unknown method %d", method);
  const int64_t n = s->n;
  hipStream_t st = m->stream;
  double *r = s->work, *rhat = r + n, *p = r + 2 * n, *v = r + 3 * n, *sv = r + 4 * n,
         *t = r + 5 * n, *y = r + 6 * n, *b = r + 7 * n;
  double *S = s->scal;
  const dim3 block(256);
  const dim3 gn((unsigned)phx_div_up(n, 256));
  const int check_every = 8;
  PHX_CHECK(phx_begin_timing(m));
  // solver ordering; x0 = 0 => r = b; rhat = r; p = r
  k_gather<<<gn, block, 0, st>>>(n, s->perm, s->rhs, b);
  PHX_HIP(hipMemcpyAsync(r, b, sizeof(double) * n, hipMemcpyDeviceToDevice, st));
  PHX_HIP(hipMemcpyAsync(rhat, b, sizeof(double) * n, hipMemcpyDeviceToDevice, st));
  PHX_HIP(hipMemcpyAsync(p, b, sizeof(double) * n, hipMemcpyDeviceToDevice, st));
  PHX_HIP(hipMemsetAsync(y, 0, sizeof(double) * n, st));
  PHX_HIP(hipMemsetAsync(S, 0, sizeof(double) * 16, st));
  k_dot2<<<vec_grid(n), block, 0, st>>>(n, b, b, &S[S_RHO0]);  // rho_0 = (rhat, r) = (b, b)
  PHX_HIP(hipMemcpyAsync(s->scal_h, S, sizeof(double) * 16, hipMemcpyDeviceToHost, st));
  PHX_HIP(hipStreamSynchronize(st));
  const double bb = s->scal_h[S_RHO0];
  int64_t it = 0, spmvs = 0;
  double relres = 0.0;
  int rc = PHX_OK;
  if (bb == 0.0) {
    relres = 0.0;
  } else {
    relres = 1.0;
    while (it < max_iter) {
      // v = A D^-1 p, (rhat, v)
      PHX_CHECK(launch_spmv(s, s->sell_val, p, v, 1, rhat, &S[S_RV], nullptr));
      k_update_s<<<vec_grid(n), block, 0, st>>>(n, (int)(it & 1), r, v, sv, S);
      // t = A D^-1 s, (t,s), (t,t)
      PHX_CHECK(launch_spmv(s, s->sell_val, sv, t, 2, sv, &S[S_TS], &S[S_TT]));
      k_update_xr<<<vec_grid(n), block, 0, st>>>(n, (int)(it & 1), p, sv, t, rhat, y, r, S);
      spmvs += 2;
      ++it;
      const bool check = (it % check_every == 0) || it == max_iter;
      if (check) {
        PHX_HIP(hipMemcpyAsync(s->scal_h, S, sizeof(double) * 16, hipMemcpyDeviceToHost, st));
        PHX_HIP(hipStreamSynchronize(st));
        const double rr = s->scal_h[S_RR];
        relres = sqrt(rr / bb);
        if (!(rr == rr) || !(s->scal_h[S_OMEGA] == s->scal_h[S_OMEGA]) ||
            s->scal_h[((it)&1)] == 0.0) {
          phx_set_error("BiCGStab breakdown at iteration %lld (rho=%g omega=%g rr=%g)",
                        (long long)it, s->scal_h[it & 1], s->scal_h[S_OMEGA], rr);
          rc = PHX_ERR_BREAKDOWN;
          break;
        }
        if (relres <= rtol) break;
      }
      k_update_p<<<vec_grid(n), block, 0, st>>>(n, (int)((it - 1) & 1), r, v, p, S, S);
      k_clear_iter_scalars<<<1, 1, 0, st>>>((int)(it & 1), S);
    }
  }
  PHX_HIP(hipGetLastError());
  // back to full numbering, x = D^-1 y, inactive DoFs = 0 (MUMPS ICNTL(24)=1 semantics)
  double *xfull = x_out;
  double *owned = nullptr;
  if (loc != PHX_DEVICE) { PHX_HIP(hipMalloc(&owned, sizeof(double) * (size_t)s->nfull)); xfull = owned; }
  PHX_HIP(hipMemsetAsync(xfull, 0, sizeof(double) * (size_t)s->nfull, st));
  k_scatter_solution<<<gn, block, 0, st>>>(n, s->perm, s->full_of_active, s->diag, y, xfull);
  PHX_CHECK(phx_end_timing(m, 3));
  if (owned) {
    PHX_HIP(hipMemcpy(x_out, owned, sizeof(double) * (size_t)s->nfull, hipMemcpyDeviceToHost));
    PHX_HIP(hipFree(owned));
  }
  if (stats) {
    stats[0] = (double)it;
    stats[1] = relres;
    stats[2] = m->timings[3];
    stats[3] = (double)spmvs;
  }
  return rc;
}

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

extern "C" int phx_spmv_bench(phx_system *s, int reps, double *out) {
  phx_mesh *m = s->mesh;
  PHX_HIP(hipSetDevice(m->device));
  const int64_t n = s->n;
  hipStream_t st = m->stream;
  double *xs = s->work + 6 * n, *ys = s->work + 7 * n;
  PHX_HIP(hipMemcpyAsync(xs, s->rhs, sizeof(double) * n, hipMemcpyDeviceToDevice, st));
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
