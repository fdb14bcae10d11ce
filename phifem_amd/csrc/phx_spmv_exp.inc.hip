// Timing experiments for the SELL-16 x 4 branch of k_spmv_sell (phx_spmv_bench with PHX_SELL_EXP=<variant>): the same
// arithmetic with different load schedules.  Not used by the solver.
template <int VAR>
__global__ void __launch_bounds__(256)
k_sell16_exp(int64_t nslices, const int64_t *__restrict__ slice_ptr, const int32_t *__restrict__ scol,
             const double *__restrict__ sval, const double *__restrict__ x, double *__restrict__ y,
             const int32_t *__restrict__ rows) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  constexpr int NS = VAR == 3 ? 2 : 1;
  const int64_t s0 = ((int64_t)blockIdx.x * (blockDim.x >> 6) + wave) * NS;
  if (s0 >= nslices) return;
  if constexpr (VAR == 3) {
    const bool two = s0 + 1 < nslices;
    const int64_t bA = slice_ptr[s0], bB = slice_ptr[s0 + 1], bC = two ? slice_ptr[s0 + 2] : bB;
    const int tA = __builtin_amdgcn_readfirstlane((int)((bB - bA) >> 6));
    const int tB = __builtin_amdgcn_readfirstlane((int)((bC - bB) >> 6));
    const int32_t *cA = scol + bA + lane, *cB = scol + bB + lane;
    const double *vA = sval + bA + lane, *vB = sval + bB + lane;
    const int32_t rA = rows[s0 * 16 + (lane & 15)], rB = two ? rows[(s0 + 1) * 16 + (lane & 15)] : -1;
    double aA = 0.0, aB = 0.0;
    const int tm = max(tA, tB);
    for (int j = 0; j < tm; j += 4) {
      int32_t cc[2][4]; double vv[2][4], xs[2][4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (j + q < tA) { cc[0][q] = NT_LOAD(&cA[(j + q) * 64]); vv[0][q] = NT_LOAD(&vA[(j + q) * 64]); }
        if (j + q < tB) { cc[1][q] = NT_LOAD(&cB[(j + q) * 64]); vv[1][q] = NT_LOAD(&vB[(j + q) * 64]); }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (j + q < tA) xs[0][q] = x[cc[0][q]];
        if (j + q < tB) xs[1][q] = x[cc[1][q]];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (j + q < tA) aA = __builtin_fma(vv[0][q], xs[0][q], aA);
        if (j + q < tB) aB = __builtin_fma(vv[1][q], xs[1][q], aB);
      }
    }
    aA += __shfl_xor(aA, 16); aA += __shfl_xor(aA, 32);
    aB += __shfl_xor(aB, 16); aB += __shfl_xor(aB, 32);
    if (lane < 16 && rA >= 0) y[rA] = aA;
    if (lane < 16 && rB >= 0) y[rB] = aB;
    return;
  } else {
    const int64_t s = s0;
    const int64_t base = slice_ptr[s];
    const int trips = __builtin_amdgcn_readfirstlane((int)((slice_ptr[s + 1] - base) >> 6));
    const int32_t *c = scol + base + lane;
    const double *v = sval + base + lane;
    const int32_t rr = rows[s * 16 + (lane & 15)];
    double acc = 0.0;
#define LD(p) ((VAR == 2 || VAR == 4) ? *(p) : NT_LOAD(p))
    if constexpr (VAR == 0 || VAR == 2 || VAR == 7) {
      int j = 0;
      for (; j + 4 <= trips; j += 4) {
        int32_t cc[4]; double vv[4], xs[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { cc[q] = LD(&c[(j + q) * 64]); vv[q] = LD(&v[(j + q) * 64]); }
#pragma unroll
        for (int q = 0; q < 4; ++q) xs[q] = VAR == 7 ? (double)cc[q] : x[cc[q]];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_fma(vv[q], xs[q], acc);
      }
      for (; j < trips; ++j) { const int32_t c1 = LD(&c[j * 64]); acc = __builtin_fma(LD(&v[j * 64]), VAR == 7 ? (double)c1 : x[c1], acc); }
    } else {
      constexpr int R = VAR == 5 ? 16 : 8;
      for (int j = 0; j < trips; j += R) {
        int32_t cc[R]; double vv[R], xs[R];
#pragma unroll
        for (int q = 0; q < R; ++q) if (j + q < trips) { cc[q] = LD(&c[(j + q) * 64]); vv[q] = LD(&v[(j + q) * 64]); }
#pragma unroll
        for (int q = 0; q < R; ++q) if (j + q < trips) xs[q] = x[cc[q]];
#pragma unroll
        for (int q = 0; q < R; ++q) if (j + q < trips) acc = __builtin_fma(vv[q], xs[q], acc);
      }
    }
#undef LD
    acc += __shfl_xor(acc, 16);
    acc += __shfl_xor(acc, 32);
    if (lane < 16 && rr >= 0) y[rr] = acc;
  }
}

static int launch_sell16_exp(phx_system *s, int var, const double *vals, const double *x, double *y) {
  hipStream_t st = s->mesh->stream;
  const dim3 block(256);
  const int64_t per = var == 3 ? 8 : 4;
  const dim3 grid((unsigned)phx_div_up(s->nslices, per));
#define X(V) case V: k_sell16_exp<V><<<grid, block, 0, st>>>(s->nslices, s->slice_ptr, s->sell_col, vals, x, y, s->sell_rows); break;
  switch (var) { X(0) X(1) X(2) X(3) X(4) X(5) X(7) default: return PHX_ERR_VALUE; }
#undef X
  PHX_HIP(hipGetLastError());
  return PHX_OK;
}
