// z direction of the box preconditioner as a TRIDIAGONAL solve (included by phx_precond.inc.hip).
//
// After the sine transforms in x and y the lattice Laplacian decouples into one Toeplitz tridiagonal system
// per (kx, ky):   (lam_x + lam_y) u + c_z tridiag(-1, 2, -1) u = f,   i.e.  T u = f / c_z,
// T = tridiag(-1, beta, -1),  beta = 2 + (lam_x + lam_y) / c_z > 2.   With rho = 1 / r the smaller root of
// r + 1/r = beta,  T = L U + rho e_1 e_1^T  with  L = I - rho S (S: shift down),  U = r (I - rho S^T):
//   w_k = f_k + rho w_{k-1}               (forward,  w_0 = 0)
//   y_k = rho (w_k + y_{k+1})             (backward, y_{N+1} = 0)        y = (L U)^-1 f
//   u_k = y_k - coef g_k,   g = (L U)^-1 e_1,  g_k = (rho^k - E rho^(N+1-k)) / (1 - rho^2),  E = rho^(N+1),
//   coef = rho y_1 / (1 + rho g_1)        (Sherman-Morrison for the corner entry)
// Both recurrences are contractions (rho < 1): stable for every beta > 2 (numpy prototype: error <= eps * cond
// against a dense solve for N = 63 ... 1023, beta - 2 = 1e-6 ... 6).  Constant coefficients make them trivially
// parallel: a column of N values is cut into P chunks of C, one wavefront per chunk with the chunk in REGISTERS
// (lane = column: every global access is a coalesced row of 64 columns); the chunk-end carries travel through
// LDS (P values per column) and enter the next chunk as  rho^j x carry.   One read and one write of the
// lattice, no transform, no division per point: the pass runs at the speed of a copy, where the sine
// transform pair it replaces (forward, 1/lambda, inverse in LDS) was the slowest pass of the preconditioner.
// The same recurrences continue ACROSS slabs of a partitioned box (W_in / Y_in below): the z coupling between
// GPUs is solved exactly from one small all-gather of two carries per column (phx_dist.inc.hip).
//
// MODE 0: one rank, everything in one launch.  MODE 1: carries only (Wl = w at the last owned plane, Yl = y at
// the first, both with zero inflow; no lattice store).  MODE 2: with inflows W_in = true w below the first owned
// plane, Y_in = true y above the last, y1 = true y_1 of the global column.

__device__ __forceinline__ double dpow(double x, int n) {  // x^n, n >= 0 (square and multiply)
  double r = 1.0;
  while (n > 0) {
    if (n & 1) r *= x;
    x *= x;
    n >>= 1;
  }
  return r;
}

struct TriArgs {
  int m0, m1, pitch;        // lattice columns: e = x + pitch y, x < m0 (x = m0 .. pitch-1: padding)
  int64_t plane;            // elements between consecutive z planes
  int nloc;                 // planes held here
  int k0;                   // global (1-based) index of the first local plane
  int N;                    // planes of the global column
  double cz, scale_xy;      // c_z, (2 / L0)(2 / L1)
  const double *lamx, *lamy;
  const double *w_in, *y_in, *y1;   // MODE 2, one value per column e
  double *wl, *yl;                  // MODE 1
  int cl;                   // rows per chunk (<= C, the register array of the kernel): ceil(nloc / chunks), so the
                            // chunks are balanced (194 planes: 8 x 25 rows instead of 6 x 32 + 2 + 0)
};

// NT = threads of the block (64 x chunks): with 512 threads the compiler may keep a chunk of 32 rows in registers
// (__launch_bounds__(1024) capped it at 128 VGPRs: C = 32 spilled 36 registers, C = 48 145 -- the pass ran at 3.1 TB/s on the
// 768 x 768 x 194 lattice and at 1.9 TB/s on 384 x 384 x 354 against 4.8 TB/s where C = 8)
// CW = columns of a block (64, or 32: the two halves of a wavefront then take different chunks of the same 32 columns --
// twice the chunks for long columns, rows of 256 bytes).
template <typename T, int C, int MODE, int NT, int CW = 64>
__global__ void __launch_bounds__(NT)
k_tri_z(TriArgs a, T *__restrict__ G) {
  extern __shared__ double tri_lds[];
  constexpr int SUB = 64 / CW;
  const int lane = (int)(threadIdx.x & 63) % CW, c = (int)(threadIdx.x >> 6) * SUB + (int)(threadIdx.x & 63) / CW;
  const int P = (int)(blockDim.x >> 6) * SUB;
  const int64_t e = (int64_t)blockIdx.x * CW + lane;
  const int x = (int)(e % a.pitch), yy = (int)(e / a.pitch);
  const bool colok = x < a.m0 && yy < a.m1;
  double *sW = tri_lds, *sY = tri_lds + (size_t)P * CW;       // [P][CW] each
  const int CL = a.cl;
  const int r0 = c * CL, len = max(0, min(CL, a.nloc - r0));   // local rows r0 .. r0 + len - 1
  // ---- column constants
  double rho = 0.0, om = 1.0;  // om = 1 - rho^2
  if (colok) {
    const double t = (a.lamx[x + 1] + a.lamy[yy + 1]) / a.cz;   // beta - 2 > 0
    const double sq = sqrt(t * (t + 4.0));                       // r - rho
    rho = 2.0 / (2.0 + t + sq);
    om = rho * sq;
  }
  double v[C];
  {
    const T *gp = G + ((int64_t)r0 * a.plane + e);
#pragma unroll
    for (int i = 0; i < C; ++i) {
      v[i] = (colok && i < len) ? (double)*gp : 0.0;
      gp += a.plane;
    }
  }
  // ---- forward, chunk-local
  double acc = 0.0;
#pragma unroll
  for (int i = 0; i < C; ++i) {
    if (i < len) { acc = __builtin_fma(rho, acc, v[i]); v[i] = acc; }
  }
  sW[c * CW + lane] = acc;
  const double rhoC = dpow(rho, CL);
  __syncthreads();
  // true w below this chunk: inflow of the rank, then the chunks before this one (all of full length CL)
  double win = MODE == 2 && colok ? a.w_in[e] : 0.0;
  for (int j = 0; j < c; ++j) win = __builtin_fma(rhoC, win, sW[j * CW + lane]);
  {
    double pw = win;
#pragma unroll
    for (int i = 0; i < C; ++i) {
      pw *= rho;
      if (i < len) v[i] += pw;
    }
  }
  // ---- backward, chunk-local (on the true w)
  acc = 0.0;
#pragma unroll
  for (int i = C - 1; i >= 0; --i) {
    if (i < len) { acc = rho * (v[i] + acc); v[i] = acc; }
  }
  sY[c * CW + lane] = acc;
  __syncthreads();
  // true y above this chunk, and y at the first local plane: chain the chunks from the top (a chunk of length
  // l passes an inflow on as rho^l x inflow; only the last non-empty chunk can be short)
  double yin = MODE == 2 && colok ? a.y_in[e] : 0.0, ynext = 0.0;
  for (int j = P - 1; j >= 0; --j) {
    if (j == c) ynext = yin;
    const int lj = max(0, min(CL, a.nloc - j * CL));
    yin = sY[j * CW + lane] + (lj == CL ? rhoC : dpow(rho, lj)) * yin;
  }
  if (MODE == 1) {
    // carries with zero inflow: w at the last local plane = what the chunk scan hands on at the top
    if (c == 0 && colok) {
      double wtop = 0.0;
      for (int j = 0; j < P; ++j) {
        const int lj = max(0, min(CL, a.nloc - j * CL));
        wtop = sW[j * CW + lane] + (lj == CL ? rhoC : dpow(rho, lj)) * wtop;
      }
      a.wl[e] = wtop;
      a.yl[e] = yin;
    }
    return;
  }
  if (!colok) return;
  // ---- corner correction and store
  const double y1 = MODE == 2 ? a.y1[e] : yin;
  const double E = dpow(rho, a.N + 1);
  const double iom = 1.0 / om;
  const double g1 = (rho - E * E / rho) * iom;
  const double coef = rho * y1 / (1.0 + rho * g1) * iom;       // includes 1 / (1 - rho^2) of g_k
  const double s = a.scale_xy / a.cz;
  const int kend = a.k0 + r0 + len - 1;                        // global index of this chunk's last row
  double ak = dpow(rho, kend), bk = dpow(rho, a.N + 1 - kend) * E, bp = ynext;
  const double r = 1.0 / rho;
  T *gp = G + ((int64_t)(r0 + len - 1) * a.plane + e);
#pragma unroll
  for (int i = C - 1; i >= 0; --i) {
    if (i < len) {
      bp *= rho;
      *gp = (T)(s * (v[i] + bp - coef * (ak - bk)));
      gp -= a.plane;
      ak *= r;
      bk *= rho;
    }
  }
}

// c_z = 0 (2-D lattices carried through the 3-D kernels with one real plane): the z direction decouples
template <typename T>
__global__ void k_scale_xy(TriArgs a, T *__restrict__ G) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int x = (int)(e % a.pitch), yy = (int)(e / a.pitch);
  if (x >= a.m0 || yy >= a.m1) return;
  const double f = a.scale_xy / (a.lamx[x + 1] + a.lamy[yy + 1]);
  for (int k = 0; k < a.nloc; ++k) G[(int64_t)k * a.plane + e] = (T)(f * (double)G[(int64_t)k * a.plane + e]);
}

// interface recurrences of the slab-exact solve, one thread per column: gathered[r] = {Wl_r[ncol], Yl_r[ncol]} of
// rank r (zero-inflow carries), planes[r] = planes of rank r.  For this rank:
//   W_r = Wl_r + rho^len_r W_{r-1};   Y_r = Yl_r + kappa_r W_{r-1} + rho^len_r Y_{r+1},
//   kappa_r = rho^2 (1 - rho^(2 len_r)) / (1 - rho^2)   (what an inflow W contributes to y at the rank's first plane)
struct TriRanks { int nranks, rank; int planes[16]; };
__global__ void k_tri_interface(TriArgs a, TriRanks R, const double *__restrict__ gathered, int64_t ncol,
                                double *__restrict__ w_in, double *__restrict__ y_in, double *__restrict__ y1) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= ncol) return;
  const int x = (int)(e % a.pitch), yy = (int)(e / a.pitch);
  if (x >= a.m0 || yy >= a.m1) { w_in[e] = 0.0; y_in[e] = 0.0; y1[e] = 0.0; return; }
  const double t = (a.lamx[x + 1] + a.lamy[yy + 1]) / a.cz;
  const double sq = sqrt(t * (t + 4.0));
  const double rho = 2.0 / (2.0 + t + sq), om = rho * sq;
  double W[17];   // W[r + 1] = true w at the last plane of rank r
  W[0] = 0.0;
  for (int r = 0; r < R.nranks; ++r)
    W[r + 1] = gathered[(2 * (int64_t)r) * ncol + e] + dpow(rho, R.planes[r]) * W[r];
  double Y = 0.0, ymine = 0.0;   // Y: true y at the first plane of rank r, chained from the top
  for (int r = R.nranks - 1; r >= 0; --r) {
    if (r == R.rank) ymine = Y;
    const double pl = dpow(rho, R.planes[r]);
    const double kappa = rho * rho * (1.0 - pl * pl) / om;
    Y = gathered[(2 * (int64_t)r + 1) * ncol + e] + kappa * W[r] + pl * Y;
  }
  w_in[e] = W[R.rank];
  y_in[e] = ymine;
  y1[e] = Y;
}

// register rows per lane C, chunks per column P (8 or 16 wavefronts of 64 columns, or 16 wavefronts of 2 x 32 columns) and
// rows per chunk CL for `nloc` planes
static inline void tri_shape(int nloc, int *C, int *P, int *CL) {
  *P = nloc > 384 ? 32 : (nloc > 256 ? 16 : 8);
  *CL = std::max(1, (nloc + *P - 1) / *P);
  const int cand[] = {8, 16, 24, 32, 48, 64};
  *C = 64;
  for (int c : cand) if (c >= *CL) { *C = c; break; }
}

template <typename T, int MODE>
static int tri_launch(const TriArgs &a0, T *G, hipStream_t st) {
  TriArgs a = a0;
  if (a.cz == 0.0) {
    const int64_t ncol = (int64_t)a.pitch * a.m1;
    if (MODE != 1) k_scale_xy<T><<<dim3((unsigned)phx_div_up(ncol, 256)), dim3(256), 0, st>>>(a, G);
    PHX_HIP(hipGetLastError());
    return PHX_OK;
  }
  int C, P, CL;
  tri_shape(a.nloc, &C, &P, &CL);
  PHX_REQUIRE(a.nloc <= C * P, PHX_ERR_VALUE, "tridiagonal z pass: %d planes exceed %d x %d", a.nloc, P, C);
  a.cl = CL;
  const int64_t ncol = (int64_t)a.pitch * a.m1;
  const int cw = P == 32 ? 32 : 64;
  const dim3 grid((unsigned)phx_div_up(ncol, cw)), block((unsigned)(P == 8 ? 512 : 1024));
  const size_t lds = sizeof(double) * 2 * (size_t)cw * (size_t)P;
  if (P == 8) {
    switch (C) {
      // (up to 24 rows the 128-register cap of 1024 threads costs nothing and keeps four waves per SIMD)
      case 8: k_tri_z<T, 8, MODE, 1024><<<grid, block, lds, st>>>(a, G); break;
      case 16: k_tri_z<T, 16, MODE, 1024><<<grid, block, lds, st>>>(a, G); break;
      case 24: k_tri_z<T, 24, MODE, 1024><<<grid, block, lds, st>>>(a, G); break;
      default: k_tri_z<T, 32, MODE, 512><<<grid, block, lds, st>>>(a, G); break;
    }
  } else if (P == 16) {
    k_tri_z<T, 24, MODE, 1024><<<grid, block, lds, st>>>(a, G);   // 257 .. 384 planes
  } else {
    switch (C) {
      case 8: case 16: case 24: k_tri_z<T, 24, MODE, 1024, 32><<<grid, block, lds, st>>>(a, G); break;   // .. 768 planes
      case 32: k_tri_z<T, 32, MODE, 1024, 32><<<grid, block, lds, st>>>(a, G); break;
      case 48: k_tri_z<T, 48, MODE, 1024, 32><<<grid, block, lds, st>>>(a, G); break;
      default: k_tri_z<T, 64, MODE, 1024, 32><<<grid, block, lds, st>>>(a, G); break;
    }
  }
  PHX_HIP(hipGetLastError());
  return PHX_OK;
}
