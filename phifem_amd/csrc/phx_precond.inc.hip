// Fictitious-domain preconditioner for systems assembled on a Kuhn box (included by phx_solve.hip).
//
// The u-u block of the phi-FEM matrix is, away from Gamma_h, the P1 stiffness matrix of the uniform
// lattice = the 7-point finite-difference Laplacian  K = cx T_x + cy T_y + cz T_z,  T = tridiag(-1,2,-1),
// c_a = h_b h_c / h_a.  M^-1 = R K_box^-1 R^T (R: restriction of a lattice box around the active vertices
// to the active u DoFs; homogeneous Dirichlet on the box faces) makes the Krylov iteration count
// independent of h (CPU prototype, sphere: 43/49/50/51/49 iterations for n = 16..64 against
// 78/121/164/194/230 with Jacobi); the p block keeps Jacobi.  K_box is diagonalised by the type-I
// discrete sine transform in every axis:  u = S_x S_y S_z diag(1/lambda) S_z S_y S_x r.
//
// DST-I of two real lines a, b at a time through ONE complex FFT of length L (not 2L): with
//   y_j = sin(pi j / L) (x_j + x_{L-j}) + (x_j - x_{L-j}) / 2,   w = y^a + i y^b,   W = FFT_L(w),
//   Y^a_k = (W_k + conj W_{L-k}) / 2,  Y^b_k = (W_k - conj W_{L-k}) / (2i)
// the sine coefficients F_k = sum_j x_j sin(pi j k / L) are  F_{2k} = -Im Y_k  and
// F_{2k+1} = F_{2k-1} + Re Y_k  (F_1 = Re Y_0 / 2): a prefix sum, done per thread and across the threads
// of the pair in LDS.  The FFT is a Stockham autosort in LDS (radices 8/4/2, a radix-3 stage last so that
// every sub-transform size is a power of two), twiddles from a table.  x lines are contiguous; y and z lines
// are walked as tiles of W adjacent columns so that every global access is a contiguous run of W doubles.
// Five passes over the lattice per application: x (gather from the Krylov vector), y, z (forward +
// 1/lambda + inverse fused in LDS), y, x (scatter).  Templated on the precision of lattice and transforms:
// f64 is the default; f32 is an opt-in (faster, but BiCGStab is not a flexible method and the half-length
// transform amplifies rounding by O(L): erratic on some problems).
#include <limits.h>
#include <math.h>

#include <map>
#include <tuple>

#define PHX_PRECOND_MARGIN 4   // lattice planes between the active vertices and the Dirichlet faces of the box

struct DstPlan {
  int L = 0, nstage = 0, pairs = 0, tp = 0;  // pairs per block, threads per pair (L / 8)
  int slot = 0;                 // lanes reserved per pair: tp, or 32 / 64 when the pair lives inside one wave
  int wave = 0;                 // 1: pairs never straddle a wavefront (wave-local synchronisation)
  int radix[8], pw[8], tws[8];  // per stage: radix R, sub-transform size p so far (a power of two), L / (p R)
  int scr = 0;                  // scan scratch per pair (complex doubles)
  int tab_off = 0;              // offset (complex values) of the table copies in LDS
  int lds_elems = 0;            // complex values of dynamic LDS per block
  double2 *tw = nullptr;        // device, exp(-2 pi i j / L), j < L
  double *sintab = nullptr;     // device, sin(pi j / L), j <= L / 2
  float2 *tw_f = nullptr;       // the same tables rounded to f32
  float *sintab_f = nullptr;
};

static std::map<std::pair<int, int>, DstPlan> g_dst_plans;  // (device, L or -L for f32 tiles) -> plan

// L = 2^a 3^b, b <= 1, 64 <= L <= 1024, 64 | L
static int dst_pick_length(int64_t need) {
  const int cand[] = {64, 128, 192, 256, 384, 512, 768, 1024};  // L / R is a multiple of 8 in every stage
  for (int c : cand)
    if (c >= need) return c;
  return -1;
}

// LDS index padding: one extra 16-byte element after every 8, so that the stride-8 / stride-64 write
// patterns of the first Stockham stages spread over the banks (unpadded: up to 32-way conflicts)
#define ZP(n) ((n) + ((n) >> 3))
#define ZLEN(N) ((N) + ((N) >> 3) + 2)  // even: the scan scratch behind it holds doubles

static int dst_get_plan(int device, int L, bool f32, DstPlan *out) {
  auto key = std::make_pair(device, f32 ? -L : L);
  auto it = g_dst_plans.find(key);
  if (it != g_dst_plans.end()) { *out = it->second; return PHX_OK; }
  DstPlan P;
  P.L = L;
  int rest = L;
  P.nstage = 0;
  // radix 3 (if any) goes LAST so that every stage's p is a power of two (k = i & (p - 1))
  const bool three = rest % 3 == 0;
  if (three) rest /= 3;
  while (rest % 8 == 0) { P.radix[P.nstage++] = 8; rest /= 8; }
  if (rest % 4 == 0) { P.radix[P.nstage++] = 4; rest /= 4; }
  if (rest % 2 == 0) { P.radix[P.nstage++] = 2; rest /= 2; }
  PHX_REQUIRE(rest == 1, PHX_ERR_VALUE, "unsupported transform length %d", L);
  if (three) P.radix[P.nstage++] = 3;
  for (int st = 0, pp = 1; st < P.nstage; ++st) {
    P.pw[st] = pp;
    P.tws[st] = L / (pp * P.radix[st]);
    pp *= P.radix[st];
  }
  P.tp = L / 8;
  // A pair whose L / 8 threads fill a 32- or 64-lane slot of one wavefront synchronises wave-locally inside
  // a transform, and the block can then be as wide as the memory system likes: W = 2 pairs adjacent columns
  // per strided tile (memory phase alone at 192^3, f32: 32-byte rows 31 us, 64-byte 18.9, 128-byte 12.6).
  // Measured, block -> wave mode (x / y / z pass, us): L = 256: 42/66/88 -> 38/43/70; 512x512x128 (y): 263 -> 86.
  // Only where no lane idles (L = 256, 512): at L = 192 / 384 / 128 the 25-50 % empty slot lanes cost as
  // much issue time as the barriers save (the transforms are VALU/LDS-issue bound).
  P.wave = (P.tp == 32 || P.tp == 64 || (!f32 && P.tp == 24)) ? 1 : 0;  // f64, L = 192: 190 -> 185 us per application
  if (const char *e = getenv("PHX_DST_WAVE")) P.wave = (P.tp <= 64) && atoi(e) != 0;  // tuning aid
  P.slot = P.wave ? (P.tp <= 32 ? 32 : 64) : P.tp;
  const int el = f32 ? (int)sizeof(float2) : (int)sizeof(double2);
  // long f64 transforms take (almost) a whole CU's LDS per block to reach 128-byte tile rows
  // (768 x 768 x 192 lattice, y pass: 80 KB 1057 us, 156 KB 669 us)
  int budget = (P.wave ? 64 : (L >= 768 ? (f32 ? 80 : 156) : (f32 ? 20 : 40))) * 1024;
  int max_threads = P.wave ? 1024 : (L >= 768 ? 1024 : 512);
  int max_pairs = P.wave ? (f32 ? 16 : 8) : 1024;
  if (const char *e = getenv("PHX_DST_LDS_KB")) budget = atoi(e) > 0 ? atoi(e) * 1024 : budget;  // tuning aids
  if (const char *e = getenv("PHX_DST_PAIRS")) max_pairs = atoi(e) > 0 ? atoi(e) : max_pairs;
  P.pairs = 1;
  while ((2 * P.pairs * ZLEN(L) + 2 * L) * el <= budget && 2 * P.pairs * P.slot <= max_threads &&
         2 * P.pairs <= max_pairs)
    P.pairs *= 2;
  // scan scratch, in complex values of the transform type (the scan runs in f64); none in wave mode (shuffles)
  P.scr = P.wave ? 0 : 2 * (P.tp + (P.tp + 7) / 8 + 1);
  P.tab_off = P.pairs * (ZLEN(L) + P.scr);            // LDS copies of the tables: twiddles, then sines
  P.lds_elems = P.tab_off + L + (L / 2 + 2 + 1) / 2;
  std::vector<double2> tw((size_t)L);
  const long double pi = 3.141592653589793238462643383279502884L;
  for (int j = 0; j < L; ++j) {
    const long double a = -2.0L * pi * (long double)j / (long double)L;
    tw[j] = make_double2((double)cosl(a), (double)sinl(a));
  }
  std::vector<double> st((size_t)L / 2 + 1);
  for (int j = 0; j <= L / 2; ++j) st[j] = (double)sinl(pi * (long double)j / (long double)L);
  PHX_HIP(hipMalloc(&P.tw, sizeof(double2) * tw.size()));  // live as long as the process (a few KB per length)
  PHX_HIP(hipMalloc(&P.sintab, sizeof(double) * st.size()));
  PHX_HIP(hipMemcpy(P.tw, tw.data(), sizeof(double2) * tw.size(), hipMemcpyHostToDevice));
  PHX_HIP(hipMemcpy(P.sintab, st.data(), sizeof(double) * st.size(), hipMemcpyHostToDevice));
  std::vector<float2> twf(tw.size());
  std::vector<float> stf(st.size());
  for (size_t j = 0; j < tw.size(); ++j) twf[j] = make_float2((float)tw[j].x, (float)tw[j].y);
  for (size_t j = 0; j < st.size(); ++j) stf[j] = (float)st[j];
  PHX_HIP(hipMalloc(&P.tw_f, sizeof(float2) * twf.size()));
  PHX_HIP(hipMalloc(&P.sintab_f, sizeof(float) * stf.size()));
  PHX_HIP(hipMemcpy(P.tw_f, twf.data(), sizeof(float2) * twf.size(), hipMemcpyHostToDevice));
  PHX_HIP(hipMemcpy(P.sintab_f, stf.data(), sizeof(float) * stf.size(), hipMemcpyHostToDevice));
  g_dst_plans[key] = P;
  *out = P;
  return PHX_OK;
}

// complex pair of T (float: 8-byte, double: 16-byte LDS accesses)
template <typename T> struct alignas(2 * sizeof(T)) C2 { T x, y; };
template <typename T> __device__ __forceinline__ C2<T> mk(T x, T y) { C2<T> r; r.x = x; r.y = y; return r; }
template <typename T> __device__ __forceinline__ C2<T> cmul(C2<T> a, C2<T> b) {
  return mk<T>(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
template <typename T> __device__ __forceinline__ C2<T> cadd(C2<T> a, C2<T> b) { return mk<T>(a.x + b.x, a.y + b.y); }
template <typename T> __device__ __forceinline__ C2<T> csub(C2<T> a, C2<T> b) { return mk<T>(a.x - b.x, a.y - b.y); }
template <typename T> __device__ __forceinline__ C2<T> mul_mi(C2<T> a) { return mk<T>(a.y, -a.x); }  // a * (-i)

template <typename T> __device__ __forceinline__ void dft2(C2<T> *v) {
  const C2<T> a = v[0], b = v[1];
  v[0] = cadd(a, b); v[1] = csub(a, b);
}
template <typename T> __device__ __forceinline__ void dft3(C2<T> *v) {
  const T s = T(0.86602540378443864676), hf = T(0.5);  // sin(2 pi / 3)
  const C2<T> t = cadd(v[1], v[2]), d = csub(v[1], v[2]);
  const C2<T> m = mk<T>(v[0].x - hf * t.x, v[0].y - hf * t.y);
  const C2<T> r = mk<T>(s * d.y, -s * d.x);  // -i s d
  v[0] = cadd(v[0], t); v[1] = cadd(m, r); v[2] = csub(m, r);
}
template <typename T> __device__ __forceinline__ void dft4(C2<T> *v) {
  const C2<T> t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]);
  const C2<T> t2 = cadd(v[1], v[3]), t3 = mul_mi(csub(v[1], v[3]));
  v[0] = cadd(t0, t2); v[1] = cadd(t1, t3); v[2] = csub(t0, t2); v[3] = csub(t1, t3);
}
template <typename T> __device__ __forceinline__ void dft8(C2<T> *v) {
  C2<T> e[4] = {v[0], v[2], v[4], v[6]}, o[4] = {v[1], v[3], v[5], v[7]};
  dft4(e); dft4(o);
  const T h = T(0.70710678118654752440);
  o[1] = mk<T>(h * (o[1].x + o[1].y), h * (o[1].y - o[1].x));    // * (1 - i)/sqrt2
  o[2] = mul_mi(o[2]);
  o[3] = mk<T>(h * (o[3].y - o[3].x), -h * (o[3].x + o[3].y));   // * (-1 - i)/sqrt2
  for (int k = 0; k < 4; ++k) { v[k] = cadd(e[k], o[k]); v[k + 4] = csub(e[k], o[k]); }
}

// tables of the plan in the precision of the transform
template <typename T> struct PlanTab;
template <> struct PlanTab<double> {
  static __device__ __forceinline__ const C2<double> *tw(const DstPlan &P) { return reinterpret_cast<const C2<double> *>(P.tw); }
  static __device__ __forceinline__ const double *sn(const DstPlan &P) { return P.sintab; }
};
template <> struct PlanTab<float> {
  static __device__ __forceinline__ const C2<float> *tw(const DstPlan &P) { return reinterpret_cast<const C2<float> *>(P.tw_f); }
  static __device__ __forceinline__ const float *sn(const DstPlan &P) { return P.sintab_f; }
};

// Synchronisation among the threads of ONE pair.  WAVE: the pair's threads sit inside a single wavefront
// (plan.slot = 32 or 64 lanes per pair), LDS operations of a wave execute in order, so a compiler fence is
// all that is needed; otherwise the pair spans wavefronts and the whole block meets at a barrier.
template <bool WAVE> __device__ __forceinline__ void psync() {
  if (WAVE) {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
  } else {
    __syncthreads();
  }
}

// Length-specialised kernels (template parameter LL > 0): transform length, threads per pair and lanes per pair are
// compile-time constants (the divisions and multiplications by them fold); the stage schedule stays in the plan
// (kernel argument, scalar registers).  LL = 0: everything from the run-time plan.  [Overwriting the plan fields
// of the by-value kernel argument instead sent its arrays to scratch memory: 29 -> 46 us per pass.]
template <int LL> __device__ __forceinline__ int plan_L(const DstPlan &P) { return LL > 0 ? LL : P.L; }
template <int LL> __device__ __forceinline__ int plan_tp(const DstPlan &P) { return LL > 0 ? LL / 8 : P.tp; }
template <int LL, bool WAVE> __device__ __forceinline__ int plan_slot(const DstPlan &P) {
  return LL > 0 ? (WAVE ? (LL / 8 <= 32 ? 32 : 64) : LL / 8) : P.slot;
}

// one Stockham stage of radix R on the L-point sequence `z` of this pair: `t` = thread within the pair.
// All inputs are read into registers, the block synchronises, then the outputs are written in place.
template <typename T, int R, bool WAVE>
__device__ __forceinline__ void fft_stage(C2<T> *z, int nb, int tp, int t, int p, int tws,
                                          const C2<T> *__restrict__ tw) {
  constexpr int MAXB = (8 + R - 1) / R;
  C2<T> u[MAXB][R];
  // padded LDS indices are linear in q: nb and p (p > 1) are multiples of 8, the first stage (p = 1) has
  // radix 8, so ZP(i + q nb) = ZP(i) + q ZP'(nb) and ZP(j + q p) = ZP(j) + q ZP'(p)   (ZP'(n) = n + n / 8)
  const int nbp = nb + (nb >> 3);
  const int pp = p == 1 ? 1 : p + (p >> 3);
#pragma unroll
  for (int b = 0; b < MAXB; ++b) {
    const int i = t + b * tp;
    if (i < nb) {
      const int k = i & (p - 1);
      const int step = tws * k;  // twiddle exponent of q = 1; q * step < L for q < R
      // running indices: integer multiplies are quarter rate, the transforms are issue bound
      const C2<T> *zr = z + ZP(i);
      const C2<T> *twq = tw;
#pragma unroll
      for (int q = 0; q < R; ++q) {
        C2<T> w = *zr;
        // every stage behind the first multiplies unconditionally (tw[0] = 1 exactly): a per-lane `k > 0` test put each
        // of the R - 1 twiddle reads behind its own branch and its own LDS wait
        if (q > 0 && p > 1) w = cmul(w, *twq);
        u[b][q] = w;
        zr += nbp;
        twq += step;
      }
      if constexpr (R == 8) dft8(u[b]);
      else if constexpr (R == 4) dft4(u[b]);
      else if constexpr (R == 3) dft3(u[b]);
      else dft2(u[b]);
    }
  }
  psync<WAVE>();
#pragma unroll
  for (int b = 0; b < MAXB; ++b) {
    const int i = t + b * tp;
    if (i < nb) {
      const int k = i & (p - 1);
      const int j = (i - k) * R + k;
      C2<T> *zw = z + ZP(j);
#pragma unroll
      for (int q = 0; q < R; ++q) { *zw = u[b][q]; zw += pp; }
    }
  }
  psync<WAVE>();
}

// forward complex FFT of this pair's sequence `z` in LDS.  Every thread of the block must call it (block
// barriers inside); threads of an idle pair slot pass live = false and do no work.
// stage schedule of dst_get_plan as constant expressions: radix 3 (if any) last, then 8s, a 4, a 2
constexpr int sched_n(int L) {
  int r = L % 3 == 0 ? L / 3 : L, n = 0;
  while (r % 8 == 0) { r /= 8; ++n; }
  if (r % 4 == 0) { r /= 4; ++n; }
  if (r % 2 == 0) { r /= 2; ++n; }
  return n + (L % 3 == 0 ? 1 : 0);
}
constexpr int sched_radix(int L, int s) {
  int r = L % 3 == 0 ? L / 3 : L, n = 0;
  while (r % 8 == 0) { if (n == s) return 8; r /= 8; ++n; }
  if (r % 4 == 0) { if (n == s) return 4; r /= 4; ++n; }
  if (r % 2 == 0) { if (n == s) return 2; r /= 2; ++n; }
  return 3;
}
constexpr int sched_pw(int L, int s) {
  int p = 1;
  for (int q = 0; q < s; ++q) p *= sched_radix(L, q);
  return p;
}
template <typename T, bool WAVE, int LL, int S>
__device__ __forceinline__ void fft_stages_c(C2<T> *z, int tt, const C2<T> *tw) {
  if constexpr (S < sched_n(LL)) {
    constexpr int R = sched_radix(LL, S), p = sched_pw(LL, S), tws = LL / (p * R);
    fft_stage<T, R, WAVE>(z, LL / R, LL / 8, tt, p, tws, tw);   // every index expression folds
    fft_stages_c<T, WAVE, LL, S + 1>(z, tt, tw);
  }
}

template <typename T, bool WAVE, int LL>
__device__ __forceinline__ void fft_pairs(C2<T> *z, const DstPlan &P, int t, bool live, const C2<T> *tw) {
  const int L = plan_L<LL>(P), tp = plan_tp<LL>(P);
  const int tt = live ? t : L;  // empty butterfly range
  if constexpr (LL > 0) {
    fft_stages_c<T, WAVE, LL, 0>(z, tt, tw);
    return;
  }
  for (int s = 0; s < P.nstage; ++s) {
    const int R = P.radix[s], p = P.pw[s], tws = P.tws[s];
    if (R == 8) fft_stage<T, 8, WAVE>(z, L >> 3, tp, tt, p, tws, tw);
    else if (R == 4) fft_stage<T, 4, WAVE>(z, L >> 2, tp, tt, p, tws, tw);
    else if (R == 3) fft_stage<T, 3, WAVE>(z, tws * p, tp, tt, p, tws, tw);  // L / 3 butterflies
    else fft_stage<T, 2, WAVE>(z, L >> 1, tp, tt, p, tws, tw);
  }
}

// DPP move of a complex double: lane l reads the value of the lane CTRL selects (0x110 + n: row_shr n, 0x138: wave_shr 1,
// 0x142 / 0x143: row_bcast 15 / 31), 0 where there is none or where ROWS masks the row out
template <int CTRL, int ROWS = 0xf>
__device__ __forceinline__ C2<double> dpp_c2(C2<double> v) {
  int w[4] = {__double2loint(v.x), __double2hiint(v.x), __double2loint(v.y), __double2hiint(v.y)};
#pragma unroll
  for (int q = 0; q < 4; ++q) w[q] = __builtin_amdgcn_update_dpp(0, w[q], CTRL, ROWS, 0xf, true);
  return mk<double>(__hiloint2double(w[1], w[0]), __hiloint2double(w[3], w[2]));
}

// In:  w[ZP(j)] = (a_j, b_j), j = 1 .. L-1 (w[0] arbitrary).   Out: w[ZP(k)] = (F^a_k, F^b_k), k = 1 .. L-1,
// F_k = sum_j x_j sin(pi j k / L).  `scr`: tp + tp/8 + 1 complex values of scan scratch of this pair.
// Block-wide barriers inside: every thread of the block calls it.
// PRE = false: the caller has already written the folded sequence y_j = sin(pi j / L)(x_j + x_{L-j}) + (x_j - x_{L-j}) / 2
// (w[0] = 0, w[H] = 2 x_H) -- the loaders of the x and y passes fold in registers, which saves a read and a write of
// every LDS element and one synchronisation per transform.
template <typename T, bool WAVE, int LL, bool PRE = true>
__device__ __forceinline__ void dst_core(C2<T> *w, C2<T> *scr, const DstPlan &P, int t, bool live,
                                         const C2<T> *tw, const T *sn) {
  const int L = plan_L<LL>(P), tp = plan_tp<LL>(P), H = L >> 1, slot = plan_slot<LL, WAVE>(P);
  const T hf = T(0.5);
  if (PRE && live) {
    for (int j = 1 + t; j < H; j += tp) {
      const C2<T> X = w[ZP(j)], Y = w[ZP(L - j)];
      const T s = sn[j];
      const C2<T> e = mk<T>(s * (X.x + Y.x), s * (X.y + Y.y));
      const C2<T> o = mk<T>(hf * (X.x - Y.x), hf * (X.y - Y.y));
      w[ZP(j)] = cadd(e, o);
      w[ZP(L - j)] = csub(e, o);
    }
    if (t == 0) {
      w[0] = mk<T>(T(0), T(0));
      const C2<T> X = w[ZP(H)];
      w[ZP(H)] = mk<T>(X.x + X.x, X.y + X.y);
    }
  }
  if (PRE) psync<WAVE>();
  fft_pairs<T, WAVE, LL>(w, P, t, live, tw);
  // thread t owns k = 4 t .. 4 t + 3  (k < L / 2)
  // The running sums are kept in f64 whatever the transform precision: in f32 they are what turns the
  // O(eps log L) error of the FFT into O(eps L) (2-D flower problem: 114-582 erratic iterations and
  // occasional breakdowns with f32 sums against 38 with f64 transforms).
  C2<T> Wk[4], Wm[4];
  C2<double> c[4];
  if (live) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = 4 * t + i;
      Wk[i] = w[ZP(k)];
      Wm[i] = w[ZP(k == 0 ? 0 : L - k)];
    }
  }
  psync<WAVE>();
  C2<double> run = mk<double>(0.0, 0.0);
  if (live) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = 4 * t + i;
      C2<T> R;
      if (k == 0) {
        R = mk<T>(hf * Wk[i].x, hf * Wk[i].y);  // F_1 = Re Y_0 / 2 starts the running sum
      } else {
        R = mk<T>(hf * (Wk[i].x + Wm[i].x), hf * (Wk[i].y + Wm[i].y));
        w[ZP(2 * k)] = mk<T>(-hf * (Wk[i].y - Wm[i].y), hf * (Wk[i].x - Wm[i].x));
      }
      run = cadd(run, mk<double>((double)R.x, (double)R.y));
      c[i] = run;
    }
  }
  C2<double> E = mk<double>(0.0, 0.0);   // sum of the totals of the threads before this one
  if constexpr (WAVE) {
    // the pair sits inside one wavefront: prefix sum of the per-thread totals by shuffles -- no scratch in LDS
    // (which lets a fifth block fit a CU at L = 192), no synchronisation.  Idle lanes carry zeros.
    // Data-parallel-primitive moves instead of ds_bpermute: a VALU move with a lane-shifted source, no LDS round trip,
    // and lanes without a source read 0 (bound_ctrl), which spares the `t >= d` selects.  Inclusive scan inside each
    // row of 16 lanes (row_shr 1, 2, 4, 8), then the row totals cross the rows (row_bcast15 into rows 1 and 3,
    // row_bcast31 into rows 2 and 3 when the pair fills the wave).
    C2<double> inc = run;
    inc = cadd(inc, dpp_c2<0x111>(inc));
    inc = cadd(inc, dpp_c2<0x112>(inc));
    inc = cadd(inc, dpp_c2<0x114>(inc));
    inc = cadd(inc, dpp_c2<0x118>(inc));
    inc = cadd(inc, dpp_c2<0x142, 0xa>(inc));
    if (slot == 64) inc = cadd(inc, dpp_c2<0x143, 0xc>(inc));
    // exclusive prefix = the inclusive one of the lane below (no subtraction: that would cost the low bits)
    const C2<double> ex = dpp_c2<0x138>(inc);   // wave_shr:1
    E = t == 0 ? mk<double>(0.0, 0.0) : ex;
  } else {
    C2<double> *tot = reinterpret_cast<C2<double> *>(scr), *gt = tot + tp;
    if (live) tot[t] = run;
    psync<WAVE>();
    if (live && t < (tp + 7) / 8) {
      C2<double> g = mk<double>(0.0, 0.0);
      for (int q = 8 * t; q < min(8 * t + 8, tp); ++q) g = cadd(g, tot[q]);
      gt[t] = g;
    }
    psync<WAVE>();
    if (live) {
      for (int g = 0; g < (t >> 3); ++g) E = cadd(E, gt[g]);
      for (int q = t & ~7; q < t; ++q) E = cadd(E, tot[q]);
    }
  }
  if (live) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const C2<double> F = cadd(E, c[i]);
      w[ZP(2 * (4 * t + i) + 1)] = mk<T>((T)F.x, (T)F.y);
    }
  }
  psync<WAVE>();
}

// copies the twiddle and sine tables of the plan into LDS (the stage loops read them with LDS latency
// instead of L1/L2 latency); the caller synchronises before the first use
template <typename T>
__device__ __forceinline__ void stage_tables(C2<T> *zs, const DstPlan &P, const C2<T> **tw, const T **sn) {
  C2<T> *ltw = zs + P.tab_off;
  T *lsn = reinterpret_cast<T *>(ltw + P.L);
  const C2<T> *gtw = PlanTab<T>::tw(P);
  const T *gsn = PlanTab<T>::sn(P);
  for (int j = threadIdx.x; j < P.L; j += blockDim.x) ltw[j] = gtw[j];
  for (int j = threadIdx.x; j <= P.L / 2; j += blockDim.x) lsn[j] = gsn[j];
  *tw = ltw;
  *sn = lsn;
}

struct BoxGrid {
  int m[3];          // stored interior points per axis (= L - 1)
  int L[3];          // transform lengths
  int64_t pitch, plane;
  double scale;      // (2/Lx)(2/Ly)(2/Lz): the three inverse transforms
  const double *lam[3];  // c_a (2 - 2 cos(pi k / L_a)), k = 0 .. L_a - 1 (device)
  double c[3];       // coefficient of tridiag(-1, 2, -1) per axis
};

#include "phx_tridiag.inc.hip"

// ---- x lines (contiguous): pair = two consecutive lines of the flattened (y, z) index.
// IO = 1: the input is gathered from the Krylov vector through gmap (solver position of the u DoF at the
// lattice point, -1: none), times dscale when that is given; IO = 2: the result is scattered out the same
// way, times dscale.
// T: precision of the lattice array and of the transform (the Krylov vectors stay f64).
template <typename T, int IO, bool WAVE, int LL = 0>
__global__ void __launch_bounds__(1024)
k_dst_x(BoxGrid g, DstPlan P, T *__restrict__ G, const int32_t *__restrict__ gmap,
        const double *__restrict__ vin, double *__restrict__ vout, const double *__restrict__ dscale,
        const uint8_t *__restrict__ line_any) {
  extern __shared__ double2 zs_raw[];
  C2<T> *zs = reinterpret_cast<C2<T> *>(zs_raw);
  const int PL = plan_L<LL>(P), Ptp = plan_tp<LL>(P), Pslot = plan_slot<LL, WAVE>(P);
  const int pr = threadIdx.x / Pslot, t = threadIdx.x % Pslot;  // slot >= tp lanes per pair
  const int64_t nlines = (int64_t)g.m[1] * g.m[2];
  if ((IO == 1 || IO == 2) && line_any) {
    // x lines that hold no active vertex (63 % of the lattice lies outside the domain): their input is zero (IO = 1)
    // and their output is never gathered (IO = 2).  A block whose lines are all of that kind skips the transform:
    // the forward pass only clears its lines, the backward pass does nothing.  Uniform over the block.
    const int64_t l0 = (int64_t)blockIdx.x * P.pairs * 2, l1 = min(l0 + 2 * P.pairs, nlines);
    bool any = false;
    for (int64_t l = l0; l < l1; ++l) any |= line_any[l] != 0;
    if (!any) return;   // forward: the y pass takes such rows as zero (row_any); backward: nothing to scatter
  }
  const int64_t line0 = ((int64_t)blockIdx.x * P.pairs + pr) * 2;
  const bool live = line0 < nlines && t < Ptp;
  C2<T> *w = zs + (size_t)pr * ZLEN(PL);
  C2<T> *scr = zs + (size_t)P.pairs * ZLEN(PL) + (size_t)pr * P.scr;
  const int mx = g.m[0], L = PL;
  const C2<T> *tw;
  const T *sn;
  stage_tables<T>(zs, P, &tw, &sn);
  int64_t base[2] = {0, 0};
  bool has[2] = {false, false};
  if (live) {
    for (int c = 0; c < 2; ++c) {
      const int64_t l = line0 + c;
      // a line without a mapped point: its rows are never written by the y pass before (backward) and never read by
      // the y pass after (forward) -- it takes part in the pair's transform as zeros
      has[c] = l < nlines && !((IO == 1 || IO == 2) && line_any && !line_any[l]);
      base[c] = has[c] ? (l % g.m[1]) * g.pitch + (l / g.m[1]) * g.plane : 0;
    }
    // the thread takes the PAIRS (j, L - j), j = 1 + t + i tp (4 trips: j = 1 .. L / 2) of both lines and folds them in
    // registers (dst_core<..., PRE = false>); all loads are issued before the first LDS write
    const int H = L >> 1;
    const T *sng = PlanTab<T>::sn(P);
    T va[4], vb[4], ua[4], ub[4], sj[4];   // line a / b at j (v) and at L - j (u)
    if (IO == 1) {
      int32_t qa[4], qb[4], pa[4], pb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j = 1 + t + i * Ptp;
        qa[i] = has[0] ? gmap[base[0] + j - 1] : -1;
        qb[i] = has[1] ? gmap[base[1] + j - 1] : -1;
        pa[i] = has[0] && j < H ? gmap[base[0] + L - j - 1] : -1;
        pb[i] = has[1] && j < H ? gmap[base[1] + L - j - 1] : -1;
        sj[i] = sng[j];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        va[i] = qa[i] >= 0 ? (T)(dscale ? vin[qa[i]] * dscale[qa[i]] : vin[qa[i]]) : T(0);
        vb[i] = qb[i] >= 0 ? (T)(dscale ? vin[qb[i]] * dscale[qb[i]] : vin[qb[i]]) : T(0);
        ua[i] = pa[i] >= 0 ? (T)(dscale ? vin[pa[i]] * dscale[pa[i]] : vin[pa[i]]) : T(0);
        ub[i] = pb[i] >= 0 ? (T)(dscale ? vin[pb[i]] * dscale[pb[i]] : vin[pb[i]]) : T(0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j = 1 + t + i * Ptp;
        va[i] = has[0] ? G[base[0] + j - 1] : T(0);
        vb[i] = has[1] ? G[base[1] + j - 1] : T(0);
        ua[i] = has[0] && j < H ? G[base[0] + L - j - 1] : T(0);
        ub[i] = has[1] && j < H ? G[base[1] + L - j - 1] : T(0);
        sj[i] = sng[j];
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = 1 + t + i * Ptp;
      if (j < H) {
        const C2<T> e = mk<T>(sj[i] * (va[i] + ua[i]), sj[i] * (vb[i] + ub[i]));
        const C2<T> o = mk<T>(T(0.5) * (va[i] - ua[i]), T(0.5) * (vb[i] - ub[i]));
        w[ZP(j)] = cadd(e, o);
        w[ZP(L - j)] = csub(e, o);
      } else {
        w[ZP(H)] = mk<T>(va[i] + va[i], vb[i] + vb[i]);
      }
    }
    if (t == 0) w[0] = mk<T>(T(0), T(0));
  }
  __syncthreads();
  dst_core<T, WAVE, LL, false>(w, scr, P, t, live, tw, sn);
  if (!live) return;
  if (IO == 2) {
    // k = t + 1 + i tp: all map loads, then all scale loads, then the stores
    int32_t qa[8], qb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = t + 1 + i * Ptp;
      qa[i] = k <= mx && has[0] ? gmap[base[0] + k - 1] : -1;
      qb[i] = k <= mx && has[1] ? gmap[base[1] + k - 1] : -1;
    }
    double da[8], db[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      da[i] = qa[i] >= 0 ? (dscale ? dscale[qa[i]] : 1.0) : 0.0;
      db[i] = qb[i] >= 0 ? (dscale ? dscale[qb[i]] : 1.0) : 0.0;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = t + 1 + i * Ptp;
      if (k <= mx) {
        const C2<T> F = w[ZP(k)];
        if (qa[i] >= 0) vout[qa[i]] = (double)F.x * da[i];
        if (qb[i] >= 0) vout[qb[i]] = (double)F.y * db[i];
      }
    }
  } else {
    for (int k = t + 1; k <= mx; k += Ptp) {
      const C2<T> F = w[ZP(k)];
      if (has[0]) G[base[0] + k - 1] = F.x;
      if (has[1]) G[base[1] + k - 1] = F.y;
    }
  }
}

// ---- y / z lines (strided): a block takes W = 2 * pairs adjacent x columns of one `outer` index, so every
// global access is a run of W consecutive values.  AXIS = 1: lines along y (outer = z), AXIS = 2: lines
// along z (outer = y).  SOLVE (z only): forward transform, times scale / lambda, inverse transform, all in LDS.
// row_any (AXIS = 1 only, nullable): row_any[z] = {first, last} row of plane z whose x line holds a mapped lattice
// point (one block-uniform pair: scalar loads and two compares per row; a flag load per row in front of every tile
// load made the pass slower than the traffic it saved).  dir = 1
// (forward, after the gathering x pass): such a row holds nothing -- it is not even written by the x pass -- and is taken
// as zero; dir = 2 (backward, before the scattering x pass): nobody reads such a row, so it is not stored.  37 % of the
// rows of the ball's box: that much less of the pass's traffic.
template <typename T, int AXIS, bool SOLVE, bool WAVE, int LL = 0>
__global__ void __launch_bounds__(1024)
k_dst_s(BoxGrid g, DstPlan P, T *__restrict__ G, const int2 *__restrict__ row_any, int dir) {
  extern __shared__ double2 zs_raw[];
  C2<T> *zs = reinterpret_cast<C2<T> *>(zs_raw);
  const int Ptp = plan_tp<LL>(P), Pslot = plan_slot<LL, WAVE>(P);
  const int pr = threadIdx.x / Pslot, t = threadIdx.x % Pslot;  // slot >= tp lanes per pair
  const int W = 2 * P.pairs, L = plan_L<LL>(P);
  const int mx = g.m[0];
  const int ncb = (mx + W - 1) / W;                 // column blocks
  const int col0 = (int)(blockIdx.x % ncb) * W;
  const int64_t outer = blockIdx.x / ncb;
  const int len = g.m[AXIS];
  const int64_t estride = AXIS == 1 ? g.pitch : g.plane;
  const int64_t base = col0 + outer * (AXIS == 1 ? g.plane : g.pitch);
  const int ncols = min(W, mx - col0);
  const bool live = 2 * pr < ncols && t < Ptp;
  C2<T> *w = zs + (size_t)pr * ZLEN(L);
  C2<T> *scr = zs + (size_t)P.pairs * ZLEN(L) + (size_t)pr * P.scr;
  const C2<T> *tw;
  const T *sn;
  stage_tables<T>(zs, P, &tw, &sn);
  // cooperative tile load: thread -> fixed column, rows row0, row0 + rstep, ...  (blockDim = W tp / 2,
  // rstep = L / 16: 16 trips, all loads in flight before the first LDS write)
  const int tcol = threadIdx.x % W, row0 = threadIdx.x / W, rstep = blockDim.x / W;
  T *wcol = reinterpret_cast<T *>(zs + (size_t)(tcol >> 1) * ZLEN(L)) + (tcol & 1);
  if constexpr (!SOLVE) {
    // plain transform: the thread takes the PAIRS (j, L - j), j = 1 + row0 + i rstep <= L / 2, and folds them in
    // registers (dst_core<..., PRE = false>)
    constexpr int NT = 8;
    const int H = L >> 1;
    T va[NT], vb[NT], sj[NT];
    const T *sng = PlanTab<T>::sn(P);
    int rlo = 0, rhi = len - 1;
    if (AXIS == 1 && row_any && dir == 1) { const int2 iv = row_any[outer]; rlo = iv.x; rhi = iv.y; }
    const bool colok = tcol < ncols;
    const T *g0 = G + (base + tcol);
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int j = 1 + row0 + i * rstep;
      const int ra = j - 1, rb = L - 1 - j;
      const bool ok = j <= H && colok;
      va[i] = (ok && ra >= rlo && ra <= rhi) ? g0[(int64_t)ra * estride] : T(0);
      vb[i] = (ok && j < H && rb >= rlo && rb <= rhi) ? g0[(int64_t)rb * estride] : T(0);
      sj[i] = j <= H ? sng[j] : T(0);
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int j = 1 + row0 + i * rstep;
      if (j < H) {
        const T e = sj[i] * (va[i] + vb[i]), o = T(0.5) * (va[i] - vb[i]);
        wcol[2 * ZP(j)] = e + o;
        wcol[2 * ZP(L - j)] = e - o;
      } else if (j == H) {
        wcol[2 * ZP(H)] = va[i] + va[i];
      }
    }
    if (row0 == 0) wcol[0] = T(0);
  } else {
    T vv[16];
    const T *gp = G + (base + (int64_t)row0 * estride + tcol);
    const int64_t gstep = (int64_t)rstep * estride;
    const bool colok = tcol < ncols;
    int row = row0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      vv[i] = (row < len && colok) ? *gp : T(0);
      gp += gstep;
      row += rstep;
    }
    row = row0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (row < L - 1) wcol[2 * ZP(row + 1)] = vv[i];
      row += rstep;
    }
  }
  __syncthreads();
  dst_core<T, WAVE, LL, SOLVE>(w, scr, P, t, live, tw, sn);
  if (SOLVE) {
    if (live) {
      const double *lx = g.lam[0], *ly = g.lam[1], *lz = g.lam[2];
      const int kx = col0 + 2 * pr + 1;
      const double lxy0 = lx[kx] + ly[outer + 1];
      const double lxy1 = (kx + 1 < g.L[0] ? lx[kx + 1] : lx[kx]) + ly[outer + 1];
      // the divide runs in the precision of the transform (an f64 division costs ~10x an f32 one)
      const T sc = (T)g.scale, l0 = (T)lxy0, l1 = (T)lxy1;
      for (int k = t + 1; k < L; k += Ptp) {
        const C2<T> F = w[ZP(k)];
        const T lzk = (T)lz[k];
        w[ZP(k)] = mk<T>(F.x * sc / (l0 + lzk), F.y * sc / (l1 + lzk));
      }
    }
    psync<WAVE>();
    dst_core<T, WAVE, LL>(w, scr, P, t, live, tw, sn);
  }
  __syncthreads();
  if (tcol < ncols) {
    const T *wc = reinterpret_cast<const T *>(zs + (size_t)(tcol >> 1) * ZLEN(L)) + (tcol & 1);
    T *gp = G + (base + (int64_t)row0 * estride + tcol);
    const int64_t gstep = (int64_t)rstep * estride;
    int rlo = 0, rhi = len - 1;
    if (AXIS == 1 && row_any && dir == 2) { const int2 iv = row_any[outer]; rlo = iv.x; rhi = iv.y; }
    for (int row = row0; row < len; row += rstep) {
      if (row >= rlo && row <= rhi) *gp = wc[2 * ZP(row + 1)];
      gp += gstep;
    }
  }
}

constexpr bool dst_wave_f64(int L) { return L / 8 == 32 || L / 8 == 64 || L / 8 == 24; }
#include "phx_dst_wave.inc.hip"
#include "phx_dst_pair.inc.hip"

// --------------------------------------------------------------------------------------------------
struct phx_box_precond {
  BoxGrid g;
  DstPlan plan[3];
  void *G = nullptr;         // lattice array, f32 or f64
  bool f32 = false;          // precision of the lattice array and the transforms
  int32_t *gmap = nullptr;   // [plane * m2] solver position of the u DoF, -1 none
  double *dscale = nullptr;  // [n] diag of A in solver order (weighted systems: sqrt(diag A * diag K_box))
  double *iscale = nullptr;  // [n] weighted systems only: sqrt(diag K_box / diag A), applied to the input
  const uint8_t *own_ptr = nullptr;  // ownership mask the maps were built for
  int64_t nvec = 0;          // entries of the Krylov vectors the x passes gather from / scatter to
  double *lam[3] = {nullptr, nullptr, nullptr};
  int lo[3] = {0, 0, 0};     // lattice index of the lower Dirichlet face
  // slab-exact mode (multi-GPU): this rank holds `g.m[2]` planes of a GLOBAL column of zN planes starting at the
  // 1-based global index zk0; carries of the tridiagonal recurrences cross ranks through one all-gather
  bool dist = false;
  int zN = 0, zk0 = 1, nranks = 1, rank = 0;
  int planes[16] = {0};      // planes held by every rank
  bool carry_borrowed = false;    // carry_send / carry_recv belong to the driver (torch tensors)
  double *carry_send = nullptr;   // [2][ncol]  {Wl, Yl} of this rank (zero-inflow carries)
  double *carry_recv = nullptr;   // [nranks][2][ncol]
  double *tri_in = nullptr;       // [3][ncol]  {W_in, Y_in, y1}
  uint8_t *line_any = nullptr;  // [m1 * m2] x line holds at least one mapped lattice point (nullptr: all do)
  int2 *line_iv = nullptr;      // [m2] {first, last} row of the plane with such a line ({1, 0}: none); line_any is widened to
                                // the whole interval, so "outside the interval" and "line_any = 0" say the same
  bool ztri = true;          // z direction: tridiagonal solve (default) or forward / inverse sine transform in LDS
  bool rowskip = false;      // an APPLICATION is running (gathering / scattering x passes around the middle passes): rows of
                             // x lines without a mapped point are neither written by the forward passes nor read back
};

static void box_precond_free(phx_box_precond *bp) {
  if (!bp) return;
  (void)phx_free(bp->G); (void)phx_free(bp->gmap); (void)phx_free(bp->dscale); (void)phx_free(bp->iscale); (void)phx_free(bp->line_any); (void)phx_free(bp->line_iv);
  if (!bp->carry_borrowed) { (void)phx_free(bp->carry_send); (void)phx_free(bp->carry_recv); }
  (void)phx_free(bp->tri_in);
  for (int a = 0; a < 3; ++a) if (bp->lam[a]) (void)hipFree(bp->lam[a]);   // only tables the cache did not take
  delete bp;
}

template <typename T, bool WAVE>
static int dst_allow_lds_t() {
  const int bytes = 160 * 1024;  // padded transform data + scan scratch may exceed the 64 KB default
  PHX_HIP(hipFuncSetAttribute((const void *)k_dst_x<T, 0, WAVE>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  PHX_HIP(hipFuncSetAttribute((const void *)k_dst_x<T, 1, WAVE>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  PHX_HIP(hipFuncSetAttribute((const void *)k_dst_x<T, 2, WAVE>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  PHX_HIP(hipFuncSetAttribute((const void *)k_dst_s<T, 1, false, WAVE>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  PHX_HIP(hipFuncSetAttribute((const void *)k_dst_s<T, 2, true, WAVE>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  return PHX_OK;
}
// f64 kernels specialised for the transform length, in the synchronisation mode dst_get_plan picks for it
#define PHX_DST_LENGTHS(X) X(64) X(128) X(192) X(256) X(384) X(512) X(768) X(1024)
template <int LL>
static int dst_allow_lds_len() {
  const int bytes = 160 * 1024;
  constexpr bool WV = dst_wave_f64(LL);
  PHX_HIP(hipFuncSetAttribute((const void *)k_dst_x<double, 0, WV, LL>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  PHX_HIP(hipFuncSetAttribute((const void *)k_dst_x<double, 1, WV, LL>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  PHX_HIP(hipFuncSetAttribute((const void *)k_dst_x<double, 2, WV, LL>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  PHX_HIP(hipFuncSetAttribute((const void *)k_dst_s<double, 1, false, WV, LL>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  return PHX_OK;
}
static int dst_allow_lds() {
  static bool done = false;
  if (done) return PHX_OK;
#define X(L_) PHX_CHECK((dst_allow_lds_len<L_>()));
  PHX_DST_LENGTHS(X)
#undef X
#define X(L_) PHX_CHECK((dst_wave_allow_lds<L_>()));
  PHX_DST_WAVE_LENGTHS(X)
#undef X
#define X(L_) PHX_CHECK((dst_pair_allow_lds<L_>()));
  PHX_DST_PAIR_LENGTHS(X)
#undef X
  PHX_CHECK((dst_allow_lds_t<double, true>()));
  PHX_CHECK((dst_allow_lds_t<double, false>()));
  PHX_CHECK((dst_allow_lds_t<float, true>()));
  PHX_CHECK((dst_allow_lds_t<float, false>()));
  done = true;
  return PHX_OK;
}

// c[a]: coefficient of tridiag(-1, 2, -1) along axis a (3-D: h_b h_c / h_a; 2-D: h_y / h_x, h_x / h_y, 0)
static int box_grid_setup(phx_box_precond *bp, int device, const int L[3], const double c[3], bool f32) {
  bp->f32 = f32;
  PHX_CHECK(dst_allow_lds());
  BoxGrid &g = bp->g;
  if (const char *e = getenv("PHX_Z_TRIDIAG")) bp->ztri = atoi(e) != 0;  // A/B aid: 0 = sine transforms in z
  for (int a = 0; a < 3; ++a) {
    g.L[a] = L[a];
    g.m[a] = L[a] - 1;
    g.c[a] = c[a];
    // no transform runs along a tridiagonal z axis: its length is free (L[2] >= 2)
    if (a < 2 || !bp->ztri) PHX_CHECK(dst_get_plan(device, L[a], f32, &bp->plan[a]));
  }
  g.pitch = L[0];
  g.plane = g.pitch * g.m[1];
  g.scale = (2.0 / L[0]) * (2.0 / L[1]) * (2.0 / L[2]);
  for (int a = 0; a < 3; ++a) {
    // eigenvalue tables: constants of (length, coefficient) like the twiddle tables of a plan -- kept per process (a
    // step that re-assembles the same box found three blocking uploads and three hipMalloc / hipFree pairs here)
    static std::map<std::tuple<int, int, uint64_t>, double *> lam_cache;
    uint64_t cbits;
    memcpy(&cbits, &c[a], sizeof(cbits));
    const auto key = std::make_tuple(device, L[a], cbits);
    auto it = lam_cache.find(key);
    if (it != lam_cache.end()) {
      g.lam[a] = it->second;
      continue;
    }
    std::vector<double> lam((size_t)L[a]);
    for (int k = 0; k < L[a]; ++k)
      lam[k] = c[a] * (2.0 - 2.0 * cos(3.14159265358979323846 * (double)k / (double)L[a]));
    double *d = nullptr;
    PHX_HIP(hipMalloc(&d, sizeof(double) * (size_t)L[a]));
    PHX_HIP(hipMemcpy(d, lam.data(), sizeof(double) * (size_t)L[a], hipMemcpyHostToDevice));
    if (lam_cache.size() < 256) lam_cache[key] = d;   // beyond that the table lives and dies with this lattice
    else bp->lam[a] = d;
    g.lam[a] = d;
  }
  PHX_HIP(phx_malloc(&bp->G, (f32 ? sizeof(float) : sizeof(double)) * (size_t)(g.plane * g.m[2])));
  return PHX_OK;
}

static TriArgs box_tri_args(const phx_box_precond *bp) {
  const BoxGrid &g = bp->g;
  TriArgs a;
  memset(&a, 0, sizeof(a));
  a.m0 = g.m[0]; a.m1 = g.m[1]; a.pitch = (int)g.pitch; a.plane = g.plane;
  a.nloc = g.m[2];
  a.k0 = bp->dist ? bp->zk0 : 1;
  a.N = bp->dist ? bp->zN : g.m[2];
  a.cz = g.c[2]; a.scale_xy = (2.0 / g.L[0]) * (2.0 / g.L[1]);
  a.lamx = g.lam[0]; a.lamy = g.lam[1];
  if (bp->dist) {
    const int64_t ncol = g.pitch * g.m[1];
    a.wl = bp->carry_send; a.yl = bp->carry_send + ncol;
    a.w_in = bp->tri_in; a.y_in = bp->tri_in + ncol; a.y1 = bp->tri_in + 2 * ncol;
  }
  return a;
}

// z pass on G: tridiagonal solve per (kx, ky) column, or forward sine transform, 1 / lambda, inverse in LDS
template <typename T>
static int box_pass_z_t(phx_box_precond *bp, hipStream_t st) {
  const BoxGrid &g = bp->g;
  T *G = static_cast<T *>(bp->G);
  if (bp->ztri) return tri_launch<T, 0>(box_tri_args(bp), G, st);
  const DstPlan &pz = bp->plan[2];
  const int W = 2 * pz.pairs, ncb = (g.m[0] + W - 1) / W;
  const dim3 grid((unsigned)((int64_t)ncb * g.m[1])), block((unsigned)(pz.pairs * pz.slot));
  const size_t lds = (size_t)pz.lds_elems * sizeof(T) * 2;
  if (pz.wave) k_dst_s<T, 2, true, true><<<grid, block, lds, st>>>(g, pz, G, nullptr, 0);
  else k_dst_s<T, 2, true, false><<<grid, block, lds, st>>>(g, pz, G, nullptr, 0);
  PHX_HIP(hipGetLastError());
  return PHX_OK;
}

// dir: 0 plain transform of every row, 1 / 2 forward / backward pass of an application (rows of x lines without a
// mapped point are not read / not written, see k_dst_s)
template <typename T>
static int box_pass_y_t(phx_box_precond *bp, hipStream_t st, phx_system *prof, int dir = 0) {
  const BoxGrid &g = bp->g;
  if (g.m[2] <= 0) return PHX_OK;
  const int2 *ra = dir != 0 ? bp->line_iv : nullptr;
  const DstPlan &py = bp->plan[1];
  T *G = static_cast<T *>(bp->G);
  const size_t el = sizeof(T) * 2;
  const int W = 2 * py.pairs, ncb = (g.m[0] + W - 1) / W;
  const dim3 grid((unsigned)((int64_t)ncb * g.m[2])), block((unsigned)(py.pairs * py.slot));
  if (prof) PHX_CHECK(prof_begin(prof, 1));
  bool done = false;
  if constexpr (sizeof(T) == 8) {
    if (dst_wave_fast(g, py, 0)) {
      const dim3 g2((unsigned)ncb, (unsigned)g.m[2]);
      switch (py.L) {
#define X(L_) case L_: k_dst_yw<L_><<<g2, block, (size_t)py.lds_elems * el, st>>>(g, py, G, ra, dir); done = true; break;
        PHX_DST_WAVE_LENGTHS(X)
#undef X
        default: break;
      }
    }
  }
  if constexpr (sizeof(T) == 8) {
    if (!done && dst_pair_fast(g, py, 0)) {
      static const int ydbg = getenv("PHX_DST_YDBG") ? atoi(getenv("PHX_DST_YDBG")) : 0;   // experiment: 1 no transform, 2 no memory
      dir |= ydbg << 8;
      switch (py.L) {
#define X(L_) case L_: { \
          using S = PairShape<L_, 0>; \
          const int ncbp = (g.m[0] + S::W - 1) / S::W, ntiles = ncbp * g.m[2]; \
          k_dst_yp<L_, 0><<<dim3((unsigned)dst_pair_grid<L_>(ntiles)), dim3(S::NTHR), (size_t)S::LDS_ELEMS * 16, st>>>(g, py, G, ra, dir, ncbp, ntiles); \
          done = true; } break;
        PHX_DST_PAIR_LENGTHS(X)
#undef X
        default: break;
      }
    }
  }
  if constexpr (sizeof(T) == 8) {
    if (!done && !getenv("PHX_DST_GENERIC")) {
      switch (py.L) {
#define X(L_) case L_: if ((py.wave != 0) == dst_wave_f64(L_)) { \
          k_dst_s<double, 1, false, dst_wave_f64(L_), L_><<<grid, block, (size_t)py.lds_elems * el, st>>>(g, py, G, ra, dir); done = true; } break;
        PHX_DST_LENGTHS(X)
#undef X
        default: break;
      }
    }
  }
  if (!done) {
    if (py.wave) k_dst_s<T, 1, false, true><<<grid, block, (size_t)py.lds_elems * el, st>>>(g, py, G, ra, dir);
    else k_dst_s<T, 1, false, false><<<grid, block, (size_t)py.lds_elems * el, st>>>(g, py, G, ra, dir);
  }
  if (prof) PHX_CHECK(prof_end(prof, 1));
  PHX_HIP(hipGetLastError());
  return PHX_OK;
}

// slab-exact mode, first half of the middle passes: y transform, then the zero-inflow carries of this rank's
// planes into carry_send (the driver all-gathers them into carry_recv)
template <typename T>
static int box_middle_A_t(phx_box_precond *bp, hipStream_t st, phx_system *prof) {
  PHX_CHECK(box_pass_y_t<T>(bp, st, prof, bp->rowskip ? 1 : 0));
  return tri_launch<T, 1>(box_tri_args(bp), static_cast<T *>(bp->G), st);
}
// second half: interface recurrences over the gathered carries, z solve with inflows, inverse y transform
template <typename T>
static int box_middle_B_t(phx_box_precond *bp, hipStream_t st, phx_system *prof) {
  const BoxGrid &g = bp->g;
  const int64_t ncol = g.pitch * g.m[1];
  TriArgs a = box_tri_args(bp);
  TriRanks R;
  memset(&R, 0, sizeof(R));
  R.nranks = bp->nranks; R.rank = bp->rank;
  for (int r = 0; r < bp->nranks; ++r) R.planes[r] = bp->planes[r];
  if (a.cz != 0.0)
    k_tri_interface<<<dim3((unsigned)phx_div_up(ncol, 256)), dim3(256), 0, st>>>(a, R, bp->carry_recv, ncol, bp->tri_in,
                                                                                bp->tri_in + ncol, bp->tri_in + 2 * ncol);
  PHX_HIP(hipGetLastError());
  if (g.m[2] > 0) PHX_CHECK((tri_launch<T, 2>(a, static_cast<T *>(bp->G), st)));
  return box_pass_y_t<T>(bp, st, prof, bp->rowskip ? 2 : 0);
}

// the three middle passes (y, z solve, y) on G
template <typename T>
static int box_solve_middle_t(phx_box_precond *bp, hipStream_t st, phx_system *prof) {
  PHX_CHECK(box_pass_y_t<T>(bp, st, prof, bp->rowskip ? 1 : 0));
  PHX_CHECK(box_pass_z_t<T>(bp, st));
  return box_pass_y_t<T>(bp, st, prof, bp->rowskip ? 2 : 0);
}
static int box_solve_middle(phx_box_precond *bp, hipStream_t st, phx_system *prof = nullptr) {
  return bp->f32 ? box_solve_middle_t<float>(bp, st, prof) : box_solve_middle_t<double>(bp, st, prof);
}

template <typename T, int IO>
static int box_pass_x_t(phx_box_precond *bp, hipStream_t st, const double *vin, double *vout) {
  const BoxGrid &g = bp->g;
  const DstPlan &px = bp->plan[0];
  if (g.m[2] <= 0) return PHX_OK;
  const int64_t npairs = ((int64_t)g.m[1] * g.m[2] + 1) / 2;
  const dim3 grid((unsigned)phx_div_up(npairs, px.pairs)), block((unsigned)(px.pairs * px.slot));
  const size_t lds = (size_t)px.lds_elems * sizeof(T) * 2;
  const double *sc = IO == 1 ? bp->iscale : bp->dscale;
  bool done = false;
  if constexpr (sizeof(T) == 8) {
    if (dst_wave_fast(g, px, bp->nvec)) {
      const uint32_t vb = (uint32_t)(bp->nvec * 8);
      double *Gd = static_cast<double *>(bp->G);
      switch (px.L) {
#define X(L_) case L_: \
          if (sc) k_dst_xw<L_, IO, IO != 0><<<grid, block, lds, st>>>(g, px, Gd, bp->gmap, vin, vout, sc, bp->line_any, vb); \
          else k_dst_xw<L_, IO, false><<<grid, block, lds, st>>>(g, px, Gd, bp->gmap, vin, vout, sc, bp->line_any, vb); \
          done = true; break;
        PHX_DST_WAVE_LENGTHS(X)
#undef X
        default: break;
      }
    }
  }
  if constexpr (sizeof(T) == 8) {
    if (!done && dst_pair_fast(g, px, bp->nvec)) {
      const uint32_t vb = (uint32_t)(bp->nvec * 8);
      double *Gd = static_cast<double *>(bp->G);
      switch (px.L) {
#define X(L_) case L_: { \
          using S = PairShape<L_, 0>; \
          const int ngroups = (int)phx_div_up(npairs, S::PAIRS); \
          const dim3 gp((unsigned)dst_pair_grid<L_>(ngroups)), bk(S::NTHR); \
          const size_t ldsp = (size_t)S::LDS_ELEMS * 16; \
          if (sc) k_dst_xp<L_, 0, IO, IO != 0><<<gp, bk, ldsp, st>>>(g, px, Gd, bp->gmap, vin, vout, sc, bp->line_any, vb, ngroups); \
          else k_dst_xp<L_, 0, IO, false><<<gp, bk, ldsp, st>>>(g, px, Gd, bp->gmap, vin, vout, sc, bp->line_any, vb, ngroups); \
          done = true; } break;
        PHX_DST_PAIR_LENGTHS(X)
#undef X
        default: break;
      }
    }
  }
  if constexpr (sizeof(T) == 8) {
    if (!done && !getenv("PHX_DST_GENERIC")) {
      switch (px.L) {
#define X(L_) case L_: if ((px.wave != 0) == dst_wave_f64(L_)) { \
          k_dst_x<double, IO, dst_wave_f64(L_), L_><<<grid, block, lds, st>>>(g, px, static_cast<double *>(bp->G), bp->gmap, vin, vout, sc, bp->line_any); \
          done = true; } break;
        PHX_DST_LENGTHS(X)
#undef X
        default: break;
      }
    }
  }
  if (!done) {
    if (px.wave)
      k_dst_x<T, IO, true><<<grid, block, lds, st>>>(g, px, static_cast<T *>(bp->G), bp->gmap, vin, vout, sc, bp->line_any);
    else
      k_dst_x<T, IO, false><<<grid, block, lds, st>>>(g, px, static_cast<T *>(bp->G), bp->gmap, vin, vout, sc, bp->line_any);
  }
  PHX_HIP(hipGetLastError());
  return PHX_OK;
}
template <int IO>
static int box_pass_x(phx_box_precond *bp, hipStream_t st, const double *vin, double *vout) {
  return bp->f32 ? box_pass_x_t<float, IO>(bp, st, vin, vout) : box_pass_x_t<double, IO>(bp, st, vin, vout);
}

void phx_box_precond_destroy(phx_box_precond *bp) { box_precond_free(bp); }

// lattice bounding box of the (owned) active u vertices: out[0..2] = min, out[3..5] = max.
// Grid-stride with private bounds, then one atomic per block and axis (same-address atomics from every
// wavefront serialise: the z maximum improves with almost every wave in index order, 3.7 ms at 256^3).
__global__ void __launch_bounds__(256)
k_active_bbox(int64_t nv, int64_t n0, int64_t n1, const int32_t *__restrict__ du,
              const int32_t *__restrict__ iperm, const uint8_t *__restrict__ own,
              const int32_t *__restrict__ v2lat, int *__restrict__ out) {
  __shared__ int red[6][4];
  int lo[3] = {INT_MAX, INT_MAX, INT_MAX}, hi[3] = {-1, -1, -1};
  const uint32_t m0 = (uint32_t)n0, m1 = (uint32_t)n1;  // nv < 2^31
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < nv; v += (int64_t)gridDim.x * blockDim.x) {
    const int32_t d = du[v];
    if (d >= 0 && (!own || own[iperm[d]])) {
      const uint32_t w = v2lat ? (uint32_t)v2lat[v] : (uint32_t)v, q = w / m0;
      const int idx[3] = {(int)(w - q * m0), (int)(q % m1), (int)(q / m1)};
      for (int a = 0; a < 3; ++a) { lo[a] = min(lo[a], idx[a]); hi[a] = max(hi[a], idx[a]); }
    }
  }
  for (int a = 0; a < 3; ++a) {
    for (int o = 32; o > 0; o >>= 1) {
      lo[a] = min(lo[a], __shfl_xor(lo[a], o));
      hi[a] = max(hi[a], __shfl_xor(hi[a], o));
    }
    if ((threadIdx.x & 63) == 0) { red[a][threadIdx.x >> 6] = lo[a]; red[3 + a][threadIdx.x >> 6] = hi[a]; }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int a = threadIdx.x;
    const int l = min(min(red[a][0], red[a][1]), min(red[a][2], red[a][3]));
    const int h = max(max(red[3 + a][0], red[3 + a][1]), max(red[3 + a][2], red[3 + a][3]));
    if (h >= 0) { atomicMin(&out[a], l); atomicMax(&out[3 + a], h); }
  }
}

// ---- P2: the DoFs (vertices and edge midpoints of the Kuhn box) are exactly the points of the lattice of
// spacing h / 2 (every face and every cube holds one diagonal, so each face and cube centre carries one edge
// DoF); P1 on that lattice is spectrally equivalent to P2 on the box.  Entity e < nv: vertex (2i, 2j, 2k);
// e >= nv: midpoint of edges[e - nv] = sum of its endpoints' lattice coordinates.
__device__ __forceinline__ void p2_lattice_point(int64_t e, int64_t nv, int64_t n0, int64_t n1,
                                                 const int32_t *__restrict__ edges, int *q) {
  const uint32_t m0 = (uint32_t)n0, m1 = (uint32_t)n1;
  auto vert = [&](uint32_t w, int *o) {
    const uint32_t r = w / m0;
    o[0] = (int)(w - r * m0); o[1] = (int)(r % m1); o[2] = (int)(r / m1);
  };
  if (e < nv) {
    vert((uint32_t)e, q);
    for (int a = 0; a < 3; ++a) q[a] *= 2;
  } else {
    int a0[3], a1[3];
    vert((uint32_t)edges[2 * (e - nv)], a0);
    vert((uint32_t)edges[2 * (e - nv) + 1], a1);
    for (int a = 0; a < 3; ++a) q[a] = a0[a] + a1[a];
  }
}

__global__ void __launch_bounds__(256)
k_active_bbox_p2(int64_t nent, int64_t nv, int64_t n0, int64_t n1, const int32_t *__restrict__ edges,
                 const int32_t *__restrict__ du, const int32_t *__restrict__ iperm,
                 const uint8_t *__restrict__ own, int *__restrict__ out) {
  __shared__ int red[6][4];
  int lo[3] = {INT_MAX, INT_MAX, INT_MAX}, hi[3] = {-1, -1, -1};
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < nent; e += (int64_t)gridDim.x * blockDim.x) {
    const int32_t d = du[e];
    if (d >= 0 && (!own || own[iperm[d]])) {
      int q[3];
      p2_lattice_point(e, nv, n0, n1, edges, q);
      for (int a = 0; a < 3; ++a) { lo[a] = min(lo[a], q[a]); hi[a] = max(hi[a], q[a]); }
    }
  }
  for (int a = 0; a < 3; ++a) {
    for (int o = 32; o > 0; o >>= 1) {
      lo[a] = min(lo[a], __shfl_xor(lo[a], o));
      hi[a] = max(hi[a], __shfl_xor(hi[a], o));
    }
    if ((threadIdx.x & 63) == 0) { red[a][threadIdx.x >> 6] = lo[a]; red[3 + a][threadIdx.x >> 6] = hi[a]; }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int a = threadIdx.x;
    const int l = min(min(red[a][0], red[a][1]), min(red[a][2], red[a][3]));
    const int h = max(max(red[3 + a][0], red[3 + a][1]), max(red[3 + a][2], red[3 + a][3]));
    if (h >= 0) { atomicMin(&out[a], l); atomicMax(&out[3 + a], h); }
  }
}

// gmap (pre-set to -1) filled from the entities
__global__ void k_box_gmap_p2(BoxGrid g, int lo0, int lo1, int lo2, int64_t nent, int64_t nv, int64_t n0, int64_t n1,
                              const int32_t *__restrict__ edges, const int32_t *__restrict__ du,
                              const int32_t *__restrict__ iperm, const uint8_t *__restrict__ own,
                              int32_t *__restrict__ gmap) {
  const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= nent) return;
  const int32_t d = du[e];
  if (d < 0) return;
  const int32_t pos = iperm[d];
  if (own && !own[pos]) return;
  int q[3];
  p2_lattice_point(e, nv, n0, n1, edges, q);
  const int x = q[0] - lo0 - 1, y = q[1] - lo1 - 1, z = q[2] - lo2 - 1;
  if (x < 0 || x >= g.m[0] || y < 0 || y >= g.m[1] || z < 0 || z >= g.m[2]) return;
  gmap[x + g.pitch * y + g.plane * z] = pos;
}

__global__ void k_box_gmap(BoxGrid g, int lo0, int lo1, int lo2, int64_t n0, int64_t n1, int64_t n2,
                           const int32_t *__restrict__ du, const int32_t *__restrict__ iperm,
                           const uint8_t *__restrict__ own, const int32_t *__restrict__ lat2v,
                           int32_t *__restrict__ gmap) {
  const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= g.plane * g.m[2]) return;
  const int x = (int)(e % g.pitch), y = (int)((e / g.pitch) % g.m[1]), z = (int)(e / g.plane);
  int32_t q = -1;
  const int64_t i = lo0 + 1 + x, j = lo1 + 1 + y, k = lo2 + 1 + z;
  if (x < g.m[0] && i >= 0 && i < n0 && j >= 0 && j < n1 && k >= 0 && k < n2) {
    int64_t v = i + n0 * (j + n1 * k);
    if (lat2v) v = lat2v[v];  // sub-mesh of the box: its own vertex numbering (-1: not in the sub-mesh)
    const int32_t d = v >= 0 ? du[v] : -1;
    if (d >= 0) {
      const int32_t pos = iperm[d];
      if (!own || own[pos]) q = pos;
    }
  }
  gmap[e] = q;
}

// one wavefront per x line: does any lattice point of the line carry a DoF?
__global__ void k_line_any(BoxGrid g, const int32_t *__restrict__ gmap, uint8_t *__restrict__ line_any) {
  const int64_t l = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (l >= (int64_t)g.m[1] * g.m[2]) return;
  const int32_t *row = gmap + ((l % g.m[1]) * g.pitch + (l / g.m[1]) * g.plane);
  bool any = false;
  for (int k = lane; k < g.m[0]; k += 64) any |= row[k] >= 0;
  const unsigned long long b = __ballot(any);
  if (lane == 0) line_any[l] = b != 0ull;
}

// per plane: the interval of rows with such a line; the flags are widened to it (a line inside the interval without
// a point is transformed as zeros: rare, and the passes then agree on ONE notion of "row not stored")
__global__ void k_line_intervals(int m1, int m2, uint8_t *__restrict__ line_any, int2 *__restrict__ iv) {
  const int z = blockIdx.x, lane = threadIdx.x;   // one wavefront per plane
  if (z >= m2) return;
  uint8_t *la = line_any + (int64_t)m1 * z;
  int lo = m1, hi = -1;
  for (int y = lane; y < m1; y += 64)
    if (la[y]) { lo = min(lo, y); hi = max(hi, y); }
  for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o)); hi = max(hi, __shfl_xor(hi, o)); }
  if (hi < 0) { lo = 1; hi = 0; }
  for (int y = lo + lane; y <= hi; y += 64) la[y] = 1;
  if (lane == 0) iv[z] = make_int2(lo, hi);
}

__global__ void k_dscale(int64_t n, const int32_t *__restrict__ perm, const double *__restrict__ diag,
                         double *__restrict__ dscale) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) dscale[i] = diag[perm[i]];
}

// Weighted systems A ~ S K S (strong Dirichlet: S ~ |phi_h|, regularised as S^2 = diag A / diag K_box):
// M^-1 = S^-1 K_box^-1 S^-1, so P = D M^-1 scales by sqrt(kd / D) on the way in and sqrt(D kd) on the way out.
__global__ void k_dscale_weighted(int64_t n, const int32_t *__restrict__ perm, const double *__restrict__ diag,
                                  double kd, double *__restrict__ dscale, double *__restrict__ iscale) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double D = fabs(diag[perm[i]]);
  dscale[i] = sqrt(D * kd);
  iscale[i] = D > 0.0 ? sqrt(kd / D) : 0.0;
}

// z margin on a side where the active vertices reach the face of the mesh box (an OPEN end: phi-FEM imposes nothing
// on the box boundary, a natural condition): the Dirichlet face of the lattice moves this many planes away so that
// the lattice operator does not clamp what the real operator leaves free.  1024 x 1024 x 128 slab through the
// sphere: 54 iterations with ~32 planes, 96 with the closed-side margin of 4.  z planes are cheap (tridiagonal
// solve, any length).
#define PHX_PRECOND_MARGIN_OPEN 32
// lattice bounding box of this rank's (owned) active u DoFs in LOCAL lattice coordinates; hbb[3] < 0: none
__global__ void k_bbox_init(int *bb) { bb[threadIdx.x] = threadIdx.x < 3 ? INT_MAX : -1; }
static int box_local_bbox(phx_system *s, bool p2, int hbb[6]) {
  phx_mesh *m = s->mesh;
  hipStream_t st = m->stream;
  const int64_t n0 = m->box_n[0] + 1, n1 = m->box_n[1] + 1;
  int *dbb = nullptr;
  const int init[6] = {INT_MAX, INT_MAX, INT_MAX, -1, -1, -1};
  for (int i = 0; i < 6; ++i) hbb[i] = init[i];
  PHX_HIP(phx_malloc(&dbb, sizeof(init)));
  k_bbox_init<<<1, 6, 0, st>>>(dbb);   // (an upload from a pageable host array is a host round trip)
  if (p2)
    k_active_bbox_p2<<<dim3((unsigned)std::min<int64_t>(phx_div_up(s->nent, 256), 1024)), dim3(256), 0, st>>>(
        s->nent, m->nv, n0, n1, m->edges, s->dof_of_vertex_u, s->iperm, s->own, dbb);
  else
    k_active_bbox<<<dim3((unsigned)std::min<int64_t>(phx_div_up(m->nv, 256), 1024)), dim3(256), 0, st>>>(
        m->nv, n0, n1, s->dof_of_vertex_u, s->iperm, s->own, m->v2lat, dbb);
  PHX_HIP(hipMemcpyAsync(hbb, dbb, sizeof(init), hipMemcpyDeviceToHost, st));
  PHX_HIP(hipStreamSynchronize(st));
  PHX_HIP(phx_free(dbb));
  return PHX_OK;
}

// Builds the lattice of transform lengths L (interior points L - 1) whose lower Dirichlet face sits at the LOCAL
// lattice index lo, with its maps and scalings, and installs it as the preconditioner of `s` (state 1).
static int box_precond_build(phx_system *s, bool p2, const int L[3], const int lo[3], phx_box_precond **out) {
  phx_mesh *m = s->mesh;
  hipStream_t st = m->stream;
  const int64_t n0 = m->box_n[0] + 1, n1 = m->box_n[1] + 1, n2 = m->gdim == 3 ? m->box_n[2] + 1 : 1;
  phx_box_precond *bp = new phx_box_precond();
  // 2-D: the lattice gets a dummy third axis with coefficient 0 (one real plane; the z pass only scales)
  // P2: the lattice has spacing h / 2
  const double hs = p2 ? 0.5 : 1.0;
  const double h[3] = {hs * m->box_h[0], hs * m->box_h[1], hs * m->box_h[2]};
  const double c3[3] = {h[1] * h[2] / h[0], h[0] * h[2] / h[1], h[0] * h[1] / h[2]};
  const double c2[3] = {h[1] / h[0], h[0] / h[1], 0.0};
  int rc = box_grid_setup(bp, m->device, L, m->gdim == 3 ? c3 : c2, m->precond == 2);
  if (rc != PHX_OK) { box_precond_free(bp); return rc; }
  for (int a = 0; a < 3; ++a) bp->lo[a] = lo[a];
  const int64_t tot = bp->g.plane * bp->g.m[2];
  if (phx_malloc(&bp->gmap, sizeof(int32_t) * (size_t)std::max<int64_t>(tot, 1)) != hipSuccess ||
      (!s->u_unscaled && phx_malloc(&bp->dscale, sizeof(double) * (size_t)std::max<int64_t>(s->n, 1)) != hipSuccess)) {
    box_precond_free(bp);
    return PHX_ERR_HIP;
  }
  if (tot > 0) {
    if (p2) {
      if (hipMemsetAsync(bp->gmap, 0xff, sizeof(int32_t) * (size_t)tot, st) != hipSuccess) { box_precond_free(bp); return PHX_ERR_HIP; }
      k_box_gmap_p2<<<dim3((unsigned)phx_div_up(s->nent, 256)), dim3(256), 0, st>>>(
          bp->g, lo[0], lo[1], lo[2], s->nent, m->nv, n0, n1, m->edges, s->dof_of_vertex_u, s->iperm, s->own, bp->gmap);
    } else {
      k_box_gmap<<<dim3((unsigned)phx_div_up(tot, 256)), dim3(256), 0, st>>>(
          bp->g, lo[0], lo[1], lo[2], n0, n1, n2, s->dof_of_vertex_u, s->iperm, s->own, m->lat2v, bp->gmap);
    }
    const int64_t nlines = (int64_t)bp->g.m[1] * bp->g.m[2];
    if (phx_malloc(&bp->line_any, (size_t)nlines) != hipSuccess) { box_precond_free(bp); return PHX_ERR_HIP; }
    k_line_any<<<dim3((unsigned)phx_div_up(nlines * 64, 256)), dim3(256), 0, st>>>(bp->g, bp->gmap, bp->line_any);
    if (phx_malloc(&bp->line_iv, sizeof(int2) * (size_t)bp->g.m[2]) != hipSuccess) { box_precond_free(bp); return PHX_ERR_HIP; }
    k_line_intervals<<<dim3((unsigned)bp->g.m[2]), dim3(64), 0, st>>>(bp->g.m[1], bp->g.m[2], bp->line_any, bp->line_iv);
  }
  if (s->n > 0) {
    if (s->u_weighted) {
      if (phx_malloc(&bp->iscale, sizeof(double) * (size_t)s->n) != hipSuccess) { box_precond_free(bp); return PHX_ERR_HIP; }
      const double *cc = m->gdim == 3 ? c3 : c2;
      k_dscale_weighted<<<dim3((unsigned)phx_div_up(s->n, 256)), dim3(256), 0, st>>>(
          s->n, s->perm, s->diag, 2.0 * (cc[0] + cc[1] + cc[2]), bp->dscale, bp->iscale);
    } else if (!s->u_unscaled) {   // unscaled u columns: P = K_box^-1 itself, nothing to multiply on the way out
      k_dscale<<<dim3((unsigned)phx_div_up(s->n, 256)), dim3(256), 0, st>>>(s->n, s->perm, s->diag, bp->dscale);
    }
    if (hipGetLastError() != hipSuccess) { box_precond_free(bp); return PHX_ERR_HIP; }
    // rows outside the u block: P is the identity there, written by the producers of p and s (RestOut,
    // phx_solve.hip).  The other entries of phat / shat (u rows this rank does not own) are never written and stay zero.
  }
  if (hipGetLastError() != hipSuccess) { box_precond_free(bp); return PHX_ERR_HIP; }
  bp->own_ptr = s->own;
  bp->nvec = s->n;
  *out = bp;
  return PHX_OK;
}

static inline bool box_precond_kinds(const phx_system *s, bool *p2) {
  const phx_mesh *m = s->mesh;
  *p2 = s->u_p2_block && m->is_box && m->edges != nullptr;
  const bool p1 = (m->is_box || m->on_box_lattice) && s->u_vertex_block;
  return m->precond != 0 && (p1 || *p2);
}

// Builds the (rank-local) preconditioner of system `s` (state 1) or marks it not applicable (state -1).
static int box_precond_setup(phx_system *s) {
  phx_mesh *m = s->mesh;
  s->precond_state = -1;
  s->precond_veto = true;   // until built, or found to have nothing to build (multi-GPU vote, phx_solve.hip)
  bool p2;
  if (!box_precond_kinds(s, &p2)) return PHX_OK;
  int hbb[6];
  PHX_CHECK(box_local_bbox(s, p2, hbb));
  if (hbb[3] < 0) { s->precond_veto = false; return PHX_OK; }  // no (owned) active u DoF here: nothing to precondition
  int L[3], lo[3];
  bool ztri = true;
  if (const char *e = getenv("PHX_Z_TRIDIAG")) ztri = atoi(e) != 0;
  for (int a = 0; a < 3; ++a) {
    const int extent = hbb[3 + a] - hbb[a] + 1;
    // z (tridiagonal solve, no transform): exactly extent + margins planes; a 2-D lattice keeps its one real plane
    const bool free_len = a == 2 && ztri;
    if (free_len && m->gdim == 3) {
      const int top = (m->is_box || m->on_box_lattice) ? (int)m->box_n[2] : -1;   // last vertex plane of the mesh box
      int mlo = hbb[2] == 0 ? PHX_PRECOND_MARGIN_OPEN : PHX_PRECOND_MARGIN;
      int mhi = hbb[5] == top ? PHX_PRECOND_MARGIN_OPEN : PHX_PRECOND_MARGIN;
      if (extent + mlo + mhi + 1 > 1025) mlo = mhi = PHX_PRECOND_MARGIN;
      L[a] = extent + mlo + mhi + 1;
      if (L[a] > 1025) return PHX_OK;
      lo[a] = hbb[a] - 1 - mlo;
      continue;
    }
    if (free_len) L[a] = 2;
    else L[a] = dst_pick_length(extent + 2 * PHX_PRECOND_MARGIN + 1);
    if (L[a] < 0 || L[a] > 1025) return PHX_OK;  // larger than the longest transform / column: stay with Jacobi
    lo[a] = hbb[a] - 1 - (L[a] - 1 - extent) / 2;
  }
  phx_box_precond *bp = nullptr;
  PHX_CHECK(box_precond_build(s, p2, L, lo, &bp));
  s->precond = bp;
  s->precond_state = 1;
  s->precond_veto = false;
  return PHX_OK;
}

// ---- slab-exact preconditioner of a partitioned box (multi-GPU) -------------------------------------------------
// Every rank holds whole x-y planes of ONE global lattice box: the sine transforms in x and y are rank-local, and
// the tridiagonal recurrences in z (phx_tridiag.inc.hip) continue across ranks through two carries per (kx, ky)
// column and rank -- one all-gather of 16 bytes x columns per application, issued by the driver between the two
// halves (phases 7 / 9 and 8 / 10).  The result is K_box^-1 of the GLOBAL box, not a block-Jacobi over slabs
// (whose iteration count grew from 51 to 78-97 on 2-8 thin slabs).
extern "C" int phx_precond_local_bbox(phx_system *s, int64_t *out6) {
  PHX_HIP(hipSetDevice(s->mesh->device));
  for (int a = 0; a < 3; ++a) { out6[a] = INT64_MAX; out6[3 + a] = -1; }
  bool p2;
  if (!box_precond_kinds(s, &p2) || p2 || s->n == 0) return PHX_OK;
  int hbb[6];
  PHX_CHECK(box_local_bbox(s, false, hbb));
  if (hbb[3] < 0) return PHX_OK;
  for (int a = 0; a < 3; ++a) {   // GLOBAL vertex indices
    out6[a] = hbb[a] + s->mesh->box_off[a];
    out6[3 + a] = hbb[3 + a] + s->mesh->box_off[a];
  }
  return PHX_OK;
}

// bbox6: min / max GLOBAL vertex indices of the active u DoFs of ALL ranks; zb[nranks + 1]: rank r owns the global
// vertex planes [zb[r], zb[r+1]).  *ncol_out: columns (kx, ky) of the lattice = doubles per carry array, 0 when the
// preconditioner cannot be built (a transform longer than 1024, a rank with more than 1024 planes, P2, ...): then
// every rank -- they all see the same numbers -- keeps Jacobi.
extern "C" int phx_precond_setup_global(phx_system *s, const int64_t *bbox6, int nranks, int rank, const int64_t *zb,
                                        int64_t *ncol_out) {
  phx_mesh *m = s->mesh;
  PHX_HIP(hipSetDevice(m->device));
  *ncol_out = 0;
  if (s->precond_state == 1) { phx_box_precond_destroy(s->precond); s->precond = nullptr; }
  s->precond_state = -1;
  s->precond_veto = true;
  bool p2;
  const bool kinds = s->n > 0 ? box_precond_kinds(s, &p2) : (m->precond != 0 && m->is_box);
  PHX_REQUIRE(nranks >= 1 && nranks <= 16 && rank >= 0 && rank < nranks, PHX_ERR_VALUE, "bad rank layout");
  if (!kinds || (s->n > 0 && p2) || m->gdim != 3 || bbox6[5] < 0) return PHX_OK;
  int L[3], lo_g[3];
  for (int a = 0; a < 2; ++a) {
    const int extent = (int)(bbox6[3 + a] - bbox6[a] + 1);
    L[a] = dst_pick_length(extent + 2 * PHX_PRECOND_MARGIN + 1);
    if (L[a] < 0) return PHX_OK;
    lo_g[a] = (int)bbox6[a] - 1 - (L[a] - 1 - extent) / 2;
  }
  // open ends of the global box (see PHX_PRECOND_MARGIN_OPEN); zb[nranks] - 1 is its last vertex plane
  const int mlo = bbox6[2] == 0 ? PHX_PRECOND_MARGIN_OPEN : PHX_PRECOND_MARGIN;
  const int mhi = bbox6[5] == zb[nranks] - 1 ? PHX_PRECOND_MARGIN_OPEN : PHX_PRECOND_MARGIN;
  const int zext = (int)(bbox6[5] - bbox6[2] + 1), N = zext + mlo + mhi;
  lo_g[2] = (int)bbox6[2] - 1 - mlo;
  int planes[16], k0r = 1, first_local = 0;
  for (int r = 0; r < nranks; ++r) {
    const int64_t kf = r == 0 ? lo_g[2] + 1 : std::max<int64_t>(zb[r], lo_g[2] + 1);
    const int64_t ke = r == nranks - 1 ? lo_g[2] + N + 1 : std::min<int64_t>(zb[r + 1], lo_g[2] + N + 1);
    planes[r] = (int)std::max<int64_t>(0, ke - kf);
    if (planes[r] > 1024) return PHX_OK;
    if (r == rank) { k0r = (int)(kf - lo_g[2]); first_local = (int)kf; }
  }
  L[2] = planes[rank] + 1;
  const int lo[3] = {lo_g[0] - (int)m->box_off[0], lo_g[1] - (int)m->box_off[1], first_local - 1 - (int)m->box_off[2]};
  phx_box_precond *bp = nullptr;
  PHX_CHECK(box_precond_build(s, false, L, lo, &bp));
  bp->dist = true;
  bp->zN = N; bp->zk0 = k0r; bp->nranks = nranks; bp->rank = rank;
  for (int r = 0; r < nranks; ++r) bp->planes[r] = planes[r];
  const int64_t ncol = bp->g.pitch * bp->g.m[1];
  if (phx_malloc(&bp->tri_in, sizeof(double) * 3 * (size_t)ncol) != hipSuccess) { box_precond_free(bp); return PHX_ERR_HIP; }
  PHX_HIP(hipMemsetAsync(bp->tri_in, 0, sizeof(double) * 3 * (size_t)ncol, m->stream));
  s->precond = bp;
  s->precond_state = 1;
  s->precond_veto = false;
  *ncol_out = ncol;
  return PHX_OK;
}

// the all-gather buffers of the carries (caller-owned device memory, alive as long as the system's preconditioner):
// send[2 ncol], recv[nranks][2 ncol]
extern "C" int phx_precond_set_carry_buffers(phx_system *s, double *send, double *recv) {
  PHX_REQUIRE(s->precond_state == 1 && s->precond->dist, PHX_ERR_VALUE, "no slab-exact preconditioner is set up");
  s->precond->carry_send = send;
  s->precond->carry_recv = recv;
  s->precond->carry_borrowed = true;
  return PHX_OK;
}

// out[4] = {slab-exact preconditioner active (0/1), doubles each rank contributes to the all-gather, planes held here,
//           planes of the global column}
extern "C" int phx_precond_dist_info(const phx_system *s, int64_t *out4) {
  for (int i = 0; i < 4; ++i) out4[i] = 0;
  if (s->precond_state != 1 || !s->precond->dist) return PHX_OK;
  const phx_box_precond *bp = s->precond;
  out4[0] = 1; out4[1] = 2 * bp->g.pitch * bp->g.m[1]; out4[2] = bp->g.m[2]; out4[3] = bp->zN;
  return PHX_OK;
}

// vout = P vin:  u rows: K_box^-1 (times D where the SELL copy holds A D^-1), all other rows: identity.
// part 0: everything (rank-local preconditioner); parts 1 / 2: the halves of the slab-exact preconditioner before /
// after the all-gather of the carries.
static int box_precond_apply(phx_system *s, const double *vin, double *vout, int part = 0) {
  phx_box_precond *bp = s->precond;
  hipStream_t st = s->mesh->stream;
  if (part == 0 && bp->dist) {
    phx_set_error("the slab-exact preconditioner is applied in two halves around an all-gather");
    return PHX_ERR_VALUE;
  }
  // the identity part (rows outside the u block) is written by the kernels that produce p and s (RestOut, phx_solve.hip)
  bp->rowskip = bp->line_iv != nullptr;
  if (part != 2) PHX_CHECK(box_pass_x<1>(bp, st, vin, nullptr));
  if (part == 0) PHX_CHECK(box_solve_middle(bp, st, s));
  else if (part == 1) PHX_CHECK(bp->f32 ? box_middle_A_t<float>(bp, st, s) : box_middle_A_t<double>(bp, st, s));
  else PHX_CHECK(bp->f32 ? box_middle_B_t<float>(bp, st, s) : box_middle_B_t<double>(bp, st, s));
  if (part != 1) PHX_CHECK(box_pass_x<2>(bp, st, nullptr, vout));
  bp->rowskip = false;
  return PHX_OK;
}

// Test / inspection entry: solve K_box u = f on an (L0-1) x (L1-1) x (L2-1) interior lattice with spacings h
// (x fastest, no padding in `f`) in f64 (f32 = 0) or f32 transforms; u overwrites f.
extern "C" int phx_box_poisson_solve(int device, const int *L, const double *h, int f32, double *f_host) {
  PHX_HIP(hipSetDevice(device));
  {
    bool ztri = true;
    if (const char *e = getenv("PHX_Z_TRIDIAG")) ztri = atoi(e) != 0;
    for (int a = 0; a < 3; ++a)
      PHX_REQUIRE(dst_pick_length(L[a]) == L[a] || (a == 2 && ztri && L[a] >= 2 && L[a] <= 1025), PHX_ERR_VALUE,
                  "L[%d] = %d is not a supported transform length", a, L[a]);
  }
  phx_box_precond *bp = new phx_box_precond();
  const double c[3] = {h[1] * h[2] / h[0], h[0] * h[2] / h[1], h[0] * h[1] / h[2]};
  int rc = box_grid_setup(bp, device, L, c, f32 != 0);
  if (rc != PHX_OK) { box_precond_free(bp); return rc; }
  const BoxGrid &g = bp->g;
  const size_t tot = (size_t)(g.plane * g.m[2]);
  std::vector<double> tmp(tot, 0.0);
  for (int64_t z = 0; z < g.m[2]; ++z)
    for (int64_t y = 0; y < g.m[1]; ++y)
      memcpy(&tmp[z * g.plane + y * g.pitch], &f_host[(z * g.m[1] + y) * g.m[0]], sizeof(double) * (size_t)g.m[0]);
  std::vector<float> tmpf;
  if (f32) { tmpf.resize(tot); for (size_t i = 0; i < tot; ++i) tmpf[i] = (float)tmp[i]; }
  const void *src = f32 ? (const void *)tmpf.data() : (const void *)tmp.data();
  const size_t bytes = (f32 ? sizeof(float) : sizeof(double)) * tot;
  hipStream_t st = nullptr;
  if (hipMemcpy(bp->G, src, bytes, hipMemcpyHostToDevice) != hipSuccess) { box_precond_free(bp); return PHX_ERR_HIP; }
  rc = box_pass_x<0>(bp, st, nullptr, nullptr);
  if (rc == PHX_OK) rc = box_solve_middle(bp, st);
  if (rc == PHX_OK) rc = box_pass_x<0>(bp, st, nullptr, nullptr);
  if (rc == PHX_OK && hipDeviceSynchronize() != hipSuccess) rc = PHX_ERR_HIP;
  void *dst = f32 ? (void *)tmpf.data() : (void *)tmp.data();
  if (rc == PHX_OK && hipMemcpy(dst, bp->G, bytes, hipMemcpyDeviceToHost) != hipSuccess) rc = PHX_ERR_HIP;
  if (rc == PHX_OK) {
    if (f32) for (size_t i = 0; i < tot; ++i) tmp[i] = (double)tmpf[i];
    for (int64_t z = 0; z < g.m[2]; ++z)
      for (int64_t y = 0; y < g.m[1]; ++y)
        memcpy(&f_host[(z * g.m[1] + y) * g.m[0]], &tmp[z * g.plane + y * g.pitch], sizeof(double) * (size_t)g.m[0]);
  }
  box_precond_free(bp);
  return rc;
}

// Timing aid (tools/dst_bench.py): average microseconds of the x, y, z (solve) passes on a zero lattice.
extern "C" int phx_box_dst_bench(int device, const int *L, int f32, int reps, double *out_us3) {
  PHX_HIP(hipSetDevice(device));
  for (int a = 0; a < 2; ++a)
    PHX_REQUIRE(dst_pick_length(L[a]) == L[a], PHX_ERR_VALUE, "L[%d] = %d is not a supported transform length", a, L[a]);
  phx_box_precond *bp = new phx_box_precond();
  const double c[3] = {1.0, 1.0, 1.0};
  int rc = box_grid_setup(bp, device, L, c, f32 != 0);
  if (rc != PHX_OK) { box_precond_free(bp); return rc; }
  const BoxGrid &g = bp->g;
  const size_t bytes = (f32 ? sizeof(float) : sizeof(double)) * (size_t)(g.plane * g.m[2]);
  (void)hipMemset(bp->G, 0, bytes);
  hipStream_t st = nullptr;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int which = 0; which < 3 && rc == PHX_OK; ++which) {
    auto run = [&]() -> int {
      if (which == 0) return box_pass_x<0>(bp, st, nullptr, nullptr);
      if (which == 2) return f32 ? box_pass_z_t<float>(bp, st) : box_pass_z_t<double>(bp, st);
      const int rc_y = f32 ? box_pass_y_t<float>(bp, st, nullptr) : box_pass_y_t<double>(bp, st, nullptr);
      if (rc_y != PHX_OK) return rc_y;
      return hipGetLastError() == hipSuccess ? PHX_OK : PHX_ERR_HIP;
    };
    for (int i = 0; i < 3 && rc == PHX_OK; ++i) rc = run();
    (void)hipEventRecord(e0, st);
    for (int i = 0; i < reps && rc == PHX_OK; ++i) rc = run();
    (void)hipEventRecord(e1, st);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    out_us3[which] = 1e3 * ms / reps;
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  box_precond_free(bp);
  return rc;
}
