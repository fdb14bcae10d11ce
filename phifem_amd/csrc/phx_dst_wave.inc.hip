// Sine-transform passes of the lattice preconditioner, f64, for the transform lengths whose pairs live inside one
// wavefront (L = 192, 256, 512) -- included by phx_precond.inc.hip behind k_dst_x / k_dst_s, which stay for every
// other length, for f32 and as the A/B reference (PHX_DST_OLD=1).
//
// Same algorithm and the same LDS layout as k_dst_x / k_dst_s (two real lines per complex FFT of length L, Stockham
// stages in LDS, dst_core); what differs is everything around the transform.  The instruction stream of the old
// kernels (llvm-objdump, L = 192, y pass: 1537 instructions, 825 VALU of which 296 floating point, 476 SALU, 77
// exec-mask regions) was two thirds address arithmetic and predication: 64-bit multiplies per tile access, a branch
// around every guarded load, run-time divisions by the tile width.  Here
//  * block shape, tile width, trip counts and every LDS index step are compile-time constants (WaveShape);
//  * global accesses go through buffer descriptors with 32-bit byte offsets: a masked access selects an offset
//    beyond the descriptor's range (the load returns 0, the store is dropped) -- no branch, no exec-mask region;
//    the gathers of the x pass need no mask at all: map entry -1 shifts to an out-of-range offset;
//  * the lanes of a pair run the whole transform inside ONE predicated region (wave-local synchronisation only).
// Host side: box_pass_x_t / box_pass_y_t take this path when dst_wave_fast() says the lattice fits 31-bit offsets.
typedef unsigned phx_v2u __attribute__((ext_vector_type(2)));
#define PHX_BUF_OOB 0x80000000u   // beyond every descriptor built here (num_records <= 2^31 - 1)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const void *p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ double buf_ld_f64(__amdgpu_buffer_rsrc_t rs, uint32_t off) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, 0, 0));
}
__device__ __forceinline__ int32_t buf_ld_i32(__amdgpu_buffer_rsrc_t rs, uint32_t off) {
  return (int32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, 0);
}
__device__ __forceinline__ void buf_st_f64(double v, __amdgpu_buffer_rsrc_t rs, uint32_t off) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(phx_v2u, v), rs, (int)off, 0, 0);
}

template <int LL> struct WaveShape {
  static constexpr int TP = LL / 8;                  // lanes that work on a pair
  static constexpr int SLOT = TP <= 32 ? 32 : 64;    // lanes reserved per pair
  static constexpr int PAIRS = LL >= 512 ? 4 : 8;    // = dst_get_plan's choice for these lengths (checked on the host)
  static constexpr int NTHR = PAIRS * SLOT;
  static constexpr int W = 2 * PAIRS;                // columns of a y tile
  static constexpr int RSTEP = NTHR / W;             // rows of a y tile one trip of the block covers
  static constexpr int H = LL / 2;
  static constexpr int ZL = ZLEN(LL);
  static_assert(dst_wave_f64(LL) && RSTEP % 8 == 0 && TP % 8 == 0 && H % RSTEP == 0, "shape");
};

// The twiddle and sine tables of the plan go to LDS behind the padded pair arrays (the stage loops read them with LDS
// latency).  Two halves: the global loads are ISSUED behind the tile loads of the block and the LDS writes follow the
// tile's own -- staging them first (as k_dst_x / k_dst_s do) put a full global round trip in front of the tile loads.
template <int LL> struct WaveTables {
  using S = WaveShape<LL>;
  static constexpr int NTW = (LL + S::NTHR - 1) / S::NTHR, NSN = (LL / 2 + 1 + S::NTHR - 1) / S::NTHR;
  C2<double> tw[NTW];
  double sn[NSN];
  __device__ __forceinline__ void load(const DstPlan &P) {
    const C2<double> *gtw = PlanTab<double>::tw(P);
    const double *gsn = P.sintab;
#pragma unroll
    for (int q = 0; q < NTW; ++q) {
      const int j = q * S::NTHR + (int)threadIdx.x;
      tw[q] = gtw[min(j, LL - 1)];
    }
#pragma unroll
    for (int q = 0; q < NSN; ++q) {
      const int j = q * S::NTHR + (int)threadIdx.x;
      sn[q] = gsn[min(j, LL / 2)];
    }
  }
  __device__ __forceinline__ void store(C2<double> *zs, const C2<double> **twp, const double **snp) const {
    C2<double> *ltw = zs + S::PAIRS * S::ZL;   // = P.tab_off (no scan scratch in wave mode)
    double *lsn = reinterpret_cast<double *>(ltw + LL);
#pragma unroll
    for (int q = 0; q < NTW; ++q) {
      const int j = q * S::NTHR + (int)threadIdx.x;
      if (j < LL) ltw[j] = tw[q];
    }
#pragma unroll
    for (int q = 0; q < NSN; ++q) {
      const int j = q * S::NTHR + (int)threadIdx.x;
      if (j <= LL / 2) lsn[j] = sn[q];
    }
    *twp = ltw;
    *snp = lsn;
  }
};

// ---- y pass: a block transforms the W columns [col0, col0 + W) of plane `outer` (grid: column blocks x planes).
// row_any / dir as in k_dst_s: rows outside the plane's interval are taken as zero (dir = 1) / not stored (dir = 2).
template <int LL>
__global__ void __launch_bounds__(WaveShape<LL>::NTHR)
k_dst_yw(BoxGrid g, DstPlan P, double *__restrict__ G, const int2 *__restrict__ row_any, int dir) {
  using S = WaveShape<LL>;
  extern __shared__ double2 zs_raw[];
  C2<double> *zs = reinterpret_cast<C2<double> *>(zs_raw);
  constexpr int len = LL - 1;                        // rows of a plane (g.m[1])
  constexpr int ZS = S::RSTEP + S::RSTEP / 8;        // ZP(j + RSTEP) - ZP(j)
  const int tid = (int)threadIdx.x;
  const int col0 = (int)blockIdx.x * S::W, outer = (int)blockIdx.y;
  const int ncols = min(S::W, g.m[0] - col0);
  const int pitch8 = (int)g.pitch * 8;
  const __amdgpu_buffer_rsrc_t rs =
      buf_rsrc(G + ((int64_t)outer * g.plane + col0), (uint32_t)((len - 1) * pitch8 + ncols * 8));
  const C2<double> *tw;
  const double *sn;
  WaveTables<LL> tab;
  const int tcol = tid % S::W, row0 = tid / S::W;
  const bool colok = tcol < ncols;
  double *wcol = reinterpret_cast<double *>(zs + (tcol >> 1) * S::ZL) + (tcol & 1);
  const uint32_t dstep = (uint32_t)(S::RSTEP * pitch8);
  // rows [rlo, rlo + span] of the plane are read (forward pass of an application) / written (backward pass); every row
  // otherwise.  An empty interval {1, 0} becomes rlo = 2^30, span = 0.  (Block-uniform: scalar registers.)
  int lrlo = 0, srlo = 0;
  uint32_t lspan = len - 1, sspan = len - 1;
  if (row_any && dir != 0) {
    const int2 iv = row_any[outer];
    const int rlo = iv.y >= iv.x ? iv.x : (1 << 30);
    const uint32_t span = iv.y >= iv.x ? (uint32_t)(iv.y - iv.x) : 0u;
    if (dir == 1) { lrlo = rlo; lspan = span; } else { srlo = rlo; sspan = span; }
  }
  {
    // the thread takes the PAIRS of rows (j - 1, L - 1 - j), j = 1 + row0 + i RSTEP <= L / 2, of its column and folds them
    // in registers: y_j = sin(pi j / L)(x_j + x_{L-j}) + (x_j - x_{L-j}) / 2.  j = L / 2 pairs a row with itself:
    // sin = 1, the difference vanishes, both writes put 2 x_H into the same element.
    constexpr int NT = S::H / S::RSTEP;
    double va[NT], vb[NT], sj[NT];
    const double *sng = P.sintab;
    // a column beyond the lattice starts out of range and stays there (offsets move by less than 2^30)
    const uint32_t offa = colok ? (uint32_t)(tcol * 8 + row0 * pitch8) : PHX_BUF_OOB;
    const uint32_t offb = colok ? (uint32_t)(tcol * 8 + (LL - 2 - row0) * pitch8) : 0xc0000000u;
    const uint32_t da = (uint32_t)(row0 - lrlo), db = (uint32_t)(LL - 2 - row0 - lrlo);
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const bool oka = da + (uint32_t)(i * S::RSTEP) <= lspan;
      const bool okb = db - (uint32_t)(i * S::RSTEP) <= lspan;
      va[i] = buf_ld_f64(rs, oka ? offa + (uint32_t)i * dstep : PHX_BUF_OOB);
      vb[i] = buf_ld_f64(rs, okb ? offb - (uint32_t)i * dstep : PHX_BUF_OOB);
      sj[i] = sng[1 + row0 + i * S::RSTEP];
    }
    tab.load(P);
    const int zpa = ZP(1 + row0), zpb = ZP(LL - 1 - row0);
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const double e = sj[i] * (va[i] + vb[i]), o = 0.5 * (va[i] - vb[i]);
      wcol[2 * (zpa + i * ZS)] = e + o;
      wcol[2 * (zpb - i * ZS)] = e - o;
    }
    if (row0 == 0) wcol[0] = 0.0;
    tab.store(zs, &tw, &sn);
  }
  __syncthreads();
  {
    const int pr = tid / S::SLOT, t = tid % S::SLOT;
    if (2 * pr < ncols && t < S::TP) dst_core<double, true, LL, false>(zs + pr * S::ZL, nullptr, P, t, true, tw, sn);
  }
  __syncthreads();
  {
    constexpr int NS = (len + S::RSTEP - 1) / S::RSTEP;
    const int zr = ZP(row0 + 1);
    const uint32_t offs = colok ? (uint32_t)(tcol * 8 + row0 * pitch8) : PHX_BUF_OOB;
    const uint32_t ds = (uint32_t)(row0 - srlo);
    double v[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) v[i] = wcol[2 * (zr + i * ZS)];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const bool ok = ds + (uint32_t)(i * S::RSTEP) <= sspan;   // implies row < len
      buf_st_f64(v[i], rs, ok ? offs + (uint32_t)i * dstep : PHX_BUF_OOB);
    }
  }
}

// ---- x pass: pair = two consecutive x lines (line l starts at element l * pitch: planes are not padded).
// IO = 0: lattice -> lattice; IO = 1: gathered from the Krylov vector through gmap (times dscale if SC);
// IO = 2: scattered out through gmap (times dscale if SC).  line_any as in k_dst_x.
template <int LL, int IO, bool SC>
__global__ void __launch_bounds__(WaveShape<LL>::NTHR)
k_dst_xw(BoxGrid g, DstPlan P, double *__restrict__ G, const int32_t *__restrict__ gmap,
         const double *__restrict__ vin, double *__restrict__ vout, const double *__restrict__ dscale,
         const uint8_t *__restrict__ line_any, uint32_t vec_bytes) {
  using S = WaveShape<LL>;
  extern __shared__ double2 zs_raw[];
  C2<double> *zs = reinterpret_cast<C2<double> *>(zs_raw);
  constexpr int TP = S::TP;
  constexpr int ZS = TP + TP / 8;                    // ZP(j + TP) - ZP(j)
  const int tid = (int)threadIdx.x;
  const int pr = tid / S::SLOT, t = tid % S::SLOT;
  const int nlines = g.m[1] * g.m[2];
  if (IO != 0 && line_any) {
    // a block whose lines hold no mapped point: forward, the y pass takes its rows as zero; backward, nothing to scatter.
    // Every wave looks at the 2 PAIRS flags of the block with its first lanes (one load, one ballot).
    const int lane = tid & 63, l = (int)blockIdx.x * S::PAIRS * 2 + lane;
    const bool mine = lane < 2 * S::PAIRS && l < nlines && line_any[l] != 0;
    if (__ballot(mine) == 0) return;
  }
  const int line0 = ((int)blockIdx.x * S::PAIRS + pr) * 2;
  const bool live = line0 < nlines && t < TP;
  C2<double> *w = zs + pr * S::ZL;
  const C2<double> *tw;
  const double *sn;
  WaveTables<LL> tab;
  const uint32_t lat = (uint32_t)nlines * (uint32_t)g.pitch;   // lattice elements
  const __amdgpu_buffer_rsrc_t rsG = buf_rsrc(G, lat * 8u);
  const __amdgpu_buffer_rsrc_t rsM = buf_rsrc(gmap, IO != 0 ? lat * 4u : 0u);
  const __amdgpu_buffer_rsrc_t rsV = buf_rsrc(IO == 1 ? (const void *)vin : (const void *)vout, IO != 0 ? vec_bytes : 0u);
  const __amdgpu_buffer_rsrc_t rsS = buf_rsrc(dscale, SC ? vec_bytes : 0u);
  bool has[2] = {false, false};
  uint32_t base[2] = {0, 0};
  double va[4], vb[4], ua[4], ub[4], sj[4];   // line a / b at j (v) and at L - j (u)
  if (live) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int l = line0 + c;
      has[c] = l < nlines && !(IO != 0 && line_any && !line_any[l]);
      base[c] = (uint32_t)l * (uint32_t)g.pitch;
    }
    // the lane takes the pairs (j, L - j), j = 1 + t + i TP (j = 1 .. L / 2) of both lines; j = L / 2 pairs an element
    // with itself (see k_dst_yw)
    const double *sng = P.sintab;
    const uint32_t ea = base[0] + (uint32_t)t, eb = base[1] + (uint32_t)t;                         // element of j = 1 + t
    const uint32_t fa = base[0] + (uint32_t)(LL - 2 - t), fb = base[1] + (uint32_t)(LL - 2 - t);   // element of L - j
    if (IO == 1) {
      int32_t qa[4], qb[4], pa[4], pb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        qa[i] = buf_ld_i32(rsM, has[0] ? (ea + (uint32_t)(i * TP)) * 4u : PHX_BUF_OOB);
        qb[i] = buf_ld_i32(rsM, has[1] ? (eb + (uint32_t)(i * TP)) * 4u : PHX_BUF_OOB);
        pa[i] = buf_ld_i32(rsM, has[0] ? (fa - (uint32_t)(i * TP)) * 4u : PHX_BUF_OOB);
        pb[i] = buf_ld_i32(rsM, has[1] ? (fb - (uint32_t)(i * TP)) * 4u : PHX_BUF_OOB);
        sj[i] = sng[1 + t + i * TP];
      }
      // map entry -1 (no DoF at the lattice point) becomes byte offset 0xfffffff8: out of range, the gather returns 0
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t oqa = has[0] ? (uint32_t)qa[i] << 3 : PHX_BUF_OOB, oqb = has[1] ? (uint32_t)qb[i] << 3 : PHX_BUF_OOB;
        const uint32_t opa = has[0] ? (uint32_t)pa[i] << 3 : PHX_BUF_OOB, opb = has[1] ? (uint32_t)pb[i] << 3 : PHX_BUF_OOB;
        va[i] = buf_ld_f64(rsV, oqa); vb[i] = buf_ld_f64(rsV, oqb);
        ua[i] = buf_ld_f64(rsV, opa); ub[i] = buf_ld_f64(rsV, opb);
        if (SC) {
          va[i] *= buf_ld_f64(rsS, oqa); vb[i] *= buf_ld_f64(rsS, oqb);
          ua[i] *= buf_ld_f64(rsS, opa); ub[i] *= buf_ld_f64(rsS, opb);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        va[i] = buf_ld_f64(rsG, has[0] ? (ea + (uint32_t)(i * TP)) * 8u : PHX_BUF_OOB);
        vb[i] = buf_ld_f64(rsG, has[1] ? (eb + (uint32_t)(i * TP)) * 8u : PHX_BUF_OOB);
        ua[i] = buf_ld_f64(rsG, has[0] ? (fa - (uint32_t)(i * TP)) * 8u : PHX_BUF_OOB);
        ub[i] = buf_ld_f64(rsG, has[1] ? (fb - (uint32_t)(i * TP)) * 8u : PHX_BUF_OOB);
        sj[i] = sng[1 + t + i * TP];
      }
    }
  }
  tab.load(P);   // every lane of the block, behind the lattice loads
  if (live) {
    const int zpa = ZP(1 + t), zpb = ZP(LL - 1 - t);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const C2<double> e = mk<double>(sj[i] * (va[i] + ua[i]), sj[i] * (vb[i] + ub[i]));
      const C2<double> o = mk<double>(0.5 * (va[i] - ua[i]), 0.5 * (vb[i] - ub[i]));
      w[zpa + i * ZS] = cadd(e, o);
      w[zpb - i * ZS] = csub(e, o);
    }
    if (t == 0) w[0] = mk<double>(0.0, 0.0);
  }
  tab.store(zs, &tw, &sn);
  __syncthreads();   // the tables; the pair's own elements are wave-local
  if (!live) return;
  dst_core<double, true, LL, false>(w, nullptr, P, t, true, tw, sn);
  // k = 1 + t + i TP, i < 8: k = 1 .. L; the lattice holds k <= L - 1
  const int zk = ZP(1 + t);
  C2<double> F[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) F[i] = w[zk + i * ZS];
  if (IO == 2) {
    int32_t qa[8], qb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const bool kin = i < 7 || t < TP - 1;
      qa[i] = buf_ld_i32(rsM, has[0] && kin ? (base[0] + (uint32_t)(t + i * TP)) * 4u : PHX_BUF_OOB);
      qb[i] = buf_ld_i32(rsM, has[1] && kin ? (base[1] + (uint32_t)(t + i * TP)) * 4u : PHX_BUF_OOB);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const bool kin = i < 7 || t < TP - 1;
      const uint32_t oa = has[0] && kin ? (uint32_t)qa[i] << 3 : PHX_BUF_OOB, ob = has[1] && kin ? (uint32_t)qb[i] << 3 : PHX_BUF_OOB;
      double xa = F[i].x, xb = F[i].y;
      if (SC) { xa *= buf_ld_f64(rsS, oa); xb *= buf_ld_f64(rsS, ob); }
      buf_st_f64(xa, rsV, oa);
      buf_st_f64(xb, rsV, ob);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const bool kin = i < 7 || t < TP - 1;
      buf_st_f64(F[i].x, rsG, has[0] && kin ? (base[0] + (uint32_t)(t + i * TP)) * 8u : PHX_BUF_OOB);
      buf_st_f64(F[i].y, rsG, has[1] && kin ? (base[1] + (uint32_t)(t + i * TP)) * 8u : PHX_BUF_OOB);
    }
  }
}

// lengths served by the kernels above
#define PHX_DST_WAVE_LENGTHS(X) X(192) X(256) X(512)

template <int LL>
static int dst_wave_allow_lds() {
  const int bytes = 160 * 1024;
  PHX_HIP(hipFuncSetAttribute((const void *)k_dst_yw<LL>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  PHX_HIP(hipFuncSetAttribute((const void *)k_dst_xw<LL, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  PHX_HIP(hipFuncSetAttribute((const void *)k_dst_xw<LL, 1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  PHX_HIP(hipFuncSetAttribute((const void *)k_dst_xw<LL, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  PHX_HIP(hipFuncSetAttribute((const void *)k_dst_xw<LL, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  PHX_HIP(hipFuncSetAttribute((const void *)k_dst_xw<LL, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  return PHX_OK;
}

// the plan of axis `a` has the shape the kernels above are compiled for, and every byte offset fits 31 bits
template <int LL>
static bool dst_wave_shape_ok(const DstPlan &p) {
  using S = WaveShape<LL>;
  return p.L == LL && p.wave && p.pairs == S::PAIRS && p.slot == S::SLOT && p.scr == 0 && p.tab_off == S::PAIRS * S::ZL;
}
static bool dst_wave_fast(const BoxGrid &g, const DstPlan &p, int64_t nvec) {
  static const bool off = getenv("PHX_DST_OLD") != nullptr || getenv("PHX_DST_GENERIC") != nullptr;
  if (off) return false;
  const int64_t lat = g.plane * g.m[2];
  if (g.plane != g.pitch * g.m[1] || lat * 8 >= (int64_t)PHX_BUF_OOB || nvec * 8 >= (int64_t)PHX_BUF_OOB) return false;
  bool ok = false;
#define X(L_) ok = ok || dst_wave_shape_ok<L_>(p);
  PHX_DST_WAVE_LENGTHS(X)
#undef X
  return ok;
}
