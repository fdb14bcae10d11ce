// Sine-transform passes for the LONG f64 lengths (384, 768, 1024) -- included by phx_precond.inc.hip behind
// phx_dst_wave.inc.hip.  These are the lengths of the 1024^3 slabs (BASELINE configs[4]) and of the P2 fine lattice
// (configs[2]), where the x / y passes ARE the solve (ten passes of 1.8 GB per iteration on the 768 x 768 x 192 lattice).
//
// A pair of such a transform spans several wavefronts (L / 8 threads, block-wide barriers inside dst_core) and the padded
// tile of 16 columns takes most of a CU's LDS (L = 768: 154 KB), so ONE block lives on a CU and k_dst_s ran its three
// phases -- tile load, transform, tile store -- strictly one after the other: 2.7 TB/s where the short lengths, with five
// blocks per CU to overlap, reach 4.8.  Here a block is PERSISTENT and software-pipelined over its tiles:
//     transform tile n in LDS        | the loads of tile n+1 are in flight (into registers), the stores of tile n-1 drain
//     results of tile n -> registers | (16 LDS reads per thread)
//     fold tile n+1 registers -> LDS | (waits for its loads; they had a whole transform)
//     issue the stores of tile n, the loads of tile n+2
// so the global traffic of a CU runs behind its LDS / VALU work instead of between it.  Loader / storer idioms as in
// phx_dst_wave.inc.hip: compile-time shapes, buffer descriptors, out-of-range offsets for masked accesses, gathers
// through the map without a mask.

template <int LL> struct LongShape {
  static constexpr int TP = LL / 8;                                   // threads of a pair
  static constexpr int PAIRS = LL >= 1024 ? 4 : (LL >= 768 ? 8 : 4);  // = dst_get_plan's choice (checked on the host)
  static constexpr int NTHR = PAIRS * TP;
  static constexpr int W = 2 * PAIRS;                                 // columns of a y tile
  static constexpr int RSTEP = NTHR / W;                              // = TP / 2
  static constexpr int H = LL / 2;
  static constexpr int ZL = ZLEN(LL);
  static constexpr int SCR = 2 * (TP + (TP + 7) / 8 + 1);             // scan scratch per pair (DstPlan::scr)
  static constexpr int TAB = PAIRS * (ZL + SCR);                      // DstPlan::tab_off
  static_assert(!dst_wave_f64(LL) && RSTEP % 8 == 0 && TP % 8 == 0 && H % RSTEP == 0, "shape");
};

template <int LL>
__device__ __forceinline__ void stage_tables_long(C2<double> *zs, const DstPlan &P, const C2<double> **tw, const double **sn) {
  using S = LongShape<LL>;
  C2<double> *ltw = zs + S::TAB;
  double *lsn = reinterpret_cast<double *>(ltw + LL);
  const C2<double> *gtw = PlanTab<double>::tw(P);
  const double *gsn = P.sintab;
  for (int j = (int)threadIdx.x; j < LL; j += S::NTHR) ltw[j] = gtw[j];
  for (int j = (int)threadIdx.x; j <= LL / 2; j += S::NTHR) lsn[j] = gsn[j];
  *tw = ltw;
  *sn = lsn;
}

// ---- y pass.  Tile q = (column block q % ncb, plane q / ncb); block b takes tiles b, b + gridDim.x, ...
template <int LL>
__global__ void __launch_bounds__(LongShape<LL>::NTHR)
k_dst_yl(BoxGrid g, DstPlan P, double *__restrict__ G, const int2 *__restrict__ row_any, int dir, int ncb, int ntiles) {
  using S = LongShape<LL>;
  extern __shared__ double2 zs_raw[];
  C2<double> *zs = reinterpret_cast<C2<double> *>(zs_raw);
  constexpr int len = LL - 1;
  constexpr int ZS = S::RSTEP + S::RSTEP / 8;
  constexpr int NT = S::H / S::RSTEP;
  constexpr int NS = (len + S::RSTEP - 1) / S::RSTEP;
  const int tid = (int)threadIdx.x;
  const int tcol = tid % S::W, row0 = tid / S::W;
  const int pr = tid / S::TP, t = tid % S::TP;
  const int pitch8 = (int)g.pitch * 8;
  const uint32_t dstep = (uint32_t)(S::RSTEP * pitch8);
  const C2<double> *tw;
  const double *sn;
  stage_tables_long<LL>(zs, P, &tw, &sn);
  double *wcol = reinterpret_cast<double *>(zs + (tcol >> 1) * S::ZL) + (tcol & 1);
  C2<double> *w = zs + pr * S::ZL, *scr = zs + S::PAIRS * S::ZL + pr * S::SCR;
  const int zpa = ZP(1 + row0), zpb = ZP(LL - 1 - row0), zr = ZP(row0 + 1);
  double va[NT], vb[NT], out[NS];

  // per-tile scalars (block-uniform)
  struct Tile { __amdgpu_buffer_rsrc_t rs; int ncols; int lrlo, srlo; uint32_t lspan, sspan; };
  auto tile_of = [&](int q) {
    Tile T;
    const int cb = q % ncb, outer = q / ncb;
    const int col0 = cb * S::W;
    T.ncols = min(S::W, g.m[0] - col0);
    T.rs = buf_rsrc(G + ((int64_t)outer * g.plane + col0), (uint32_t)((len - 1) * pitch8 + T.ncols * 8));
    T.lrlo = 0; T.srlo = 0; T.lspan = len - 1; T.sspan = len - 1;
    if (row_any && dir != 0) {
      const int2 iv = row_any[outer];
      const int rlo = iv.y >= iv.x ? iv.x : (1 << 30);
      const uint32_t span = iv.y >= iv.x ? (uint32_t)(iv.y - iv.x) : 0u;
      if (dir == 1) { T.lrlo = rlo; T.lspan = span; } else { T.srlo = rlo; T.sspan = span; }
    }
    return T;
  };
  auto issue_loads = [&](const Tile &T) {
    const bool colok = tcol < T.ncols;
    const uint32_t offa = colok ? (uint32_t)(tcol * 8 + row0 * pitch8) : PHX_BUF_OOB;
    const uint32_t offb = colok ? (uint32_t)(tcol * 8 + (LL - 2 - row0) * pitch8) : 0xc0000000u;
    const uint32_t da = (uint32_t)(row0 - T.lrlo), db = (uint32_t)(LL - 2 - row0 - T.lrlo);
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const bool oka = da + (uint32_t)(i * S::RSTEP) <= T.lspan;
      const bool okb = db - (uint32_t)(i * S::RSTEP) <= T.lspan;
      va[i] = buf_ld_f64(T.rs, oka ? offa + (uint32_t)i * dstep : PHX_BUF_OOB);
      vb[i] = buf_ld_f64(T.rs, okb ? offb - (uint32_t)i * dstep : PHX_BUF_OOB);
    }
  };
  auto fold = [&]() {   // registers -> LDS (see k_dst_yw)
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const double sj = sn[1 + row0 + i * S::RSTEP];
      const double e = sj * (va[i] + vb[i]), o = 0.5 * (va[i] - vb[i]);
      wcol[2 * (zpa + i * ZS)] = e + o;
      wcol[2 * (zpb - i * ZS)] = e - o;
    }
    if (row0 == 0) wcol[0] = 0.0;
  };
  auto issue_stores = [&](const Tile &T) {
    const bool colok = tcol < T.ncols;
    const uint32_t offs = colok ? (uint32_t)(tcol * 8 + row0 * pitch8) : PHX_BUF_OOB;
    const uint32_t ds = (uint32_t)(row0 - T.srlo);
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const bool ok = ds + (uint32_t)(i * S::RSTEP) <= T.sspan;
      buf_st_f64(out[i], T.rs, ok ? offs + (uint32_t)i * dstep : PHX_BUF_OOB);
    }
  };

  int q = (int)blockIdx.x;
  if (q >= ntiles) return;
  const int stride = (int)gridDim.x;
  Tile cur = tile_of(q);
  issue_loads(cur);
  __syncthreads();   // tables
  fold();
  Tile nxt = cur;
  bool has_next = q + stride < ntiles;
  if (has_next) { nxt = tile_of(q + stride); issue_loads(nxt); }
  __syncthreads();
  for (;;) {
    dst_core<double, false, LL, false>(w, scr, P, t, 2 * pr < cur.ncols, tw, sn);   // ends with a block barrier
#pragma unroll
    for (int i = 0; i < NS; ++i) out[i] = wcol[2 * (zr + i * ZS)];
    __syncthreads();   // every result is in registers before the next tile overwrites the arrays
    if (has_next) fold();
    issue_stores(cur);
    if (!has_next) break;
    cur = nxt;
    q += stride;
    has_next = q + stride < ntiles;
    if (has_next) { nxt = tile_of(q + stride); issue_loads(nxt); }
    __syncthreads();
  }
}

// ---- x pass.  Group q = PAIRS consecutive pairs of x lines; block b takes groups b, b + gridDim.x, ...
template <int LL, int IO, bool SC>
__global__ void __launch_bounds__(LongShape<LL>::NTHR)
k_dst_xl(BoxGrid g, DstPlan P, double *__restrict__ G, const int32_t *__restrict__ gmap,
         const double *__restrict__ vin, double *__restrict__ vout, const double *__restrict__ dscale,
         const uint8_t *__restrict__ line_any, uint32_t vec_bytes, int ngroups) {
  using S = LongShape<LL>;
  extern __shared__ double2 zs_raw[];
  C2<double> *zs = reinterpret_cast<C2<double> *>(zs_raw);
  constexpr int TP = S::TP;
  constexpr int ZS = TP + TP / 8;
  const int tid = (int)threadIdx.x;
  const int pr = tid / TP, t = tid % TP;
  const int nlines = g.m[1] * g.m[2];
  const C2<double> *tw;
  const double *sn;
  stage_tables_long<LL>(zs, P, &tw, &sn);
  C2<double> *w = zs + pr * S::ZL, *scr = zs + S::PAIRS * S::ZL + pr * S::SCR;
  const uint32_t lat = (uint32_t)nlines * (uint32_t)g.pitch;
  const __amdgpu_buffer_rsrc_t rsG = buf_rsrc(G, lat * 8u);
  const __amdgpu_buffer_rsrc_t rsM = buf_rsrc(gmap, IO != 0 ? lat * 4u : 0u);
  const __amdgpu_buffer_rsrc_t rsV = buf_rsrc(IO == 1 ? (const void *)vin : (const void *)vout, IO != 0 ? vec_bytes : 0u);
  const __amdgpu_buffer_rsrc_t rsS = buf_rsrc(dscale, SC ? vec_bytes : 0u);
  const int zpa = ZP(1 + t), zpb = ZP(LL - 1 - t);
  double va[4], vb[4], ua[4], ub[4];
  C2<double> F[8];

  struct Grp { bool has0, has1, any; uint32_t base0, base1; };
  auto group_of = [&](int q) {
    Grp Q;
    const int l = (q * S::PAIRS + pr) * 2;
    Q.has0 = l < nlines && !(IO != 0 && line_any && !line_any[min(l, nlines - 1)]);
    Q.has1 = l + 1 < nlines && !(IO != 0 && line_any && !line_any[min(l + 1, nlines - 1)]);
    Q.base0 = (uint32_t)l * (uint32_t)g.pitch;
    Q.base1 = Q.base0 + (uint32_t)g.pitch;
    // a group whose lines hold no mapped point does nothing (block-uniform: every thread counts the group's flags)
    Q.any = true;
    if (IO != 0 && line_any) {
      Q.any = false;
      const int l0 = q * S::PAIRS * 2, l1 = min(l0 + 2 * S::PAIRS, nlines);
      for (int k = l0; k < l1; ++k) Q.any |= line_any[k] != 0;
    }
    return Q;
  };
  auto issue_loads = [&](const Grp &Q) {
    const uint32_t ea = Q.base0 + (uint32_t)t, eb = Q.base1 + (uint32_t)t;
    const uint32_t fa = Q.base0 + (uint32_t)(LL - 2 - t), fb = Q.base1 + (uint32_t)(LL - 2 - t);
    const bool h0 = Q.has0 && Q.any, h1 = Q.has1 && Q.any;
    if (IO == 1) {
      int32_t qa[4], qb[4], pa[4], pb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        qa[i] = buf_ld_i32(rsM, h0 ? (ea + (uint32_t)(i * TP)) * 4u : PHX_BUF_OOB);
        qb[i] = buf_ld_i32(rsM, h1 ? (eb + (uint32_t)(i * TP)) * 4u : PHX_BUF_OOB);
        pa[i] = buf_ld_i32(rsM, h0 ? (fa - (uint32_t)(i * TP)) * 4u : PHX_BUF_OOB);
        pb[i] = buf_ld_i32(rsM, h1 ? (fb - (uint32_t)(i * TP)) * 4u : PHX_BUF_OOB);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t oqa = h0 ? (uint32_t)qa[i] << 3 : PHX_BUF_OOB, oqb = h1 ? (uint32_t)qb[i] << 3 : PHX_BUF_OOB;
        const uint32_t opa = h0 ? (uint32_t)pa[i] << 3 : PHX_BUF_OOB, opb = h1 ? (uint32_t)pb[i] << 3 : PHX_BUF_OOB;
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
        va[i] = buf_ld_f64(rsG, h0 ? (ea + (uint32_t)(i * TP)) * 8u : PHX_BUF_OOB);
        vb[i] = buf_ld_f64(rsG, h1 ? (eb + (uint32_t)(i * TP)) * 8u : PHX_BUF_OOB);
        ua[i] = buf_ld_f64(rsG, h0 ? (fa - (uint32_t)(i * TP)) * 8u : PHX_BUF_OOB);
        ub[i] = buf_ld_f64(rsG, h1 ? (fb - (uint32_t)(i * TP)) * 8u : PHX_BUF_OOB);
      }
    }
  };
  auto fold = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const double sj = sn[1 + t + i * TP];
      const C2<double> e = mk<double>(sj * (va[i] + ua[i]), sj * (vb[i] + ub[i]));
      const C2<double> o = mk<double>(0.5 * (va[i] - ua[i]), 0.5 * (vb[i] - ub[i]));
      w[zpa + i * ZS] = cadd(e, o);
      w[zpb - i * ZS] = csub(e, o);
    }
    if (t == 0) w[0] = mk<double>(0.0, 0.0);
  };
  auto issue_stores = [&](const Grp &Q) {
    const bool h0 = Q.has0 && Q.any, h1 = Q.has1 && Q.any;
    if (IO == 2) {
      int32_t qa[8], qb[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const bool kin = i < 7 || t < TP - 1;
        qa[i] = buf_ld_i32(rsM, h0 && kin ? (Q.base0 + (uint32_t)(t + i * TP)) * 4u : PHX_BUF_OOB);
        qb[i] = buf_ld_i32(rsM, h1 && kin ? (Q.base1 + (uint32_t)(t + i * TP)) * 4u : PHX_BUF_OOB);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const bool kin = i < 7 || t < TP - 1;
        const uint32_t oa = h0 && kin ? (uint32_t)qa[i] << 3 : PHX_BUF_OOB, ob = h1 && kin ? (uint32_t)qb[i] << 3 : PHX_BUF_OOB;
        double xa = F[i].x, xb = F[i].y;
        if (SC) { xa *= buf_ld_f64(rsS, oa); xb *= buf_ld_f64(rsS, ob); }
        buf_st_f64(xa, rsV, oa);
        buf_st_f64(xb, rsV, ob);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const bool kin = i < 7 || t < TP - 1;
        buf_st_f64(F[i].x, rsG, h0 && kin ? (Q.base0 + (uint32_t)(t + i * TP)) * 8u : PHX_BUF_OOB);
        buf_st_f64(F[i].y, rsG, h1 && kin ? (Q.base1 + (uint32_t)(t + i * TP)) * 8u : PHX_BUF_OOB);
      }
    }
  };

  int q = (int)blockIdx.x;
  if (q >= ngroups) return;
  const int stride = (int)gridDim.x;
  const int zk = ZP(1 + t);
  Grp cur = group_of(q);
  issue_loads(cur);
  __syncthreads();   // tables
  fold();
  Grp nxt = cur;
  bool has_next = q + stride < ngroups;
  if (has_next) { nxt = group_of(q + stride); issue_loads(nxt); }
  __syncthreads();
  for (;;) {
    // (a group without a mapped point skips its transform -- block-uniform -- and stores nothing)
    if (cur.any) dst_core<double, false, LL, false>(w, scr, P, t, (q * S::PAIRS + pr) * 2 < nlines, tw, sn);
#pragma unroll
    for (int i = 0; i < 8; ++i) F[i] = w[zk + i * ZS];
    __syncthreads();
    if (has_next) fold();
    issue_stores(cur);
    if (!has_next) break;
    cur = nxt;
    q += stride;
    has_next = q + stride < ngroups;
    if (has_next) { nxt = group_of(q + stride); issue_loads(nxt); }
    __syncthreads();
  }
}

#define PHX_DST_LONG_LENGTHS(X) X(384) X(768) X(1024)

template <int LL>
static int dst_long_allow_lds() {
  const int bytes = 160 * 1024;
  PHX_HIP(hipFuncSetAttribute((const void *)k_dst_yl<LL>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  PHX_HIP(hipFuncSetAttribute((const void *)k_dst_xl<LL, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  PHX_HIP(hipFuncSetAttribute((const void *)k_dst_xl<LL, 1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  PHX_HIP(hipFuncSetAttribute((const void *)k_dst_xl<LL, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  PHX_HIP(hipFuncSetAttribute((const void *)k_dst_xl<LL, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  PHX_HIP(hipFuncSetAttribute((const void *)k_dst_xl<LL, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  return PHX_OK;
}
template <int LL>
static bool dst_long_shape_ok(const DstPlan &p) {
  using S = LongShape<LL>;
  return p.L == LL && !p.wave && p.pairs == S::PAIRS && p.slot == S::TP && p.scr == S::SCR && p.tab_off == S::TAB;
}
// blocks a CU holds (by LDS) x CUs: the persistent grid
static int dst_long_grid(const DstPlan &p, int64_t nwork) {
  const size_t lds = (size_t)p.lds_elems * 16;
  const int per_cu = (int)std::max<size_t>(1, std::min<size_t>((size_t)(160 * 1024) / std::max<size_t>(lds, 1), 2048 / (size_t)(p.pairs * p.slot)));
  return (int)std::min<int64_t>(nwork, (int64_t)per_cu * 256);
}
static bool dst_long_fast(const BoxGrid &g, const DstPlan &p, int64_t nvec) {
  static const bool off = getenv("PHX_DST_OLD") != nullptr || getenv("PHX_DST_GENERIC") != nullptr;
  if (off) return false;
  const int64_t lat = g.plane * g.m[2];
  if (g.plane != g.pitch * g.m[1] || lat * 8 >= (int64_t)PHX_BUF_OOB || nvec * 8 >= (int64_t)PHX_BUF_OOB) return false;
  bool ok = false;
#define X(L_) ok = ok || dst_long_shape_ok<L_>(p);
  PHX_DST_LONG_LENGTHS(X)
#undef X
  return ok;
}
