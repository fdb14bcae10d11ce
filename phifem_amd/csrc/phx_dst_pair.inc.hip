// Sine-transform passes for the LONG f64 lengths (384, 768, 1024): ONE WAVEFRONT PER PAIR of lines -- included by
// phx_precond.inc.hip behind phx_dst_wave.inc.hip.  (PHX_DST_OLD=1: the generic kernels k_dst_x / k_dst_s, the A/B reference.)
//
// Why: the kernels of round 3 (k_dst_xl / k_dst_yl, docs/HISTORY.md) spread a pair over L / 8 threads = 1.5 wavefronts (L = 768), so every Stockham stage and every
// step of the prefix sum met at a BLOCK barrier: eleven barriers over twelve wavefronts per tile, 15 us of transform time per
// tile against 10 us of memory time and 5 us of LDS pipe time.  Here a pair lives inside one wavefront (64 lanes x 12
// elements at L = 768, 32 x 12 at 384 -- two pairs per wavefront --, 64 x 16 at 1024), the stages synchronise wave-locally
// (LDS operations of a wave execute in order), the prefix sum runs on DPP moves, and the waves of a block drift apart
// freely: the x pass has no block barrier inside its loop at all, the y pass three per tile (around the tile's fold and
// read-out, whose thread <-> element map differs from the transform's).
//   schedule (radix, sub-transform size so far):  768: (4,1) (4,4) (4,16) (12,64)    384: (4,1) (4,4) (2,16) (12,32)
//                                                1024: (8,1) (8,8) (4,64) (4,256)
// The last stage of the 12-element shapes is one radix-12 butterfly per lane, done in registers as a prime-factor 3 x 4
// transform (no internal twiddles) and IN PLACE (p = L / 12 = lanes of the pair: input and output positions coincide).
// Blocks are PERSISTENT and software-pipelined over their tiles (one block per CU: the tile takes most of the LDS): while
// tile n is transformed in LDS the loads of tile n + 1 are in flight into registers and the stores of tile n - 1 drain.
// Loader / storer idioms as in phx_dst_wave.inc.hip: compile-time shapes, buffer descriptors, out-of-range offsets for
// masked accesses, gathers through the map without a mask.

// LDS layout of a pair's sequence (16-byte elements).  L = 384 / 768: NO padding, an XOR swizzle of the low four index bits
//   S(n) = n ^ (14 [n bit 4] | [n bit 3])
// under which every access of the transform is free of bank conflicts on gfx950 -- the reads of ds_read_b128 are served in
// the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32) over 64 banks, the writes of ds_write_b128 in groups of
// eight consecutive lanes over 32 banks (MI355X_MICROARCH.md, LDS): aligned runs of consecutive elements stay runs, the
// stride-4 and stride-16 scatters of the radix-4 stages and the stride-6 / stride-12 accesses of the post-processing spread
// over all banks (checked by enumeration, tools/r04/lds_conflicts.py; the one-in-eight padding of k_dst_x / k_dst_s made 43 %
// of the LDS cycles of the first version of these kernels conflict cycles, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE).
// S(n + m) = S(n) + m for m a multiple of 32.  L = 1024 (radix-8 stages, eight k per lane) keeps the padding.
template <int LL> struct PairLay {
  static constexpr bool SWZ = LL != 1024;
  static constexpr int ZL = SWZ ? LL + 1 : ZLEN(LL);     // odd pair stride: the columns of a y tile spread over the banks
  static __device__ __forceinline__ int idx(int n) { return SWZ ? (n ^ ((0xFE10 >> ((n >> 1) & 12)) & 15)) : ZP(n); }
  static constexpr int lin(int m) { return SWZ ? m : m + m / 8; }   // idx(n + m) - idx(n), m a multiple of 32 (SWZ) / 8
};

template <int LL, int NP = 0> struct PairShape {
  static constexpr int TP = LL >= 768 ? 64 : 32;                      // lanes of a pair
  static constexpr int EPT = LL / TP;                                 // elements per lane: 12, 12, 16
  static constexpr int KPT = EPT / 2;                                 // spectral indices k per lane in the post-processing
  static constexpr int PAIRS = NP ? NP : (LL == 384 ? 16 : (LL == 768 ? 8 : 6));
  static constexpr int NTHR = PAIRS * TP;
  static constexpr int W = 2 * PAIRS;                                 // columns of a y tile
  static constexpr int RSTEP = NTHR / W;                              // = TP / 2
  static constexpr int H = LL / 2;
  static constexpr int ZL = PairLay<LL>::ZL;
  static constexpr int TAB = PAIRS * ZL + (PAIRS * ZL & 1);           // table copies behind the pair arrays
  static constexpr int LDS_ELEMS = TAB + LL + (LL / 2 + 2 + 1) / 2;   // complex doubles
  static_assert(EPT * TP == LL && RSTEP % 8 == 0 && H % RSTEP == 0 && (TP == 64 || PAIRS % 2 == 0), "shape");
};

// prime-factor 12-point transform: n = (4 n1 + 3 n2) mod 12, k = (4 k1 + 9 k2) mod 12 (k1 = k mod 3, k2 = k mod 4)
__device__ __forceinline__ void dft12(C2<double> *v) {
  C2<double> b[4][3];
#pragma unroll
  for (int n2 = 0; n2 < 4; ++n2) {
    C2<double> a[3] = {v[(3 * n2) % 12], v[(4 + 3 * n2) % 12], v[(8 + 3 * n2) % 12]};
    dft3(a);
    b[n2][0] = a[0]; b[n2][1] = a[1]; b[n2][2] = a[2];
  }
#pragma unroll
  for (int k1 = 0; k1 < 3; ++k1) {
    C2<double> c[4] = {b[0][k1], b[1][k1], b[2][k1], b[3][k1]};
    dft4(c);
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) v[(4 * k1 + 9 * k2) % 12] = c[k2];
  }
}

// one Stockham stage of radix R, sub-transform size P so far, on the LL-point sequence `z` of this pair (`t` = lane within
// the pair, TP lanes, EPT / R butterflies each).  Wave-local synchronisation.
template <int LL, int TP, int R, int P>
__device__ __forceinline__ void pair_stage(C2<double> *z, int t, const C2<double> *__restrict__ tw) {
  constexpr int EPT = LL / TP, NB = LL / R, MAXB = EPT / R, TWS = LL / (P * R);
  using Y = PairLay<LL>;
  constexpr int NBP = Y::lin(NB);   // idx(i + q NB) = idx(i) + q NBP
  static_assert(NB % 32 == 0 && EPT % R == 0 && NB == MAXB * TP && (P & (P - 1)) == 0, "stage");
  constexpr bool INPLACE = NB == P;  // last stage: k = i, j = k, outputs at the inputs' positions
  C2<double> u[MAXB][R];
#pragma unroll
  for (int b = 0; b < MAXB; ++b) {
    const int i = t + b * TP;
    const int k = i & (P - 1);
    const C2<double> *zr = z + Y::idx(i);
    const C2<double> *twq = tw;
    const int step = TWS * k;
#pragma unroll
    for (int q = 0; q < R; ++q) {
      C2<double> w = zr[q * NBP];
      if (q > 0 && P > 1) w = cmul(w, *twq);
      u[b][q] = w;
      twq += step;
    }
    if constexpr (R == 12) dft12(u[b]);
    else if constexpr (R == 8) dft8(u[b]);
    else if constexpr (R == 4) dft4(u[b]);
    else dft2(u[b]);
  }
  if (!INPLACE) psync<true>();
#pragma unroll
  for (int b = 0; b < MAXB; ++b) {
    const int i = t + b * TP;
    const int k = i & (P - 1);
    const int j = (i - k) * R + k;
#pragma unroll
    for (int q = 0; q < R; ++q) z[Y::idx(j + q * P)] = u[b][q];
  }
  psync<true>();
}

template <int LL, int TP>
__device__ __forceinline__ void pair_fft(C2<double> *z, int t, const C2<double> *tw) {
  if constexpr (LL == 768) {
    pair_stage<LL, TP, 4, 1>(z, t, tw); pair_stage<LL, TP, 4, 4>(z, t, tw);
    pair_stage<LL, TP, 4, 16>(z, t, tw); pair_stage<LL, TP, 12, 64>(z, t, tw);
  } else if constexpr (LL == 384) {
    pair_stage<LL, TP, 4, 1>(z, t, tw); pair_stage<LL, TP, 4, 4>(z, t, tw);
    pair_stage<LL, TP, 2, 16>(z, t, tw); pair_stage<LL, TP, 12, 32>(z, t, tw);
  } else {
    static_assert(LL == 1024, "length");
    pair_stage<LL, TP, 8, 1>(z, t, tw); pair_stage<LL, TP, 8, 8>(z, t, tw);
    pair_stage<LL, TP, 4, 64>(z, t, tw); pair_stage<LL, TP, 4, 256>(z, t, tw);
  }
}

// In: w[idx(j)] = folded sequence y_j of the two lines (dst_core with PRE = false).  Out: w[idx(k)] = (F^a_k, F^b_k), k = 1 .. L-1.
// Every lane of the wavefront calls it (pairs of a half-filled wave transform zeros).
template <int LL, int TP>
__device__ __forceinline__ void pair_core(C2<double> *w, int t, const C2<double> *tw) {
  constexpr int KPT = LL / TP / 2;
  using Y = PairLay<LL>;
  pair_fft<LL, TP>(w, t, tw);
  C2<double> Wk[KPT], Wm[KPT], c[KPT];
#pragma unroll
  for (int i = 0; i < KPT; ++i) {
    const int k = KPT * t + i;
    Wk[i] = w[Y::idx(k)];
    Wm[i] = w[Y::idx(k == 0 ? 0 : LL - k)];
  }
  psync<true>();
  C2<double> run = mk<double>(0.0, 0.0);
#pragma unroll
  for (int i = 0; i < KPT; ++i) {
    const int k = KPT * t + i;
    C2<double> R;
    if (k == 0) {
      R = mk<double>(0.5 * Wk[i].x, 0.5 * Wk[i].y);
    } else {
      R = mk<double>(0.5 * (Wk[i].x + Wm[i].x), 0.5 * (Wk[i].y + Wm[i].y));
      w[Y::idx(2 * k)] = mk<double>(-0.5 * (Wk[i].y - Wm[i].y), 0.5 * (Wk[i].x - Wm[i].x));
    }
    run = cadd(run, R);
    c[i] = run;
  }
  // inclusive scan of the lane totals inside each row of 16 lanes, then across the rows of the pair (see dst_core)
  C2<double> inc = run;
  inc = cadd(inc, dpp_c2<0x111>(inc));
  inc = cadd(inc, dpp_c2<0x112>(inc));
  inc = cadd(inc, dpp_c2<0x114>(inc));
  inc = cadd(inc, dpp_c2<0x118>(inc));
  inc = cadd(inc, dpp_c2<0x142, 0xa>(inc));
  if (TP == 64) inc = cadd(inc, dpp_c2<0x143, 0xc>(inc));
  const C2<double> ex = dpp_c2<0x138>(inc);   // wave_shr:1
  const C2<double> E = t == 0 ? mk<double>(0.0, 0.0) : ex;
#pragma unroll
  for (int i = 0; i < KPT; ++i) w[Y::idx(2 * (KPT * t + i) + 1)] = cadd(E, c[i]);
  psync<true>();
}

template <int LL, int NP>
__device__ __forceinline__ void stage_tables_pair(C2<double> *zs, const DstPlan &P, const C2<double> **tw, const double **sn) {
  using S = PairShape<LL, NP>;
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
// A thread owns, of its column, the row pairs (j - 1, L - 1 - j), j = 1 + row0 + i RSTEP <= L / 2: it loads them, folds them
// into LDS, and -- after the transform -- reads the SAME elements back and stores them: read-out and next fold touch only the
// thread's own LDS elements, so no barrier stands between them (two block barriers per tile: before and after the transform).
template <int LL, int NP>
__global__ void __launch_bounds__((PairShape<LL, NP>::NTHR))
k_dst_yp(BoxGrid g, DstPlan P, double *__restrict__ G, const int2 *__restrict__ row_any, int dir, int ncb, int ntiles) {
  using S = PairShape<LL, NP>;
  using Y = PairLay<LL>;
  extern __shared__ double2 zs_raw[];
  C2<double> *zs = reinterpret_cast<C2<double> *>(zs_raw);
  constexpr int len = LL - 1;
  constexpr int NT = S::H / S::RSTEP;
  const int tid = (int)threadIdx.x;
  const int tcol = tid % S::W, row0 = tid / S::W;
  const int pr = tid / S::TP, t = tid % S::TP;
  const int pitch8 = (int)g.pitch * 8;
  const uint32_t dstep = (uint32_t)(S::RSTEP * pitch8);
  const C2<double> *tw;
  const double *sn;
  stage_tables_pair<LL, NP>(zs, P, &tw, &sn);
  double *wcol = reinterpret_cast<double *>(zs + (tcol >> 1) * S::ZL) + (tcol & 1);
  C2<double> *w = zs + pr * S::ZL;
  double va[NT], vb[NT];

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
      if ((dir & 3) == 1) { T.lrlo = rlo; T.lspan = span; } else { T.srlo = rlo; T.sspan = span; }
    }
    return T;
  };
  auto issue_loads = [&](const Tile &T) {
    const bool colok = tcol < T.ncols && !(dir & 0x200);
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
  // (optionally) the results of the finished tile out of LDS into oa / ob, then the fold of va / vb into the same elements
  double oa[NT], ob[NT];
  auto swap_tile = [&](bool read_out, bool fold_in) {
    // the swizzled LDS indices are recomputed per tile: hoisted out of the tile loop they (and those of the stages) cost
    // some 100 registers and spilled
    int r0 = row0;
    asm volatile("" : "+v"(r0));
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int j = 1 + r0 + i * S::RSTEP;
      double *pa = wcol + 2 * Y::idx(j), *pb = wcol + 2 * Y::idx(LL - j);
      if (read_out) { oa[i] = *pa; ob[i] = *pb; }
      if (fold_in) {
        const double sj = sn[j];
        const double e = sj * (va[i] + vb[i]), o = 0.5 * (va[i] - vb[i]);
        *pa = e + o;
        *pb = e - o;   // j = L / 2: the same element, e + o = e - o = 2 x_H
      }
    }
    if (fold_in && row0 == 0) wcol[0] = 0.0;
  };
  auto issue_stores = [&](const Tile &T) {
    const bool colok = tcol < T.ncols && !(dir & 0x200);
    const uint32_t offa = colok ? (uint32_t)(tcol * 8 + row0 * pitch8) : PHX_BUF_OOB;
    const uint32_t offb = colok ? (uint32_t)(tcol * 8 + (LL - 2 - row0) * pitch8) : 0xc0000000u;
    const uint32_t da = (uint32_t)(row0 - T.srlo), db = (uint32_t)(LL - 2 - row0 - T.srlo);
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const bool oka = da + (uint32_t)(i * S::RSTEP) <= T.sspan;
      const bool okb = db - (uint32_t)(i * S::RSTEP) <= T.sspan;
      buf_st_f64(oa[i], T.rs, oka ? offa + (uint32_t)i * dstep : PHX_BUF_OOB);
      buf_st_f64(ob[i], T.rs, okb ? offb - (uint32_t)i * dstep : PHX_BUF_OOB);
    }
  };

  // blocks b and b + 8 share an XCD (round-robin placement; speed only): consecutive tiles of a sweep go to one XCD
  int q = (int)blockIdx.x;
  if ((gridDim.x & 7) == 0) q = (q & 7) * (int)(gridDim.x >> 3) + (q >> 3);
  if (q >= ntiles) return;
  const int stride = (int)gridDim.x;
  Tile cur = tile_of(q);
  issue_loads(cur);
  __syncthreads();   // tables
  swap_tile(false, true);
  Tile nxt = cur;
  bool has_next = q + stride < ntiles;
  if (has_next) { nxt = tile_of(q + stride); issue_loads(nxt); }
  __syncthreads();
  for (;;) {
    // a wavefront whose pair(s) lie beyond the tile's columns transforms nothing (wave-uniform)
    if (2 * (pr - (S::TP == 32 ? (pr & 1) : 0)) < cur.ncols && !(dir & 0x100)) {
      int tt = t;
      asm volatile("" : "+v"(tt));
      pair_core<LL, S::TP>(w, tt, tw);
    }
    __syncthreads();   // every pair of the tile is transformed
    swap_tile(true, has_next);
    issue_stores(cur);
    if (!has_next) break;
    cur = nxt;
    q += stride;
    has_next = q + stride < ntiles;
    if (has_next) { nxt = tile_of(q + stride); issue_loads(nxt); }
    __syncthreads();   // the next tile is folded
  }
}

// ---- x pass.  Wave w of block b takes the pairs (q PAIRS + w) (x TP / 64 pairs), q = b, b + gridDim.x, ...; no block
// barrier behind the staging of the tables.
template <int LL, int NP, int IO, bool SC>
__global__ void __launch_bounds__((PairShape<LL, NP>::NTHR))
k_dst_xp(BoxGrid g, DstPlan P, double *__restrict__ G, const int32_t *__restrict__ gmap,
         const double *__restrict__ vin, double *__restrict__ vout, const double *__restrict__ dscale,
         const uint8_t *__restrict__ line_any, uint32_t vec_bytes, int ngroups) {
  using S = PairShape<LL, NP>;
  extern __shared__ double2 zs_raw[];
  C2<double> *zs = reinterpret_cast<C2<double> *>(zs_raw);
  using Y = PairLay<LL>;
  constexpr int TP = S::TP, EPT = S::EPT, HP = EPT / 2;
  constexpr int ZS = Y::lin(TP);   // idx(n + TP) - idx(n)
  const int tid = (int)threadIdx.x;
  const int pr = tid / TP, t = tid % TP;
  const int nlines = g.m[1] * g.m[2];
  const C2<double> *tw;
  const double *sn;
  stage_tables_pair<LL, NP>(zs, P, &tw, &sn);
  C2<double> *w = zs + pr * S::ZL;
  const uint32_t lat = (uint32_t)nlines * (uint32_t)g.pitch;
  const __amdgpu_buffer_rsrc_t rsG = buf_rsrc(G, lat * 8u);
  const __amdgpu_buffer_rsrc_t rsM = buf_rsrc(gmap, IO != 0 ? lat * 4u : 0u);
  const __amdgpu_buffer_rsrc_t rsV = buf_rsrc(IO == 1 ? (const void *)vin : (const void *)vout, IO != 0 ? vec_bytes : 0u);
  const __amdgpu_buffer_rsrc_t rsS = buf_rsrc(dscale, SC ? vec_bytes : 0u);
  const int zpa = Y::idx(1 + t), zpb = Y::idx(LL - 1 - t);
  double va[HP], vb[HP], ua[HP], ub[HP];
  C2<double> F[EPT];

  struct Grp { bool has0, has1, any; uint32_t base0, base1; };
  auto group_of = [&](int q) {
    Grp Q;
    const int l = (q * S::PAIRS + pr) * 2;
    Q.has0 = l < nlines && !(IO != 0 && line_any && !line_any[min(l, nlines - 1)]);
    Q.has1 = l + 1 < nlines && !(IO != 0 && line_any && !line_any[min(l + 1, nlines - 1)]);
    Q.base0 = (uint32_t)l * (uint32_t)g.pitch;
    Q.base1 = Q.base0 + (uint32_t)g.pitch;
    // a wavefront whose lines hold no mapped point does nothing (wave-uniform)
    Q.any = __ballot(Q.has0 || Q.has1) != 0;
    return Q;
  };
  auto issue_loads = [&](const Grp &Q) {
    const uint32_t ea = Q.base0 + (uint32_t)t, eb = Q.base1 + (uint32_t)t;
    const uint32_t fa = Q.base0 + (uint32_t)(LL - 2 - t), fb = Q.base1 + (uint32_t)(LL - 2 - t);
    const bool h0 = Q.has0, h1 = Q.has1;
    if (IO == 1) {
      int32_t qa[HP], qb[HP], pa[HP], pb[HP];
#pragma unroll
      for (int i = 0; i < HP; ++i) {
        qa[i] = buf_ld_i32(rsM, h0 ? (ea + (uint32_t)(i * TP)) * 4u : PHX_BUF_OOB);
        qb[i] = buf_ld_i32(rsM, h1 ? (eb + (uint32_t)(i * TP)) * 4u : PHX_BUF_OOB);
        pa[i] = buf_ld_i32(rsM, h0 ? (fa - (uint32_t)(i * TP)) * 4u : PHX_BUF_OOB);
        pb[i] = buf_ld_i32(rsM, h1 ? (fb - (uint32_t)(i * TP)) * 4u : PHX_BUF_OOB);
      }
#pragma unroll
      for (int i = 0; i < HP; ++i) {
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
      for (int i = 0; i < HP; ++i) {
        va[i] = buf_ld_f64(rsG, h0 ? (ea + (uint32_t)(i * TP)) * 8u : PHX_BUF_OOB);
        vb[i] = buf_ld_f64(rsG, h1 ? (eb + (uint32_t)(i * TP)) * 8u : PHX_BUF_OOB);
        ua[i] = buf_ld_f64(rsG, h0 ? (fa - (uint32_t)(i * TP)) * 8u : PHX_BUF_OOB);
        ub[i] = buf_ld_f64(rsG, h1 ? (fb - (uint32_t)(i * TP)) * 8u : PHX_BUF_OOB);
      }
    }
  };
  // the lane holds the pairs (j, L - j), j = 1 + t + i TP, i < EPT / 2 (j = 1 .. L / 2; j = L / 2 pairs an element with itself)
  auto fold = [&]() {
#pragma unroll
    for (int i = 0; i < HP; ++i) {
      const double sj = sn[1 + t + i * TP];
      const C2<double> e = mk<double>(sj * (va[i] + ua[i]), sj * (vb[i] + ub[i]));
      const C2<double> o = mk<double>(0.5 * (va[i] - ua[i]), 0.5 * (vb[i] - ub[i]));
      w[zpa + i * ZS] = cadd(e, o);
      w[zpb - i * ZS] = csub(e, o);
    }
    if (t == 0) w[0] = mk<double>(0.0, 0.0);
  };
  auto issue_stores = [&](const Grp &Q) {
    const bool h0 = Q.has0, h1 = Q.has1;
    if (IO == 2) {
      int32_t qa[EPT], qb[EPT];
#pragma unroll
      for (int i = 0; i < EPT; ++i) {
        const bool kin = i < EPT - 1 || t < TP - 1;
        qa[i] = buf_ld_i32(rsM, h0 && kin ? (Q.base0 + (uint32_t)(t + i * TP)) * 4u : PHX_BUF_OOB);
        qb[i] = buf_ld_i32(rsM, h1 && kin ? (Q.base1 + (uint32_t)(t + i * TP)) * 4u : PHX_BUF_OOB);
      }
#pragma unroll
      for (int i = 0; i < EPT; ++i) {
        const bool kin = i < EPT - 1 || t < TP - 1;
        const uint32_t oa = h0 && kin ? (uint32_t)qa[i] << 3 : PHX_BUF_OOB, ob = h1 && kin ? (uint32_t)qb[i] << 3 : PHX_BUF_OOB;
        double xa = F[i].x, xb = F[i].y;
        if (SC) { xa *= buf_ld_f64(rsS, oa); xb *= buf_ld_f64(rsS, ob); }
        buf_st_f64(xa, rsV, oa);
        buf_st_f64(xb, rsV, ob);
      }
    } else {
#pragma unroll
      for (int i = 0; i < EPT; ++i) {
        const bool kin = i < EPT - 1 || t < TP - 1;
        buf_st_f64(F[i].x, rsG, h0 && kin ? (Q.base0 + (uint32_t)(t + i * TP)) * 8u : PHX_BUF_OOB);
        buf_st_f64(F[i].y, rsG, h1 && kin ? (Q.base1 + (uint32_t)(t + i * TP)) * 8u : PHX_BUF_OOB);
      }
    }
  };

  int q = (int)blockIdx.x;
  __syncthreads();   // tables (every thread of the block, before any wave leaves)
  if (q >= ngroups) return;
  const int stride = (int)gridDim.x;
  const int zk = Y::idx(1 + t);
  Grp cur = group_of(q);
  if (cur.any) issue_loads(cur);
  if (cur.any) fold();
  Grp nxt = cur;
  bool has_next = q + stride < ngroups;
  if (has_next) { nxt = group_of(q + stride); if (nxt.any) issue_loads(nxt); }
  psync<true>();
  for (;;) {
    if (cur.any) {
      int tt = t;
      asm volatile("" : "+v"(tt));   // LDS indices of the stages recomputed per pair, not kept in registers across the loop
      pair_core<LL, TP>(w, tt, tw);
#pragma unroll
      for (int i = 0; i < EPT; ++i) F[i] = w[zk + i * ZS];
      psync<true>();
    }
    if (has_next && nxt.any) fold();
    if (cur.any) issue_stores(cur);
    if (!has_next) break;
    cur = nxt;
    q += stride;
    has_next = q + stride < ngroups;
    if (has_next) { nxt = group_of(q + stride); if (nxt.any) issue_loads(nxt); }
    psync<true>();
  }
}

#define PHX_DST_PAIR_LENGTHS(X) X(384) X(768) X(1024)
// alternative y-tile widths (PHX_DST_YALT=1): as many pairs as the LDS holds
template <int LL>
static int dst_pair_allow_lds() {
  const int bytes = 160 * 1024;
  PHX_HIP(hipFuncSetAttribute((const void *)k_dst_yp<LL, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  PHX_HIP(hipFuncSetAttribute((const void *)k_dst_xp<LL, 0, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  PHX_HIP(hipFuncSetAttribute((const void *)k_dst_xp<LL, 0, 1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  PHX_HIP(hipFuncSetAttribute((const void *)k_dst_xp<LL, 0, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  PHX_HIP(hipFuncSetAttribute((const void *)k_dst_xp<LL, 0, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  PHX_HIP(hipFuncSetAttribute((const void *)k_dst_xp<LL, 0, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  return PHX_OK;
}
// blocks a CU holds (by LDS and threads) x CUs: the persistent grid
template <int LL, int NP = 0>
static int dst_pair_grid(int64_t nwork) {
  using S = PairShape<LL, NP>;
  const size_t lds = (size_t)S::LDS_ELEMS * 16;
  const int per_cu = (int)std::max<size_t>(1, std::min<size_t>((size_t)(160 * 1024) / lds, 2048 / (size_t)S::NTHR));
  return (int)std::min<int64_t>(nwork, (int64_t)per_cu * 256);
}
static bool dst_pair_fast(const BoxGrid &g, const DstPlan &p, int64_t nvec) {
  static const bool off = getenv("PHX_DST_OLD") != nullptr || getenv("PHX_DST_GENERIC") != nullptr;
  if (off) return false;
  const int64_t lat = g.plane * g.m[2];
  if (g.plane != g.pitch * g.m[1] || lat * 8 >= (int64_t)PHX_BUF_OOB || nvec * 8 >= (int64_t)PHX_BUF_OOB) return false;
  return p.L == 384 || p.L == 768 || p.L == 1024;
}
