// artn_xgemm_pc_kernel.h -- artn_k_xgemm_pc: the extent-based GEMM (artn_xgemm_kernel.h, artn_xgemm_plan.h) with the waves of a
// workgroup SPLIT BY ROLE.  Included by artn_kernels.hip after artn_xgemm_kernel.h.
//
// artn_k_xgemm runs two 4-wave workgroups per CU; every wave copies AND multiplies, and the two workgroups fall into lockstep:
// both issue the next chunk's loads (2 600-3 000 cycles: table reads, 64-bit addresses, a dozen loads per lane), then both
// multiply (sharing the matrix pipe), then both stand at the barrier -- a full chunk costs 11 300 cycles for 3 072 cycles of
// MFMA issue per wave (in-kernel marks).  Issuing the loads from inside the multiply loop failed twice (the
// compiler waits for ALL loads in flight at every trip of a rolled loop; the unrolled form spills).  Here ONE 8-wave
// workgroup per CU:
//   waves 0-3  CONSUMERS  one per SIMD: operands from LDS, 3 NB accumulators of v_mfma_f32_32x32x2_f32 (3M), the epilogue;
//                          nothing else -- the matrix pipe of a SIMD belongs to one instruction stream of independent MFMAs;
//   waves 4-7  PRODUCERS  one per SIMD beside it: tile tables, address arithmetic, global loads two chunks ahead (two
//                          register sets), registers -> LDS one chunk ahead.  Their vector work issues between the
//                          consumer's MFMAs.
// One workgroup barrier per chunk: behind it the consumers own the LDS stage the producers have just filled, and the
// producers the stage the consumers have just read.  Same plan (ArtnXGemmPlan, kc = 16), same tiles (128 x 32 NB), same tile
// order, same LDS images and tables as artn_k_xgemm -- tests/csrc/plan_emulate.cpp::run_xgemm replays both.

#define ARTN_XGPC_THREADS 512

template <int NB, bool TRANS>
__global__ __launch_bounds__(ARTN_XGPC_THREADS, 2) void artn_k_xgemm_pc(const float2 *__restrict__ A, const float2 *__restrict__ B,
                                                                       float2 *__restrict__ C, const ArtnXGemmPlan P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if ((unsigned)(unsigned long)(lds_byte_t *)smem != 0) __builtin_trap(); // LDS is addressed by raw byte offsets
  constexpr int TM = ARTN_XG_TM, TN = 32 * NB, KC = ARTN_XG_KC, KCL = 4, RSTEP = 256 / KC;
  constexpr int PA = TM + 2, PB = TN + 2;
  constexpr unsigned A_BYTES = KC * PA * 8, B_BYTES = KC * PB * 8, STAGE = A_BYTES + B_BYTES;
  constexpr unsigned LEV = 2 * STAGE;
  constexpr unsigned T_MA0 = LEV, T_MA1 = LEV + 2048, T_NB0 = LEV + 4096, T_NB1 = LEV + 6144, T_KA = LEV + 8192, T_KB = T_KA + ARTN_XG_KTAB * 4;
  constexpr unsigned TT = T_KB + ARTN_XG_KTAB * 4; // tile tables: 4 row sets (rowA, rowC) then 4 column sets (colB, colC), 1 KiB each
  constexpr int NA = TM * KC / 256, NBL = TN * KC / 256; // loads per producer thread and chunk: 8 and 2 NB
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave >= 4;
  const int ptid = tid & 255; // thread index inside its role
  const int j = lane & 31, h = lane >> 5;

  // ---- level tables (once per workgroup)
  auto level_tables = [&](const ArtnXSide &S, unsigned t0, unsigned t1, bool two) {
    if (ptid < S.L0) {
      unsigned o0, o1;
      artn_xg_decode(S, 0, S.n0, (unsigned)ptid, o0, o1);
      lds_write4(t0 + 4u * ptid, o0);
      lds_write4((two ? t0 + 1024u : T_KB) + 4u * ptid, o1);
    }
    if (two && ptid < S.L1) {
      unsigned o0, o1;
      artn_xg_decode(S, S.n0, S.n1, (unsigned)ptid, o0, o1);
      lds_write4(t1 + 4u * ptid, o0);
      lds_write4(t1 + 1024u + 4u * ptid, o1);
    }
  };
  if (!producer) {
    level_tables(P.m, T_MA0, T_MA1, true);
    level_tables(P.n, T_NB0, T_NB1, true);
  } else {
    level_tables(P.k, T_KA, 0u, false);
    if (ptid < ARTN_XG_KC) { // (padding of the k tables: a chunk reads kbase .. kbase + 15 without clamping)
      unsigned o0, o1;
      artn_xg_decode(P.k, 0, P.k.n0, (unsigned)P.k.L0 - 1u, o0, o1);
      lds_write4(T_KA + 4u * (P.k.L0 + ptid), o0);
      lds_write4(T_KB + 4u * (P.k.L0 + ptid), o1);
    }
  }

  // ---- the workgroup's sequence of items (tile, chunk): every wave walks it with its own iterators
  const unsigned G = gridDim.x, tiles_n = (unsigned)P.tiles_n;
  const unsigned total_hm = (unsigned)(P.n_tiles / P.tiles_n);
  unsigned run = (unsigned)(P.n_tiles / ((long)G * 4));
  run = run < 1 ? 1 : (run > 16 ? 16 : run);
  const unsigned n_super = (total_hm + run - 1) / run, n_runs = n_super * tiles_n;
  const unsigned wg = (G & 7) == 0 ? (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3) : blockIdx.x;
  const unsigned K0 = (unsigned)P.k.L0;
  const int cpg = P.cpg;
  const unsigned n_chunks = (unsigned)((long)P.k_groups * cpg);
  struct It {
    unsigned R, pos, r_tn, r_hm0; // run, position inside it
    unsigned c, iq, ig;           // chunk of the tile, its position inside its group, the group
    unsigned seq, cseq;           // tiles / column tiles met so far (table sets: seq & 3, cseq & 3)
    unsigned gA, gB;              // element offsets of the group in A / B (the contracted labels above the k table)
    bool valid;
    XgTile T;
  };
  auto it_init = [&](It &I) {
    I.R = wg; I.pos = 0; I.c = 0; I.iq = 0; I.ig = 0; I.seq = 0; I.cseq = 0; I.gA = 0; I.gB = 0;
    I.valid = I.R < n_runs;
    if (I.valid) {
      I.r_tn = I.R % tiles_n;
      I.r_hm0 = (I.R / tiles_n) * run;
      I.T = xg_tile(P, I.r_hm0, I.r_tn);
    }
  };
  auto it_next = [&](It &I, bool with_groups = false) {
    if (!I.valid) return;
    if (++I.c < n_chunks) {
      if (++I.iq == (unsigned)cpg) {
        I.iq = 0;
        ++I.ig;
        if (with_groups) { // (the producers' issue iterator only: a mixed-radix decode, once per group)
          unsigned o0, o1;
          artn_xg_decode(P.k, P.k.n0, P.k.n_lab - P.k.n0, I.ig, o0, o1);
          I.gA = __builtin_amdgcn_readfirstlane(o0);
          I.gB = __builtin_amdgcn_readfirstlane(o1);
        }
      }
      return;
    }
    I.c = 0; I.iq = 0; I.ig = 0; I.gA = 0; I.gB = 0;
    ++I.seq;
    if (I.pos + 1 < run && I.r_hm0 + I.pos + 1 < total_hm) ++I.pos;
    else {
      I.R += G;
      I.pos = 0;
      if (I.R >= n_runs) { I.valid = false; return; }
      I.r_tn = I.R % tiles_n;
      I.r_hm0 = (I.R / tiles_n) * run;
    }
    const unsigned n0_before = I.T.n0;
    I.T = xg_tile(P, I.r_hm0 + I.pos, I.r_tn);
    if (I.T.n0 != n0_before) ++I.cseq;
  };

  // ---- per-tile tables, built by the producers (thread p < 128: row p; 128 <= p < 128 + TN: column p - 128)
  unsigned c_q1 = 0xffffffffu, c_o0 = 0, c_o1 = 0;
  unsigned s_first = 0xffffffffu, s_i0b = 0, s_i1b = 0, s_q1b = 0;
  auto build_side = [&](const ArtnXSide &S, unsigned first, int loc, unsigned t0, unsigned t1, unsigned dst) {
    const unsigned tot = (unsigned)S.total, L0 = (unsigned)S.L0, L1 = (unsigned)S.L1;
    unsigned posn = (unsigned)loc;
    if (first + posn >= tot) posn = tot - 1 - first;
    unsigned i0b, i1b, q1b;
    if (first == s_first + ARTN_XG_TM && L0 >= 32u) {
      i0b = s_i0b + ARTN_XG_TM; i1b = s_i1b; q1b = s_q1b;
      while (i0b >= L0) { i0b -= L0; ++i1b; }
      while (i1b >= L1) { i1b -= L1; ++q1b; }
    } else {
      const unsigned q0b = first / L0;
      i0b = first - q0b * L0;
      q1b = q0b / L1;
      i1b = q0b - q1b * L1;
    }
    s_first = first; s_i0b = i0b; s_i1b = i1b; s_q1b = q1b;
    unsigned i0 = i0b + posn, i1 = i1b, q1 = q1b;
    while (i0 >= L0) { i0 -= L0; ++i1; }
    while (i1 >= L1) { i1 -= L1; ++q1; }
    if (q1 != c_q1) {
      c_q1 = q1;
      artn_xg_decode(S, S.n0 + S.n1, S.n_lab - S.n0 - S.n1, q1, c_o0, c_o1);
    }
    const unsigned o0 = c_o0 + lds_read4(t0 + 4u * i0) + lds_read4(t1 + 4u * i1);
    const unsigned o1 = c_o1 + lds_read4(t0 + 1024u + 4u * i0) + lds_read4(t1 + 1024u + 4u * i1);
    lds_write4(dst + 4u * loc, o0);
    lds_write4(dst + 512u + 4u * loc, o1);
  };
  auto build_tile = [&](const It &I, bool cols) { // (producer threads only)
    if (ptid < TM) build_side(P.m, I.T.m0, ptid, T_MA0, T_MA1, TT + (I.seq & 3u) * 1024u);
    else if (cols && ptid - TM < TN) build_side(P.n, I.T.n0, ptid - TM, T_NB0, T_NB1, TT + 4096u + (I.cseq & 3u) * 1024u);
  };

  const char *Ac = reinterpret_cast<const char *>(A), *Bc = reinterpret_cast<const char *>(B);
  const int amode = P.amode, bmode = P.bmode;
  __syncthreads(); // level tables are in LDS

  if (producer) {
    // =============================== PRODUCERS ===============================
    It I2, I3; // the item whose loads are issued next; the item after it (its tile's tables are built one step earlier)
    it_init(I2);
    if (!I2.valid) return; // (never: the grid is at most n_runs)
    I3 = I2;
    v2f_t va[2][NA], vb[2][NBL];
    int kv[2] = {0, 0}; // valid contracted values of the chunk in each register set
    auto ld = [&](const char *base, unsigned off) {
#ifdef XG_ABLATE_MEM // (timing-only probe builds)
      v2f_t v;
      asm volatile("" : "=v"(v) : "s"(base), "v"(off));
      return v;
#else
      return *reinterpret_cast<const v2f_t *>(base + ((unsigned long)off << 3));
#endif
    };
    auto issue = [&](const It &I, v2f_t (&xa)[NA], v2f_t (&xb)[NBL], int &kvalid) {
      const unsigned hA = I.T.hA + I.gA, hB = I.T.hB + I.gB;
      const unsigned kbase = I.iq * KC;
      kvalid = (int)(K0 - kbase < (unsigned)KC ? K0 - kbase : (unsigned)KC);
      const unsigned ttr = TT + (I.seq & 3u) * 1024u, ttc = TT + 4096u + (I.cseq & 3u) * 1024u;
      const unsigned t = (unsigned)ptid;
      if (amode == 0) {
        const unsigned base = hA + lds_read4(ttr + 4u * (t & 127u));
        const unsigned ka = T_KA + 4u * (kbase + (t >> 7));
#pragma unroll
        for (int u = 0; u < NA; ++u) xa[u] = ld(Ac, base + lds_read4(ka + 8u * u));
      } else {
        const unsigned base = hA + lds_read4(T_KA + 4u * (kbase + (t & (KC - 1u))));
        const unsigned ra = ttr + 4u * (t >> KCL);
#pragma unroll
        for (int u = 0; u < NA; ++u) xa[u] = ld(Ac, base + lds_read4(ra + 4u * RSTEP * u));
      }
      if (bmode == 0) {
        const unsigned kb = T_KB + 4u * (kbase + (t >> 5)), cb = ttc + 4u * (t & 31u);
        const unsigned k0 = hB + lds_read4(kb), k1 = hB + lds_read4(kb + 32u);
        unsigned cv[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) cv[b] = lds_read4(cb + 128u * b);
#pragma unroll
        for (int u = 0; u < NBL; ++u) xb[u] = ld(Bc, (u / NB ? k1 : k0) + cv[u % NB]);
      } else {
        const unsigned base = hB + lds_read4(T_KB + 4u * (kbase + (t & (KC - 1u))));
        const unsigned cb = ttc + 4u * (t >> KCL);
#pragma unroll
        for (int u = 0; u < NBL; ++u) xb[u] = ld(Bc, base + lds_read4(cb + 4u * RSTEP * u));
      }
    };
    auto fill = [&](unsigned buf, const v2f_t (&xa)[NA], const v2f_t (&xb)[NBL], int kvalid) {
      const unsigned t = (unsigned)ptid;
      const bool part = kvalid < KC;
      if (amode == 0) {
        const unsigned d = buf + ((t >> 7) * PA + (t & 127u)) * 8u;
        const int kh = (int)(t >> 7);
#pragma unroll
        for (int u = 0; u < NA; ++u) {
          v2f_t v = xa[u];
          if (part && kh + 2 * u >= kvalid) v = v2f_t{0.f, 0.f};
          lds_write8(d + (unsigned)u * (2u * PA * 8u), v);
        }
      } else {
        const unsigned d = buf + ((t & (KC - 1u)) * PA + (t >> KCL)) * 8u;
        const bool z = part && (int)(t & (KC - 1u)) >= kvalid;
#pragma unroll
        for (int u = 0; u < NA; ++u) lds_write8(d + 8u * RSTEP * u, z ? v2f_t{0.f, 0.f} : xa[u]);
      }
      if (bmode == 0) {
        const unsigned d = buf + A_BYTES + ((t >> 5) * PB + (t & 31u)) * 8u;
        const int kh = (int)(t >> 5);
#pragma unroll
        for (int u = 0; u < NBL; ++u) {
          v2f_t v = xb[u];
          if (part && kh + 8 * (u / NB) >= kvalid) v = v2f_t{0.f, 0.f};
          lds_write8(d + (unsigned)(u / NB) * (8u * PB * 8u) + (unsigned)(u % NB) * 256u, v);
        }
      } else {
        const unsigned d = buf + A_BYTES + ((t & (KC - 1u)) * PB + (t >> KCL)) * 8u;
        const bool z = part && (int)(t & (KC - 1u)) >= kvalid;
#pragma unroll
        for (int u = 0; u < NBL; ++u) lds_write8(d + 8u * RSTEP * u, z ? v2f_t{0.f, 0.f} : xb[u]);
      }
    };
    // the tables of the tiles of items 0, 1, 2 (an item's tables are built one step before its loads are issued)
    build_tile(I3, true);
    unsigned built_seq = I3.seq, built_cseq = I3.cseq;
    auto build_if_new = [&](const It &I) {
      if (I.valid && I.seq != built_seq) {
        build_tile(I, I.cseq != built_cseq);
        built_seq = I.seq;
        built_cseq = I.cseq;
      }
    };
    it_next(I3); build_if_new(I3); // item 1
    it_next(I3); build_if_new(I3); // item 2
    __syncthreads();               // (P1) tile tables of items 0..2
    issue(I2, va[0], vb[0], kv[0]); // item 0 -> set 0
    it_next(I2, true);
    const bool have1 = I2.valid;
    if (have1) issue(I2, va[1], vb[1], kv[1]); // item 1 -> set 1
    it_next(I2, true);              // I2: item 2 (I3 too: step i builds the tables of item i + 3's tile, never further ahead --
                                    //  four table sets hold the tiles of items i .. i + 3)
    fill(0u, va[0], vb[0], kv[0]);  // item 0 -> stage 0 (waits for its loads)
    __syncthreads();                // (P2) item 0 is in LDS
    // step i: loads of item i + 2 -> register set i & 1 FIRST, then stage (i + 1) & 1 <- item i + 1 (set (i + 1) & 1, loaded a
    // whole step ago), then the tables of item i + 3's tile.  The loads are UNCONDITIONAL (past the end of the sequence the
    // last item is loaded again and never used): only then does the compiler count them -- s_waitcnt vmcnt(NA + NBL) in
    // front of the LDS writes instead of vmcnt(0), which would wait for the loads just issued (a memory latency per step:
    // measured, the first version of this loop).
    bool more = have1; // item i + 1 exists
    It Iq = I2;        // what is issued: I2 while it is valid, else the last valid item again
    int kdummy = 0;
    for (;;) {
      // ---- even step: item i + 1 sits in set 1, item i + 2 goes to set 0
      const bool m2 = I2.valid; // item i + 2 exists
      if (m2) Iq = I2;
      issue(Iq, va[0], vb[0], m2 ? kv[0] : kdummy);
      if (more) fill(STAGE, va[1], vb[1], kv[1]);
      it_next(I2, true);
      it_next(I3);
      build_if_new(I3);
      __syncthreads();
      if (!more) break;           // (item i was the last: the consumers leave after this barrier too)
      // ---- odd step (i + 1): item i + 2 sits in set 0, item i + 3 goes to set 1
      more = I2.valid;            // item i + 3 exists
      if (more) Iq = I2;
      issue(Iq, va[1], vb[1], more ? kv[1] : kdummy);
      if (m2) fill(0u, va[0], vb[0], kv[0]);
      it_next(I2, true);
      it_next(I3);
      build_if_new(I3);
      __syncthreads();
      if (!m2) break;
    }
    return;
  }

  // =============================== CONSUMERS ===============================
  It Ic;
  it_init(Ic);
  if (!Ic.valid) return;
  __syncthreads(); // (P1)
  __syncthreads(); // (P2)
  const unsigned lane_x = (unsigned)(h * PA + 32 * wave + j) * 8u;
  const unsigned lane_w = A_BYTES + (unsigned)(h * PB + j) * 8u;
  const int flush_chunks = P.flush_chunks;
  f32x16 acc[NB * 3];
#pragma unroll
  for (int b = 0; b < NB * 3; ++b)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;
  int since_flush = 0;
  bool flushed_before = false;
  unsigned cur = 0;
  for (;;) {
    const bool last = Ic.c + 1 == n_chunks;
    const unsigned kbase = Ic.iq * KC;
    const int kvalid = (int)(K0 - kbase < (unsigned)KC ? K0 - kbase : (unsigned)KC);
    {
      auto mac = [&](const v2f_t &x, const v2f_t (&w)[NB]) {
#ifdef XG_ABLATE_MFMA
#pragma unroll
        for (int b = 0; b < NB; ++b) asm volatile("" ::"v"(x), "v"(w[b]));
        return;
#endif
        const float xs = x.x + x.y;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const float ws = w[b].x + w[b].y;
          if constexpr (TRANS) {
            acc[3 * b] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.x, w[b].x, acc[3 * b], 0, 0, 0);
            acc[3 * b + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.y, w[b].y, acc[3 * b + 1], 0, 0, 0);
            acc[3 * b + 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(xs, ws, acc[3 * b + 2], 0, 0, 0);
          } else {
            acc[3 * b] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[b].x, x.x, acc[3 * b], 0, 0, 0);
            acc[3 * b + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[b].y, x.y, acc[3 * b + 1], 0, 0, 0);
            acc[3 * b + 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(ws, xs, acc[3 * b + 2], 0, 0, 0);
          }
        }
      };
      auto load_ops = [&](unsigned xo, unsigned wo, v2f_t &x, v2f_t (&w)[NB]) {
        x = lds_read8(xo);
#pragma unroll
        for (int b = 0; b < NB; ++b) w[b] = lds_read8(wo + (unsigned)b * 256u);
      };
      constexpr unsigned XS = 2u * PA * 8u, WS = 2u * PB * 8u;
      v2f_t X0, X1, W0[NB], W1[NB];
      unsigned xo = cur * STAGE + lane_x, wo = cur * STAGE + lane_w;
      load_ops(xo, wo, X0, W0);
      const int trips = (kvalid + 3) >> 2;
#pragma unroll 1
      for (int q = 0; q < trips; ++q) {
        load_ops(xo + XS, wo + WS, X1, W1);
        __builtin_amdgcn_sched_barrier(0);
        mac(X0, W0);
        xo += 2u * XS;
        wo += 2u * WS;
        load_ops(xo, wo, X0, W0); // (after the last trip: a pair of the next region, read and dropped)
        __builtin_amdgcn_sched_barrier(0);
        mac(X1, W1);
      }
    }
    ++since_flush;
    const bool flush = last || (flush_chunks > 0 && since_flush == flush_chunks);
    if (flush) {
      const XgTile &T = Ic.T;
      const unsigned ttr = TT + (Ic.seq & 3u) * 1024u + 512u, ttc = TT + 4096u + (Ic.cseq & 3u) * 1024u + 512u; // rowC, colC
      const unsigned Mtot = (unsigned)P.m.total, Ntot = (unsigned)P.n.total;
      const unsigned rows_left = Mtot - T.m0, cols_left = Ntot - T.n0;
      const bool full = rows_left >= (unsigned)TM && cols_left >= (unsigned)TN;
      char *Cc = reinterpret_cast<char *>(C) + ((unsigned long)T.hC << 3);
      unsigned jj = (unsigned)j, hh4 = 4u * (unsigned)h;
      OPAQUE_V(jj);
      OPAQUE_V(hh4);
      const unsigned lane_left = TRANS ? cols_left : rows_left, reg_left = TRANS ? rows_left : cols_left;
      const unsigned lane_tab = TRANS ? ttc : ttr, reg_tab = TRANS ? ttr : ttc;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const unsigned lane_loc = TRANS ? 32u * b + jj : 32u * wave + jj;
        const bool lane_ok = full || lane_loc < lane_left;
        char *lp = Cc + ((unsigned long)lds_read4(lane_tab + 4u * lane_loc) << 3);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const unsigned reg0 = (TRANS ? 32u * wave : 32u * b) + 8u * g;
          if (!full && reg0 >= reg_left) continue;
          const u32x4_t tab = *(__attribute__((address_space(3))) u32x4_t *)(unsigned long)(reg_tab + 4u * (reg0 + hh4));
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int r = 4 * g + q;
            const float t1 = acc[3 * b][r], t2 = acc[3 * b + 1][r], t3 = acc[3 * b + 2][r];
            v2f_t val = {t1 - t2, t3 - t1 - t2};
            if (lane_ok && (full || reg0 + hh4 + (unsigned)q < reg_left)) {
              v2f_t *dst = reinterpret_cast<v2f_t *>(lp + ((unsigned long)tab[q] << 3));
              if (flushed_before) val += __builtin_nontemporal_load(dst);
              *dst = val;
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      flushed_before = !last;
      since_flush = 0;
#pragma unroll
      for (int b = 0; b < NB * 3; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;
    }
    it_next(Ic);
    __syncthreads();
    cur ^= 1u;
    if (!Ic.valid) break;
  }
}
