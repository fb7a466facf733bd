// artn_xgemm_kernel.h -- the EXTENT-based two-operand LDS GEMM of libartn_hip.so (included by artn_kernels.hip).
//
// One pairwise contraction whose labels have ANY extents (bond dimension 3, 5, 6 ...: the reference's torch.einsum at
// artensor/contraction.py:70 contracts whatever bond_dims the network has, tensor_network.py:4-30), as a GEMM over flattened
// mixed-radix indices (artn_xgemm_plan.h):
//
//   workgroup  = a C tile of 128 consecutive values of m x 32 NB consecutive values of n; wave w owns rows 32 w .. 32 w + 31
//                and all NB blocks of 32 columns: 3 NB accumulators of v_mfma_f32_32x32x2_f32 (3M arithmetic: T1 = A_re B_re,
//                T2 = A_im B_im, T3 = (A_re + A_im)(B_re + B_im); C_re = T1 - T2, C_im = T3 - T1 - T2 -- as artn_k_gemm<.., M3>);
//   chunk      = 16 contracted values: images [16][130] of the first operand and [16][32 NB + 2] of the second, 8-byte
//                elements, copied global -> registers -> LDS one element per lane and load (odd extents leave nothing
//                16-byte aligned); the copy lanes of an operand run along its free index or along k, whichever its fastest
//                label belongs to (ArtnXGemmPlan::amode / bmode).  Element offsets are off(row) + off(k): per-tile row tables and
//                per-level k tables in LDS.  Double buffered: the loads of chunk c + 1 fly while chunk c is multiplied;
//   k loop     = groups of k.L0 values (the innermost contracted labels, one level table) x the remaining labels decoded per
//                group; the last chunk of a group is zero-padded to an even count; every 4096 values the partial sum goes
//                to C (read-add-write) and the registers restart from zero;
//   epilogue   = straight from the accumulators: lane j of a store instruction is row m (TRANS = false) or column n (TRANS = true:
//                the MFMA roles of the two operands are swapped), whichever C's fastest label belongs to; 8 bytes per lane.
// Lane roles of one MFMA (contracted pair s of a chunk, kk = 2 s + h, h = lane >> 5, j = lane & 31):
//   TRANS = false:  srcA = W (B image [kk][32 b + j])  rows i <-> n;  srcB = X (A image [kk][32 w + j])  columns <-> m
//   TRANS = true:   srcA = X                           rows i <-> m;  srcB = W                           columns <-> n
//   accumulator register r of lane (j, h): row i = (r & 3) + 8 (r >> 2) + 4 h, column j.

struct XgTile {
  unsigned m0, n0;       // first row / column of the tile
  unsigned hA, hB, hC;   // element offsets of the tile's batch value
};

__device__ __forceinline__ XgTile xg_tile(const ArtnXGemmPlan &P, unsigned hm, unsigned tn) { // hm = batch value x tiles_m + tile of m
  XgTile T;
  const unsigned tsm = (unsigned)P.tiles_m;
  unsigned hh = hm / tsm;
  const unsigned tm = hm - hh * tsm;
  T.m0 = tm * ARTN_XG_TM;
  T.n0 = tn * 32u * (unsigned)P.nb;
  unsigned a = 0, b = 0, c = 0;
  for (int i = 0; i < P.n_h; ++i) {
    const unsigned e = (unsigned)P.h_ext[i], q = hh / e, d = hh - q * e;
    a += d * (unsigned)P.h_sA[i];
    b += d * (unsigned)P.h_sB[i];
    c += d * (unsigned)P.h_sC[i];
    hh = q;
  }
  T.hA = __builtin_amdgcn_readfirstlane(a);
  T.hB = __builtin_amdgcn_readfirstlane(b);
  T.hC = __builtin_amdgcn_readfirstlane(c);
  T.m0 = __builtin_amdgcn_readfirstlane(T.m0);
  T.n0 = __builtin_amdgcn_readfirstlane(T.n0);
  return T;
}

__device__ __forceinline__ unsigned lds_read4(unsigned a) { return *(__attribute__((address_space(3))) unsigned *)(unsigned long)a; }

template <int NB, bool TRANS>
__global__ __launch_bounds__(ARTN_WG_THREADS, 2) void artn_k_xgemm(const float2 *__restrict__ A, const float2 *__restrict__ B,
                                                                  float2 *__restrict__ C, const ArtnXGemmPlan P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if ((unsigned)(unsigned long)(lds_byte_t *)smem != 0) __builtin_trap(); // LDS is addressed by raw byte offsets
  constexpr int TM = ARTN_XG_TM, TN = 32 * NB, KC = ARTN_XG_KC;
  constexpr int PA = TM + 2, PB = TN + 2;
  constexpr unsigned A_BYTES = KC * PA * 8, B_BYTES = KC * PB * 8, STAGE = A_BYTES + B_BYTES;
  constexpr unsigned LEV = 2 * STAGE;                       // level tables: 10 x 256 x 4 bytes
  constexpr unsigned T_MA0 = LEV, T_MC0 = LEV + 1024, T_MA1 = LEV + 2048, T_MC1 = LEV + 3072, T_NB0 = LEV + 4096, T_NC0 = LEV + 5120,
                     T_NB1 = LEV + 6144, T_NC1 = LEV + 7168, T_KA = LEV + 8192, T_KB = LEV + 9216;
  constexpr unsigned TT = LEV + 10240;                      // tile tables: 2 sets x (rowA, rowC, colB, colC) x 128 x 4 bytes
  constexpr int NA = TM * KC / ARTN_WG_THREADS, NBL = TN * KC / ARTN_WG_THREADS; // loads per thread and chunk: 8 and 2 NB
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;

  // ---- level tables (once per workgroup; one pass per flattened index, so the plan is read with scalar loads)
  auto level_tables = [&](const ArtnXSide &S, unsigned t0, unsigned t1, bool two) {
    if (tid < S.L0) {
      unsigned o0, o1;
      artn_xg_decode(S, 0, S.n0, (unsigned)tid, o0, o1);
      lds_write4(t0 + 4u * tid, o0);
      lds_write4(t0 + 1024u + 4u * tid, o1);
    }
    if (two && tid < S.L1) {
      unsigned o0, o1;
      artn_xg_decode(S, S.n0, S.n1, (unsigned)tid, o0, o1);
      lds_write4(t1 + 4u * tid, o0);
      lds_write4(t1 + 1024u + 4u * tid, o1);
    }
  };
  level_tables(P.m, T_MA0, T_MA1, true);
  level_tables(P.n, T_NB0, T_NB1, true);
  level_tables(P.k, T_KA, 0u, false);
  // ---- per-tile tables: element offsets of the tile's rows in A and C, of its columns in B and C (set `s`)
  // Rows (columns) of a tile are consecutive values of the flattened index: index = ((q1 L1) + i1) L0 + i0.  The tile's first
  // value is split by two uniform divisions; a row adds its position to i0 and carries.  What lies above the two table
  // levels (q1) is decoded label by label -- but a workgroup walks CONSECUTIVE tiles, so q1 changes once in L0 L1 / 128
  // tiles and every thread keeps the decode of the q1 it met last.
  unsigned c_q1 = 0xffffffffu, c_o0 = 0, c_o1 = 0;
  auto build_side = [&](const ArtnXSide &S, unsigned first, int loc, unsigned t0, unsigned t1, unsigned dst) {
    const unsigned tot = (unsigned)S.total, L0 = (unsigned)S.L0, L1 = (unsigned)S.L1;
    unsigned pos = (unsigned)loc;
    if (first + pos >= tot) pos = tot - 1 - first; // rows / columns past the end read valid memory and are never stored
    const unsigned q0b = first / L0, i0b = first - q0b * L0, q1b = q0b / L1, i1b = q0b - q1b * L1;
    unsigned i0 = i0b + pos, i1 = i1b, q1 = q1b;
    while (i0 >= L0) { i0 -= L0; ++i1; }
    while (i1 >= L1) { i1 -= L1; ++q1; }
    if (q1 != c_q1) {
      c_q1 = q1;
      artn_xg_decode(S, S.n0 + S.n1, S.n_lab - S.n0 - S.n1, q1, c_o0, c_o1);
    }
    const unsigned o0 = c_o0 + lds_read4(t0 + 4u * i0) + lds_read4(t1 + 4u * i1);
    const unsigned o1 = c_o1 + lds_read4(t0 + 1024u + 4u * i0) + lds_read4(t1 + 1024u + 4u * i1);
    lds_write4(dst + 4u * loc, o0);        // rowA / colB
    lds_write4(dst + 512u + 4u * loc, o1); // rowC / colC
  };
  auto build_tile = [&](const XgTile &T, unsigned set) { // waves 0, 1: the rows; waves 2, 3: the columns
    if (wave < 2) build_side(P.m, T.m0, tid, T_MA0, T_MA1, TT + set * 2048u);
    else if (tid - TM < TN) build_side(P.n, T.n0, tid - TM, T_NB0, T_NB1, TT + set * 2048u + 1024u);
  };

  // ---- copy slots: element (row, kk) of slot u of this thread
  const int amode = P.amode, bmode = P.bmode;
  // (the thread id is passed in, opaque per call: otherwise every slot's row, kk, LDS address and table address -- 100+
  //  registers -- is hoisted out of the chunk loop and the accumulators spill; shifts and masks instead of branches)
  const int a_rs = amode ? 4 : 0, a_rm = amode ? 15 : TM - 1, a_rstep = amode ? 16 : 0;
  const int a_ks = amode ? 0 : 7, a_km = amode ? 15 : 1, a_kstep = amode ? 0 : 2;
  auto a_slot = [&](int t, int u, int &row, int &kk) {
    row = ((t >> a_rs) & a_rm) + a_rstep * u;
    kk = ((t >> a_ks) & a_km) + a_kstep * u;
  };
  auto b_slot = [&](int t, int u, int &col, int &kk) {
    const int e = t + ARTN_WG_THREADS * u, k0 = e / TN, c0 = e - k0 * TN;
    const int k1 = t & 15, c1 = (t >> 4) + 16 * u;
    kk = bmode ? k1 : k0;
    col = bmode ? c1 : c0;
  };

  const unsigned K0 = (unsigned)P.k.L0;
  const int cpg = P.cpg;
  const long n_chunks = (long)P.k_groups * cpg;
  // state of the chunk whose loads were issued last: position inside the group, group offsets
  int iq = 0;          // chunk inside the group
  unsigned ig = 0;     // group index
  unsigned gA = 0, gB = 0;
  int kvalid_next = 0; // valid contracted values of the chunk in flight
  v2f_t va[NA], vb[NBL];
  const char *Ac = reinterpret_cast<const char *>(A), *Bc = reinterpret_cast<const char *>(B);

  auto issue = [&](const XgTile &T, unsigned set, bool first_of_tile) {
    if (first_of_tile) { iq = 0; ig = 0; gA = 0; gB = 0; }
    else if (++iq == cpg) {
      iq = 0;
      ++ig;
      unsigned o0, o1;
      artn_xg_decode(P.k, P.k.n0, P.k.n_lab - P.k.n0, ig, o0, o1);
      gA = __builtin_amdgcn_readfirstlane(o0);
      gB = __builtin_amdgcn_readfirstlane(o1);
    }
    const unsigned kbase = (unsigned)iq * KC;
    kvalid_next = (int)(K0 - kbase < (unsigned)KC ? K0 - kbase : (unsigned)KC);
    const unsigned tt = TT + set * 2048u;
    int t = tid;
    OPAQUE_V(t);
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      int row, kk;
      a_slot(t, u, row, kk);
      unsigned kc = kbase + (unsigned)kk;
      if (kc >= K0) kc = K0 - 1;
      const unsigned off = T.hA + gA + lds_read4(tt + 4u * row) + lds_read4(T_KA + 4u * kc);
      va[u] = *reinterpret_cast<const v2f_t *>(Ac + ((unsigned long)off << 3));
    }
#pragma unroll
    for (int u = 0; u < NBL; ++u) {
      int col, kk;
      b_slot(t, u, col, kk);
      unsigned kc = kbase + (unsigned)kk;
      if (kc >= K0) kc = K0 - 1;
      const unsigned off = T.hB + gB + lds_read4(tt + 1024u + 4u * col) + lds_read4(T_KB + 4u * kc);
      vb[u] = *reinterpret_cast<const v2f_t *>(Bc + ((unsigned long)off << 3));
    }
  };
  auto fill = [&](unsigned buf) { // registers -> LDS images; contracted values past the end of a group are zeros
    int t = tid;
    OPAQUE_V(t);
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      int row, kk;
      a_slot(t, u, row, kk);
      v2f_t v = va[u];
      if (kk >= kvalid_next) v = v2f_t{0.f, 0.f};
      lds_write8(buf + (unsigned)(kk * PA + row) * 8u, v);
    }
#pragma unroll
    for (int u = 0; u < NBL; ++u) {
      int col, kk;
      b_slot(t, u, col, kk);
      v2f_t v = vb[u];
      if (kk >= kvalid_next) v = v2f_t{0.f, 0.f};
      lds_write8(buf + A_BYTES + (unsigned)(kk * PB + col) * 8u, v);
    }
  };

  // Tile order.  A RUN is `run` consecutive row tiles (same batch value or the next) of ONE column tile: the workgroup that
  // walks it keeps its column tables and, most of the time, the decode cache above.  Runs are numbered column tile fastest
  // and dealt round-robin: the workgroups that run side by side hold the tiles_n column tiles of the same rows (their
  // reads of those rows meet in L2 instead of following each other through HBM) and neighbouring rows.  Measured on a
  const unsigned G = gridDim.x, tiles_n = (unsigned)P.tiles_n;
  const unsigned total_hm = (unsigned)(P.n_tiles / P.tiles_n);
  unsigned run = (unsigned)(P.n_tiles / ((long)G * 4));
  run = run < 1 ? 1 : (run > 16 ? 16 : run);
  const unsigned n_super = (total_hm + run - 1) / run, n_runs = n_super * tiles_n; // (< 2^31: n_tiles is)
  __syncthreads(); // level tables are in LDS
  // (consecutive runs on the SAME XCD -- workgroup b runs on XCD b mod 8 -- so that those shared rows meet in ONE L2)
  const unsigned wg = (G & 7) == 0 ? (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3) : blockIdx.x;
  unsigned R = wg, pos = 0; // current run, position inside it
  if (R >= n_runs) return;
  unsigned r_tn = R % tiles_n, r_hm0 = (R / tiles_n) * run;
  XgTile T = xg_tile(P, r_hm0, r_tn), Tn = T;
  unsigned set = 0;
  build_tile(T, 0u);
  __syncthreads();
  issue(T, 0u, true);
  int kvalid = kvalid_next;
  fill(0u);
  __syncthreads();
  unsigned cur = 0;
  const unsigned lane_x = (unsigned)(h * PA + 32 * wave + j) * 8u;
  const unsigned lane_w = A_BYTES + (unsigned)(h * PB + j) * 8u;
  const int flush_chunks = P.flush_chunks;

  for (;;) {
    // the successor of this tile in the workgroup's sequence
    bool more_tiles = true;
    if (pos + 1 < run && r_hm0 + pos + 1 < total_hm) ++pos;
    else {
      R += G;
      pos = 0;
      more_tiles = R < n_runs;
      if (more_tiles) { r_tn = R % tiles_n; r_hm0 = (R / tiles_n) * run; }
    }
    if (more_tiles) Tn = xg_tile(P, r_hm0 + pos, r_tn);
    f32x16 acc[NB * 3];
#pragma unroll
    for (int b = 0; b < NB * 3; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;
    int since_flush = 0;
    bool flushed_before = false;
    for (long c = 0; c < n_chunks; ++c) {
      const bool last = c + 1 == n_chunks;
      bool have_next = true;
      if (!last) {
        issue(T, set, false);
      } else {
        have_next = more_tiles;
        if (have_next) { // (the other table set was last read in the previous tile's epilogue, a barrier ago)
          build_tile(Tn, set ^ 1u);
          __syncthreads();
          issue(Tn, set ^ 1u, true);
        }
      }
      // ---- multiply chunk `cur`: pairs of contracted values, two per trip (a group's last chunk is zero-padded in LDS, so a
      //      trip may run one pair past ceil(kvalid / 2); the operands of pair s + 1 are read under the MFMAs of pair s)
      {
        const unsigned xa = cur * STAGE + lane_x, wa = cur * STAGE + lane_w;
        auto mac = [&](const v2f_t &x, const v2f_t (&w)[NB]) {
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
        constexpr unsigned XS = 2u * PA * 8u, WS = 2u * PB * 8u; // bytes between two pairs
        const int trips = (kvalid + 3) >> 2;
        v2f_t X0, X1, W0[NB], W1[NB];
        unsigned xo = xa, wo = wa;
        load_ops(xo, wo, X0, W0);
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
        // ---- epilogue: accumulators -> C (a later partial sum of the tile is added to what the earlier ones left)
        const unsigned tt = TT + set * 2048u;
        const unsigned Mtot = (unsigned)P.m.total, Ntot = (unsigned)P.n.total;
        char *Cc = reinterpret_cast<char *>(C);
        unsigned jj = (unsigned)j, hh4 = 4u * (unsigned)h; // (opaque: no store address is computed before its turn)
        OPAQUE_V(jj);
        OPAQUE_V(hh4);
        if constexpr (!TRANS) {
          const unsigned m_loc = 32u * wave + jj;
          const bool m_ok = T.m0 + m_loc < Mtot;
          const unsigned rowc = T.hC + lds_read4(tt + 512u + 4u * m_loc);
#pragma unroll
          for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const unsigned n_loc = 32u * b + (unsigned)((r & 3) + 8 * (r >> 2)) + hh4;
              const unsigned off = rowc + lds_read4(tt + 1536u + 4u * n_loc);
              const float t1 = acc[3 * b][r], t2 = acc[3 * b + 1][r], t3 = acc[3 * b + 2][r];
              v2f_t val = {t1 - t2, t3 - t1 - t2};
              if (m_ok && T.n0 + n_loc < Ntot) {
                v2f_t *dst = reinterpret_cast<v2f_t *>(Cc + ((unsigned long)off << 3));
                if (flushed_before) val += __builtin_nontemporal_load(dst); // (written by this lane at the previous flush: read past the L1)
                *dst = val;
              }
              if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0); // (no more than four addresses alive at a time)
            }
        } else {
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            const unsigned n_loc = 32u * b + jj;
            const bool n_ok = T.n0 + n_loc < Ntot;
            const unsigned colc = T.hC + lds_read4(tt + 1536u + 4u * n_loc);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const unsigned m_loc = 32u * wave + (unsigned)((r & 3) + 8 * (r >> 2)) + hh4;
              const unsigned off = colc + lds_read4(tt + 512u + 4u * m_loc);
              const float t1 = acc[3 * b][r], t2 = acc[3 * b + 1][r], t3 = acc[3 * b + 2][r];
              v2f_t val = {t1 - t2, t3 - t1 - t2};
              if (n_ok && T.m0 + m_loc < Mtot) {
                v2f_t *dst = reinterpret_cast<v2f_t *>(Cc + ((unsigned long)off << 3));
                if (flushed_before) val += __builtin_nontemporal_load(dst); // (written by this lane at the previous flush: read past the L1)
                *dst = val;
              }
              if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
        flushed_before = true;
        since_flush = 0;
        if (!last) {
#pragma unroll
          for (int b = 0; b < NB * 3; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;
        }
      }
      if (have_next) fill((cur ^ 1u) * STAGE);
      kvalid = kvalid_next;
      __syncthreads();
      cur ^= 1u;
    }
    if (!more_tiles) break;
    T = Tn;
    set ^= 1u;
  }
}
