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
  T.n0 = (unsigned)P.col0 + tn * 32u * (unsigned)P.nb; // (col0: the tail launch of a step whose last column tile is narrower)
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

#ifdef XG_STAMPS // (timing probe builds only: phase marks of workgroup 0, tiles 8 and 9)
__device__ unsigned long long xg_stamp_buf[64];
#define XG_MARK(k)                                                                             \
  if (blockIdx.x == 0 && tid == 0 && tile_count >= 8 && tile_count < 10) {                      \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    xg_stamp_buf[(tile_count - 8) * 16 + (k)] = __builtin_amdgcn_s_memtime();                  \
    __builtin_amdgcn_sched_barrier(0);                                                         \
  }
#else
#define XG_MARK(k)
#endif

// KC: contracted values per chunk.  16; 8 for steps of a few contracted values and at most 32 columns (KC = 8, NB = 1: 36 KiB of
// LDS and under 128 registers, so FOUR workgroups per CU: those steps are chains of LDS and memory latencies per tile --
// 10 000 cycles per tile of a 9 x 9 step for 1 150 cycles of MFMAs -- and only more waves hide them).
template <int NB, bool TRANS, int KC = ARTN_XG_KC>
__global__ __launch_bounds__(ARTN_WG_THREADS, (KC == 8 ? 4 : 2)) void artn_k_xgemm(const float2 *__restrict__ A, const float2 *__restrict__ B,
                                                                  float2 *__restrict__ C, const ArtnXGemmPlan P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if ((unsigned)(unsigned long)(lds_byte_t *)smem != 0) __builtin_trap(); // LDS is addressed by raw byte offsets
  constexpr int TM = ARTN_XG_TM, TN = 32 * NB;
  static_assert(KC == 16 || KC == 8, "chunks of 16 or 8 contracted values");
  constexpr int KCL = KC == 16 ? 4 : 3, RSTEP = ARTN_WG_THREADS / KC; // log2 KC; rows between two slots of a thread in mode 1
  constexpr int PA = TM + 2, PB = TN + 2;
  constexpr unsigned A_BYTES = KC * PA * 8, B_BYTES = KC * PB * 8, STAGE = A_BYTES + B_BYTES;
  constexpr unsigned LEV = 2 * STAGE;                       // level tables: 8 x 256 x 4 bytes (m, n) + 2 x 272 x 4 (k)
  constexpr unsigned T_MA0 = LEV, T_MC0 = LEV + 1024, T_MA1 = LEV + 2048, T_MC1 = LEV + 3072, T_NB0 = LEV + 4096, T_NC0 = LEV + 5120,
                     T_NB1 = LEV + 6144, T_NC1 = LEV + 7168, T_KA = LEV + 8192, T_KB = T_KA + ARTN_XG_KTAB * 4;
  // (the k tables carry 16 more entries, copies of the last one: a chunk reads kbase .. kbase + 15 without clamping)
  constexpr unsigned TT = T_KB + ARTN_XG_KTAB * 4;          // tile tables: 2 sets x (rowA, rowC, colB, colC) x 128 x 4 bytes
  constexpr int NA = TM * KC / ARTN_WG_THREADS, NBL = TN * KC / ARTN_WG_THREADS; // loads per thread and chunk: 8 and 2 NB (KC = 16)
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
      lds_write4((two ? t0 + 1024u : T_KB) + 4u * tid, o1);
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
  if (tid < ARTN_XG_KC) { // (padding of the k tables)
    unsigned o0, o1;
    artn_xg_decode(P.k, 0, P.k.n0, (unsigned)P.k.L0 - 1u, o0, o1);
    lds_write4(T_KA + 4u * (P.k.L0 + tid), o0);
    lds_write4(T_KB + 4u * (P.k.L0 + tid), o1);
  }
  // ---- per-tile tables: element offsets of the tile's rows in A and C, of its columns in B and C (set `s`)
  // Rows (columns) of a tile are consecutive values of the flattened index: index = ((q1 L1) + i1) L0 + i0.  The tile's first
  // value is split by two uniform divisions; a row adds its position to i0 and carries.  What lies above the two table
  // levels (q1) is decoded label by label -- but a workgroup walks CONSECUTIVE tiles, so q1 changes once in L0 L1 / 128
  // tiles and every thread keeps the decode of the q1 it met last.
  // The two divisions themselves are skipped when the tile follows the one this thread built last (first + 128: a run of
  // row tiles), and the column tables are rebuilt only when the column tile changes (once per run).
  unsigned c_q1 = 0xffffffffu, c_o0 = 0, c_o1 = 0;
  unsigned s_first = 0xffffffffu, s_i0b = 0, s_i1b = 0, s_q1b = 0;
  auto build_side = [&](const ArtnXSide &S, unsigned first, int loc, unsigned t0, unsigned t1, unsigned dst) {
    const unsigned tot = (unsigned)S.total, L0 = (unsigned)S.L0, L1 = (unsigned)S.L1;
    unsigned pos = (unsigned)loc;
    if (first + pos >= tot) pos = tot - 1 - first; // rows / columns past the end read valid memory and are never stored
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
  // tile tables: rows (rowA, rowC) in set `rs` at TT + 1024 rs, columns (colB, colC) in set `cs` at TT + 2048 + 1024 cs
  auto build_tile = [&](const XgTile &T, unsigned rs, unsigned cs, bool cols) { // waves 0, 1: the rows; waves 2, 3: the columns
    if (wave < 2) build_side(P.m, T.m0, tid, T_MA0, T_MA1, TT + rs * 1024u);
    else if (cols && tid - TM < TN) build_side(P.n, T.n0, tid - TM, T_NB0, T_NB1, TT + 2048u + cs * 1024u);
  };

  // ---- copy slots.  An operand's copy lanes run along its free index (mode 0) or along k (mode 1):
  //   A, mode 0: row = t & 127,                 kk = (t >> 7) + 2 u     A, mode 1: kk = t & 15, row = (t >> 4) + 16 u    (u < 8)
  //   B, mode 0: col = (t & 31) + 32 (u % NB),  kk = (t >> 5) + 8 (u / NB)   B, mode 1: kk = t & 15, col = (t >> 4) + 16 u  (u < 2 NB)
  //   (chunks of 16; with chunks of 8 a thread has half the slots and mode 1 reads kk = t & 7, row / col = (t >> 3) + 32 u)
  // so that per chunk a thread reads one or two table entries per load from LDS at compile-time offsets and its LDS
  // destinations differ by compile-time offsets too: the copy costs a handful of vector instructions per load (the first
  // version recomputed row, kk, clamps and addresses per slot: 7.5 VALU instructions per MFMA, MFMA busy 0.53).
  const int amode = P.amode, bmode = P.bmode;
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
  auto ld = [&](const char *base, unsigned off) {
#ifdef XG_ABLATE_MEM // (timing-only probe builds, tools/probes/xgemm_probe.hip: never in the library)
    v2f_t v;
    asm volatile("" : "=v"(v) : "s"(base), "v"(off));
    return v;
#else
    return *reinterpret_cast<const v2f_t *>(base + ((unsigned long)off << 3));
#endif
  };

  auto issue = [&](const XgTile &T, unsigned rs, unsigned cs, bool first_of_tile) {
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
    const unsigned ttr = TT + rs * 1024u, ttc = TT + 2048u + cs * 1024u;
    unsigned t = (unsigned)tid;
    OPAQUE_V(t); // (nothing derived from the thread id is hoisted out of the chunk loop: the accumulators need the registers)
    if (amode == 0) {
      const unsigned base = T.hA + gA + lds_read4(ttr + 4u * (t & 127u));
      const unsigned ka = T_KA + 4u * (kbase + (t >> 7));
#pragma unroll
      for (int u = 0; u < NA; ++u) va[u] = ld(Ac, base + lds_read4(ka + 8u * u));
    } else {
      const unsigned base = T.hA + gA + lds_read4(T_KA + 4u * (kbase + (t & (KC - 1u))));
      const unsigned ra = ttr + 4u * (t >> KCL);
#pragma unroll
      for (int u = 0; u < NA; ++u) va[u] = ld(Ac, base + lds_read4(ra + 4u * RSTEP * u));
    }
    if (bmode == 0) {
      const unsigned kb = T_KB + 4u * (kbase + (t >> 5)), cb = ttc + 4u * (t & 31u);
      const unsigned k0 = T.hB + gB + lds_read4(kb), k1 = KC == 16 ? T.hB + gB + lds_read4(kb + 32u) : 0u;
      unsigned cv[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) cv[b] = lds_read4(cb + 128u * b);
#pragma unroll
      for (int u = 0; u < NBL; ++u) vb[u] = ld(Bc, (u / NB ? k1 : k0) + cv[u % NB]);
    } else {
      const unsigned base = T.hB + gB + lds_read4(T_KB + 4u * (kbase + (t & (KC - 1u))));
      const unsigned cb = ttc + 4u * (t >> KCL);
#pragma unroll
      for (int u = 0; u < NBL; ++u) vb[u] = ld(Bc, base + lds_read4(cb + 4u * RSTEP * u));
    }
  };
  auto fill = [&](unsigned buf) { // registers -> LDS images; contracted values past the end of a group are zeros
    unsigned t = (unsigned)tid;
    OPAQUE_V(t);
    const bool part = kvalid_next < KC; // (uniform: the last chunk of a group)
    const int kv = kvalid_next;
    // (round 6: the zero padding is a UNIFORM branch -- as `part && k >= kv` selects inside one body hipcc emitted four
    //  v_cndmask per element for EVERY chunk, ~45 vector instructions next to the 48 MFMAs of a chunk, although only one
    //  chunk in cpg is partial: SQ_INSTS_VALU per MFMA 5.2 -> see profiles/r06_xgemm_pmc.md)
    if (!part) {
      if (amode == 0) {
        const unsigned d = buf + ((t >> 7) * PA + (t & 127u)) * 8u;
#pragma unroll
        for (int u = 0; u < NA; ++u) lds_write8(d + (unsigned)u * (2u * PA * 8u), va[u]);
      } else {
        const unsigned d = buf + ((t & (KC - 1u)) * PA + (t >> KCL)) * 8u;
#pragma unroll
        for (int u = 0; u < NA; ++u) lds_write8(d + 8u * RSTEP * u, va[u]);
      }
      if (bmode == 0) {
        const unsigned d = buf + A_BYTES + ((t >> 5) * PB + (t & 31u)) * 8u;
#pragma unroll
        for (int u = 0; u < NBL; ++u) lds_write8(d + (unsigned)(u / NB) * (8u * PB * 8u) + (unsigned)(u % NB) * 256u, vb[u]);
      } else {
        const unsigned d = buf + A_BYTES + ((t & (KC - 1u)) * PB + (t >> KCL)) * 8u;
#pragma unroll
        for (int u = 0; u < NBL; ++u) lds_write8(d + 8u * RSTEP * u, vb[u]);
      }
      return;
    }
    if (amode == 0) {
      const unsigned d = buf + ((t >> 7) * PA + (t & 127u)) * 8u;
      const int kh = (int)(t >> 7);
#pragma unroll
      for (int u = 0; u < NA; ++u) {
        v2f_t v = va[u];
        if (kh + 2 * u >= kv) v = v2f_t{0.f, 0.f};
        lds_write8(d + (unsigned)u * (2u * PA * 8u), v);
      }
    } else {
      const unsigned d = buf + ((t & (KC - 1u)) * PA + (t >> KCL)) * 8u;
      const bool z = (int)(t & (KC - 1u)) >= kv;
#pragma unroll
      for (int u = 0; u < NA; ++u) lds_write8(d + 8u * RSTEP * u, z ? v2f_t{0.f, 0.f} : va[u]);
    }
    if (bmode == 0) {
      const unsigned d = buf + A_BYTES + ((t >> 5) * PB + (t & 31u)) * 8u;
      const int kh = (int)(t >> 5);
#pragma unroll
      for (int u = 0; u < NBL; ++u) {
        v2f_t v = vb[u];
        if (kh + 8 * (u / NB) >= kv) v = v2f_t{0.f, 0.f};
        lds_write8(d + (unsigned)(u / NB) * (8u * PB * 8u) + (unsigned)(u % NB) * 256u, v);
      }
    } else {
      const unsigned d = buf + A_BYTES + ((t & (KC - 1u)) * PB + (t >> KCL)) * 8u;
      const bool z = (int)(t & (KC - 1u)) >= kv;
#pragma unroll
      for (int u = 0; u < NBL; ++u) lds_write8(d + 8u * RSTEP * u, z ? v2f_t{0.f, 0.f} : vb[u]);
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
  unsigned set = 0, cset = 0; // table sets of the current tile: rows, columns
  build_tile(T, 0u, 0u, true);
  __syncthreads();
  issue(T, 0u, 0u, true);
  int kvalid = kvalid_next;
  fill(0u);
  __syncthreads();
  unsigned cur = 0;
  const unsigned lane_x = (unsigned)(h * PA + 32 * wave + j) * 8u;
  const unsigned lane_w = A_BYTES + (unsigned)(h * PB + j) * 8u;
  const int flush_chunks = P.flush_chunks;
  const int fill_sel = P.prio; // (development: the trip after which the next chunk goes to LDS; 0: after the second)
  int tile_count = 0;
  for (;; ++tile_count) {
    XG_MARK(0);
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
    XG_MARK(1);
    f32x16 acc[NB * 3];
#pragma unroll
    for (int b = 0; b < NB * 3; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;
    int since_flush = 0;
    bool flushed_before = false, cols_change = false;
    for (long c = 0; c < n_chunks; ++c) {
      const bool last = c + 1 == n_chunks;
      bool have_next = true, new_cols = false;
      // the first operands of this chunk are read while the next chunk's loads are being issued
      v2f_t X0, X1, W0[NB], W1[NB];
      unsigned xo = cur * STAGE + lane_x, wo = cur * STAGE + lane_w;
      X0 = lds_read8(xo);
#pragma unroll
      for (int b = 0; b < NB; ++b) W0[b] = lds_read8(wo + (unsigned)b * 256u);
      XG_MARK(2);
      if (!last) {
        issue(T, set, cset, false);
      } else {
        have_next = more_tiles;
        if (have_next) { // (the other table sets were last read in an earlier tile's epilogue, at least a barrier ago)
          new_cols = Tn.n0 != T.n0;
          cols_change = new_cols;
#ifndef XG_ABLATE_BUILD // (timing probe: every tile uses stale tables)
          build_tile(Tn, set ^ 1u, cset ^ 1u, new_cols);
#endif
          XG_MARK(3);
          __syncthreads();
          XG_MARK(4);
          issue(Tn, set ^ 1u, new_cols ? cset ^ 1u : cset, true);
        }
      }
      // ---- multiply chunk `cur`: pairs of contracted values, two per trip (a group's last chunk is zero-padded in LDS, so a
      //      trip may run one pair past ceil(kvalid / 2); the operands of pair s + 1 are read under the MFMAs of pair s).
      //      The next chunk goes registers -> LDS in the MIDDLE of the loop: its loads were issued a thousand cycles ago,
      //      and what is left between the last MFMA and the barrier is nothing (two workgroups of a CU fall into lockstep --
      //      both in their MFMA loops, then both copying -- so whatever a wave does outside the loop, the pipe is idle for).
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
        constexpr unsigned XS = 2u * PA * 8u, WS = 2u * PB * 8u; // bytes between two pairs
        XG_MARK(5);
        const int trips = (kvalid + 3) >> 2;
        const int fill_at = fill_sel == 0 ? (trips > 1 ? 1 : 0) : (fill_sel < trips ? fill_sel : trips - 1);
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
          if (q == fill_at && have_next) fill((cur ^ 1u) * STAGE);
        }
      }
      XG_MARK(6);
      ++since_flush;
      const bool flush = last || (flush_chunks > 0 && since_flush == flush_chunks);
#ifdef XG_ABLATE_EPI
      if (false) {
#else
      if (flush) {
#endif
        // ---- epilogue: accumulators -> C (a later partial sum of the tile is added to what the earlier ones left).
        //      Four consecutive table entries per LDS read; groups of four registers that lie past the last row / column are
        //      skipped as a whole (a 9-column step stores 2 of its 4 groups); no predicates inside full tiles.  (The first
        //      version -- one table read, one 64-bit address and two compares per stored element -- was 0.8 of the 1.9 ms
        //      of a 27 x 27 step and 1.3 of the 2.8 ms of a 9 x 9 step, WITHOUT its stores.)
        const unsigned ttr = TT + set * 1024u + 512u, ttc = TT + 2048u + cset * 1024u + 512u; // rowC, colC
        const unsigned Mtot = (unsigned)P.m.total, Ntot = (unsigned)P.n.total;
        const unsigned rows_left = Mtot - T.m0, cols_left = Ntot - T.n0; // (uniform; may exceed the tile)
        const bool full = rows_left >= (unsigned)TM && cols_left >= (unsigned)TN;
        char *Cc = reinterpret_cast<char *>(C) + ((unsigned long)T.hC << 3);
        unsigned jj = (unsigned)j, hh4 = 4u * (unsigned)h; // (opaque: no store address is computed before its turn)
        OPAQUE_V(jj);
        OPAQUE_V(hh4);
        // lanes run along `lane side` (rows if !TRANS, columns if TRANS), registers along the other (`reg side`)
        const unsigned lane_left = TRANS ? cols_left : rows_left, reg_left = TRANS ? rows_left : cols_left;
        const unsigned lane_tab = TRANS ? ttc : ttr, reg_tab = TRANS ? ttr : ttc;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          // (TRANS: the lane's column depends on the block; !TRANS: its row does not)
          const unsigned lane_loc = TRANS ? 32u * b + jj : 32u * wave + jj;
          const bool lane_ok = full || lane_loc < lane_left;
          char *lp = Cc + ((unsigned long)lds_read4(lane_tab + 4u * lane_loc) << 3);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const unsigned reg0 = (TRANS ? 32u * wave : 32u * b) + 8u * g; // first row / column of this group of registers (+ 4 h)
            if (!full && reg0 >= reg_left) continue; // (uniform)
            const u32x4_t tab = *(__attribute__((address_space(3))) u32x4_t *)(unsigned long)(reg_tab + 4u * (reg0 + hh4));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int r = 4 * g + q;
              const float t1 = acc[3 * b][r], t2 = acc[3 * b + 1][r], t3 = acc[3 * b + 2][r];
              v2f_t val = {t1 - t2, t3 - t1 - t2};
              if (lane_ok && (full || reg0 + hh4 + (unsigned)q < reg_left)) {
                v2f_t *dst = reinterpret_cast<v2f_t *>(lp + ((unsigned long)tab[q] << 3));
                if (flushed_before) val += __builtin_nontemporal_load(dst); // (written by this lane at the previous flush: read past the L1)
#ifdef XG_ABLATE_STORE
                asm volatile("" ::"v"(val), "v"(dst));
#else
                *dst = val;
#endif
              }
            }
            __builtin_amdgcn_sched_barrier(0); // (no more than four addresses alive at a time)
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
      XG_MARK(7);
      kvalid = kvalid_next;
      __syncthreads();
      XG_MARK(8);
      cur ^= 1u;
    }
    if (!more_tiles) break;
    T = Tn;
    set ^= 1u;
    if (cols_change) cset ^= 1u;
  }
}
