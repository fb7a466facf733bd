// artn_xgemm128_kernel.h -- the extent-based GEMM in complex128 (included by artn_kernels.hip; round 5).
//
// artn_k_xgemm (artn_xgemm_kernel.h) for 16-byte elements on v_mfma_f64_16x16x4_f64: a step whose labels have ANY extents,
// run in the precision the caller's tensors have (the reference executes whatever dtype it is given:
// /root/reference/artensor/simulation.py:90, contraction.py:70).  Same plan (ArtnXGemmPlan: flattened mixed-radix indices m, n, k,
// separable element offsets, level tables + per-tile row / column tables in LDS, runs of row tiles), same tile order; what differs:
//
//   tile       = 128 consecutive values of m x 32 NB of n (NB = 1: two blocks need 128 accumulator registers and spill); wave w owns rows 32 w .. 32 w + 31 as two blocks of 16
//                and all 4 NB blocks of 8 complex columns: 8 NB accumulators f64x4 (four real products per complex product);
//   chunk      = 8 contracted values: images [8][130] and [8][32 NB + 2] of 16-byte elements, one element per lane and load,
//                double buffered (55 KiB of LDS with the tables: two workgroups per CU);
//   MFMA       = D[i][j] += sum_kk Aop[i][kk] Bop[kk][j], lane group g = lane >> 4 carries kk = g = 2 kcl + p (contracted value
//                2 s + kcl of pair s, p = 0 re / 1 im of the X element), j = lane & 15:
//                  W side (A operand): row i = j = 2 n_in + ro: value (ro, p) = (0,0) re b, (0,1) -im b, (1,0) im b, (1,1) re b
//                  X side (B operand): column j = row m of the block: p ? im a : re a
//                  accumulator register r of lane (j, g): row i = g + 4 r -> component g & 1 of column n_in = (g >> 1) + 2 r
//                (the lane map of artn_k_gemm128);
//   epilogue   = straight from the accumulators, one f64 component per lane and store (the real and the imaginary part of a
//                result element sit in neighbouring lane groups); partial sums every 4 096 contracted values as in complex64.
// The chunk loop is the plain one -- issue the next chunk's loads, multiply this one, registers -> LDS, barrier: complex128 is the
// accuracy path, not the throughput path (f64 MFMA peak 78.6 TFLOP/s).

#define ARTN_XG128_KC 8

template <int NB>
__global__ __launch_bounds__(ARTN_WG_THREADS, 2) void artn_k_xgemm128(const double2 *__restrict__ A, const double2 *__restrict__ B,
                                                                     double2 *__restrict__ C, const ArtnXGemmPlan P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if ((unsigned)(unsigned long)(lds_byte_t *)smem != 0) __builtin_trap(); // LDS is addressed by raw byte offsets
  constexpr int TM = ARTN_XG_TM, TN = 32 * NB, KC = ARTN_XG128_KC;
  constexpr int KCL = 3, RSTEP = ARTN_WG_THREADS / KC;
  constexpr int PA = TM + 2, PB = TN + 2;
  constexpr unsigned A_BYTES = KC * PA * 16, B_BYTES = KC * PB * 16, STAGE = A_BYTES + B_BYTES;
  constexpr unsigned LEV = 2 * STAGE;
  constexpr unsigned T_MA0 = LEV, T_MA1 = LEV + 2048, T_NB0 = LEV + 4096, T_NB1 = LEV + 6144, T_KA = LEV + 8192, T_KB = T_KA + ARTN_XG_KTAB * 4;
  constexpr unsigned TT = T_KB + ARTN_XG_KTAB * 4;
  constexpr int NA = TM * KC / ARTN_WG_THREADS, NBL = TN * KC / ARTN_WG_THREADS; // 4 and NB loads per thread and chunk
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4, ro = j & 1, p = g & 1;

  // ---- level tables (as artn_k_xgemm)
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
  if (tid < ARTN_XG_KC) { // (padding of the k tables: a chunk reads kbase .. kbase + 7 without clamping)
    unsigned o0, o1;
    artn_xg_decode(P.k, 0, P.k.n0, (unsigned)P.k.L0 - 1u, o0, o1);
    lds_write4(T_KA + 4u * (P.k.L0 + tid), o0);
    lds_write4(T_KB + 4u * (P.k.L0 + tid), o1);
  }
  // ---- per-tile tables (as artn_k_xgemm: the outer digits cached per thread, incremental along a run of row tiles)
  unsigned c_q1 = 0xffffffffu, c_o0 = 0, c_o1 = 0;
  unsigned s_first = 0xffffffffu, s_i0b = 0, s_i1b = 0, s_q1b = 0;
  auto build_side = [&](const ArtnXSide &S, unsigned first, int loc, unsigned t0, unsigned t1, unsigned dst) {
    const unsigned tot = (unsigned)S.total, L0 = (unsigned)S.L0, L1 = (unsigned)S.L1;
    unsigned pos = (unsigned)loc;
    if (first + pos >= tot) pos = tot - 1 - first;
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
    lds_write4(dst + 4u * loc, o0);
    lds_write4(dst + 512u + 4u * loc, o1);
  };
  auto build_tile = [&](const XgTile &T, unsigned rs, unsigned cs, bool cols) {
    if (wave < 2) build_side(P.m, T.m0, tid, T_MA0, T_MA1, TT + rs * 1024u);
    else if (cols && tid - TM < TN) build_side(P.n, T.n0, tid - TM, T_NB0, T_NB1, TT + 2048u + cs * 1024u);
  };

  // ---- copy slots (chunks of 8):  A, mode 0: row = t & 127, kk = (t >> 7) + 2 u;  mode 1: kk = t & 7, row = (t >> 3) + 32 u  (u < 4)
  //                                 B, mode 0: col = (t & 31) + 32 u, kk = t >> 5;   mode 1: kk = t & 7, col = (t >> 3) + 32 u  (u < NB)
  const int amode = P.amode, bmode = P.bmode;
  const unsigned K0 = (unsigned)P.k.L0;
  const int cpg = P.cpg;
  const long n_chunks = (long)P.k_groups * cpg;
  int iq = 0;
  unsigned ig = 0, gA = 0, gB = 0;
  int kvalid_next = 0;
  f32x4 va[NA], vb[NBL];
  const char *Ac = reinterpret_cast<const char *>(A), *Bc = reinterpret_cast<const char *>(B);
  auto ld = [&](const char *base, unsigned off) { return *reinterpret_cast<const f32x4 *>(base + ((unsigned long)off << 4)); };
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
    const unsigned t = (unsigned)tid;
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
      const unsigned k0 = T.hB + gB + lds_read4(T_KB + 4u * (kbase + (t >> 5)));
      const unsigned cb = ttc + 4u * (t & 31u);
#pragma unroll
      for (int u = 0; u < NBL; ++u) vb[u] = ld(Bc, k0 + lds_read4(cb + 128u * u));
    } else {
      const unsigned base = T.hB + gB + lds_read4(T_KB + 4u * (kbase + (t & (KC - 1u))));
      const unsigned cb = ttc + 4u * (t >> KCL);
#pragma unroll
      for (int u = 0; u < NBL; ++u) vb[u] = ld(Bc, base + lds_read4(cb + 4u * RSTEP * u));
    }
  };
  auto fill = [&](unsigned buf) { // registers -> LDS images; contracted values past the end of a group are zeros
    const unsigned t = (unsigned)tid;
    const bool part = kvalid_next < KC;
    const int kv = kvalid_next;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    if (amode == 0) {
      const unsigned d = buf + ((t >> 7) * PA + (t & 127u)) * 16u;
      const int kh = (int)(t >> 7);
#pragma unroll
      for (int u = 0; u < NA; ++u) lds_write16(d + (unsigned)u * (2u * PA * 16u), (part && kh + 2 * u >= kv) ? zero : va[u]);
    } else {
      const unsigned d = buf + ((t & (KC - 1u)) * PA + (t >> KCL)) * 16u;
      const bool z = part && (int)(t & (KC - 1u)) >= kv;
#pragma unroll
      for (int u = 0; u < NA; ++u) lds_write16(d + 16u * RSTEP * u, z ? zero : va[u]);
    }
    if (bmode == 0) {
      const unsigned d = buf + A_BYTES + ((t >> 5) * PB + (t & 31u)) * 16u;
      const bool z = part && (int)(t >> 5) >= kv;
#pragma unroll
      for (int u = 0; u < NBL; ++u) lds_write16(d + (unsigned)u * 512u, z ? zero : vb[u]);
    } else {
      const unsigned d = buf + A_BYTES + ((t & (KC - 1u)) * PB + (t >> KCL)) * 16u;
      const bool z = part && (int)(t & (KC - 1u)) >= kv;
#pragma unroll
      for (int u = 0; u < NBL; ++u) lds_write16(d + 16u * RSTEP * u, z ? zero : vb[u]);
    }
  };

  // ---- tile order (as artn_k_xgemm: runs of row tiles of one column tile, dealt to XCD-contiguous workgroups)
  const unsigned G = gridDim.x, tiles_n = (unsigned)P.tiles_n;
  const unsigned total_hm = (unsigned)(P.n_tiles / P.tiles_n);
  unsigned run = (unsigned)(P.n_tiles / ((long)G * 4));
  run = run < 1 ? 1 : (run > 16 ? 16 : run);
  const unsigned n_super = (total_hm + run - 1) / run, n_runs = n_super * tiles_n;
  __syncthreads(); // level tables are in LDS
  const unsigned wg = (G & 7) == 0 ? (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3) : blockIdx.x;
  unsigned R = wg, pos = 0;
  if (R >= n_runs) return;
  unsigned r_tn = R % tiles_n, r_hm0 = (R / tiles_n) * run;
  XgTile T = xg_tile(P, r_hm0, r_tn), Tn = T;
  unsigned set = 0, cset = 0;
  build_tile(T, 0u, 0u, true);
  __syncthreads();
  issue(T, 0u, 0u, true);
  int kvalid = kvalid_next;
  fill(0u);
  __syncthreads();
  unsigned cur = 0;
  // X: element (kk = 2 s + (g >> 1), row 32 wave + 16 mb + j), component p;  W: element (kk, column 8 nbk + (j >> 1)), component ro ^ p
  const unsigned lane_x = (unsigned)((g >> 1) * PA + 32 * wave + j) * 16u + (unsigned)p * 8u;
  const unsigned lane_w = A_BYTES + (unsigned)((g >> 1) * PB + (j >> 1)) * 16u + (unsigned)(ro ^ p) * 8u;
  const double w_sign = (ro == 0 && p == 1) ? -1.0 : 1.0;
  const int flush_chunks = P.flush_chunks;
  for (;;) {
    bool more_tiles = true;
    if (pos + 1 < run && r_hm0 + pos + 1 < total_hm) ++pos;
    else {
      R += G;
      pos = 0;
      more_tiles = R < n_runs;
      if (more_tiles) { r_tn = R % tiles_n; r_hm0 = (R / tiles_n) * run; }
    }
    if (more_tiles) Tn = xg_tile(P, r_hm0 + pos, r_tn);
    f64x4 acc[2][4 * NB];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 4 * NB; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};
    int since_flush = 0;
    bool flushed_before = false, cols_change = false;
    for (long c = 0; c < n_chunks; ++c) {
      const bool last = c + 1 == n_chunks;
      bool have_next = true, new_cols = false;
      if (!last) {
        issue(T, set, cset, false);
      } else {
        have_next = more_tiles;
        if (have_next) { // (the other table sets were last read in an earlier tile's epilogue, at least a barrier ago)
          new_cols = Tn.n0 != T.n0;
          cols_change = new_cols;
          build_tile(Tn, set ^ 1u, cset ^ 1u, new_cols);
          __syncthreads();
          issue(Tn, set ^ 1u, new_cols ? cset ^ 1u : cset, true);
        }
      }
      // ---- multiply chunk `cur`: pairs of contracted values (a group's last chunk is zero-padded to an even count in LDS)
      {
        const unsigned xo = cur * STAGE + lane_x, wo = cur * STAGE + lane_w;
        const int pairs = (kvalid + 1) >> 1;
#pragma unroll 1
        for (int s = 0; s < pairs; ++s) {
          double x[2], w[4 * NB];
#pragma unroll
          for (int a = 0; a < 2; ++a) x[a] = lds_read_f64(xo + (unsigned)s * (2u * PA * 16u) + (unsigned)a * 256u);
#pragma unroll
          for (int b = 0; b < 4 * NB; ++b) w[b] = w_sign * lds_read_f64(wo + (unsigned)s * (2u * PB * 16u) + (unsigned)b * 128u);
#pragma unroll
          for (int b = 0; b < 4 * NB; ++b)
#pragma unroll
            for (int a = 0; a < 2; ++a) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(w[b], x[a], acc[a][b], 0, 0, 0);
        }
      }
      if (have_next) fill((cur ^ 1u) * STAGE);
      ++since_flush;
      const bool flush = last || (flush_chunks > 0 && since_flush == flush_chunks);
      if (flush) {
        // ---- epilogue: lane (j, g) holds component g & 1 of C[row 32 wave + 16 a + j][column 8 b + (g >> 1) + 2 r]
        const unsigned ttr = TT + set * 1024u + 512u, ttc = TT + 2048u + cset * 1024u + 512u; // rowC, colC
        const unsigned rows_left = (unsigned)P.m.total - T.m0, cols_left = (unsigned)P.n.total - T.n0;
        char *Cc = reinterpret_cast<char *>(C) + ((unsigned long)T.hC << 4) + (unsigned)p * 8u;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const unsigned m_loc = 32u * wave + 16u * a + (unsigned)j;
          const bool row_ok = m_loc < rows_left;
          char *rp = Cc + ((unsigned long)lds_read4(ttr + 4u * m_loc) << 4);
#pragma unroll
          for (int b = 0; b < 4 * NB; ++b) {
            if (8u * b >= cols_left) continue; // (uniform)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const unsigned n_loc = 8u * b + (unsigned)(g >> 1) + 2u * r;
              if (row_ok && n_loc < cols_left) {
                double *dst = reinterpret_cast<double *>(rp + ((unsigned long)lds_read4(ttc + 4u * n_loc) << 4));
                double val = acc[a][b][r];
                if (flushed_before) val += __builtin_nontemporal_load(dst); // (written by this lane at the previous flush: read past the L1)
                *dst = val;
              }
            }
          }
        }
        flushed_before = true;
        since_flush = 0;
        if (!last) {
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4 * NB; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};
        }
      }
      kvalid = kvalid_next;
      __syncthreads();
      cur ^= 1u;
    }
    if (!more_tiles) break;
    T = Tn;
    set ^= 1u;
    if (cols_change) cset ^= 1u;
  }
}
