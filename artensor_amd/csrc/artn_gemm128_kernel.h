// artn_gemm128_kernel.h -- complex128 on the matrix cores (included by artn_kernels.hip).
//
// The reference takes any dtype (`TensorNetworkSimulation.contraction(dtype=...)`,
// /root/reference/artensor/simulation.py:90; its own test runs complex128 through the same executors).
// Round 1 sent every complex128 step to the strided one-thread-per-element kernel; this is the two-operand
// LDS GEMM of artn_gemm_kernel.h with 16-byte elements on v_mfma_f64_16x16x4_f64:
//
//   workgroup  = a C tile of 2^mt x 2^nt elements, 4 waves in a wm x wn grid, each 2 x NB MFMA blocks of
//                16 rows (m) x 8 complex columns (n);
//   chunk      = 2^3 contracted values: [8][2^mt] of the first operand and [8][2^nt] of the second, one
//                16-byte element per copy lane, global -> registers -> LDS, double buffered; image rows
//                are 2^7 elements apart (compile-time LDS offsets in the MFMA loop);
//   k loop     = remaining contracted bits in Gray-code order, accumulators (f64) in registers throughout;
//   epilogue   = accumulators -> LDS in C order (XOR-swizzled) -> 16-byte stores, passes of 2^12 elements.
//
// Lane roles of one v_mfma_f64_16x16x4_f64 (D[i][j] += sum_kk Aop[i][kk] Bop[kk][j], one f64 per lane and operand;
// lane group g = lane >> 4 carries kk = g = 2 * kcl + p: contracted value kc = 2s + kcl, p = 0 re / 1 im of the X element):
//   W side (A operand): row i = lane & 15 = 2 * n_in + ro     value (ro, p): (0,0) re(b)  (0,1) -im(b)  (1,0) im(b)  (1,1) re(b)
//   X side (B operand): column j = lane & 15 = m_in           value p ? im(a) : re(a)
//   accumulator register r of lane (j, g): row i = g + 4r  ->  ro = g & 1, n_in = (g >> 1) + 2r   (guide: f64 C/D map)
// (f64x4, lds_read_f64, lds_write_f64: artn_kernels.hip, next to the other LDS accessors)

// GATHER: row indices on one outer axis (artn_contract_gather: the chunk loop of the sparse executor in complex128,
// /root/reference/artensor/contraction.py:140-175 with `dtype=torch.complex128`): the tile offsets of both operands are
// read through rows_a[x] / rows_b[x] (tile_offsets<true>), nothing else changes.
template <int NB, bool GATHER = false>
__global__ __launch_bounds__(ARTN_WG_THREADS, 2) void artn_k_gemm128(const double2 *__restrict__ A, const double2 *__restrict__ B,
                                                                    double2 *__restrict__ C, const ArtnGemmPlan P) {
  constexpr int MB = 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if ((unsigned)(unsigned long)(lds_byte_t *)smem != 0) __builtin_trap(); // LDS is addressed by raw byte offsets
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4, ro = j & 1, p = g & 1;
  const int mt = P.mt, nt = P.nt;
  constexpr unsigned a_bytes = 16u << (ARTN_GEMM_PITCH_LOG2 + ARTN_GEMM128_KC), stage_bytes = 2 * a_bytes;
  constexpr unsigned ROW2 = 32u << ARTN_GEMM_PITCH_LOG2; // bytes between k pairs (two image rows of 16-byte elements)
  const int epi_bits = P.tc_bits < ARTN_GEMM128_EPI_BITS ? P.tc_bits : ARTN_GEMM128_EPI_BITS;
  const unsigned epi_bytes = 16u << epi_bits;
  const unsigned tab_base = 2 * stage_bytes > epi_bytes ? 2 * stage_bytes : epi_bytes;
  long *offtab = reinterpret_cast<long *>(smem + tab_base);
  long *kotab = offtab + 512 + 128;
  if (tid < P.n_ko) {
    kotab[2 * tid] = P.ko_sA[tid] * 16;
    kotab[2 * tid + 1] = P.ko_sB[tid] * 16;
  }
  const OffTab OT = build_offset_table(P, offtab, tid);

  // ---- copy threads: element e = tid + 256 * u of an image (one 16-byte element per lane load)
  const int a_iters = P.ta_bits > 8 ? 1 << (P.ta_bits - 8) : 1, b_iters = P.tb_bits > 8 ? 1 << (P.tb_bits - 8) : 1;
  const bool a_act = P.ta_bits >= 8 || tid < (1 << P.ta_bits), b_act = P.tb_bits >= 8 || tid < (1 << P.tb_bits);
  unsigned a_gl = 0, a_ll = 0, b_gl = 0, b_ll = 0;
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    if ((tid >> b) & 1) {
      if (b < P.ta_bits) { a_gl += (unsigned)P.a_stride[b] * 16u; a_ll += (unsigned)P.a_lds[b]; }
      if (b < P.tb_bits) { b_gl += (unsigned)P.b_stride[b] * 16u; b_ll += (unsigned)P.b_lds[b]; }
    }
  }
  long a_gi[2], b_gi[2];
  unsigned a_li[2], b_li[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    a_gi[b] = 8 + b < P.ta_bits ? P.a_stride[8 + b] * 16 : 0;
    a_li[b] = 8 + b < P.ta_bits ? (unsigned)P.a_lds[8 + b] : 0u;
    b_gi[b] = 8 + b < P.tb_bits ? P.b_stride[8 + b] * 16 : 0;
    b_li[b] = 8 + b < P.tb_bits ? (unsigned)P.b_lds[8 + b] : 0u;
  }

  // ---- MFMA lanes
  const int wn = wave & ((1 << P.wn_log2) - 1), wm = wave >> P.wn_log2;
  const bool w_active = wm < (1 << P.wm_log2);
  const int n_in = j >> 1;
  const bool w_valid = nt >= 3 || n_in < (1 << nt);
  // X: element (kc = 2s + (g >> 1), row), component p;  W: element (kc, column), component ro ^ p, negated for (ro, p) = (0, 1)
  const unsigned lane_x = ((((unsigned)(g >> 1)) << ARTN_GEMM_PITCH_LOG2) + (unsigned)(wm * MB * 16 + j)) * 16u + (unsigned)p * 8u;
  const unsigned lane_w = a_bytes + ((((unsigned)(g >> 1)) << ARTN_GEMM_PITCH_LOG2) + (unsigned)(wn * NB * 8) + (unsigned)(w_valid ? n_in : 0)) * 16u +
                          (unsigned)(ro ^ p) * 8u;
  const double w_sign = !w_valid ? 0.0 : ((ro == 0 && p == 1) ? -1.0 : 1.0);

  // ---- epilogue offsets (16-byte elements of the C-ordered result image, swizzled; disjoint bit fields: XOR)
  auto m_off = [&](int m_local) {
    unsigned o = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i)
      if (i < mt && ((m_local >> i) & 1)) o |= 1u << P.m_pos[i];
    return o;
  };
  auto n_off = [&](int n_local) {
    unsigned o = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i)
      if (i < nt && ((n_local >> i) & 1)) o |= 1u << P.n_pos[i];
    return o;
  };
  // accumulator register r of lane (j, g): column n_in = (g >> 1) + 2r, parity ro' = g & 1
  const unsigned lane_c = swz_gemm(m_off(wm * MB * 16 + j) | n_off(wn * NB * 8 + (g >> 1)), P);
  unsigned c_mb[MB], c_nb[NB];
#pragma unroll
  for (int q = 0; q < MB; ++q) c_mb[q] = swz_gemm(m_off(q * 16), P);
#pragma unroll
  for (int q = 0; q < NB; ++q) c_nb[q] = swz_gemm(n_off(q * 8), P);
  const unsigned c_r0 = swz_gemm(n_off(2), P), c_r1 = swz_gemm(n_off(4), P);
  const int n_lim = nt >= 3 ? 8 : 1 << nt;
  const int o_iters = epi_bits > 8 ? 1 << (epi_bits - 8) : 1;
  const bool o_act = epi_bits >= 8 || tid < (1 << epi_bits);
  unsigned o_gl = 0;
#pragma unroll
  for (int b = 0; b < 8; ++b)
    if (((tid >> b) & 1) && b < P.tc_bits) o_gl += (unsigned)P.out_stride[b] * 16u;
  long o_gi[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) o_gi[b] = 8 + b < epi_bits ? P.out_stride[8 + b] * 16 : 0;
  const int n_pass = 1 << (P.tc_bits - epi_bits);
  const unsigned o_ll = swz_gemm((unsigned)tid, P) * 16u;

  f32x4 va[4], vb[4];
  auto issue = [&](const char *__restrict__ Ab, const char *__restrict__ Bb) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (u < a_iters && a_act) va[u] = *reinterpret_cast<const f32x4 *>(Ab + ((u & 1) ? a_gi[0] : 0) + ((u & 2) ? a_gi[1] : 0) + a_gl);
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (u < b_iters && b_act) vb[u] = *reinterpret_cast<const f32x4 *>(Bb + ((u & 1) ? b_gi[0] : 0) + ((u & 2) ? b_gi[1] : 0) + b_gl);
  };
  auto fill = [&](unsigned buf) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (u < a_iters && a_act) lds_write16(buf + a_ll + ((u & 1) ? a_li[0] : 0u) + ((u & 2) ? a_li[1] : 0u), va[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (u < b_iters && b_act) lds_write16(buf + a_bytes + b_ll + ((u & 1) ? b_li[0] : 0u) + ((u & 2) ? b_li[1] : 0u), vb[u]);
  };

  long t0 = blockIdx.x;
  const long G = gridDim.x, n_tiles = P.n_tiles;
  if ((G & 7) == 0) t0 = (long)(blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  const int n_chunks = 1 << P.n_ko;
  __syncthreads();
  TileOff off = {0, 0, 0, 0}, noff = {0, 0, 0, 0};
  const char *Ac = reinterpret_cast<const char *>(A), *Bc = reinterpret_cast<const char *>(B);
  if (t0 < n_tiles) {
    off = tile_offsets<GATHER>(P, OT, t0);
    issue(Ac + off.a * 16, Bc + off.b1 * 16);
    fill(0u);
  }
  __syncthreads();
  unsigned cur = 0;
  for (long tile = t0; tile < n_tiles; tile += G) {
    const bool more_tiles = tile + G < n_tiles;
    if (more_tiles) noff = next_offsets<GATHER>(P, OT, off, tile, G);
    f64x4 acc[MB][NB];
#pragma unroll
    for (int a = 0; a < MB; ++a)
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[a][b][e] = 0.0;
    long ka = 0, kb = 0;
    for (int c = 0; c < n_chunks; ++c) {
      const bool last = c + 1 == n_chunks;
      bool have_next = true;
      if (!last) {
        const int bit = __builtin_ctz((unsigned)(c + 1));
        const unsigned gn = (unsigned)(c + 1) ^ ((unsigned)(c + 1) >> 1);
        const long sa = kotab[2 * bit], sb = kotab[2 * bit + 1];
        if ((gn >> bit) & 1) { ka += sa; kb += sb; } else { ka -= sa; kb -= sb; }
        ka = uniform64(ka);
        kb = uniform64(kb);
        issue(Ac + off.a * 16 + ka, Bc + off.b1 * 16 + kb);
      } else {
        have_next = more_tiles;
        if (have_next) issue(Ac + noff.a * 16, Bc + noff.b1 * 16);
      }
      if (w_active) {
        const unsigned xa = cur * stage_bytes + lane_x, wa = cur * stage_bytes + lane_w;
#pragma unroll
        for (int s = 0; s < 4; ++s) { // 4 pairs of contracted values per chunk
          double X[MB], W[NB];
#pragma unroll
          for (int a = 0; a < MB; ++a) X[a] = lds_read_f64(xa + (unsigned)s * ROW2 + (unsigned)a * 256u);
#pragma unroll
          for (int b = 0; b < NB; ++b) W[b] = lds_read_f64(wa + (unsigned)s * ROW2 + (unsigned)b * 128u) * w_sign;
#pragma unroll
          for (int a = 0; a < MB; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(W[b], X[a], acc[a][b], 0, 0, 0);
        }
      }
      if (!last) {
        fill((cur ^ 1u) * stage_bytes);
        __syncthreads();
        cur ^= 1u;
        continue;
      }
      // ---- epilogue
      char *Cb = reinterpret_cast<char *>(C) + off.c * 16;
      for (int pass = 0; pass < n_pass; ++pass) {
        __syncthreads();
        unsigned lc = lane_c;
        OPAQUE_V(lc);
        if (w_active) {
#pragma unroll
          for (int a = 0; a < MB; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int n_loc = (g >> 1) + 2 * r;
                const unsigned pos = lc ^ c_mb[a] ^ c_nb[b] ^ ((r & 1) ? c_r0 : 0u) ^ ((r & 2) ? c_r1 : 0u);
                if (n_loc < n_lim && (int)(pos >> ARTN_GEMM128_EPI_BITS) == pass)
                  lds_write_f64((pos & ((1u << ARTN_GEMM128_EPI_BITS) - 1u)) * 16u + (unsigned)p * 8u, acc[a][b][r]);
              }
        }
        __syncthreads();
        char *Cp = Cb + (P.tc_bits > epi_bits ? pass * P.out_stride[epi_bits] * 16 : 0);
        unsigned oll = o_ll ^ ((swz_gemm((unsigned)pass << ARTN_GEMM128_EPI_BITS, P) & ((1u << ARTN_GEMM128_EPI_BITS) - 1u)) * 16u), ogl = o_gl;
        OPAQUE_V(oll);
        OPAQUE_V(ogl);
        for (int i0 = 0; i0 < o_iters; i0 += 4) {
          f32x4 x[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int i = i0 + u;
            if (i < o_iters && o_act) x[u] = lds_read16(oll ^ (swz_gemm((unsigned)i * 256u, P) * 16u));
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int i = i0 + u;
            if (i < o_iters && o_act) {
              long o = 0;
#pragma unroll
              for (int b = 0; b < 4; ++b)
                if ((i >> b) & 1) o += o_gi[b];
              *reinterpret_cast<f32x4 *>(Cp + o + ogl) = x[u];
            }
          }
        }
      }
      __syncthreads();
      if (have_next) {
        fill(0u);
        __syncthreads();
      }
      cur = 0;
    }
    off = noff;
  }
}
