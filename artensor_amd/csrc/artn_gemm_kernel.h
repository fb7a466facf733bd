// artn_gemm_kernel.h -- the two-operand LDS GEMM of libartn_hip.so (included by artn_kernels.hip).
//
// artn_k_bits streams ONE big operand and keeps the small one in registers; that stops paying
// when a step contracts 7+ bits (the fragments fill the register file, one workgroup per CU) or
// when the second operand is itself big (big x big steps of sliced circuits and random networks:
// the reference's torch.einsum at artensor/contraction.py:70,179,190 takes any pair of operands).
// Here both operands go through LDS:
//
//   workgroup  = a C tile of 2^mt x 2^nt elements (mt <= 7, nt <= 7), 4 waves in a wm x wn grid,
//                each wave MB x NB MFMA blocks of 32 rows (m) x 16 complex columns (n);
//   chunk      = 2^4 values of the contracted index: the [16][2^mt] piece of the first operand and
//                the [16][2^nt] piece of the second are copied global -> registers -> LDS with
//                16-byte lanes (element pairs along each operand's stride-1 bit), double buffered:
//                the loads of chunk c+1 are in flight while chunk c is multiplied;
//   k loop     = the remaining contracted bits, walked in Gray-code order (one stride added or
//                subtracted per chunk), accumulators in registers; every 2^12 contracted values the
//                partial sum is flushed into the C tile (read-add-write through the epilogue) and
//                the registers restart from zero: one chain of 2^15 fp32 additions put the n53 m20
//                big-batch slice 1.2e-5 of the typical amplitude from the reference, chains of 2^10
//                5.9e-6 (what the other 453 steps leave) at 4 % more time; 2^12 costs 1 %;
//   epilogue   = accumulators -> LDS in C order (XOR-swizzled like artn_k_bits' stage output) ->
//                16-byte coalesced stores, in passes of 2^13 elements.
//
// Arithmetic as in artn_k_bits: interleaved complex64 times the real block form of the other
// operand on v_mfma_f32_32x32x2_f32, 8 real FLOP per complex multiply-add, fp32 throughout.
// Lane roles of one MFMA pair (k pair s of a chunk, kc = 2s + h):
//   W side (MFMA A operand): row i = lane&31 = 2*n_in_block + ro, value from B image [kc][n]
//   X side (MFMA B operand): column j = lane&31 = m_in_block,     value from A image [kc][m]
//   accumulator register r of lane (j, h): n_in_block = ((r>>1)&1) + 2h + 4(r>>2), ro = r&1.

template <typename PlanT>
__device__ __forceinline__ unsigned swz_gemm(unsigned elem_off, const PlanT &P) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (i < P.swz_n && ((elem_off >> P.swz_src[i]) & 1)) elem_off ^= 1u << P.swz_dst[i];
  return elem_off;
}

// BF = true (ARTN_C64_BF16, the reduced-precision sampling mode): operands are rounded to bfloat16
// (round to nearest even) when they enter LDS, chunks are 2^5 contracted values, and the images hold per
// row and per 4 consecutive contracted values the 8 bf16 -- (re, im) x 4 -- that one lane feeds to
// v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate): image[kc >> 2][row][kc & 3], 16 bytes per lane
// read.  Lane half h carries contracted bit 2: kc = 8t + 4h + u.  The W side turns its raw (re, im)
// pairs into (re, -im) or (im, re) per output parity with one v_perm + one v_xor per pair.
// Accumulation stays fp32.

__device__ __forceinline__ void lds_write4(unsigned a, unsigned v) {
  *(__attribute__((address_space(3))) unsigned *)(unsigned long)a = v;
}

// M3 = true (fp32 only, tiles with 32+ columns): the complex product as THREE real products instead of four
// (the "3M" scheme of BLAS xGEMM3M): T1 = A_re B_re, T2 = A_im B_im, T3 = (A_re + A_im)(B_re + B_im);
// C_re = T1 - T2, C_im = T3 - T1 - T2.  An MFMA block is then 32 rows (m) x 32 complex columns (n) with three
// accumulators, three MFMAs per pair of contracted values instead of four for the same 32 x 32 outputs:
// 6 real FLOP per complex multiply-add on the matrix pipe, the other 2 become one add per operand element
// and three per result.  NB counts 32-column blocks in this mode.
// TALL (single-block tiles, fp32): images of 2^6 contracted values x 2^5 rows (ArtnGemmPlan::pitch_log2 = 5, kc = 6)
template <int MB, int NB, bool BF = false, bool M3 = false, bool TALL = false, bool GATHER = false>
__global__ __launch_bounds__(ARTN_WG_THREADS, 2) void artn_k_gemm(const float2 *__restrict__ A, const float2 *__restrict__ B,
                                                                 float2 *__restrict__ C, const ArtnGemmPlan P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if ((unsigned)(unsigned long)(lds_byte_t *)smem != 0) __builtin_trap(); // LDS is addressed by raw byte offsets
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5, ro = j & 1;
  const int mt = P.mt, nt = P.nt;
  // both images are [16][2^7] whatever mt / nt (ARTN_GEMM_PITCH_LOG2): LDS offsets of the MFMA loop are immediates
  // (fp32: [16][128] x 8 B; bf16: [8][128][4] x 4 B -- 16 KiB either way)
  static_assert(!TALL || (MB == 1 && NB == 1 && !BF), "tall chunks: one 32 x 32 block, fp32");
  constexpr int PITCH = TALL ? ARTN_GEMM_PITCH_TALL_LOG2 : ARTN_GEMM_PITCH_LOG2, KCB = TALL ? ARTN_GEMM_KC_TALL : ARTN_GEMM_KC;
  constexpr int NS = 1 << (KCB - 1); // fp32 MFMA steps (pairs of contracted values) per chunk
  constexpr unsigned a_bytes = 8u << (PITCH + KCB), b_bytes = a_bytes, stage_bytes = a_bytes + b_bytes;
  constexpr unsigned ROW2 = 16u << PITCH; // fp32: bytes between k pairs (two image rows)
  constexpr int NVA = BF ? 8 : 4;                        // 16-byte global loads per thread and chunk, first operand
  const int epi_bits = P.tc_bits < ARTN_GEMM_EPI_BITS ? P.tc_bits : ARTN_GEMM_EPI_BITS;
  const unsigned epi_bytes = 8u << epi_bits;
  const unsigned tab_base = 2 * stage_bytes > epi_bytes ? 2 * stage_bytes : epi_bytes;
  long *offtab = reinterpret_cast<long *>(smem + tab_base);
  long *kotab = offtab + 512 + 128;
  if (tid < P.n_ko) {
    kotab[2 * tid] = P.ko_sA[tid] * 8;
    kotab[2 * tid + 1] = P.ko_sB[tid] * 8;
  }
  const OffTab OT = build_offset_table(P, offtab, tid);

  // ---- copy threads: 16-byte chunk c = tid + 256 * u of an image, chunk bits 1.. -> strides
  const int a_cb = P.ta_bits - 1, b_cb = P.tb_bits - 1;      // chunk-index bits
  const int a_iters = a_cb > 8 ? 1 << (a_cb - 8) : 1, b_iters = b_cb > 8 ? 1 << (b_cb - 8) : 1;
  const bool a_act = a_cb >= 8 || tid < (1 << a_cb), b_act = b_cb >= 8 || tid < (1 << b_cb);
  unsigned a_gl = 0, a_ll = 0, b_gl = 0, b_ll = 0;
#pragma unroll
  for (int b = 1; b <= 8; ++b) {
    if ((tid >> (b - 1)) & 1) {
      if (b < P.ta_bits) { a_gl += (unsigned)P.a_stride[b] * 8u; a_ll += (unsigned)P.a_lds[b]; }
      if (b < P.tb_bits) { b_gl += (unsigned)P.b_stride[b] * 8u; b_ll += (unsigned)P.b_lds[b]; }
    }
  }
  long a_gi[3], b_gi[2];
  unsigned a_li[3], b_li[2];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    a_gi[b] = 9 + b < P.ta_bits ? P.a_stride[9 + b] * 8 : 0;
    a_li[b] = 9 + b < P.ta_bits ? (unsigned)P.a_lds[9 + b] : 0u;
  }
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    b_gi[b] = 9 + b < P.tb_bits ? P.b_stride[9 + b] * 8 : 0;
    b_li[b] = 9 + b < P.tb_bits ? (unsigned)P.b_lds[9 + b] : 0u;
  }

  const unsigned a_pair = (unsigned)P.a_lds[0], b_pair = (unsigned)P.b_lds[0]; // LDS bytes between the two elements of a lane load

  // ---- MFMA lanes
  // waves: wn x wm grid of blocks; what is left of the four (tiles with fewer than 4 blocks) splits the contracted
  // values of every chunk: wave wk takes k pairs wk, wk + WK, ...
  const int wn = wave & ((1 << P.wn_log2) - 1), wm = (wave >> P.wn_log2) & ((1 << P.wm_log2) - 1);
  // (only single-block waves ever share a block: the other instantiations carry none of this)
  constexpr bool CAN_SPLIT = MB == 1 && NB == 1;
  const int wk = CAN_SPLIT ? wave >> (P.wn_log2 + P.wm_log2) : 0, WK = CAN_SPLIT ? 1 << P.wk_log2 : 1;
  constexpr bool w_active = true;
  const int n_in = M3 ? j : j >> 1;
  const bool w_valid = M3 || nt >= 4 || n_in < (1 << nt);
  constexpr int NBW = M3 ? 32 : 16; // complex columns per MFMA block
  constexpr unsigned EB = BF ? 16u : 8u; // bytes per (row, k pair | k quad) slot
  const unsigned lane_x = (((unsigned)h << PITCH) + (unsigned)(wm * MB * 32 + j)) * EB;
  const unsigned lane_w = a_bytes + (((unsigned)h << PITCH) + (unsigned)(wn * NB * NBW) + (unsigned)(w_valid ? n_in : 0)) * EB;
  const unsigned w_sel = ro ? 0x01000302u : 0x03020100u, w_sign = ro ? 0u : 0x80000000u; // bf16 W side: (im, re) / (re, -im)

  // ---- epilogue offsets (elements of the C-ordered result image, swizzled; fields are disjoint: XOR)
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
  // (accumulator rows: 4M  n_in_block = b0 + 2h + 4q;  3M  n_in_block = (r & 3) + 4h + 8(r >> 2))
  const unsigned lane_c = swz_gemm(m_off(wm * MB * 32 + j) | n_off(wn * NB * NBW + (M3 ? 4 : 2) * h), P);
  unsigned c_mb[MB], c_nb[NB];
#pragma unroll
  for (int q = 0; q < MB; ++q) c_mb[q] = swz_gemm(m_off(q * 32), P);
#pragma unroll
  for (int q = 0; q < NB; ++q) c_nb[q] = swz_gemm(n_off(q * NBW), P);
  const unsigned c_b0 = swz_gemm(n_off(1), P), c_q0 = swz_gemm(n_off(M3 ? 8 : 4), P), c_q1 = swz_gemm(n_off(M3 ? 16 : 8), P);
  const unsigned c_b1 = swz_gemm(n_off(2), P); // 3M: row bit 1
  const int n_lim = nt >= 4 ? 16 : 1 << nt; // valid n_in_block values
  // copy-out threads
  const int o_cb = epi_bits - 1;
  const int o_iters = o_cb > 8 ? 1 << (o_cb - 8) : 1;
  const bool o_act = o_cb >= 8 || tid < (1 << o_cb);
  unsigned o_gl = 0;
#pragma unroll
  for (int b = 1; b <= 8; ++b)
    if (((tid >> (b - 1)) & 1) && b < P.tc_bits) o_gl += (unsigned)P.out_stride[b] * 8u;
  long o_gi[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) o_gi[b] = 9 + b < epi_bits ? P.out_stride[9 + b] * 8 : 0;
  const int n_pass = 1 << (P.tc_bits - epi_bits);
  const long pass_stride = P.tc_bits > epi_bits ? P.out_stride[epi_bits] * 8 : 0;
  const unsigned o_ll = swz_gemm((unsigned)tid * 2u, P) * 8u;

  f32x4 va[NVA], vb[4];
  auto issue = [&](const char *__restrict__ Ab, const char *__restrict__ Bb) {
    // (opaque per call: otherwise the tile-invariant part of every load address -- tile base + per-load offset +
    //  lane offset -- is hoisted out of the chunk loop into 64-bit VGPR pairs, 16+ registers that the 3M
    //  instantiations do not have: their spilled pairs were reloaded between the loads of one chunk, each reload
    //  waiting for the load before it.  Scalar base + 32-bit lane offset is the addressing mode wanted here.)
    unsigned agl = a_gl, bgl = b_gl;
    OPAQUE_V(agl);
    OPAQUE_V(bgl);
#pragma unroll
    for (int u = 0; u < NVA; ++u) {
      if (u < a_iters && a_act) {
        const long o = ((u & 1) ? a_gi[0] : 0) + ((u & 2) ? a_gi[1] : 0) + ((u & 4) ? a_gi[2] : 0);
#ifdef ARTN_ABLATE_MEM
        asm volatile("" : "=v"(va[u]) : "s"(Ab), "s"(o), "v"(agl));
#else
        va[u] = *reinterpret_cast<const f32x4 *>(Ab + o + agl);
#endif
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (u < b_iters && b_act) {
        const long o = ((u & 1) ? b_gi[0] : 0) + ((u & 2) ? b_gi[1] : 0);
#ifdef ARTN_ABLATE_MEM
        asm volatile("" : "=v"(vb[u]) : "s"(Bb), "s"(o), "v"(bgl));
#else
        vb[u] = *reinterpret_cast<const f32x4 *>(Bb + o + bgl);
#endif
      }
    }
  };
  auto put = [&](unsigned d, unsigned pair, const f32x4 &v) {
    if constexpr (BF) {
      const unsigned p0 = pack_bf16(v[0], v[1]), p1 = pack_bf16(v[2], v[3]);
      if (pair == 4u) *(lds_u2_t *)(unsigned long)d = u2_t{p0, p1};
      else { lds_write4(d, p0); lds_write4(d + pair, p1); }
    } else {
      if (pair == 8u) lds_write16(d, v);
      else { lds_write8(d, v2f_t{v[0], v[1]}); lds_write8(d + pair, v2f_t{v[2], v[3]}); }
    }
  };
  auto fill = [&](unsigned buf) {
#pragma unroll
    for (int u = 0; u < NVA; ++u)
      if (u < a_iters && a_act)
        put(buf + a_ll + ((u & 1) ? a_li[0] : 0u) + ((u & 2) ? a_li[1] : 0u) + ((u & 4) ? a_li[2] : 0u), a_pair, va[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (u < b_iters && b_act)
        put(buf + a_bytes + b_ll + ((u & 1) ? b_li[0] : 0u) + ((u & 2) ? b_li[1] : 0u), b_pair, vb[u]);
  };

  long t0 = blockIdx.x;
  const long G = gridDim.x, n_tiles = P.n_tiles;
  if ((G & 7) == 0) t0 = (long)(blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3); // XCD-contiguous tile ranges
  const int n_chunks = 1 << P.n_ko;
  // chunks per partial sum - 1: 2^12 contracted values in fp32; bf16 operands carry 2^-9 of rounding each, the
  // length of the fp32 chain does not matter there
  const int flush_mask = BF ? 0x7fffffff : (1 << (ARTN_GEMM_FLUSH_LOG2 - KCB)) - 1;
  __syncthreads(); // tables are in LDS
  TileOff off = {0, 0, 0, 0}, noff = {0, 0, 0, 0};
  const char *Ac = reinterpret_cast<const char *>(A), *Bc = reinterpret_cast<const char *>(B);
  if (t0 < n_tiles) {
    off = tile_offsets<GATHER>(P, OT, t0);
    issue(Ac + off.a * 8, Bc + off.b1 * 8);
    fill(0u);
  }
  __syncthreads();
  unsigned cur = 0;
  for (long tile = t0; tile < n_tiles; tile += G) {
    const bool more_tiles = tile + G < n_tiles;
    if (more_tiles) noff = next_offsets<GATHER>(P, OT, off, tile, G);
    constexpr int NACC = M3 ? 3 : 1;
    f32x16 acc[MB][NB * NACC];
#pragma unroll
    for (int a = 0; a < MB; ++a)
#pragma unroll
      for (int b = 0; b < NB * NACC; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    long ka = 0, kb = 0;
    for (int c = 0; c < n_chunks; ++c) {
      const bool last = c + 1 == n_chunks;
      bool have_next = true;
      if (!last) { // Gray code: chunk c+1 differs from chunk c in one looped bit
        const int bit = __builtin_ctz((unsigned)(c + 1));
        const unsigned gn = (unsigned)(c + 1) ^ ((unsigned)(c + 1) >> 1);
        const long sa = kotab[2 * bit], sb = kotab[2 * bit + 1];
        if ((gn >> bit) & 1) { ka += sa; kb += sb; } else { ka -= sa; kb -= sb; }
        ka = uniform64(ka);
        kb = uniform64(kb);
        issue(Ac + off.a * 8 + ka, Bc + off.b1 * 8 + kb);
      } else {
        have_next = more_tiles;
        if (have_next) issue(Ac + noff.a * 8, Bc + noff.b1 * 8);
      }
      // ---- multiply chunk `cur`
      if (w_active) {
        const unsigned base = cur * stage_bytes;
        const unsigned xa = base + lane_x, wa = base + lane_w;
        if constexpr (BF) {
          // 4 groups of 8 contracted values; operands of group t+1 are read under the MFMAs of group t
          u32x4_t X[2][MB], Wr[2][NB];
          auto load_ops = [&](int t, u32x4_t (&x)[MB], u32x4_t (&w)[NB]) {
#pragma unroll
            for (int a = 0; a < MB; ++a) x[a] = __builtin_bit_cast(u32x4_t, lds_read16(xa + (unsigned)t * 4096u + (unsigned)a * 512u));
#pragma unroll
            for (int b = 0; b < NB; ++b) w[b] = __builtin_bit_cast(u32x4_t, lds_read16(wa + (unsigned)t * 4096u + (unsigned)b * 256u));
          };
          if (CAN_SPLIT && WK > 1) {
            for (int t = wk; t < 4; t += WK) {
              load_ops(t, X[0], Wr[0]);
#pragma unroll
              for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const unsigned d = Wr[0][b][e];
                  Wr[0][b][e] = w_valid ? (__builtin_amdgcn_perm(d, d, w_sel) ^ w_sign) : 0u;
                }
#pragma unroll
              for (int a = 0; a < MB; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b)
                  acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, Wr[0][b]),
                                                                     __builtin_bit_cast(bf16x8_t, X[0][a]), acc[a][b], 0, 0, 0);
            }
          } else {
          load_ops(0, X[0], Wr[0]);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            if (t + 1 < 4) load_ops(t + 1, X[(t + 1) & 1], Wr[(t + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            u32x4_t W[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const unsigned d = Wr[t & 1][b][e];
                W[b][e] = w_valid ? (__builtin_amdgcn_perm(d, d, w_sel) ^ w_sign) : 0u;
              }
#pragma unroll
            for (int a = 0; a < MB; ++a)
#pragma unroll
              for (int b = 0; b < NB; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, W[b]),
                                                                   __builtin_bit_cast(bf16x8_t, X[t & 1][a]), acc[a][b], 0, 0, 0);
          }
          } // (WK == 1)
        } else {
          if (CAN_SPLIT && WK > 1) { // the waves of a block split the chunk (small tiles: no operand pipelining)
            for (int s = wk; s < NS; s += WK) {
              v2f_t x[MB], w[NB];
#pragma unroll
              for (int a = 0; a < MB; ++a) x[a] = lds_read8(xa + (unsigned)s * ROW2 + (unsigned)a * 256u);
#pragma unroll
              for (int b = 0; b < NB; ++b) w[b] = lds_read8(wa + (unsigned)s * ROW2 + (unsigned)b * (NBW * 8u));
#pragma unroll
              for (int a = 0; a < MB; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                  if constexpr (M3) {
                    acc[a][3 * b] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[b].x, x[a].x, acc[a][3 * b], 0, 0, 0);
                    acc[a][3 * b + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[b].y, x[a].y, acc[a][3 * b + 1], 0, 0, 0);
                    acc[a][3 * b + 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[b].x + w[b].y, x[a].x + x[a].y, acc[a][3 * b + 2], 0, 0, 0);
                  } else {
                    const float w0 = w_valid ? (ro ? w[b].y : w[b].x) : 0.f, w1 = w_valid ? (ro ? w[b].x : -w[b].y) : 0.f;
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0, x[a].x, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1, x[a].y, acc[a][b], 0, 0, 0);
                  }
                }
            }
          } else {
          v2f_t X[2][MB], Wr[2][NB];
          auto load_ops = [&](int s, v2f_t (&x)[MB], v2f_t (&w)[NB]) {
#pragma unroll
            for (int a = 0; a < MB; ++a) x[a] = lds_read8(xa + (unsigned)s * ROW2 + (unsigned)a * 256u);
#pragma unroll
            for (int b = 0; b < NB; ++b) w[b] = lds_read8(wa + (unsigned)s * ROW2 + (unsigned)b * (NBW * 8u));
          };
          load_ops(0, X[0], Wr[0]);
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            if (s + 1 < NS) load_ops(s + 1, X[(s + 1) & 1], Wr[(s + 1) & 1]);
            // (left alone, the scheduler sinks these reads to just above their first use and waits for them there:
            //  every step then exposes an LDS round trip; pinned here they fly under this step's MFMAs)
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (M3) {
              float xs[MB], wsum[NB];
#pragma unroll
              for (int a = 0; a < MB; ++a) xs[a] = X[s & 1][a].x + X[s & 1][a].y;
#pragma unroll
              for (int b = 0; b < NB; ++b) wsum[b] = Wr[s & 1][b].x + Wr[s & 1][b].y;
#pragma unroll
              for (int a = 0; a < MB; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b) {
#ifdef ARTN_ABLATE_MFMA
                  asm volatile("" ::"v"(Wr[s & 1][b]), "v"(X[s & 1][a]), "v"(wsum[b]), "v"(xs[a]));
#else
                  acc[a][3 * b] = __builtin_amdgcn_mfma_f32_32x32x2f32(Wr[s & 1][b].x, X[s & 1][a].x, acc[a][3 * b], 0, 0, 0);
                  acc[a][3 * b + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(Wr[s & 1][b].y, X[s & 1][a].y, acc[a][3 * b + 1], 0, 0, 0);
                  acc[a][3 * b + 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(wsum[b], xs[a], acc[a][3 * b + 2], 0, 0, 0);
#endif
                }
              continue;
            }
            float W0[NB], W1[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
              const v2f_t bv = Wr[s & 1][b];
              W0[b] = w_valid ? (ro ? bv.y : bv.x) : 0.f;
              W1[b] = w_valid ? (ro ? bv.x : -bv.y) : 0.f;
            }
#pragma unroll
            for (int a = 0; a < MB; ++a)
#pragma unroll
              for (int b = 0; b < NB; ++b) {
#ifdef ARTN_ABLATE_MFMA
                asm volatile("" ::"v"(W0[b]), "v"(W1[b]), "v"(X[s & 1][a]));
#else
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(W0[b], X[s & 1][a].x, acc[a][b], 0, 0, 0);
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(W1[b], X[s & 1][a].y, acc[a][b], 0, 0, 0);
#endif
              }
          }
          } // (WK == 1)
        }
      }
      const bool flush = last || ((c + 1) & flush_mask) == 0;
      if (!flush) {
        fill((cur ^ 1u) * stage_bytes);
        __syncthreads();
        cur ^= 1u;
        continue;
      }
      // ---- epilogue: accumulators -> C-ordered LDS image -> global, 2^13 elements per pass; a later partial
      //      sum of the same tile is added to what the earlier ones left in C
      const bool accumulate = c > flush_mask;
      char *Cb = reinterpret_cast<char *>(C) + off.c * 8;
      for (int pass = 0; pass < n_pass; ++pass)
       for (int round = 0; round < WK; ++round) { // waves that split a block add their partial blocks one after the other
        __syncthreads(); // chunk buffers / previous pass / previous round are no longer in use
        unsigned lc = lane_c; // (opaque: 64 hoisted scatter addresses per lane would cost the accumulators their registers)
        OPAQUE_V(lc);
        const bool mine = wk == round, add = round > 0;
        if constexpr (M3) {
          if (mine) {
#pragma unroll
            for (int a = 0; a < MB; ++a)
#pragma unroll
              for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                  const unsigned pos = lc ^ c_mb[a] ^ c_nb[b] ^ ((r & 1) ? c_b0 : 0u) ^ ((r & 2) ? c_b1 : 0u) ^ ((r & 4) ? c_q0 : 0u) ^ ((r & 8) ? c_q1 : 0u);
                  const float t1 = acc[a][3 * b][r], t2 = acc[a][3 * b + 1][r], t3 = acc[a][3 * b + 2][r];
                  if ((int)(pos >> ARTN_GEMM_EPI_BITS) == pass) {
                    const unsigned ad = (pos & ((1u << ARTN_GEMM_EPI_BITS) - 1u)) * 8u;
                    v2f_t val = {t1 - t2, t3 - t1 - t2};
                    if (add) val += lds_read8(ad);
                    lds_write8(ad, val);
                  }
                }
          }
        } else if (mine) {
#pragma unroll
          for (int a = 0; a < MB; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
              for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int b0 = 0; b0 < 2; ++b0) {
                  const int n_loc = b0 + 2 * h + 4 * q;
                  const unsigned pos = lc ^ c_mb[a] ^ c_nb[b] ^ (b0 ? c_b0 : 0u) ^ ((q & 1) ? c_q0 : 0u) ^ ((q >> 1) ? c_q1 : 0u);
                  if (n_loc < n_lim && (int)(pos >> ARTN_GEMM_EPI_BITS) == pass) {
                    const unsigned ad = (pos & ((1u << ARTN_GEMM_EPI_BITS) - 1u)) * 8u;
                    v2f_t val = {acc[a][b][4 * q + 2 * b0], acc[a][b][4 * q + 2 * b0 + 1]};
                    if (add) val += lds_read8(ad);
                    lds_write8(ad, val);
                  }
                }
        }
        if (round + 1 < WK) continue;
        __syncthreads();
        char *Cp = Cb + pass * pass_stride;
        // (a swizzle source may be the pass bit itself: its image under the swizzle belongs to every address of the pass)
        unsigned oll = o_ll ^ ((swz_gemm((unsigned)pass << ARTN_GEMM_EPI_BITS, P) & ((1u << ARTN_GEMM_EPI_BITS) - 1u)) * 8u), ogl = o_gl;
        OPAQUE_V(oll);
        OPAQUE_V(ogl);
        for (int i0 = 0; i0 < o_iters; i0 += 4) {
          f32x4 x[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int i = i0 + u;
            if (i < o_iters && o_act) x[u] = lds_read16(oll ^ (swz_gemm((unsigned)i * 512u, P) * 8u));
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int i = i0 + u;
            if (i < o_iters && o_act) {
              long o = 0;
#pragma unroll
              for (int b = 0; b < 4; ++b)
                if ((i >> b) & 1) o += o_gi[b];
              f32x4 *dst = reinterpret_cast<f32x4 *>(Cp + o + ogl);
              // (this thread wrote the same 16 bytes at the previous flush, long ago; read them past the L1)
#ifdef ARTN_ABLATE_MEM
              asm volatile("" ::"v"(x[u]), "v"(dst));
#else
              if (accumulate) x[u] += __builtin_nontemporal_load(dst);
              *dst = x[u];
#endif
            }
          }
        }
      }
      __syncthreads(); // the result image has been read; the chunk buffers are free again
      if (have_next) {
        fill(0u);
        __syncthreads();
      }
      cur = 0;
      if (!last) {
#pragma unroll
        for (int a = 0; a < MB; ++a)
#pragma unroll
          for (int b = 0; b < NB * NACC; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
      }
    }
    off = noff;
  }
}

// ----------------------------------------------------------------------------------------
// artn_k_gemm_deep<MB, NB> -- artn_k_gemm<MB, NB, false, true> (fp32, 3M) with the operand loads TWO chunks ahead.
// The memory-bound steps of the sliced circuits (7-8 contracted bits, a 2^30-element first operand, a few thousand
// elements in the second) are latency-bound with one chunk (20 KiB per workgroup) in flight: a workgroup's chunk period
// is the load latency (2.7 us measured against 0.7 us of MFMA work), 3.1 TB/s chip-wide.  Two register sets alternate
// statically (the chunk loop is unrolled by two): while chunk p is multiplied, chunk p + 1 waits in one set and the
// loads of chunk p + 2 -- of this tile or of the next one -- fly into the other.  Every step issues its loads
// UNCONDITIONALLY (past the end of the last tile a valid chunk is simply loaded again): with conditional loads the
// compiler cannot count what is in flight and falls back to s_waitcnt vmcnt(0) before the LDS fill, which waits for the
// loads just issued as well -- one chunk in flight again.  Same plan, same images, same epilogue as artn_k_gemm; used for
// 2..256 chunks per tile (no partial-sum flush inside a tile), one block column per wave, and images every thread moves
// the same number of 16-byte pieces of (NA of the first operand's, NBI of the second's: compile-time constants; where the
// second operand's image has fewer than 256 pieces the upper threads load a piece again and do not store it).
// M3 = false: the 4M arithmetic of blocks with 16 columns or fewer (chunk steps of the sparse executor: 8 columns).
// DEPTH register sets (2 or 4): chunk c multiplies while chunks c + 1 .. c + DEPTH - 1 wait or fly and chunk c + DEPTH is
// issued into the set chunk c came from.  Measured (A/B in one session): 4 sets are no faster than 2 on any workload
// (n53 26.6 ms, n30 x 10 000 57.2 ms, rand2 13.4 ms either way); only DEPTH = 2 is instantiated.
// ----------------------------------------------------------------------------------------
template <int MB, int NB, int NA, int NBI, bool GATHER = false, bool M3 = true, int DEPTH = 2>
__global__ __launch_bounds__(ARTN_WG_THREADS, 2) void artn_k_gemm_deep(const float2 *__restrict__ A, const float2 *__restrict__ B,
                                                                      float2 *__restrict__ C, const ArtnGemmPlan P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if ((unsigned)(unsigned long)(lds_byte_t *)smem != 0) __builtin_trap();
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5, ro = j & 1;
  const int mt = P.mt, nt = P.nt;
  constexpr int PITCH = ARTN_GEMM_PITCH_LOG2, KCB = ARTN_GEMM_KC;
  constexpr int NS = 1 << (KCB - 1);
  constexpr unsigned a_bytes = 8u << (PITCH + KCB), stage_bytes = 2 * a_bytes;
  constexpr unsigned ROW2 = 16u << PITCH;
  constexpr int NVA = NA, NVB = NBI;
  const int epi_bits = P.tc_bits < ARTN_GEMM_EPI_BITS ? P.tc_bits : ARTN_GEMM_EPI_BITS;
  const unsigned epi_bytes = 8u << epi_bits;
  const unsigned tab_base = 2 * stage_bytes > epi_bytes ? 2 * stage_bytes : epi_bytes;
  long *offtab = reinterpret_cast<long *>(smem + tab_base);
  long *kotab = offtab + 512 + 128;
  if (tid < P.n_ko) {
    kotab[2 * tid] = P.ko_sA[tid] * 8;
    kotab[2 * tid + 1] = P.ko_sB[tid] * 8;
  }
  const OffTab OT = build_offset_table(P, offtab, tid);
  // (the launcher checked: 2^(ta_bits - 1) = 256 NA and 2^(tb_bits - 1) <= 256 NBI 16-byte pieces per image)
  const bool b_act = P.tb_bits - 1 >= 8 || tid < (1 << (P.tb_bits - 1));
  unsigned a_gl = 0, a_ll = 0, b_gl = 0, b_ll = 0;
#pragma unroll
  for (int b = 1; b <= 8; ++b) {
    if ((tid >> (b - 1)) & 1) {
      if (b < P.ta_bits) { a_gl += (unsigned)P.a_stride[b] * 8u; a_ll += (unsigned)P.a_lds[b]; }
      if (b < P.tb_bits) { b_gl += (unsigned)P.b_stride[b] * 8u; b_ll += (unsigned)P.b_lds[b]; }
    }
  }
  long a_gi[2], b_gi[2];
  unsigned a_li[2], b_li[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    a_gi[b] = 9 + b < P.ta_bits ? P.a_stride[9 + b] * 8 : 0;
    a_li[b] = 9 + b < P.ta_bits ? (unsigned)P.a_lds[9 + b] : 0u;
    b_gi[b] = 9 + b < P.tb_bits ? P.b_stride[9 + b] * 8 : 0;
    b_li[b] = 9 + b < P.tb_bits ? (unsigned)P.b_lds[9 + b] : 0u;
  }
  const unsigned a_pair = (unsigned)P.a_lds[0], b_pair = (unsigned)P.b_lds[0];
  const int wn = wave & ((1 << P.wn_log2) - 1), wm = (wave >> P.wn_log2) & ((1 << P.wm_log2) - 1);
  constexpr int NBW = M3 ? 32 : 16, NACC = M3 ? 3 : 1;
  const int n_in = M3 ? j : j >> 1;
  const bool w_valid = M3 || nt >= 4 || n_in < (1 << nt);
  const unsigned lane_x = (((unsigned)h << PITCH) + (unsigned)(wm * MB * 32 + j)) * 8u;
  const unsigned lane_w = a_bytes + (((unsigned)h << PITCH) + (unsigned)(wn * NB * NBW) + (unsigned)(w_valid ? n_in : 0)) * 8u;
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
  const unsigned lane_c = swz_gemm(m_off(wm * MB * 32 + j) | n_off(wn * NB * NBW + (M3 ? 4 : 2) * h), P);
  unsigned c_mb[MB], c_nb[NB];
#pragma unroll
  for (int q = 0; q < MB; ++q) c_mb[q] = swz_gemm(m_off(q * 32), P);
#pragma unroll
  for (int q = 0; q < NB; ++q) c_nb[q] = swz_gemm(n_off(q * NBW), P);
  const unsigned c_b0 = swz_gemm(n_off(1), P), c_b1 = swz_gemm(n_off(2), P), c_q0 = swz_gemm(n_off(M3 ? 8 : 4), P), c_q1 = swz_gemm(n_off(M3 ? 16 : 8), P);
  const int n_lim = nt >= 4 ? 16 : 1 << nt; // 4M: valid columns of a block
  const int o_cb = epi_bits - 1;
  const int o_iters = o_cb > 8 ? 1 << (o_cb - 8) : 1;
  const bool o_act = o_cb >= 8 || tid < (1 << o_cb);
  unsigned o_gl = 0;
#pragma unroll
  for (int b = 1; b <= 8; ++b)
    if (((tid >> (b - 1)) & 1) && b < P.tc_bits) o_gl += (unsigned)P.out_stride[b] * 8u;
  long o_gi[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) o_gi[b] = 9 + b < epi_bits ? P.out_stride[9 + b] * 8 : 0;
  const int n_pass = 1 << (P.tc_bits - epi_bits);
  const long pass_stride = P.tc_bits > epi_bits ? P.out_stride[epi_bits] * 8 : 0;
  const unsigned o_ll = swz_gemm((unsigned)tid * 2u, P) * 8u;

  f32x4 va[DEPTH][NVA], vb[DEPTH][NVB]; // register sets: chunk q lives in set q mod DEPTH
  auto issue = [&](f32x4 (&va)[NVA], f32x4 (&vb)[NVB], const char *__restrict__ Ab, const char *__restrict__ Bb) {
    unsigned agl = a_gl, bgl = b_gl;
    OPAQUE_V(agl);
    OPAQUE_V(bgl);
#pragma unroll
    for (int u = 0; u < NVA; ++u) va[u] = *reinterpret_cast<const f32x4 *>(Ab + ((u & 1) ? a_gi[0] : 0) + ((u & 2) ? a_gi[1] : 0) + agl);
#pragma unroll
    for (int u = 0; u < NVB; ++u) vb[u] = *reinterpret_cast<const f32x4 *>(Bb + ((u & 1) ? b_gi[0] : 0) + ((u & 2) ? b_gi[1] : 0) + bgl);
  };
  auto put = [&](unsigned d, unsigned pair, const f32x4 &v) {
    if (pair == 8u) lds_write16(d, v);
    else { lds_write8(d, v2f_t{v[0], v[1]}); lds_write8(d + pair, v2f_t{v[2], v[3]}); }
  };
  auto fill = [&](const f32x4 (&va)[NVA], const f32x4 (&vb)[NVB], unsigned buf) {
#pragma unroll
    for (int u = 0; u < NVA; ++u) put(buf + a_ll + ((u & 1) ? a_li[0] : 0u) + ((u & 2) ? a_li[1] : 0u), a_pair, va[u]);
    if (b_act) {
#pragma unroll
      for (int u = 0; u < NVB; ++u) put(buf + a_bytes + b_ll + ((u & 1) ? b_li[0] : 0u) + ((u & 2) ? b_li[1] : 0u), b_pair, vb[u]);
    }
  };

  long t0 = blockIdx.x;
  const long G = gridDim.x, n_tiles = P.n_tiles;
  if ((G & 7) == 0) t0 = (long)(blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  const int n_chunks = 1 << P.n_ko; // a multiple of DEPTH, at most 256 (checked by the launcher)
  __syncthreads();
  if (t0 >= n_tiles) return; // (never: the grid is at most n_tiles)
  TileOff off = tile_offsets<GATHER>(P, OT, t0), noff = off;
  const char *Ac = reinterpret_cast<const char *>(A), *Bc = reinterpret_cast<const char *>(B);
  // (ka, kb): byte offsets of the most recently issued chunk inside its tile (Gray-code walk over the looped bits)
  long ka = 0, kb = 0;
  auto advance = [&](int idx_next) { // offsets of chunk idx_next from those of chunk idx_next - 1
    const int bit = __builtin_ctz((unsigned)idx_next);
    const unsigned gn = (unsigned)idx_next ^ ((unsigned)idx_next >> 1);
    const long sa = kotab[2 * bit], sb = kotab[2 * bit + 1];
    if ((gn >> bit) & 1) { ka += sa; kb += sb; } else { ka -= sa; kb -= sb; }
    ka = uniform64(ka);
    kb = uniform64(kb);
  };
  issue(va[0], vb[0], Ac + off.a * 8, Bc + off.b1 * 8);
  fill(va[0], vb[0], 0u);
#pragma unroll
  for (int q = 1; q < DEPTH; ++q) {
    advance(q);
    issue(va[q], vb[q], Ac + off.a * 8 + ka, Bc + off.b1 * 8 + kb);
  }
  __syncthreads();
  unsigned cur = 0;
  for (long tile = t0; tile < n_tiles; tile += G) {
    const bool more_tiles = tile + G < n_tiles;
    if (more_tiles) noff = next_offsets<GATHER>(P, OT, off, tile, G);
    f32x16 acc[MB][NB * NACC];
#pragma unroll
    for (int a = 0; a < MB; ++a)
#pragma unroll
      for (int b = 0; b < NB * NACC; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    // one chunk: issue chunk c + DEPTH into the set that chunk c came from, multiply chunk c, then chunk c + 1 (the next set) -> LDS
#if defined(ARTN_PHASES)
    const long it_ = (tile - t0) / G;
#define GMARK(k)                                                                                                          \
  if (tid == 0 && blockIdx.x < 64 && it_ == 3 && (c == 2 || c == 3)) {                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    artn_phase_buf[blockIdx.x * 16 + (c - 2) * 8 + (k)] = __builtin_amdgcn_s_memtime();                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
  }
#else
#define GMARK(k)
#endif
    auto step = [&](int c, f32x4 (&va_free)[NVA], f32x4 (&vb_free)[NVB], const f32x4 (&va_next)[NVA], const f32x4 (&vb_next)[NVB]) {
      GMARK(0);
      const int c2 = c + DEPTH;
      long base_a = off.a, base_b = off.b1;
      if (c2 < n_chunks) {
        advance(c2);
      } else {
        if (c2 == n_chunks) { ka = 0; kb = 0; } else advance(c2 - n_chunks);
        if (more_tiles) { base_a = noff.a; base_b = noff.b1; } // (else: a chunk of this tile again, never used)
      }
      issue(va_free, vb_free, Ac + base_a * 8 + ka, Bc + base_b * 8 + kb);
      GMARK(1);
      const unsigned base = cur * stage_bytes;
      const unsigned xa = base + lane_x, wa = base + lane_w;
      v2f_t X[2][MB], Wr[2][NB];
      auto load_ops = [&](int s, v2f_t (&x)[MB], v2f_t (&w)[NB]) {
#pragma unroll
        for (int a = 0; a < MB; ++a) x[a] = lds_read8(xa + (unsigned)s * ROW2 + (unsigned)a * 256u);
#pragma unroll
        for (int b = 0; b < NB; ++b) w[b] = lds_read8(wa + (unsigned)s * ROW2 + (unsigned)b * (NBW * 8u));
      };
      load_ops(0, X[0], Wr[0]);
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        if (s + 1 < NS) load_ops(s + 1, X[(s + 1) & 1], Wr[(s + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (M3) {
          float xs[MB], wsum[NB];
#pragma unroll
          for (int a = 0; a < MB; ++a) xs[a] = X[s & 1][a].x + X[s & 1][a].y;
#pragma unroll
          for (int b = 0; b < NB; ++b) wsum[b] = Wr[s & 1][b].x + Wr[s & 1][b].y;
#pragma unroll
          for (int a = 0; a < MB; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b) {
              acc[a][3 * b] = __builtin_amdgcn_mfma_f32_32x32x2f32(Wr[s & 1][b].x, X[s & 1][a].x, acc[a][3 * b], 0, 0, 0);
              acc[a][3 * b + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(Wr[s & 1][b].y, X[s & 1][a].y, acc[a][3 * b + 1], 0, 0, 0);
              acc[a][3 * b + 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(wsum[b], xs[a], acc[a][3 * b + 2], 0, 0, 0);
            }
        } else {
          float W0[NB], W1[NB];
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            const v2f_t bv = Wr[s & 1][b];
            W0[b] = w_valid ? (ro ? bv.y : bv.x) : 0.f;
            W1[b] = w_valid ? (ro ? bv.x : -bv.y) : 0.f;
          }
#pragma unroll
          for (int a = 0; a < MB; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b) {
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(W0[b], X[s & 1][a].x, acc[a][b], 0, 0, 0);
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(W1[b], X[s & 1][a].y, acc[a][b], 0, 0, 0);
            }
        }
      }
      GMARK(2);
      if (c + 1 < n_chunks) {
#if defined(ARTN_PHASES)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NVA + NVB) : "memory");
#endif
        GMARK(3);
        fill(va_next, vb_next, (cur ^ 1u) * stage_bytes);
        GMARK(4);
        __syncthreads();
        GMARK(5);
        cur ^= 1u;
        return;
      }
      // ---- last chunk of the tile: epilogue, then the next tile's first chunk (already in va_next / vb_next) -> buffer 0
      char *Cb = reinterpret_cast<char *>(C) + off.c * 8;
      for (int pass = 0; pass < n_pass; ++pass) {
        __syncthreads();
        unsigned lc = lane_c;
        OPAQUE_V(lc);
        if constexpr (M3) {
#pragma unroll
          for (int a = 0; a < MB; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const unsigned pos = lc ^ c_mb[a] ^ c_nb[b] ^ ((r & 1) ? c_b0 : 0u) ^ ((r & 2) ? c_b1 : 0u) ^ ((r & 4) ? c_q0 : 0u) ^ ((r & 8) ? c_q1 : 0u);
                const float t1 = acc[a][3 * b][r], t2 = acc[a][3 * b + 1][r], t3 = acc[a][3 * b + 2][r];
                if ((int)(pos >> ARTN_GEMM_EPI_BITS) == pass)
                  lds_write8((pos & ((1u << ARTN_GEMM_EPI_BITS) - 1u)) * 8u, v2f_t{t1 - t2, t3 - t1 - t2});
              }
        } else {
#pragma unroll
          for (int a = 0; a < MB; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
              for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int b0 = 0; b0 < 2; ++b0) {
                  const int n_loc = b0 + 2 * h + 4 * q;
                  const unsigned pos = lc ^ c_mb[a] ^ c_nb[b] ^ (b0 ? c_b0 : 0u) ^ ((q & 1) ? c_q0 : 0u) ^ ((q >> 1) ? c_q1 : 0u);
                  if (n_loc < n_lim && (int)(pos >> ARTN_GEMM_EPI_BITS) == pass)
                    lds_write8((pos & ((1u << ARTN_GEMM_EPI_BITS) - 1u)) * 8u, v2f_t{acc[a][b][4 * q + 2 * b0], acc[a][b][4 * q + 2 * b0 + 1]});
                }
        }
        __syncthreads();
        char *Cp = Cb + pass * pass_stride;
        unsigned oll = o_ll ^ ((swz_gemm((unsigned)pass << ARTN_GEMM_EPI_BITS, P) & ((1u << ARTN_GEMM_EPI_BITS) - 1u)) * 8u), ogl = o_gl;
        OPAQUE_V(oll);
        OPAQUE_V(ogl);
        for (int i0 = 0; i0 < o_iters; i0 += 4) {
          f32x4 x[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int i = i0 + u;
            if (i < o_iters && o_act) x[u] = lds_read16(oll ^ (swz_gemm((unsigned)i * 512u, P) * 8u));
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
      fill(va_next, vb_next, 0u); // (after the last tile: the re-loaded chunk, never read)
      __syncthreads();
      cur = 0;
    };
    for (int c = 0; c < n_chunks; c += DEPTH) {
#pragma unroll
      for (int q = 0; q < DEPTH; ++q) step(c + q, va[q], vb[q], va[(q + 1) % DEPTH], vb[(q + 1) % DEPTH]);
    }
    off = noff;
  }
}
