// artn_pgemm_kernel.h -- packed-operand GEMM of the reduced-precision mode (included by artn_kernels.hip).
//
// BASELINE configs[4] (Sycamore n53 m20 big-batch sampling, bf16-complex MFMA path) is one contraction: 2^30 x 2^29
// elements over 15 contracted bits, 98.9 % of a slice.  artn_k_gemm<.., BF> reads both operands as complex64 (8 bytes
// per element) in 128 x 64 tiles, converts them to bfloat16 on the way into LDS and runs at 15 % of the bf16 MFMA
// peak: every chunk of 32 contracted values moves 48 KiB from L2 for 16 MFMAs per wave (8-9 TB/s of L2 traffic at
// 365 TFLOP/s).  Here
//   * artn_k_pack_bf16 rounds each operand ONCE (round to nearest even, the same values the other kernel feeds its
//     MFMAs) and writes it as 4-byte (re, im) bfloat16 pairs in the exact order of the GEMM's LDS images:
//     [tile][chunk][plane g = kc >> 2 (8)][row][u = kc & 3]  -- 16 bytes per (plane, row);
//   * artn_k_pgemm: one workgroup of 8 waves (4 along m x 2 along n) per CU owns a 256 x 128 tile; a chunk is 32 KiB
//     + 16 KiB of CONTIGUOUS packed data that goes global -> LDS by LDS-DMA (global_load_lds_dwordx4: no registers,
//     no ds_write, no conversion in the loop), three chunk buffers, loads two chunks ahead, one raw s_barrier per
//     chunk with counted vmcnt waits; each wave multiplies 64 x 64 outputs (2 x 4 blocks of v_mfma_f32_32x32x16_bf16,
//     32 MFMAs per chunk).  Half the bytes per element and a quarter of the re-reads: 1/8 of the L2 traffic per FLOP.
// Lane roles of one MFMA (8 complex contracted values kc = 8t + 4h + u, h = lane >> 5):
//   W side (MFMA A operand): row i = lane & 31 = 2 * n_in_block + ro; 16 bytes of the second operand's image
//                            [2t + h][n], turned into (re, -im) / (im, re) per output parity ro
//   X side (MFMA B operand): column j = lane & 31 = m_in_block; 16 bytes of the first operand's image [2t + h][m]
//   accumulator register r of lane (j, h): n_in_block = ((r >> 1) & 1) + 2h + 4(r >> 2), ro = r & 1.

// out[16-byte unit] = 4 consecutive chunk values of one row: units ordered [tile][chunk][plane][row]
__global__ __launch_bounds__(ARTN_WG_THREADS) void artn_k_pack_bf16(const float2 *__restrict__ X, u32x4_t *__restrict__ out,
                                                                   const ArtnPackSide S, const int n_ko, const long n_units) {
  const int rb = S.n_row;
  for (long unit = (long)blockIdx.x * blockDim.x + threadIdx.x; unit < n_units; unit += (long)gridDim.x * blockDim.x) {
    long r = unit, src = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i < rb && ((r >> i) & 1)) src += S.row[i];
    r >>= rb;
#pragma unroll
    for (int q = 2; q < ARTN_PG_KC; ++q)
      if ((r >> (q - 2)) & 1) src += S.kc[q];
    r >>= ARTN_PG_KC - 2;
    for (int q = 0; q < n_ko; ++q)
      if ((r >> q) & 1) src += S.ko[q];
    r >>= n_ko;
    for (int q = 0; q < S.n_to; ++q)
      if ((r >> q) & 1) src += S.to[q];
    u32x4_t v;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float2 e = X[src + ((u & 1) ? S.kc[0] : 0) + ((u & 2) ? S.kc[1] : 0)];
      v[u] = pack_bf16(e.x, e.y);
    }
    out[unit] = v;
  }
}

// LDS-DMA: 64 lanes x 16 bytes land at lds_dst + lane * 16 (wave-uniform destination in M0, per-lane source)
__device__ __forceinline__ void pg_glds16(const void *gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}

#define ARTN_PG_THREADS 512
#define ARTN_PG_A_BYTES (4u << (ARTN_PG_MT + ARTN_PG_KC)) /* 32 KiB */
#define ARTN_PG_B_BYTES (4u << (ARTN_PG_NT + ARTN_PG_KC)) /* 16 KiB */
#define ARTN_PG_STAGE (ARTN_PG_A_BYTES + ARTN_PG_B_BYTES)

// S16 (round 4): the same tile, images, staging and epilogue on v_mfma_f32_16x16x32_bf16 -- 16 complex contracted values
// per MFMA (lane group g = lane >> 4 reads plane 4s + g), blocks of 16 rows (m) x 16 complex columns (n), each wave 4 x 4
// of them with TWO accumulators per block: real and imaginary parts come from two MFMAs on the SAME raw W fragment,
//     re(C) += W . (re x, -im x)        im(C) += W . (im x, re x)
// so the sign / swap of the complex product moves to the X side (8 VALU per X fragment instead of 8 per W fragment and
// output parity), no lane pair reads the same W bytes twice -- 8 ds_read_b128 per 32 MFMAs where the 32x32x16 form needs
// 12: LDS traffic 96 -> 64 bytes per clock and CU -- and the chip holds a higher clock on this MFMA shape under load
// (MI355X_MICROARCH.md, DVFS give-back (7): 1.12-1.15 x the FLOP/s of the 32x32x16 loop at equal cycles).
//   W side (MFMA A operand): row i = lane & 15 = n_in_block; 16 bytes of the second operand's image [4s + g][n], as packed
//   X side (MFMA B operand): column j = lane & 15 = m_in_block; 16 bytes of the first operand's image [4s + g][m], conjugated
//                            (re accumulator) or with re / im swapped (im accumulator)
//   accumulator register r of lane (j, g): row i = 4g + r = n_in_block.
// MODE 2 (round 4): the S16 arithmetic on a RING of six half-chunk slots (16 contracted values: 16 + 8 KiB) instead of
// three chunk buffers: one raw barrier per half-chunk, at which the DMA of the NEXT half-chunk is waited for (counted
// vmcnt: three younger half-chunks stay in flight) -- so the half-chunk after the one being multiplied is already visible
// to every wave and its first operand fragments are read under the MFMAs of this one: no LDS round trip after a barrier.
// DMA budget: a half-chunk is issued five barriers before it is needed by a ds_read (2.5 chunk periods).
template <int MODE>
__global__ __launch_bounds__(ARTN_PG_THREADS, 1) void artn_k_pgemm(const unsigned char *__restrict__ Ap, const unsigned char *__restrict__ Bp,
                                                                   float2 *__restrict__ C, const ArtnPackPlan P) {
  constexpr bool S16 = MODE >= 1, RING = MODE == 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if ((unsigned)(unsigned long)(lds_byte_t *)smem != 0) __builtin_trap(); // LDS is addressed by raw byte offsets
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = S16 ? lane & 15 : lane & 31, h = S16 ? lane >> 4 : lane >> 5, ro = j & 1;
  const int wm = wave & 3, wn = wave >> 2;
  constexpr int MB = S16 ? 4 : 2, NB = 4;
  constexpr unsigned RA = 1u << ARTN_PG_MT, RB = 1u << ARTN_PG_NT;
  const unsigned lane_x = ((unsigned)h * RA + (unsigned)(wm * 64 + j)) * 16u;
  const unsigned lane_w = (RING ? ARTN_PG_A_BYTES / 2 : ARTN_PG_A_BYTES) + ((unsigned)h * RB + (unsigned)(wn * 64 + (S16 ? j : (j >> 1)))) * 16u;
  const unsigned w_sel = ro ? 0x01000302u : 0x03020100u, w_sign = ro ? 0u : 0x80000000u; // (im, re) / (re, -im)
  const int n_chunks = 1 << P.n_ko;

  // ---- epilogue offsets (elements of the C-ordered result image, swizzled; fields are disjoint: XOR)
  auto m_off = [&](int m_local) {
    unsigned o = 0;
#pragma unroll
    for (int i = 0; i < ARTN_PG_MT; ++i)
      if ((m_local >> i) & 1) o |= 1u << P.m_pos[i];
    return o;
  };
  auto n_off = [&](int n_local) {
    unsigned o = 0;
#pragma unroll
    for (int i = 0; i < ARTN_PG_NT; ++i)
      if ((n_local >> i) & 1) o |= 1u << P.n_pos[i];
    return o;
  };
  const unsigned lane_c = swz_gemm(m_off(wm * 64 + j) | n_off(wn * 64 + (S16 ? 4 : 2) * h), P);
  unsigned c_mb[MB], c_nb[NB];
#pragma unroll
  for (int q = 0; q < MB; ++q) c_mb[q] = swz_gemm(m_off(q * (S16 ? 16 : 32)), P);
#pragma unroll
  for (int q = 0; q < NB; ++q) c_nb[q] = swz_gemm(n_off(q * 16), P);
  const unsigned c_r1 = swz_gemm(n_off(1), P), c_r2 = swz_gemm(n_off(2), P); // S16: accumulator register r = column 4g + r
  const unsigned c_b0 = swz_gemm(n_off(1), P), c_q0 = swz_gemm(n_off(4), P), c_q1 = swz_gemm(n_off(8), P);
  constexpr int TC = ARTN_PG_MT + ARTN_PG_NT, EPI = ARTN_PG_EPI_BITS;
  // copy-out: 16-byte unit c = tid + 512 * i of a pass (elements 2c, 2c + 1 of the image)
  unsigned o_gl = 0;
#pragma unroll
  for (int b = 1; b <= 9; ++b)
    if ((tid >> (b - 1)) & 1) o_gl += (unsigned)P.out_stride[b] * 8u;
  long o_gi[3];
#pragma unroll
  for (int b = 0; b < 3; ++b) o_gi[b] = P.out_stride[10 + b] * 8;
  const unsigned o_ll = swz_gemm((unsigned)tid * 2u, P) * 8u;

  // ---- tiles: index bits [no 0..2][mo 0..1][other no][other mo]: the 32 workgroups of an XCD (an XCD-contiguous range of
  //      32 consecutive indices per grid-stride period) form a block of 4 x 8 tiles that share operand stripes in its L2
  const long G = gridDim.x, n_tiles = P.n_tiles;
  long t0 = blockIdx.x;
  if ((G & 7) == 0) t0 = (long)(blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  const int nl = P.n_no < 3 ? P.n_no : 3, ml = P.n_mo < 2 ? P.n_mo : 2;
  const long a_tile_bytes = (long)n_chunks * ARTN_PG_A_BYTES, b_tile_bytes = (long)n_chunks * ARTN_PG_B_BYTES;

  for (long tile = t0; tile < n_tiles; tile += G) {
    // tile index -> (mo, no)
    long r = tile;
    long no = r & ((1L << nl) - 1);
    r >>= nl;
    long mo = r & ((1L << ml) - 1);
    r >>= ml;
    no |= (r & ((1L << (P.n_no - nl)) - 1)) << nl;
    r >>= P.n_no - nl;
    mo |= r << ml;
    long c_off = 0;
    for (int q = 0; q < P.n_mo; ++q)
      if ((mo >> q) & 1) c_off += P.c_mo[q];
    for (int q = 0; q < P.n_no; ++q)
      if ((no >> q) & 1) c_off += P.c_no[q];
    c_off = uniform64(c_off);
    const unsigned char *At = Ap + uniform64(mo * a_tile_bytes), *Bt = Bp + uniform64(no * b_tile_bytes);

    // chunk c -> buffer c % 3: 4 + 2 LDS-DMA instructions per thread (8 KiB per workgroup-wide instruction)
    auto stage = [&](int c, unsigned buf) {
      const unsigned char *ga = At + (long)c * ARTN_PG_A_BYTES + tid * 16, *gb = Bt + (long)c * ARTN_PG_B_BYTES + tid * 16;
      const unsigned dst = buf * ARTN_PG_STAGE + (unsigned)wave * 1024u;
#pragma unroll
      for (int q = 0; q < 4; ++q) pg_glds16(ga + q * 8192, dst + q * 8192u);
#pragma unroll
      for (int q = 0; q < 2; ++q) pg_glds16(gb + q * 8192, dst + ARTN_PG_A_BYTES + q * 8192u);
    };
    typedef float acc_t __attribute__((ext_vector_type(S16 ? 4 : 16)));
    acc_t acc[MB][NB], acci[S16 ? MB : 1][S16 ? NB : 1]; // (S16: acc = real parts, acci = imaginary parts)
#pragma unroll
    for (int a = 0; a < MB; ++a)
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int e = 0; e < (S16 ? 4 : 16); ++e) {
          acc[a][b][e] = 0.f;
          if constexpr (S16) acci[a][b][e] = 0.f;
        }
    // (every wave is past the previous tile's last LDS read: the epilogue ends with a barrier)
    if constexpr (RING) {
      constexpr unsigned SLOT = ARTN_PG_STAGE / 2; // 24 KiB: [A half: 4 planes x 256 rows x 16 B][B half: 4 planes x 128 rows x 16 B]
      const int n_half = 2 * n_chunks;
      // half-chunk q -> slot q % 6: 2 + 1 LDS-DMA instructions per thread (the packed images are contiguous in q)
      auto stage_half = [&](int q, unsigned slot) {
        const unsigned char *ga = At + (long)q * (ARTN_PG_A_BYTES / 2) + tid * 16, *gb = Bt + (long)q * (ARTN_PG_B_BYTES / 2) + tid * 16;
        const unsigned dst = slot * SLOT + (unsigned)wave * 1024u;
        pg_glds16(ga, dst);
        pg_glds16(ga + 8192, dst + 8192u);
        pg_glds16(gb, dst + ARTN_PG_A_BYTES / 2);
      };
      for (int q = 0; q < 5 && q < n_half; ++q) stage_half(q, (unsigned)q);
      u32x4_t Xf[2], Wf[2][NB];
      auto read_x = [&](unsigned slot, int a_) { return __builtin_bit_cast(u32x4_t, lds_read16(slot * SLOT + lane_x + (unsigned)a_ * 256u)); };
      auto read_w = [&](unsigned slot, int b_) { return __builtin_bit_cast(u32x4_t, lds_read16(slot * SLOT + lane_w + (unsigned)b_ * 256u)); };
      unsigned slot = 0;
      for (int qq = 0; qq < n_half; qq += 2) { // (n_half is even: two half-chunks per chunk; unrolled by two so that the
#pragma unroll                                  //  W fragment buffer of a half-chunk is a compile-time choice)
      for (int par = 0; par < 2; ++par) {
        const int q = qq + par;
        // half-chunk q + 1 (and, at q = 0, half-chunk 0) has landed for this wave: at most 3 younger half-chunks in flight ...
        const int younger = (n_half - 1 < q + 4 ? n_half - 1 : q + 4) - (q + 1); // half-chunks issued after q + 1
        if (younger >= 3) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        else if (younger == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ... and after the barrier everybody's has; everybody is also done with half-chunk q - 1, whose slot takes q + 5
        __builtin_amdgcn_s_barrier();
        if (q + 5 < n_half) stage_half(q + 5, slot == 0 ? 5u : slot - 1);
        const unsigned nslot = slot == 5 ? 0u : slot + 1;
        // (after the tile's last half-chunk the "next" fragments are stale bytes of the ring, read and never used:
        //  unconditional reads keep the loop body free of branches)
        if (q == 0) { // the very first fragments of the tile
#pragma unroll
          for (int b = 0; b < NB; ++b) Wf[0][b] = read_w(0u, b);
          Xf[0] = read_x(0u, 0);
        }
#pragma unroll
        for (int a_ = 0; a_ < 4; ++a_) {
          // next X fragment: the next row block of this half-chunk, or the first of the next one (visible since this barrier)
          if (a_ < 3) Xf[(a_ + 1) & 1] = read_x(slot, a_ + 1);
          else Xf[0] = read_x(nslot, 0);
          Wf[par ^ 1][a_] = read_w(nslot, a_); // one of the next half-chunk's W fragments per row block
          __builtin_amdgcn_sched_barrier(0);             // (the reads are ISSUED here, a row block of MFMAs ahead of their use)
          u32x4_t Xc, Xs; // (re, -im) and (im, re) of this row block's X fragment: a dword is (lo: re, hi: im) in bfloat16
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const unsigned d = Xf[a_ & 1][e];
            Xc[e] = d ^ 0x80000000u;
            Xs[e] = __builtin_amdgcn_alignbit(d, d, 16);
          }
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            acc[a_][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, Wf[par][b]), __builtin_bit_cast(bf16x8_t, Xc), acc[a_][b], 0, 0, 0);
            acci[a_][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, Wf[par][b]), __builtin_bit_cast(bf16x8_t, Xs), acci[a_][b], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        slot = nslot;
      }
      }
    }
    if constexpr (!RING) {
    stage(0, 0u);
    if (n_chunks > 1) stage(1, 1u);
    unsigned cur = 0;
    for (int c = 0; c < n_chunks; ++c) {
      // this wave's part of chunk c has landed (6 younger DMA instructions -- chunk c + 1 -- may still be in flight) ...
      if (c + 1 < n_chunks) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // ... and after the barrier everybody's has; everybody is also done reading buffer (c + 2) % 3 (chunk c - 1)
      __builtin_amdgcn_s_barrier();
      if (c + 2 < n_chunks) stage(c + 2, cur >= 1 ? cur - 1 : 2u);
      const unsigned xa = cur * ARTN_PG_STAGE + lane_x, wa = cur * ARTN_PG_STAGE + lane_w;
      if constexpr (S16) {
        // two steps of 16 contracted values (planes 4s .. 4s + 3); step s: 4 row blocks x 4 column blocks x (re, im) = 32
        // MFMAs of 16 cycles.  The W fragments of step s + 1 and the X fragment of the next row block are read under the
        // MFMAs of this one.
        u32x4_t Xf[2], Wf[2][NB];
        auto read_x = [&](int s_, int a_) { return __builtin_bit_cast(u32x4_t, lds_read16(xa + (unsigned)s_ * (4u * RA * 16u) + (unsigned)a_ * 256u)); };
        auto read_w = [&](int s_, int b_) { return __builtin_bit_cast(u32x4_t, lds_read16(wa + (unsigned)s_ * (4u * RB * 16u) + (unsigned)b_ * 256u)); };
#pragma unroll
        for (int b = 0; b < NB; ++b) Wf[0][b] = read_w(0, b);
        Xf[0] = read_x(0, 0);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int s_ = it >> 2, a_ = it & 3;
          if (it + 1 < 8) Xf[(it + 1) & 1] = read_x((it + 1) >> 2, (it + 1) & 3);
          if (s_ == 0) Wf[1][a_] = read_w(1, a_); // one of the next step's W fragments per row block
          __builtin_amdgcn_sched_barrier(0);      // (the reads are ISSUED here, a whole row block of MFMAs ahead of their use)
          u32x4_t Xc, Xs; // (re, -im) and (im, re) of this row block's X fragment: a dword is (lo: re, hi: im) in bfloat16
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const unsigned d = Xf[it & 1][e];
            Xc[e] = d ^ 0x80000000u;
            Xs[e] = __builtin_amdgcn_alignbit(d, d, 16);
          }
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            acc[a_][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, Wf[s_][b]), __builtin_bit_cast(bf16x8_t, Xc),
                                                                acc[a_][b], 0, 0, 0);
            acci[a_][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, Wf[s_][b]), __builtin_bit_cast(bf16x8_t, Xs),
                                                                 acci[a_][b], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        cur = cur == 2 ? 0u : cur + 1;
        continue;
      }
      u32x4_t X[2][MB], Wr[2][NB];
      auto load_ops = [&](int t, u32x4_t (&x)[MB], u32x4_t (&w)[NB]) {
#pragma unroll
        for (int a = 0; a < MB; ++a) x[a] = __builtin_bit_cast(u32x4_t, lds_read16(xa + (unsigned)t * (2u * RA * 16u) + (unsigned)a * 512u));
#pragma unroll
        for (int b = 0; b < NB; ++b) w[b] = __builtin_bit_cast(u32x4_t, lds_read16(wa + (unsigned)t * (2u * RB * 16u) + (unsigned)b * 256u));
      };
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
            W[b][e] = __builtin_amdgcn_perm(d, d, w_sel) ^ w_sign;
          }
        if constexpr (!S16) {
#pragma unroll
          for (int a = 0; a < MB; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, W[b]),
                                                                 __builtin_bit_cast(bf16x8_t, X[t & 1][a]), acc[a][b], 0, 0, 0);
        }
      }
      cur = cur == 2 ? 0u : cur + 1;
    }
    } // !RING
    // ---- epilogue: accumulators -> C-ordered LDS image (2^13 elements per pass) -> 16-byte coalesced stores
    char *Cb = reinterpret_cast<char *>(C) + c_off * 8;
    for (int pass = 0; pass < (1 << (TC - EPI)); ++pass) {
      __syncthreads(); // chunk buffers / the previous pass are no longer in use
      unsigned lc = lane_c;
      OPAQUE_V(lc);
      if constexpr (S16) { // register r of both accumulators: (re, im) of column 4g + r of the block
#pragma unroll
        for (int a = 0; a < MB; ++a)
#pragma unroll
          for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const unsigned pos = lc ^ c_mb[a] ^ c_nb[b] ^ ((r & 1) ? c_r1 : 0u) ^ ((r & 2) ? c_r2 : 0u);
              if ((int)(pos >> EPI) == pass)
                lds_write8((pos & ((1u << EPI) - 1u)) * 8u, v2f_t{acc[a][b][r], acci[a][b][r]});
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
              const unsigned pos = lc ^ c_mb[a] ^ c_nb[b] ^ (b0 ? c_b0 : 0u) ^ ((q & 1) ? c_q0 : 0u) ^ ((q >> 1) ? c_q1 : 0u);
              if ((int)(pos >> EPI) == pass)
                lds_write8((pos & ((1u << EPI) - 1u)) * 8u, v2f_t{acc[a][b][4 * q + 2 * b0], acc[a][b][4 * q + 2 * b0 + 1]});
            }
      }
      __syncthreads();
      unsigned oll = o_ll ^ ((swz_gemm((unsigned)pass << EPI, P) & ((1u << EPI) - 1u)) * 8u), ogl = o_gl;
      OPAQUE_V(oll);
      OPAQUE_V(ogl);
      long po = 0;
#pragma unroll
      for (int b = 0; b < TC - EPI; ++b)
        if ((pass >> b) & 1) po += P.out_stride[EPI + b] * 8;
      f32x4 x[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = lds_read16(oll ^ (swz_gemm((unsigned)i * 1024u, P) * 8u));
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        long o = po;
#pragma unroll
        for (int b = 0; b < 3; ++b)
          if ((i >> b) & 1) o += o_gi[b];
        *reinterpret_cast<f32x4 *>(Cb + o + ogl) = x[i];
      }
    }
    __syncthreads(); // the result image has been read: the chunk buffers are free for the next tile
  }
}


// ----------------------------------------------------------------------------------------
// The same scheme for complex64 ARITHMETIC (ArtnPackPlan::arith = 1): elements stay 8 bytes, so packing saves no bytes --
// what it buys is the contiguous [tile][chunk][k][row] order (LDS-DMA fills, no address arithmetic and no ds_write in the
// loop, conflict-free operand reads) and the 256 x 128 tile.  Chunks of 2^4 contracted values (the same 48 KiB stage),
// 3M arithmetic on v_mfma_f32_32x32x2_f32: each wave 2 x 2 blocks of 32 rows x 32 complex columns with three
// accumulators each, 96 MFMAs per chunk; partial sums leave the registers every 2^12 contracted values (read-add-write
// of C through the epilogue), as in artn_k_gemm.
// ----------------------------------------------------------------------------------------
// out[16-byte unit] = rows 2r, 2r + 1 of one contracted value: units ordered [tile][chunk][k][row pair]
__global__ __launch_bounds__(ARTN_WG_THREADS) void artn_k_pack_f32(const float2 *__restrict__ X, f32x4 *__restrict__ out,
                                                                  const ArtnPackSide S, const int n_ko, const long n_units) {
  const int rb = S.n_row;
  for (long unit = (long)blockIdx.x * blockDim.x + threadIdx.x; unit < n_units; unit += (long)gridDim.x * blockDim.x) {
    long r = unit, src = 0;
#pragma unroll
    for (int i = 1; i < 8; ++i)
      if (i < rb && ((r >> (i - 1)) & 1)) src += S.row[i];
    r >>= rb - 1;
#pragma unroll
    for (int q = 0; q < ARTN_PG_KC - 1; ++q)
      if ((r >> q) & 1) src += S.kc[q];
    r >>= ARTN_PG_KC - 1;
    for (int q = 0; q < n_ko; ++q)
      if ((r >> q) & 1) src += S.ko[q];
    r >>= n_ko;
    for (int q = 0; q < S.n_to; ++q)
      if ((r >> q) & 1) src += S.to[q];
    const float2 e0 = X[src], e1 = X[src + S.row[0]];
    out[unit] = f32x4{e0.x, e0.y, e1.x, e1.y};
  }
}

__global__ __launch_bounds__(ARTN_PG_THREADS, 1) void artn_k_pgemm3m(const unsigned char *__restrict__ Ap, const unsigned char *__restrict__ Bp,
                                                                     float2 *__restrict__ C, const ArtnPackPlan P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if ((unsigned)(unsigned long)(lds_byte_t *)smem != 0) __builtin_trap();
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int wm = wave & 3, wn = wave >> 2;
  constexpr int MB = 2, NB = 2; // blocks of 32 rows x 32 complex columns
  constexpr unsigned RA = 1u << ARTN_PG_MT, RB = 1u << ARTN_PG_NT;
  const unsigned lane_x = ((unsigned)h * RA + (unsigned)(wm * 64 + j)) * 8u;
  const unsigned lane_w = ARTN_PG_A_BYTES + ((unsigned)h * RB + (unsigned)(wn * 64 + j)) * 8u;
  const int n_chunks = 1 << P.n_ko;
  const int seg_len = P.flush_chunks > 0 && P.flush_chunks < n_chunks ? P.flush_chunks : n_chunks;

  auto m_off = [&](int m_local) {
    unsigned o = 0;
#pragma unroll
    for (int i = 0; i < ARTN_PG_MT; ++i)
      if ((m_local >> i) & 1) o |= 1u << P.m_pos[i];
    return o;
  };
  auto n_off = [&](int n_local) {
    unsigned o = 0;
#pragma unroll
    for (int i = 0; i < ARTN_PG_NT; ++i)
      if ((n_local >> i) & 1) o |= 1u << P.n_pos[i];
    return o;
  };
  // accumulator register r of lane (j, h): n_in_block = (r & 3) + 4h + 8(r >> 2)
  const unsigned lane_c = swz_gemm(m_off(wm * 64 + j) | n_off(wn * 64 + 4 * h), P);
  unsigned c_mb[MB], c_nb[NB];
#pragma unroll
  for (int q = 0; q < MB; ++q) c_mb[q] = swz_gemm(m_off(q * 32), P);
#pragma unroll
  for (int q = 0; q < NB; ++q) c_nb[q] = swz_gemm(n_off(q * 32), P);
  const unsigned c_b0 = swz_gemm(n_off(1), P), c_b1 = swz_gemm(n_off(2), P), c_q0 = swz_gemm(n_off(8), P), c_q1 = swz_gemm(n_off(16), P);
  constexpr int TC = ARTN_PG_MT + ARTN_PG_NT, EPI = ARTN_PG_EPI_BITS;
  unsigned o_gl = 0;
#pragma unroll
  for (int b = 1; b <= 9; ++b)
    if ((tid >> (b - 1)) & 1) o_gl += (unsigned)P.out_stride[b] * 8u;
  long o_gi[3];
#pragma unroll
  for (int b = 0; b < 3; ++b) o_gi[b] = P.out_stride[10 + b] * 8;
  const unsigned o_ll = swz_gemm((unsigned)tid * 2u, P) * 8u;

  const long G = gridDim.x, n_tiles = P.n_tiles;
  long t0 = blockIdx.x;
  if ((G & 7) == 0) t0 = (long)(blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  const int nl = P.n_no < 3 ? P.n_no : 3, ml = P.n_mo < 2 ? P.n_mo : 2;
  const long a_tile_bytes = (long)n_chunks * ARTN_PG_A_BYTES, b_tile_bytes = (long)n_chunks * ARTN_PG_B_BYTES;

  for (long tile = t0; tile < n_tiles; tile += G) {
    long r = tile;
    long no = r & ((1L << nl) - 1);
    r >>= nl;
    long mo = r & ((1L << ml) - 1);
    r >>= ml;
    no |= (r & ((1L << (P.n_no - nl)) - 1)) << nl;
    r >>= P.n_no - nl;
    mo |= r << ml;
    long c_off = 0;
    for (int q = 0; q < P.n_mo; ++q)
      if ((mo >> q) & 1) c_off += P.c_mo[q];
    for (int q = 0; q < P.n_no; ++q)
      if ((no >> q) & 1) c_off += P.c_no[q];
    c_off = uniform64(c_off);
    const unsigned char *At = Ap + uniform64(mo * a_tile_bytes), *Bt = Bp + uniform64(no * b_tile_bytes);
    auto stage = [&](int c, unsigned buf) {
      const unsigned char *ga = At + (long)c * ARTN_PG_A_BYTES + tid * 16, *gb = Bt + (long)c * ARTN_PG_B_BYTES + tid * 16;
      const unsigned dst = buf * ARTN_PG_STAGE + (unsigned)wave * 1024u;
#pragma unroll
      for (int q = 0; q < 4; ++q) pg_glds16(ga + q * 8192, dst + q * 8192u);
#pragma unroll
      for (int q = 0; q < 2; ++q) pg_glds16(gb + q * 8192, dst + ARTN_PG_A_BYTES + q * 8192u);
    };
    char *Cb = reinterpret_cast<char *>(C) + c_off * 8;
    for (int seg = 0; seg < n_chunks; seg += seg_len) {
      f32x16 acc[MB][NB * 3];
#pragma unroll
      for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b = 0; b < NB * 3; ++b)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
      const int seg_end = seg + seg_len;
      stage(seg, 0u);
      if (seg + 1 < seg_end) stage(seg + 1, 1u);
      unsigned cur = 0;
      for (int c = seg; c < seg_end; ++c) {
        if (c + 1 < seg_end) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (c + 2 < seg_end) stage(c + 2, cur >= 1 ? cur - 1 : 2u);
        const unsigned xa = cur * ARTN_PG_STAGE + lane_x, wa = cur * ARTN_PG_STAGE + lane_w;
        v2f_t X[2][MB], W[2][NB];
        auto load_ops = [&](int sidx, v2f_t (&x)[MB], v2f_t (&w)[NB]) {
#pragma unroll
          for (int a = 0; a < MB; ++a) x[a] = lds_read8(xa + (unsigned)sidx * (2u * RA * 8u) + (unsigned)a * 256u);
#pragma unroll
          for (int b = 0; b < NB; ++b) w[b] = lds_read8(wa + (unsigned)sidx * (2u * RB * 8u) + (unsigned)b * 256u);
        };
        load_ops(0, X[0], W[0]);
#pragma unroll
        for (int sidx = 0; sidx < 8; ++sidx) {
          if (sidx + 1 < 8) load_ops(sidx + 1, X[(sidx + 1) & 1], W[(sidx + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);
          float xs[MB], wsum[NB];
#pragma unroll
          for (int a = 0; a < MB; ++a) xs[a] = X[sidx & 1][a].x + X[sidx & 1][a].y;
#pragma unroll
          for (int b = 0; b < NB; ++b) wsum[b] = W[sidx & 1][b].x + W[sidx & 1][b].y;
#pragma unroll
          for (int a = 0; a < MB; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b) {
              acc[a][3 * b] = __builtin_amdgcn_mfma_f32_32x32x2f32(W[sidx & 1][b].x, X[sidx & 1][a].x, acc[a][3 * b], 0, 0, 0);
              acc[a][3 * b + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(W[sidx & 1][b].y, X[sidx & 1][a].y, acc[a][3 * b + 1], 0, 0, 0);
              acc[a][3 * b + 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(wsum[b], xs[a], acc[a][3 * b + 2], 0, 0, 0);
            }
        }
        cur = cur == 2 ? 0u : cur + 1;
      }
      // ---- epilogue of this partial sum: T1, T2, T3 -> (T1 - T2, T3 - T1 - T2) -> C-ordered LDS image -> global
      const bool accumulate = seg > 0;
      for (int pass = 0; pass < (1 << (TC - EPI)); ++pass) {
        __syncthreads();
        unsigned lc = lane_c;
        OPAQUE_V(lc);
#pragma unroll
        for (int a = 0; a < MB; ++a)
#pragma unroll
          for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
              const unsigned pos = lc ^ c_mb[a] ^ c_nb[b] ^ ((rr & 1) ? c_b0 : 0u) ^ ((rr & 2) ? c_b1 : 0u) ^ ((rr & 4) ? c_q0 : 0u) ^ ((rr & 8) ? c_q1 : 0u);
              const float t1 = acc[a][3 * b][rr], t2 = acc[a][3 * b + 1][rr], t3 = acc[a][3 * b + 2][rr];
              if ((int)(pos >> EPI) == pass) lds_write8((pos & ((1u << EPI) - 1u)) * 8u, v2f_t{t1 - t2, t3 - t1 - t2});
            }
        __syncthreads();
        unsigned oll = o_ll ^ ((swz_gemm((unsigned)pass << EPI, P) & ((1u << EPI) - 1u)) * 8u), ogl = o_gl;
        OPAQUE_V(oll);
        OPAQUE_V(ogl);
        long po = 0;
#pragma unroll
        for (int b = 0; b < TC - EPI; ++b)
          if ((pass >> b) & 1) po += P.out_stride[EPI + b] * 8;
        f32x4 x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = lds_read16(oll ^ (swz_gemm((unsigned)i * 1024u, P) * 8u));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          long o = po;
#pragma unroll
          for (int b = 0; b < 3; ++b)
            if ((i >> b) & 1) o += o_gi[b];
          f32x4 *dst = reinterpret_cast<f32x4 *>(Cb + o + ogl);
          if (accumulate) x[i] += __builtin_nontemporal_load(dst); // (this thread wrote the same 16 bytes at the previous flush)
          *dst = x[i];
        }
      }
      __syncthreads(); // the result image has been read: the chunk buffers are free again
    }
  }
}
