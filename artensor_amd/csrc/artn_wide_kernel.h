// artn_wide_kernel.h -- the big fused pairs of the state-streaming path with ONE 8-wave workgroup per CU working on ONE tile
// (included by artn_kernels.hip; round 4).
//
// Reference loop: /root/reference/artensor/contraction.py:66-70 -- two consecutive `tensors[i] = einsum(eq, tensors[i],
// tensors[j])` on the same tensors[i].  Plan: the ArtnBitsPlan of make_bits (artn_plan.h) for a fused pair whose three tiles
// are 2^12 elements, unchanged -- the same tile-local bit orders, swizzles and sub-tile tables artn_k_bits runs.
//
// artn_k_bits puts two independent 4-wave workgroups on a CU and hopes that one copies while the other multiplies; measured
// (tools/phases.py) they settle into near-lockstep instead: the matrix pipe is saturated while both are in
// their stages and idle while both copy.  Here the copy phases are taken off the waves instead of being overlapped by luck:
//
//   * all 8 waves (two per SIMD: the occupancy that saturates the pipe) run the stages of the SAME tile, every stage as 3M
//     arithmetic on v_mfma_f32_16x16x4_f32 blocks (16 results n x 16 tile columns m x 4 contracted values): a 6-bit stage of
//     a 2^12 tile is 4 x 4 such blocks x 16 chain steps -- two per wave -- where the 32 x 32 blocks of artn_k_bits give it 4
//     units of work for 4 waves;
//   * tiles arrive by LDS-DMA (global_load_lds_dwordx4) in a ring of three 32 KiB regions, two tiles ahead: no prefetch
//     registers, no refill pass, and 64 KiB per CU in flight;
//   * the result of tile t is read back from its own input region (the second stage writes there) into 4 x 16 bytes per lane
//     and stored from registers; the stores drain under the first stage of tile t + 1.
//
//   iteration t:   result of tile t - 1 -> registers | stage 1: ring[t % 3] -> mid, with the stores of tile t - 1 and the offsets
//                  of tile t + 2 in pieces behind its first chain steps | barrier A | stage 2: mid -> ring[t % 3], with the four
//                  LDS-DMA passes of tile t + 2 -> ring[(t + 2) % 3] (tile t - 1's region, read out before barrier A) behind its
//                  first chain steps | wait for the DMA of tile t + 1 (issued a period ago) | barrier B
//
// Two barriers per tile.  On the pairs artn_k_bits runs with three products it is SLOWER (ARTN_WIDE=1: 56.5 against 53.5 ms on
// n30 -- with one workgroup per CU the operand waits, scatters and bookkeeping of both waves of a SIMD coincide at every barrier;
// DESIGN section 4.1d has the ablation, counter and in-kernel-mark evidence).  It is the default (ARTN_WIDE=2) where artn_k_bits
// cannot: pairs with 11+ contracted bits (5+6, 6+5, 6+6), whose 96-128 fragment registers do not fit next to three 32 x 32
// accumulators -- here a stage needs 3 x 2^(k-2) -- so they ran as four-product chains or as two single steps: the 5+6 pair of
// n30 x 10 000 bitstrings 7.50 -> 6.05 ms.
#define ARTN_WIDE_THREADS 512
#define ARTN_WIDE_REGION (8u << ARTN_TILE_BITS_TARGET) /* 32 KiB */
#define ARTN_WIDE_NBUF 3
// Diagnostic build only (-DARTN_WIDE_MARKS; tools/wide_marks.py): lane 0 of waves 0 and 4 of workgroups 0..63 records the
// shader clock at up to 12 points of tile iterations 20..23
#ifdef ARTN_WIDE_MARKS
__device__ unsigned long long artn_wide_marks[64 * 2 * 4 * 24];
#define WIDE_MARK(k)                                                                                        \
  if (mark_slot >= 0 && mark_it >= 20 && mark_it < 24) {                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    artn_wide_marks[(mark_slot * 4 + (mark_it - 20)) * 24 + (k)] = __builtin_amdgcn_s_memtime();           \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
  }
#else
#define WIDE_MARK(k)
#endif
#ifdef ARTN_ABL_NOBAR
#define WIDE_BAR() asm volatile("" ::: "memory")
#else
#define WIDE_BAR() __syncthreads()
#endif

template <int KB>
struct WideConst {
  static constexpr int NST = 1 << (KB - 2);    // chain steps of 4 contracted values
  static constexpr int NKB = KB > 2 ? KB - 2 : 1;
  unsigned lane_in, lane_out;                  // per-lane byte offsets inside the input / output region (swizzled, no region base)
  long lane_b;                                 // per-lane byte offset into the small operand
  bool w_valid;
  unsigned kxv[NST];                           // chain step s -> byte offset of its contracted bits 2.. in the input region (swizzled);
                                               // held in VECTOR registers: as scalars they alone spill the scalar file
  long kb[NKB];                                // contracted bits 2.. : byte strides in the small operand
  unsigned mo0x, mo0y;                         // (input, output) byte offsets of this wave's first sub-tile: no table read in front of
                                               // the first operand reads of a stage
  unsigned hin, hout;                          // column bit 4 of a 32-column sub-tile (its two 16-column halves)
  unsigned o0, o1;                             // result bits 0, 1 (accumulator register r) in the output region
  int nt_eff, wm, wm_count, msubs, g;
  unsigned msub_tab;                           // LDS byte address of the sub-tile table
};

// lane (jj = l & 15, g = l >> 4): reads x[kc = 4 s + g][m = 16 half + jj], holds w[kc = 4 s + g][n = jj], accumulates
// results n = 4 g + r of column m = jj (v_mfma_f32_16x16x4_f32: A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15],
// D[i = 4 (l >> 4) + r][j = l & 15]); blocks of 16 results are dealt to the waves first, 32-column sub-tiles after
// (NW: waves of the workgroup -- 8 in artn_k_wide, 4 where artn_k_bits borrows the stage: the narrow single steps, ArtnBitsPlan::narrow3)
template <int KB, int NW = 8>
__device__ __forceinline__ WideConst<KB> wide_const(const ArtnStage &st, const ArtnStage *zin, int lane, int wave, unsigned tab) {
  WideConst<KB> L;
  const int jj = lane & 15, g = lane >> 4;
  const int wn_log2 = st.nt > 4 ? st.nt - 4 : 0;
  const int wn = wave & ((1 << wn_log2) - 1);
  L.g = g;
  L.wm = wave >> wn_log2;
  L.wm_count = NW >> wn_log2;
  L.msubs = 1 << (st.m_bits - 5);
  L.nt_eff = st.nt < 4 ? st.nt : 4;
  L.msub_tab = tab;
  unsigned li = ((unsigned)(g & 1) << (st.k_in_pos[0] + 3)) + ((unsigned)(g >> 1) << (st.k_in_pos[1] + 3)), lo = 0;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    if ((jj >> b) & 1) {
      li += 8u << st.lane_in_pos[b];
      lo += 8u << st.lane_out_pos[b];
    }
  }
  if (st.nt > 2) lo += (unsigned)(g & 1) << (st.n_out_pos[2] + 3);
  if (st.nt > 3) lo += (unsigned)(g >> 1) << (st.n_out_pos[3] + 3);
  long lb = (long)(g & 1) * st.k_b_stride[0] * 8 + (long)(g >> 1) * st.k_b_stride[1] * 8;
#pragma unroll
  for (int b = 0; b < 4; ++b)
    if (b < L.nt_eff && ((jj >> b) & 1)) lb += st.n_b_stride[b] * 8;
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    if (b < wn_log2 && ((wn >> b) & 1)) {
      lo += 8u << st.n_out_pos[4 + b];
      lb += st.n_b_stride[4 + b] * 8;
    }
  }
  L.w_valid = (jj >> L.nt_eff) == 0;
  L.lane_in = swz(li, zin);
  L.lane_out = swz(lo, &st);
  L.lane_b = lb;
#pragma unroll
  for (int b = 0; b < WideConst<KB>::NKB; ++b) L.kb[b] = b + 2 < KB ? st.k_b_stride[b + 2] * 8 : 0;
#pragma unroll
  for (int s = 0; s < WideConst<KB>::NST; ++s) {
    unsigned k = 0;
#pragma unroll
    for (int b = 2; b < KB; ++b)
      if ((s >> (b - 2)) & 1) k ^= swz(8u << st.k_in_pos[b], zin);
    L.kxv[s] = k;
    asm volatile("" : "+v"(L.kxv[s]));
  }
  {
    unsigned oi = 0, oo = 0;
#pragma unroll
    for (int b = 0; b < 9; ++b) {
      if (b < st.m_bits - 5 && ((L.wm >> b) & 1)) {
        oi += 8u << st.msub_in_pos[b];
        oo += 8u << st.msub_out_pos[b];
      }
    }
    L.mo0x = swz(oi, zin);
    L.mo0y = swz(oo, &st);
  }
  L.hin = swz(8u << st.lane_in_pos[4], zin);
  L.hout = swz(8u << st.lane_out_pos[4], &st);
  L.o0 = st.nt > 0 ? swz(8u << st.n_out_pos[0], &st) : 0u;
  L.o1 = st.nt > 1 ? swz(8u << st.n_out_pos[1], &st) : 0u;
  return L;
}

// W0[s] / W1[s] / W2[s] = re / im / re + im of w[kc = 4 s + g][n = jj (+ 16 x this wave's block)]
template <int KB>
__device__ __forceinline__ void wide_load_w(float (&W0)[1 << (KB - 2)], float (&W1)[1 << (KB - 2)], float (&W2)[1 << (KB - 2)],
                                            const char *__restrict__ Bbase, const WideConst<KB> &L) {
  constexpr int NST = 1 << (KB - 2);
#pragma unroll
  for (int s = 0; s < NST; ++s) {
    long ko = 0;
#pragma unroll
    for (int b = 2; b < KB; ++b)
      if ((s >> (b - 2)) & 1) ko += L.kb[b - 2];
    float2 bv = make_float2(0.f, 0.f);
    if (L.w_valid) bv = *reinterpret_cast<const float2 *>(Bbase + ko + L.lane_b);
    W0[s] = bv.x;
    W1[s] = bv.y;
  }
#pragma unroll
  for (int s = 0; s < NST; ++s) asm volatile("" : "+v"(W0[s]), "+v"(W1[s])); // (consumed here: see load_w)
#pragma unroll
  for (int s = 0; s < NST; ++s) {
    W2[s] = W0[s] + W1[s];
    asm volatile("" : "+v"(W2[s]));
  }
}

// One stage on this wave's sub-tiles: per 32-column sub-tile two halves of 16 columns x 2^(KB - 2) chain steps x 3 products
// (T1 = x_re w_re, T2 = x_im w_im, T3 = (x_re + x_im)(w_re + w_im); re = T1 - T2, im = T3 - T1 - T2): six independent
// accumulators per sub-tile, so consecutive MFMAs never wait on each other.
template <int KB>
struct WideStage {
  static constexpr int NST = 1 << (KB - 2);
  static constexpr int U = NST < 4 ? NST : 4;  // chain steps per unit: a unit is U steps of one 16-column half
  static constexpr int NC = NST / U;           // units per half
  static constexpr int NUNITS = 2 * NC;        // units per sub-tile (even: the ping-pong buffers keep their parity across sub-tiles)
  const WideConst<KB> &L;
  const float (&W0)[NST];
  const float (&W1)[NST];
  const float (&W2)[NST];
  unsigned in_base, out_base;
  int mark_slot, mark_it, mark_k; // (diagnostics)
  typedef float f32x4_t __attribute__((ext_vector_type(4)));

  // operands of unit u (half u / NC, steps U (u % NC) ...) of the sub-tile whose halves start at li[0], li[1]
  __device__ __forceinline__ void load_unit(int u, v2f_t (&b)[U], const unsigned (&li)[2]) const {
    const int hf = u / NC, c = u % NC;
#pragma unroll
    for (int s = 0; s < U; ++s) {
#ifdef ARTN_ABL_NOLOADX
      b[s] = v2f_t{__builtin_bit_cast(float, li[hf] ^ L.kxv[c * U + s]), 1.0f};
#else
      b[s] = lds_read8(li[hf] ^ L.kxv[c * U + s]);
#endif
    }
  }
  // the three MFMAs of chain step s of unit u
  __device__ __forceinline__ void mfma_step(int u, int s, f32x4_t (&t)[2][3], const v2f_t (&b)[U]) const {
    const int hf = u / NC, c = u % NC;
    const float xs = b[s].x + b[s].y;
#ifdef ARTN_ABLATE_MFMA
    asm volatile("" ::"v"(b[s].x), "v"(b[s].y), "v"(xs), "v"(W0[c * U + s]), "v"(W1[c * U + s]), "v"(W2[c * U + s]));
#else
    t[hf][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(W0[c * U + s], b[s].x, t[hf][0], 0, 0, 0);
    t[hf][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(W1[c * U + s], b[s].y, t[hf][1], 0, 0, 0);
    t[hf][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(W2[c * U + s], xs, t[hf][2], 0, 0, 0);
#endif
  }
  // result row r (of 4) of half hf: one ds_write_b64
  __device__ __forceinline__ void scatter_row(int hf, int r, const f32x4_t (&t)[2][3], unsigned lo) const {
#ifdef ARTN_ABL_NOSCATTER
    asm volatile("" ::"v"(t[hf][0][r]), "v"(t[hf][1][r]), "v"(t[hf][2][r]), "v"(lo));
    return;
#endif
    const unsigned o = lo ^ (hf ? L.hout : 0u) ^ ((r & 1) ? L.o0 : 0u) ^ ((r & 2) ? L.o1 : 0u);
    if (L.nt_eff == 4 || ((4 * L.g + r) >> L.nt_eff) == 0)
      lds_write8(o, v2f_t{t[hf][0][r] - t[hf][1][r], t[hf][2][r] - t[hf][0][r] - t[hf][1][r]});
  }
  // All 8 waves of the workgroup pass the same barriers, so whatever is NOT an MFMA and is done by every wave at the same
  // point of a stage leaves the matrix pipe idle on every SIMD at once (in-kernel marks, tools/wide_marks.py: the tile loop's
  // own work -- 4 global stores, the next tile's offsets, 4 LDS-DMA issues -- cost 900-1 700 cycles of a 14 600-cycle period
  // as a block behind the first unit).  An MFMA holds the vector issue port for 8 of its 32 cycles, so that work is cut into
  // pieces of a few instructions and each piece is pinned (sched_barrier) behind one chain step (3 MFMAs):
  //   * the operands of unit u + 1 -- of this sub-tile or the first of the next -- are read in front of unit u;
  //   * the result rows of half 0 (final once its units are done) are written under the steps of half 1;
  //   * `fill(p)`, p = 0 .. Fill::N - 1, runs behind step p of the wave's first sub-tile.
  // (Measured alternatives: the pieces as one block behind the first unit +9 % on the n30 pairs; the two waves of a SIMD doing
  //  them at opposite ends of the stage, so that each one's block runs while its partner has the pipe, +4 %.)
  template <bool FIRST, typename Fill>
  __device__ __forceinline__ void sub_tile(f32x4_t (&t)[2][3], v2f_t (&buf)[2][U], const unsigned (&li)[2], bool more,
                                           const unsigned (&li_n)[2], unsigned lo, Fill &fill) const {
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int q = 0; q < 3; ++q) t[hf][q] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    constexpr int ROWS_PER_STEP = (4 + NST - 1) / NST; // half 0's four rows over the NST steps of half 1
#pragma unroll
    for (int u = 0; u < NUNITS; ++u) {
      if (u + 1 < NUNITS) load_unit(u + 1, buf[(u + 1) & 1], li);
      else if (more) load_unit(0, buf[0], li_n);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < U; ++s) {
        mfma_step(u, s, t, buf[u & 1]);
        __builtin_amdgcn_sched_barrier(0);
        const int slot = u * U + s;
        if (FIRST && slot < Fill::N) fill(slot);
        if (u >= NC) {
          const int k = (u - NC) * U + s; // step of half 1
#pragma unroll
          for (int r = k * ROWS_PER_STEP; r < (k + 1) * ROWS_PER_STEP && r < 4; ++r) scatter_row(0, r, t, lo);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (FIRST && mark_k == 1) { WIDE_MARK(8 + u); }
    }
    if (FIRST) {
#pragma unroll
      for (int pp = NUNITS * U; pp < Fill::N; ++pp) fill(pp); // (stages with fewer steps than pieces)
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) scatter_row(1, r, t, lo);
  }
  template <typename Fill>
  __device__ __forceinline__ void run(Fill &fill) const {
    int msub = L.wm;
    if (msub >= L.msubs) {
#pragma unroll
      for (int pp = 0; pp < Fill::N; ++pp) fill(pp);
      return;
    }
    const unsigned lin = L.lane_in ^ in_base, lout = L.lane_out ^ out_base;
    u2_t mo = u2_t{L.mo0x, L.mo0y};
    v2f_t buf[2][U];
    f32x4_t t[2][3];
    unsigned li[2] = {lin ^ mo.x, lin ^ mo.x ^ L.hin};
    load_unit(0, buf[0], li);
    {
      const int nmsub = msub + L.wm_count;
      const bool more = nmsub < L.msubs;
      u2_t mo_n = mo;
      if (more) mo_n = lds_read_u2(L.msub_tab + nmsub * 8);
      const unsigned li_n[2] = {lin ^ mo_n.x, lin ^ mo_n.x ^ L.hin};
      sub_tile<true>(t, buf, li, more, li_n, lout ^ mo.y, fill);
      WIDE_MARK(mark_k);
      if (!more) return;
      msub = nmsub;
      mo = mo_n;
      li[0] = li_n[0];
      li[1] = li_n[1];
    }
    for (;;) {
      const int nmsub = msub + L.wm_count;
      const bool more = nmsub < L.msubs;
      u2_t mo_n = mo;
      if (more) mo_n = lds_read_u2(L.msub_tab + nmsub * 8);
      const unsigned li_n[2] = {lin ^ mo_n.x, lin ^ mo_n.x ^ L.hin};
      sub_tile<false>(t, buf, li, more, li_n, lout ^ mo.y, fill);
      if (!more) return;
      msub = nmsub;
      mo = mo_n;
      li[0] = li_n[0];
      li[1] = li_n[1];
    }
  }
};

struct WideNoFill { // (a stage with nothing to interleave)
  static constexpr int N = 0;
  __device__ __forceinline__ void operator()(int) const {}
};

// LDS-DMA: 64 lanes x 16 bytes land at lds_dst + lane * 16 (wave-uniform destination in M0, per-lane source)
__device__ __forceinline__ void wide_glds16(const void *gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}
__device__ __forceinline__ void wide_dma_tile(const char *__restrict__ Abase, const long (&hi)[2], unsigned lane_off, unsigned lds_wave) {
#ifdef ARTN_ABLATE_MEM
  asm volatile("" ::"s"(Abase), "v"(lane_off), "s"(lds_wave));
  return;
#endif
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long o = ((i & 1) ? hi[0] : 0) + ((i & 2) ? hi[1] : 0);
    wide_glds16(Abase + o + lane_off, lds_wave + (unsigned)i * (ARTN_WIDE_THREADS * 16u));
  }
}

// The tile loop's own work, in pieces for WideStage::run (stage 1 / stage 2)
      // pieces 0..3: the four 16-byte stores of tile t - 1; piece 4: the offsets of tile t + 2
      struct WideFill1 {
        static constexpr int N = 5;
        const bool pending, has2;
        const f32x4 (&x)[4];
        char *Cbase;
        const long (&out_hi)[2];
        unsigned out_lane;
        TileOff &n2off;
        const ArtnBitsPlan &P;
        const OffTab &OT;
        const TileOff &noff;
        long next, G;
        __device__ __forceinline__ void operator()(int p) const {
          if (p < 4) {
            if (pending) {
              const long o = ((p & 1) ? out_hi[0] : 0) + ((p & 2) ? out_hi[1] : 0);
#ifdef ARTN_ABLATE_MEM
              asm volatile("" ::"v"(x[p]), "s"(Cbase), "v"(out_lane));
#else
              __builtin_nontemporal_store(x[p], reinterpret_cast<f32x4 *>(Cbase + o + out_lane));
#endif
            }
          } else if (has2) {
            n2off = next_offsets<false>(P, OT, noff, next, G);
          }
        }
};
      // pieces 0..3: the four LDS-DMA passes of tile t + 2 into the region tile t - 1 has left
      struct WideFill2 {
        static constexpr int N = 4;
        const bool has2;
        const char *Abase;
        const long (&in_hi)[2];
        unsigned in_lane, lds_dst;
        __device__ __forceinline__ void operator()(int p) const {
#ifndef ARTN_ABLATE_MEM
          if (has2) {
            const long o = ((p & 1) ? in_hi[0] : 0) + ((p & 2) ? in_hi[1] : 0);
            wide_glds16(Abase + o + in_lane, lds_dst + (unsigned)p * (ARTN_WIDE_THREADS * 16u));
          }
#endif
        }
};

template <int KB1, int KB2>
__global__ __launch_bounds__(ARTN_WIDE_THREADS, 1) void artn_k_wide(const float2 *__restrict__ A, const float2 *__restrict__ B1,
                                                                   const float2 *__restrict__ B2, float2 *__restrict__ C,
                                                                   const ArtnBitsPlan P) {
  constexpr int N1 = 1 << (KB1 - 2), N2 = 1 << (KB2 - 2);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if ((unsigned)(unsigned long)(lds_byte_t *)smem != 0) __builtin_trap(); // see lds_read8
  constexpr unsigned RM = ARTN_WIDE_NBUF * ARTN_WIDE_REGION, regions_end = (ARTN_WIDE_NBUF + 1) * ARTN_WIDE_REGION;
  uint2 *tab1 = reinterpret_cast<uint2 *>(smem + regions_end);
  uint2 *tab2 = tab1 + (1 << (P.st[0].m_bits - 5));
  long *offtab = reinterpret_cast<long *>(tab2 + (1 << (P.st[1].m_bits - 5)));

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // copies: thread handles the 16-byte chunks c = tid + 512 i, i < 4 (tile-local elements 2c, 2c + 1)
  unsigned in_lane = 0, out_lane = 0;
#pragma unroll
  for (int b = 1; b <= 9; ++b) {
    if ((tid >> (b - 1)) & 1) {
      in_lane += (unsigned)P.in_stride[b] * 8u;
      out_lane += (unsigned)P.out_stride[b] * 8u;
    }
  }
  const long in_hi[2] = {P.in_stride[10] * 8, P.in_stride[11] * 8}, out_hi[2] = {P.out_stride[10] * 8, P.out_stride[11] * 8};
  const unsigned lds_wave = (unsigned)wave * 1024u; // LDS-DMA: the wave's 64 x 16 bytes of pass i land at region + 8192 i + 1024 wave

  fill_msub_table(P.st[0], nullptr, tab1, tid);
  fill_msub_table(P.st[1], &P.st[0], tab2, tid);
  const unsigned tab1_a = regions_end, tab2_a = tab1_a + (8u << (P.st[0].m_bits - 5));
  const WideConst<KB1> L1 = wide_const<KB1>(P.st[0], nullptr, lane, wave, tab1_a);
  const WideConst<KB2> L2 = wide_const<KB2>(P.st[1], &P.st[0], lane, wave, tab2_a);
  const ArtnStage *zout = &P.st[1];
  const unsigned tid16_out = swz((unsigned)tid * 16u, zout);
  unsigned out_i_swz[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) out_i_swz[i] = swz((unsigned)i * (ARTN_WIDE_THREADS * 16u), zout);
  const OffTab OT = build_offset_table(P, offtab, tid);
  float W10[N1], W11[N1], W12[N1], W20[N2], W21[N2], W22[N2];
  long prev_b1 = -1, prev_b2 = -1;
  __syncthreads(); // tables are in LDS

  long t0 = blockIdx.x, G = gridDim.x, n_tiles = P.n_tiles;
  if ((G & 7) == 0) t0 = (long)(blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3); // (one XCD: a contiguous eighth of every period)
  if (P.blocked) {
    const long per = (P.n_tiles + gridDim.x - 1) / gridDim.x;
    t0 = per * blockIdx.x;
    G = 1;
    n_tiles = t0 + per < P.n_tiles ? t0 + per : P.n_tiles;
  }
  if (t0 >= n_tiles) return;
  TileOff off = tile_offsets<false>(P, OT, t0), noff = off;
  wide_dma_tile(reinterpret_cast<const char *>(A + off.a), in_hi, in_lane, 0u * ARTN_WIDE_REGION + lds_wave);
  if (t0 + G < n_tiles) {
    noff = tile_offsets<false>(P, OT, t0 + G);
    wide_dma_tile(reinterpret_cast<const char *>(A + noff.a), in_hi, in_lane, 1u * ARTN_WIDE_REGION + lds_wave);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads(); // tile t0 is in ring[0]

  unsigned cur = 0u, prv = 2u * ARTN_WIDE_REGION; // ring regions of tile t and of tile t - 1 (= of tile t + 2)
  bool pending = false;                           // the result of tile t - 1 still sits in `prv`
  long c_prev = 0;
  auto read_result = [&](f32x4 (&x)[4], unsigned region) {
#ifdef ARTN_ABL_NODRAIN
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = f32x4{1.f, 2.f, 3.f, (float)region};
#else
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = lds_read16(region + (tid16_out ^ out_i_swz[i]));
#endif
  };
  auto store_result = [&](const f32x4 (&x)[4], long c_off) {
    char *Cbase = reinterpret_cast<char *>(C + c_off);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long o = ((i & 1) ? out_hi[0] : 0) + ((i & 2) ? out_hi[1] : 0);
#ifdef ARTN_ABLATE_MEM
      asm volatile("" ::"v"(x[i]), "s"(Cbase), "v"(out_lane));
#else
      __builtin_nontemporal_store(x[i], reinterpret_cast<f32x4 *>(Cbase + o + out_lane));
#endif
    }
  };
  const int mark_slot = (lane == 0 && (wave & 3) == 0 && blockIdx.x < 64) ? (int)blockIdx.x * 2 + (wave >> 2) : -1;
  int mark_it = -1;
  for (long tile = t0; tile < n_tiles; tile += G) {
    ++mark_it;
    WIDE_MARK(0);
    if (off.b1 != prev_b1) {
      prev_b1 = off.b1;
      wide_load_w<KB1>(W10, W11, W12, reinterpret_cast<const char *>(B1 + off.b1), L1);
    }
    if (off.b2 != prev_b2) {
      prev_b2 = off.b2;
      wide_load_w<KB2>(W20, W21, W22, reinterpret_cast<const char *>(B2 + off.b2), L2);
    }
    const long next = tile + G, next2 = tile + 2 * G;
    TileOff n2off = noff;
    // the result of tile t - 1 leaves its region under the first stage of tile t: read here, stored from the hook
    f32x4 x[4];
    if (pending) read_result(x, prv);
    {
      WideFill1 fill1{pending, next2 < n_tiles, x, reinterpret_cast<char *>(C + c_prev), out_hi, out_lane, n2off, P, OT, noff, next, G};
      WideStage<KB1> s1{L1, W10, W11, W12, cur, RM, mark_slot, mark_it, 1};
      s1.run(fill1);
    }
    WIDE_MARK(2);
    WIDE_BAR(); // A: mid is complete; every wave has read tile t - 1's result out of `prv`
    WIDE_MARK(3);
    {
      WideFill2 fill2{next2 < n_tiles, reinterpret_cast<const char *>(A + n2off.a), in_hi, in_lane, prv + lds_wave};
      WideStage<KB2> s2{L2, W20, W21, W22, RM, cur, mark_slot, mark_it, 4};
      s2.run(fill2);
    }
    WIDE_MARK(5);
    // the DMA of tile t + 1 (issued one iteration ago; younger: the stores of tile t - 1 and the DMA of tile t + 2)
    if (next < n_tiles) {
      const int younger = (pending ? 4 : 0) + (next2 < n_tiles ? 4 : 0);
      if (younger == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (younger == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    WIDE_MARK(6);
    WIDE_BAR(); // B: the result of tile t is complete in `cur`; tile t + 1 is in its region for every wave
    WIDE_MARK(7);
    pending = true;
    c_prev = off.c;
    prv = cur;
    cur = cur + ARTN_WIDE_REGION == RM ? 0u : cur + ARTN_WIDE_REGION;
    off = noff;
    noff = n2off;
  }
  {
    f32x4 x[4];
    read_result(x, prv);
    store_result(x, c_prev);
  }
}

#ifdef ARTN_WIDE_MARKS
extern "C" int artn_debug_read_wide_marks(unsigned long long *host) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(artn_wide_marks), sizeof(artn_wide_marks)) == hipSuccess ? 0 : -1;
}
#endif
