// artn_bits3_kernel.h -- THREE consecutive steps on the state tensor in one pass over HBM (included by artn_kernels.hip).
//
// Reference loop: /root/reference/artensor/contraction.py:66-70 -- `tensors[i] = einsum(eq, tensors[i], tensors[j])` three
// times on the same tensors[i].  Plan: artn_plan.h make_bits3 (all four tiles 2^12 elements, every result bit of every
// step inside the tile, 3..5 contracted bits per step).  The kernel is artn_k_bits' FULL / NT tile loop -- the same
// helpers, stage code, prefetch of the next tile in registers, refill-before-store order -- with a third stage:
//
//     region 0 (tile t) --stage 1--> region 1 --stage 2--> region 0 --stage 3--> region 1 --> registers --> global
//
// so neither intermediate touches HBM: a triple moves 16 GiB per 2^30-amplitude state where a pair and a single step move 32.
// The result leaves from region 1, so region 0 can be refilled with tile t + 1 right after the barrier that ends stage 3
// (as in the single-stage kernel; the pair kernel has to read its result out of region 0 first).
// NT: non-temporal loads of the A tiles (128-byte input runs: every line is read by exactly one tile)
template <int KB1, int KB2, int KB3, bool M3, bool NT>
__global__ __launch_bounds__(ARTN_WG_THREADS, 2) void artn_k_bits3(const float2 *__restrict__ A, const float2 *__restrict__ B1,
                                                                  const float2 *__restrict__ B2, const float2 *__restrict__ B3,
                                                                  float2 *__restrict__ C, const ArtnBitsPlan P) {
  constexpr int S1 = 1 << (KB1 - 1), S2 = 1 << (KB2 - 1), S3 = 1 << (KB3 - 1);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if ((unsigned)(unsigned long)(lds_byte_t *)smem != 0) __builtin_trap(); // see lds_read8
  constexpr unsigned R0 = 0u, R1 = 8u << ARTN_TILE_BITS_TARGET, regions_end = 16u << ARTN_TILE_BITS_TARGET;
  uint2 *tab1 = reinterpret_cast<uint2 *>(smem + regions_end);
  uint2 *tab2 = tab1 + (1 << (P.st[0].m_bits - 5));
  uint2 *tab3 = tab2 + (1 << (P.st[1].m_bits - 5));
  long *offtab = reinterpret_cast<long *>(tab3 + (1 << (P.st[2].m_bits - 5)));

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5, ro = j & 1;

  unsigned in_lane = 0, out_lane = 0;
#pragma unroll
  for (int b = 1; b <= 8; ++b) {
    if ((tid >> (b - 1)) & 1) {
      in_lane += (unsigned)P.in_stride[b] * 8u;
      out_lane += (unsigned)P.out_stride[b] * 8u;
    }
  }
  long in_hi[4], out_hi[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    in_hi[b] = 9 + b < P.T_in ? P.in_stride[9 + b] * 8 : 0;
    out_hi[b] = 9 + b < P.T_out ? P.out_stride[9 + b] * 8 : 0;
  }
  const unsigned tid16 = tid * 16;

  fill_msub_table(P.st[0], nullptr, tab1, tid);
  fill_msub_table(P.st[1], &P.st[0], tab2, tid);
  fill_msub_table(P.st[2], &P.st[1], tab3, tid);
  const unsigned tab1_a = regions_end, tab2_a = tab1_a + (8u << (P.st[0].m_bits - 5)), tab3_a = tab2_a + (8u << (P.st[1].m_bits - 5));
  const StageConst<KB1> L1 = stage_const<KB1, M3>(P.st[0], nullptr, j, h, wave, tab1_a, R0, R1, 0);
  const StageConst<KB2> L2 = stage_const<KB2, M3>(P.st[1], &P.st[0], j, h, wave, tab2_a, R1, R0, 0);
  const StageConst<KB3> L3 = stage_const<KB3, M3>(P.st[2], &P.st[1], j, h, wave, tab3_a, R0, R1, 0);
  const ArtnStage *zout = &P.st[2];
  const unsigned tid16_out = swz(tid16, zout);
  unsigned out_i_swz[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) out_i_swz[i] = swz(i * (ARTN_WG_THREADS * 16), zout);
  const OffTab OT = build_offset_table(P, offtab, tid);

  // small-operand fragments: registers for the whole kernel (the three small operands do not depend on the tile)
  float W10[S1], W11[S1], W12[1], W20[S2], W21[S2], W22[1], W30[S3], W31[S3], W32[1];
  float WH0[1][S1], WH1[1][S1], WD0[1][S2], WD1[1][S2], WE0[1][S3], WE1[1][S3]; // (unused: the big-K fragment sets of the stage code)
  u32x4_t WS1[1][KB1 >= 3 ? 1 << (KB1 - 3) : 1], WS2[1][KB2 >= 3 ? 1 << (KB2 - 3) : 1], WS3[1][KB3 >= 3 ? 1 << (KB3 - 3) : 1];
  {
    const char *Bb1 = reinterpret_cast<const char *>(B1), *Bb2 = reinterpret_cast<const char *>(B2), *Bb3 = reinterpret_cast<const char *>(B3);
    if constexpr (M3 && KB1 == 5) load_w3<KB1>(W10, W11, W12, Bb1, L1);
    else if constexpr (M3) load_w4m3<KB1>(W10, W11, Bb1, L1);
    else load_w<KB1>(W10, W11, Bb1, L1, ro);
    if constexpr (M3 && KB2 == 5) load_w3<KB2>(W20, W21, W22, Bb2, L2);
    else if constexpr (M3) load_w4m3<KB2>(W20, W21, Bb2, L2);
    else load_w<KB2>(W20, W21, Bb2, L2, ro);
    if constexpr (M3 && KB3 == 5) load_w3<KB3>(W30, W31, W32, Bb3, L3);
    else if constexpr (M3) load_w4m3<KB3>(W30, W31, Bb3, L3);
    else load_w<KB3>(W30, W31, Bb3, L3, ro);
  }

  __syncthreads(); // the sub-tile and tile-offset tables are in LDS

  constexpr int NV = 8;
  f32x4 v[NV];
  TileOff off = {0, 0, 0, 0}, noff = {0, 0, 0, 0};
  long t0 = blockIdx.x;
  const long G = gridDim.x, n_tiles = P.n_tiles;
  if ((G & 7) == 0) t0 = (long)(blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3); // XCD-contiguous tile ranges (see artn_k_bits)
  if (t0 < n_tiles) {
    off = tile_offsets<false>(P, OT, t0);
    copy_in_sync(reinterpret_cast<const char *>(A + off.a), in_hi, in_lane, R0, tid16, 8);
    if (t0 + G < n_tiles) {
      noff = tile_offsets<false>(P, OT, t0 + G);
      issue_loads<NV, NT>(v, reinterpret_cast<const char *>(A + noff.a), in_hi, in_lane);
    }
  }
  __syncthreads();
  const bool stage_prio = (P.stage_prio == 1 || P.stage_prio == 3) && (__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 1);
  const bool copy_prio = P.stage_prio >= 2;
  bool half_pending = false;
  for (long tile = t0; tile < n_tiles; tile += G) {
    const long next = tile + G, next2 = tile + 2 * G;
    TileOff n2off = noff;
    if (next2 < n_tiles) n2off = next_offsets<false>(P, OT, noff, next, G);

    if (stage_prio) __builtin_amdgcn_s_setprio(2);
    run_stage<KB1, false, 0, M3>(L1, W10, W11, W12, h, lane, WH0, WH1, WS1);
    if (half_pending) { // the second half of the next tile's loads (two bursts of 16 KiB half a stage apart: artn_k_bits)
      unsigned li2 = in_lane;
      OPAQUE_V(li2);
      issue_loads<NV, NT, NV / 2, NV>(v, reinterpret_cast<const char *>(A + noff.a), in_hi, li2);
    }
    __syncthreads();
    run_stage<KB2, false, 0, M3>(L2, W20, W21, W22, h, lane, WD0, WD1, WS2);
    __syncthreads();
    run_stage<KB3, false, 0, M3>(L3, W30, W31, W32, h, lane, WE0, WE1, WS3);
    if (stage_prio) __builtin_amdgcn_s_setprio(0);
    __syncthreads();
    if (copy_prio) __builtin_amdgcn_s_setprio(3);

    unsigned lo_in = in_lane, lo_out = out_lane, t16 = tid16, t16o = tid16_out;
    OPAQUE_V(lo_in);
    OPAQUE_V(lo_out);
    OPAQUE_V(t16);
    OPAQUE_V(t16o);
    char *Cbase = reinterpret_cast<char *>(C + off.c);
    // result tile (region 1) -> registers
    f32x4 x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = lds_read16(R1 + (t16o ^ out_i_swz[i]));
    // refill region 0 with the next tile (its loads were issued one iteration ago)
    if (next < n_tiles) store_lds(v, R0, t16);
    // stores of this tile, then the first half of the loads of the tile after next
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      long o = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b)
        if ((i >> b) & 1) o += out_hi[b];
      __builtin_nontemporal_store(x[i], reinterpret_cast<f32x4 *>(Cbase + o + lo_out));
    }
    half_pending = next2 < n_tiles;
    if (next2 < n_tiles) issue_loads<NV, NT, 0, NV / 2>(v, reinterpret_cast<const char *>(A + n2off.a), in_hi, lo_in);
    if (copy_prio) __builtin_amdgcn_s_setprio(0);
    __syncthreads(); // region 0 holds the next tile; every wave is done with the result region
    off = noff;
    noff = n2off;
  }
}
