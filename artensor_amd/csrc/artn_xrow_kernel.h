// artn_xrow_kernel.h -- the ROW-STREAMING form of the extent GEMM (round 6; included by artn_kernels.hip).
//
// The memory-bound steps of a network whose bond dimension is not a power of two contract a handful of values into a handful
// of columns on tens of millions of rows: 3^16 rows x (9 contracted values -> 9 columns), 3^15 x (27 -> 27) in the
// bond-dimension-3 benchmark network (reference torch.einsum at /root/reference/artensor/contraction.py:70).  artn_k_xgemm runs
// them at 2.3-3.6 TB/s: its cost there is per TILE (row tables, a mixed-radix decode, two barriers per chunk of 8 contracted
// values, 18 vector + 13 scalar instructions per MFMA: profiles/r06_xgemm_pmc.md).  Here
//   * the small operand (at most 48 contracted values x 48 columns) lives in REGISTERS for the whole kernel, as MFMA fragments;
//   * the rows are cut into blocks of 16, dealt round-robin to the waves of the launch (block b -> wave b mod #waves: at any time
//     the launch works on ONE window of consecutive rows -- a few MB of the operand and of every column of the result);
//   * per block every lane loads its row's contracted values STRAIGHT into the MFMA operand registers
//     (v_mfma_f32_16x16x4_f32: lane (j, g) = row 16 b + j, contracted values 4 s + g), 8 bytes per lane and load; the loads
//     run D blocks AHEAD of the block being multiplied (a ring of D + 1 register sets, the loop unrolled over the ring);
//   * the loop body has NO branch: loads and stores are buffer instructions, rows / columns that do not exist get the
//     offset 0xffffffff and the hardware's range check drops them.  (With `if (row exists)` around global stores hipcc had to
//     assume the stores might not have been issued and waited for vmcnt(0) in front of every block's MFMAs: no load was ever
//     in flight under them -- the first two builds of this kernel ran at the speed of artn_k_xgemm and at half of it.)
//   * no LDS staging, no barrier in the loop; three levels of row-offset tables in LDS (built once per workgroup), the lane's
//     row position advanced by the launch's stride without a division (artn_xrow_advance);
//   * 3M arithmetic as in artn_k_xgemm (T1 = A_re B_re, T2 = A_im B_im, T3 = (A_re + A_im)(B_re + B_im));
//   * the result leaves from the accumulators: register r of lane (j, g) is column 4 g + r (+ 16 per column block) of row 16 b + j.
// Taken by make_xgemm (ArtnXGemmPlan::rowmode) for complex64 steps without batch labels whose contracted and free-B indices are
// at most 48 values each, whose result's fastest label is a free label of the first operand, on 2^15+ rows, tensors below 4 GiB.
// S: MFMA steps (four contracted values each) held in registers, 1..12; NBK: column blocks of 16 (1..3); prefetch distance and
// waves per SIMD: artn_xrow_depth / artn_xrow_waves (artn_xgemm_plan.h).

template <int S, int NBK>
__global__ __launch_bounds__(ARTN_WG_THREADS, artn_xrow_waves(S, NBK)) void artn_k_xrow(const float2 *__restrict__ A, const float2 *__restrict__ B,
                                                                                           float2 *__restrict__ C, const ArtnXGemmPlan P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if ((unsigned)(unsigned long)(lds_byte_t *)smem != 0) __builtin_trap(); // LDS is addressed by raw byte offsets
  constexpr int D = artn_xrow_depth(S);
  constexpr unsigned T0 = 0, T1 = 2048, T2 = 4096; // level tables of the row index: entries of (A byte offset, C byte offset)
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
  typedef int v2i_t __attribute__((ext_vector_type(2)));
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned j = (unsigned)(lane & 15), g = (unsigned)(lane >> 4);
  const unsigned Mtot = (unsigned)P.m.total, Ktot = (unsigned)P.k.total, Ntot = (unsigned)P.n.total;
  const unsigned L0 = (unsigned)P.m.L0, L1 = (unsigned)P.m.L1, L2 = Mtot / (L0 * L1);
  if (tid < P.m.L0) {
    unsigned o0, o1;
    artn_xg_decode(P.m, 0, P.m.n0, (unsigned)tid, o0, o1);
    lds_write4(T0 + 8u * tid, o0 << 3);
    lds_write4(T0 + 8u * tid + 4u, o1 << 3);
  }
  if (tid < P.m.L1) {
    unsigned o0, o1;
    artn_xg_decode(P.m, P.m.n0, P.m.n1, (unsigned)tid, o0, o1);
    lds_write4(T1 + 8u * tid, o0 << 3);
    lds_write4(T1 + 8u * tid + 4u, o1 << 3);
  }
  for (unsigned i = (unsigned)tid; i < L2; i += ARTN_WG_THREADS) {
    unsigned o0, o1;
    artn_xg_decode(P.m, P.m.n0 + P.m.n1, P.m.n_lab - P.m.n0 - P.m.n1, i, o0, o1);
    lds_write4(T2 + 8u * i, o0 << 3);
    lds_write4(T2 + 8u * i + 4u, o1 << 3);
  }
  // ---- per-lane constants.  Steps past the last contracted value load a valid element again and meet zero fragments.
  float wr[NBK][S], wi[NBK][S]; // the small operand: row n = 16 blk + j of its [column][contracted value] form, contracted value 4 s + g
  unsigned ka[S];               // byte offsets of this lane's contracted values in the first operand
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const unsigned k = 4u * s + g;
    unsigned oA, oB;
    artn_xg_decode(P.k, 0, P.k.n_lab, k < Ktot ? k : Ktot - 1u, oA, oB);
    ka[s] = oA << 3;
#pragma unroll
    for (int blk = 0; blk < NBK; ++blk) {
      const unsigned n = 16u * blk + j;
      unsigned nB, nC;
      artn_xg_decode(P.n, 0, P.n.n_lab, n < Ntot ? n : 0u, nB, nC);
      float2 w = float2{0.f, 0.f};
      if (k < Ktot && n < Ntot) w = B[nB + oB];
      wr[blk][s] = w.x;
      wi[blk][s] = w.y;
    }
  }
  unsigned cc[NBK][4]; // byte offset in the result of accumulator register r's column 16 blk + 4 g + r
  bool col_ok[NBK][4];
#pragma unroll
  for (int blk = 0; blk < NBK; ++blk)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const unsigned n = 16u * blk + 4u * g + (unsigned)r;
      unsigned nB, nC;
      artn_xg_decode(P.n, 0, P.n.n_lab, n < Ntot ? n : 0u, nB, nC);
      cc[blk][r] = nC << 3;
      col_ok[blk][r] = n < Ntot;
    }
  __syncthreads(); // level tables are in LDS (the only barrier of the kernel)

  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(A), 0, (int)P.row_bytes_a, 0x00020000);
  const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(C, 0, (int)P.row_bytes_c, 0x00020000);
  // ---- this wave's blocks: b = 4 (wg + it gridDim.x) + wave, it = 0 .. n_it - 1 (the same count for every wave: blocks
  //      past the end load rows that exist -- clamped table reads -- and store nothing)
  const unsigned n_blocks = (Mtot + 15u) >> 4, per_it = 4u * gridDim.x;
  const unsigned n_it = ((n_blocks + per_it - 1u) / per_it + (unsigned)D) / (unsigned)(D + 1) * (unsigned)(D + 1);
  // (workgroups go round-robin to the 8 XCDs: the ones of one XCD take neighbouring blocks -- a cache line that straddles two
  //  blocks is fetched into one L2)
  const unsigned wg = gridDim.x % 8u == 0u ? (blockIdx.x % 8u) * (gridDim.x / 8u) + blockIdx.x / 8u : blockIdx.x;
  unsigned m = 16u * (4u * wg + (unsigned)wave) + j; // row of the block whose loads are issued next
  ArtnXRowPos pos, step;
  artn_xrow_place(m < Mtot ? m : Mtot - 1u, L0, L1, pos);
  artn_xrow_place(16u * per_it, L0, L1, step);
  unsigned rc_ring[D + 1];
  v2f_t x[D + 1][S];
  auto issue = [&](int slot) { // offsets of row m, its S loads into ring slot `slot`, then on to the wave's next block
    const unsigned i2 = pos.i2 < L2 ? pos.i2 : L2 - 1u;
    const unsigned ra = lds_read4(T0 + 8u * pos.i0) + lds_read4(T1 + 8u * pos.i1) + lds_read4(T2 + 8u * i2);
    const unsigned rc = lds_read4(T0 + 8u * pos.i0 + 4u) + lds_read4(T1 + 8u * pos.i1 + 4u) + lds_read4(T2 + 8u * i2 + 4u);
    rc_ring[slot] = m < Mtot ? rc : 0xffffffffu;
#pragma unroll
    for (int s = 0; s < S; ++s) x[slot][s] = __builtin_bit_cast(v2f_t, __builtin_amdgcn_raw_buffer_load_b64(rA, (int)(ra + ka[s]), 0, 0));
    m += 16u * per_it;
    artn_xrow_advance(pos, step, L0, L1);
  };
#pragma unroll
  for (int d = 0; d < D; ++d) issue(d);
  for (unsigned it = 0; it < n_it; it += (unsigned)(D + 1)) {
#pragma unroll
    for (int u = 0; u <= D; ++u) { // block it + u lives in ring slot u; the loads of block it + u + D go to slot (u + D) mod (D + 1)
      issue((u + D) % (D + 1));
      const unsigned rc = rc_ring[u];
      const bool row_ok = rc != 0xffffffffu;
      f32x4_t t1[NBK], t2[NBK], t3[NBK];
#pragma unroll
      for (int blk = 0; blk < NBK; ++blk) t1[blk] = t2[blk] = t3[blk] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const float xs = x[u][s].x + x[u][s].y;
#pragma unroll
        for (int blk = 0; blk < NBK; ++blk) {
          const float ws = wr[blk][s] + wi[blk][s];
          t1[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[blk][s], x[u][s].x, t1[blk], 0, 0, 0);
          t2[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(wi[blk][s], x[u][s].y, t2[blk], 0, 0, 0);
          t3[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(ws, xs, t3[blk], 0, 0, 0);
        }
      }
#pragma unroll
      for (int blk = 0; blk < NBK; ++blk)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const v2f_t val = {t1[blk][r] - t2[blk][r], t3[blk][r] - t1[blk][r] - t2[blk][r]};
          const unsigned off = (row_ok && col_ok[blk][r]) ? rc + cc[blk][r] : 0xffffffffu;
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i_t, val), rC, (int)off, 0, 0); // (no nontemporal hint: a 128-byte segment of an odd-extent tensor straddles two cache lines, the L2 has to merge the halves -- tools/probes/plane_probe: 2.4 TB/s with the hint, 3.6 without)
        }
      __builtin_amdgcn_sched_barrier(0); // one block's accumulators live at a time (hipcc interleaved the unrolled blocks and spilled)
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// artn_k_xrow64 -- the same step with a lane per ROW (round 6, after tools/probes/xrow64_probe.hip: 9 -> 9 on 3^16 rows in 1.27-1.32
// ms = 4.8 TB/s where the 16-row shape above takes 1.70 and artn_k_xgemm 2.30; 27 -> 27 on 3^15 rows 1.38 ms against 1.70-1.76).
// A wave takes SUPERBLOCKS of 64 consecutive rows, lane l = row 64 b + l: a load instruction fetches ONE contracted value of 64
// rows (512 contiguous bytes where the operand's rows are contiguous; the contracted value's offset is uniform: the instruction's
// scalar offset), a store instruction writes one column of 64 rows.  In registers, four loads (contracted values 4 s .. 4 s + 3)
// are a 4 x 4 matrix of 16-lane groups (register = contracted value, group = rows 16 q .. 16 q + 15); its transpose -- two
// v_permlane16_swap and two v_permlane32_swap, the butterfly below -- is the four MFMA operands of the 16-row blocks q = 0..3
// (register = block, group g = contracted value 4 s + g: the layout of v_mfma_f32_16x16x4_f32's second operand).  The same
// butterfly turns the four blocks' results (register = block, group g = column 4 g + r) into four columns of 64 rows.  All
// twelve accumulators x NBK of a superblock live at once (4 blocks x 3 products): the loads of the NEXT superblock are issued
// group by group right behind the MFMAs that free their registers -- one register set, a full superblock in flight.
// Same tables, OOB rules (offset 0xffffffff: dropped) and 3M arithmetic as artn_k_xrow; S <= 8, NBK <= 2 (ArtnXGemmPlan::rowmode 2).

// out[i] group q = in[q] group i (registers i, 16-lane groups q).  v_permlane16_swap: odd groups of the first register <-> even
// groups of the second; v_permlane32_swap: upper 32 lanes of the first <-> lower 32 of the second.  Inline assembly: this hipcc's
// __builtin_amdgcn_permlane16_swap loses the second result (only `extractvalue 0` of the intrinsic's pair reaches the IR); the
// s_nop cover the VALU -> permlane -> VALU / MFMA wait states the hazard recognizer cannot see inside an asm statement.
__device__ __forceinline__ void xrow_butterfly(float &a0, float &a1, float &a2, float &a3) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\ts_nop 1\n\t"
               "v_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\ts_nop 1"
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
}

template <int S, int NBK>
__global__ __launch_bounds__(ARTN_WG_THREADS, artn_xrow64_waves(S, NBK)) void artn_k_xrow64(const float2 *__restrict__ A, const float2 *__restrict__ B,
                                                                                               float2 *__restrict__ C, const ArtnXGemmPlan P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if ((unsigned)(unsigned long)(lds_byte_t *)smem != 0) __builtin_trap(); // LDS is addressed by raw byte offsets
  constexpr unsigned T0 = 0, T1 = 2048, T2 = 4096; // level tables of the row index: entries of (A byte offset, C byte offset)
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
  typedef int v2i_t __attribute__((ext_vector_type(2)));
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned j = (unsigned)(lane & 15), g = (unsigned)(lane >> 4);
  const unsigned Mtot = (unsigned)P.m.total, Ktot = (unsigned)P.k.total, Ntot = (unsigned)P.n.total;
  const unsigned L0 = (unsigned)P.m.L0, L1 = (unsigned)P.m.L1, L2 = Mtot / (L0 * L1);
  if (tid < P.m.L0) {
    unsigned o0, o1;
    artn_xg_decode(P.m, 0, P.m.n0, (unsigned)tid, o0, o1);
    lds_write4(T0 + 8u * tid, o0 << 3);
    lds_write4(T0 + 8u * tid + 4u, o1 << 3);
  }
  if (tid < P.m.L1) {
    unsigned o0, o1;
    artn_xg_decode(P.m, P.m.n0, P.m.n1, (unsigned)tid, o0, o1);
    lds_write4(T1 + 8u * tid, o0 << 3);
    lds_write4(T1 + 8u * tid + 4u, o1 << 3);
  }
  for (unsigned i = (unsigned)tid; i < L2; i += ARTN_WG_THREADS) {
    unsigned o0, o1;
    artn_xg_decode(P.m, P.m.n0 + P.m.n1, P.m.n_lab - P.m.n0 - P.m.n1, i, o0, o1);
    lds_write4(T2 + 8u * i, o0 << 3);
    lds_write4(T2 + 8u * i + 4u, o1 << 3);
  }
  // ---- the small operand as MFMA fragments (lane (j, g): column 16 blk + j, contracted value 4 s + g) and the contracted
  //      values' byte offsets in the first operand: uniform, one scalar register each
  float wr[NBK][S], wi[NBK][S];
  unsigned ka[4 * S];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const unsigned k = 4u * s + g;
    unsigned oA, oB;
    artn_xg_decode(P.k, 0, P.k.n_lab, k < Ktot ? k : Ktot - 1u, oA, oB);
#pragma unroll
    for (int blk = 0; blk < NBK; ++blk) {
      const unsigned n = 16u * blk + j;
      unsigned nB, nC;
      artn_xg_decode(P.n, 0, P.n.n_lab, n < Ntot ? n : 0u, nB, nC);
      float2 w = float2{0.f, 0.f};
      if (k < Ktot && n < Ntot) w = B[nB + oB];
      wr[blk][s] = w.x;
      wi[blk][s] = w.y;
    }
#pragma unroll
    for (int gg = 0; gg < 4; ++gg) {
      const unsigned ku = 4u * s + gg;
      unsigned uA, uB;
      artn_xg_decode(P.k, 0, P.k.n_lab, ku < Ktot ? ku : Ktot - 1u, uA, uB);
      ka[4 * s + gg] = __builtin_amdgcn_readfirstlane(uA << 3);
    }
  }
  // ... and the columns' byte offsets in the result: uniform too (from an LDS table every store waited ~100 cycles for its entry:
  // eight lgkmcnt waits per superblock next to 1 150 cycles of MFMAs; past the scalar registers hipcc keeps them in VGPR lanes)
  unsigned cn[16 * NBK];
#pragma unroll
  for (int n = 0; n < 16 * NBK; ++n) {
    unsigned nB, nC;
    artn_xg_decode(P.n, 0, P.n.n_lab, (unsigned)n < Ntot ? (unsigned)n : 0u, nB, nC);
    cn[n] = __builtin_amdgcn_readfirstlane(nC << 3);
  }
  __syncthreads(); // tables are in LDS (the only barrier of the kernel)

  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(A), 0, (int)P.row_bytes_a, 0x00020000);
  const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(C, 0, (int)P.row_bytes_c, 0x00020000);
  // ---- this wave's superblocks: b = 4 (wg + it gridDim.x) + wave, it = 0 .. n_it - 1
  const unsigned n_sb = (Mtot + 63u) >> 6, per_it = 4u * gridDim.x;
  const unsigned n_it = (n_sb + per_it - 1u) / per_it;
  const unsigned wg = gridDim.x % 8u == 0u ? (blockIdx.x % 8u) * (gridDim.x / 8u) + blockIdx.x / 8u : blockIdx.x;
  unsigned m = 64u * (4u * wg + (unsigned)wave) + (unsigned)lane; // this lane's row of the superblock whose loads are issued next
  ArtnXRowPos pos, step;
  artn_xrow_place(m < Mtot ? m : Mtot - 1u, L0, L1, pos);
  artn_xrow_place(64u * per_it, L0, L1, step);
  auto offsets = [&](unsigned &ra, unsigned &rc) { // byte offsets of row m in the operand and in the result (0xffffffff: no such row)
    const unsigned i2 = pos.i2 < L2 ? pos.i2 : L2 - 1u;
    const unsigned a = lds_read4(T0 + 8u * pos.i0) + lds_read4(T1 + 8u * pos.i1) + lds_read4(T2 + 8u * i2);
    const unsigned c = lds_read4(T0 + 8u * pos.i0 + 4u) + lds_read4(T1 + 8u * pos.i1 + 4u) + lds_read4(T2 + 8u * i2 + 4u);
    ra = m < Mtot ? a : 0xffffffffu;
    rc = m < Mtot ? c : 0xffffffffu;
  };
  float xr[S][4], xi[S][4];
  auto issue = [&](int s, unsigned ra) { // the four loads of contracted values 4 s .. 4 s + 3 (a value that does not exist: zeros)
#pragma unroll
    for (int gg = 0; gg < 4; ++gg) {
      const unsigned off = 4u * s + gg < Ktot ? ra : 0xffffffffu;
      const v2f_t v = __builtin_bit_cast(v2f_t, __builtin_amdgcn_raw_buffer_load_b64(rA, (int)off, (int)ka[4 * s + gg], 0));
      xr[s][gg] = v.x;
      xi[s][gg] = v.y;
    }
  };
  unsigned ra, rc;
  offsets(ra, rc);
#pragma unroll
  for (int s = 0; s < S; ++s) issue(s, ra);
  for (unsigned it = 0; it < n_it; ++it) {
    const unsigned rc_cur = rc;
    m += 64u * per_it;
    artn_xrow_advance(pos, step, L0, L1);
    offsets(ra, rc);
    f32x4_t t1[4][NBK], t2[4][NBK], t3[4][NBK];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int blk = 0; blk < NBK; ++blk) t1[q][blk] = t2[q][blk] = t3[q][blk] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < S; ++s) {
      xrow_butterfly(xr[s][0], xr[s][1], xr[s][2], xr[s][3]); // x[s][q]: rows 16 q .. 16 q + 15, group g = contracted value 4 s + g
      xrow_butterfly(xi[s][0], xi[s][1], xi[s][2], xi[s][3]);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float xs = xr[s][q] + xi[s][q];
#pragma unroll
        for (int blk = 0; blk < NBK; ++blk) {
          const float ws = wr[blk][s] + wi[blk][s];
          t1[q][blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[blk][s], xr[s][q], t1[q][blk], 0, 0, 0);
          t2[q][blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(wi[blk][s], xi[s][q], t2[q][blk], 0, 0, 0);
          t3[q][blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(ws, xs, t3[q][blk], 0, 0, 0);
        }
      }
      issue(s, ra); // the next superblock's loads of this group: its registers are free
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int blk = 0; blk < NBK; ++blk)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float re[4], im[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          re[q] = t1[q][blk][r] - t2[q][blk][r];
          im[q] = t3[q][blk][r] - t1[q][blk][r] - t2[q][blk][r];
        }
        xrow_butterfly(re[0], re[1], re[2], re[3]); // re[gg]: column 16 blk + 4 gg + r of the 64 rows
        xrow_butterfly(im[0], im[1], im[2], im[3]);
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
          const int n = 16 * blk + 4 * gg + r;
          const v2f_t val = {re[gg], im[gg]};
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i_t, val), rC, (int)((unsigned)n < Ntot ? rc_cur : 0xffffffffu), (int)cn[n], 0);
        }
      }
    __builtin_amdgcn_sched_barrier(0);
  }
}
