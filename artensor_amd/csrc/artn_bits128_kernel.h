// artn_bits128_kernel.h -- complex128 state-streaming kernel (included by artn_kernels.hip).
//
// The reference takes any dtype (`TensorNetworkSimulation.contraction(dtype=...)`,
// /root/reference/artensor/simulation.py:90).  Until round 3 every complex128 step was one pass of the two-operand GEMM
// artn_k_gemm128: 28 passes over a 16 GiB state for the n30 contraction, each of them HBM-bound (34 GB per pass against
// 3-6 ms of f64 MFMA work).  This is artn_k_bits for 16-byte elements: the same plans (ArtnBitsPlan from make_bits, tiles
// of 2^11 elements = 32 KiB per LDS region, two workgroups per CU), the same software pipeline over tiles, one or two
// fused stages on v_mfma_f64_16x16x4_f64 with the small operands in registers -- a fused pair reads and writes the state
// once instead of twice.
//
//   copy phases : thread t moves elements t, t + 256, ... of a tile (one 16-byte element per lane and chunk)
//   stage       : out[n][m] = sum_kc w[kc][n] x[kc][m] over the tile-local bit positions of the plan.  One MFMA block =
//                 8 complex columns n (16 real rows: i = 2 n_in + ro) x 16 m (lane & 15) x 2 contracted values
//                 (lane group g = lane >> 4 carries kk = g = 2 kcl + p: contracted value kc = 2 s + kcl, p = 0 re / 1 im
//                 of the x element); a chain of 2^(k-1) MFMAs per block; the blocks of 8 columns (at most 4: nt <= 5) are
//                 dealt to the waves, the 16-m sub-tiles to the remaining waves; a wave runs two sub-tiles interleaved
//                 (two independent accumulation chains).
//   W operand   : row (n_in, ro) and group (kcl, p) take value (ro, p): (0,0) re  (0,1) -im  (1,0) im  (1,1) re of
//                 w[kc][n] -- one f64 per lane and chain step, loaded once per workgroup (or when the tile's small
//                 operand changes)
//   accumulator : register r of lane (m, g) is row i = g + 4 r: component g & 1 of column n_in = (g >> 1) + 2 r
//                 (the f64 C/D map of the guide, as in artn_gemm128_kernel.h)
//
// LDS swizzle: ArtnStage::swz_* in units of 16-byte elements (positions 0..3 = the 256 bytes one half-wave of
// ds_read_b64 covers).

// XOR swizzle of an LDS region of 16-byte elements, applied to byte offsets (see swz())
__device__ __forceinline__ unsigned swz16(unsigned byte_off, const ArtnStage *z) {
  if (z) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < z->swz_n && ((byte_off >> (z->swz_src[i] + 4)) & 1)) byte_off ^= 16u << z->swz_dst[i];
  }
  return byte_off;
}

template <int KB>
struct Stage128 {
  unsigned lane_in, lane_out;    // per-lane LDS byte offsets (region base, swizzle and the re/im half folded in)
  long lane_b;                   // per-lane byte offset into the small operand
  double w_sign;                 // +1 / -1, 0 for rows beyond the stage's columns
  unsigned kin[KB > 1 ? KB : 2]; // byte offset of contracted bit b in the LDS input tile (swizzled)
  long kb[KB > 1 ? KB : 2];      // byte stride of contracted bit b in the small operand
  unsigned o1, o2;               // output byte offsets of column bits 1, 2 (accumulator register r = bit 1 + 2 bit 2)
  int n_lim;                     // valid columns of a block: min(8, 2^nt)
  int wm, wm_count, msubs;
  unsigned tab;                  // LDS byte address of the sub-tile table: sub-tile -> (input, output) byte offsets
};

template <int KB>
__device__ __forceinline__ Stage128<KB> stage_const128(const ArtnStage &st, const ArtnStage *zin, int j, int g, int wave, unsigned tab,
                                                       unsigned in_base, unsigned out_base) {
  Stage128<KB> L;
  const int ro = j & 1, n_in = j >> 1, p = g & 1, kcl = g >> 1;
  const int wn = wave & ((1 << st.wn_log2) - 1);
  L.wm = wave >> st.wn_log2;
  L.wm_count = 4 >> st.wn_log2;
  L.msubs = 1 << (st.m_bits - 4);
  L.tab = tab;
  const int nt3 = st.nt < 3 ? st.nt : 3;
  L.n_lim = 1 << nt3;
  unsigned li = (unsigned)kcl << (st.k_in_pos[0] + 4), lo = 0;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    if ((j >> b) & 1) {
      li += 16u << st.lane_in_pos[b];
      lo += 16u << st.lane_out_pos[b];
    }
  }
  if (st.nt > 0) lo += (unsigned)kcl << (st.n_out_pos[0] + 4); // column bit 0 of the accumulator rows = g >> 1
  long lb = (long)kcl * st.k_b_stride[0] * 16;
#pragma unroll
  for (int b = 0; b < 3; ++b)
    if (b < nt3 && ((n_in >> b) & 1)) lb += st.n_b_stride[b] * 16;
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    if (b < st.wn_log2 && ((wn >> b) & 1)) {
      lo += 16u << st.n_out_pos[3 + b];
      lb += st.n_b_stride[3 + b] * 16;
    }
  }
  const bool valid = (n_in >> nt3) == 0;
  L.w_sign = !valid ? 0.0 : ((ro == 0 && p == 1) ? -1.0 : 1.0);
  L.lane_b = valid ? lb + (long)(ro ^ p) * 8 : 0;
  L.lane_in = (swz16(li, zin) ^ in_base) + (unsigned)p * 8u;
  L.lane_out = (swz16(lo, &st) ^ out_base) + (unsigned)p * 8u; // component of the accumulator rows = g & 1
#pragma unroll
  for (int b = 0; b < (KB > 1 ? KB : 2); ++b) {
    L.kin[b] = b < KB ? swz16(16u << st.k_in_pos[b], zin) : 0u;
    L.kb[b] = b < KB ? st.k_b_stride[b] * 16 : 0;
  }
  L.o1 = st.nt > 1 ? swz16(16u << st.n_out_pos[1], &st) : 0u;
  L.o2 = st.nt > 2 ? swz16(16u << st.n_out_pos[2], &st) : 0u;
  return L;
}

__device__ __forceinline__ void fill_msub_table128(const ArtnStage &st, const ArtnStage *zin, uint2 *tab, int tid) {
  const int msubs = 1 << (st.m_bits - 4);
  for (int m = tid; m < msubs; m += ARTN_WG_THREADS) {
    unsigned oi = 0, oo = 0;
    for (int b = 0; b < st.m_bits - 4; ++b) {
      if ((m >> b) & 1) {
        oi += 16u << st.msub_in_pos[b];
        oo += 16u << st.msub_out_pos[b];
      }
    }
    tab[m] = make_uint2(swz16(oi, zin), swz16(oo, &st));
  }
}

// W[s]: this lane's value of the small operand for chain step s (contracted value kc = 2 s + kcl)
template <int KB>
__device__ __forceinline__ void load_w128(double (&W)[1 << (KB - 1)], const char *__restrict__ Bbase, const Stage128<KB> &L) {
  constexpr int S = 1 << (KB - 1);
#pragma unroll
  for (int s = 0; s < S; ++s) {
    long off = 0;
#pragma unroll
    for (int b = 1; b < KB; ++b)
      if ((s >> (b - 1)) & 1) off += L.kb[b];
    W[s] = L.w_sign != 0.0 ? *reinterpret_cast<const double *>(Bbase + L.lane_b + off) * L.w_sign : 0.0;
  }
}

template <int KB>
__device__ __forceinline__ void run_stage128(const Stage128<KB> &L, const double (&W)[1 << (KB - 1)], int g) {
  constexpr int S = 1 << (KB - 1);
  auto store = [&](const f64x4 &acc, unsigned oa) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if ((g >> 1) + 2 * r < L.n_lim) lds_write_f64(oa ^ ((r & 1) ? L.o1 : 0u) ^ ((r & 2) ? L.o2 : 0u), acc[r]);
  };
  for (int ms = L.wm; ms < L.msubs; ms += 2 * L.wm_count) {
    const bool two = ms + L.wm_count < L.msubs; // (uniform)
    const u2_t e0 = lds_read_u2(L.tab + (unsigned)ms * 8u);
    const u2_t e1 = lds_read_u2(L.tab + (unsigned)(two ? ms + L.wm_count : ms) * 8u);
    const unsigned xa0 = L.lane_in ^ e0.x, xa1 = L.lane_in ^ e1.x;
    f64x4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    if (two) {
#pragma unroll
      for (int s = 0; s < S; ++s) {
        unsigned kx = 0;
#pragma unroll
        for (int b = 1; b < KB; ++b)
          if ((s >> (b - 1)) & 1) kx ^= L.kin[b];
        const double x0 = lds_read_f64(xa0 ^ kx), x1 = lds_read_f64(xa1 ^ kx);
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(W[s], x0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(W[s], x1, acc1, 0, 0, 0);
      }
      store(acc0, L.lane_out ^ e0.y);
      store(acc1, L.lane_out ^ e1.y);
    } else {
#pragma unroll
      for (int s = 0; s < S; ++s) {
        unsigned kx = 0;
#pragma unroll
        for (int b = 1; b < KB; ++b)
          if ((s >> (b - 1)) & 1) kx ^= L.kin[b];
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(W[s], lds_read_f64(xa0 ^ kx), acc0, 0, 0, 0);
      }
      store(acc0, L.lane_out ^ e0.y);
    }
  }
}

// ACC: C += result (ArtnBitsPlan::accumulate: the slice loop's `collect += ...`, reference simulation.py:114, in the store phase of
// a complex128 slice's last launch): the accumulator's 16-byte chunks are read and added before the stores -- every result
// element belongs to exactly one lane of one tile.  A separate set of instantiations (its own translation unit): behind a
// run-time flag the conditional loads cost every launch of the complex64 kernel 40 %.
typedef double f64x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 add_c128(f32x4 x, f32x4 c) {
  return __builtin_bit_cast(f32x4, __builtin_bit_cast(f64x2_t, x) + __builtin_bit_cast(f64x2_t, c));
}
template <int KB1, int KB2, bool ACC = false>
__global__ __launch_bounds__(ARTN_WG_THREADS, 2) void artn_k_bits128(const double2 *__restrict__ A, const double2 *__restrict__ B1,
                                                                    const double2 *__restrict__ B2, double2 *__restrict__ C,
                                                                    const ArtnBitsPlan P) {
  constexpr int S1 = 1 << (KB1 - 1);
  constexpr int KB2e = KB2 > 0 ? KB2 : 1;
  constexpr int S2 = 1 << (KB2e - 1);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if ((unsigned)(unsigned long)(lds_byte_t *)smem != 0) __builtin_trap(); // LDS is addressed by raw byte offsets
  // the larger region first: each region base is then a multiple of that region's size (XOR-ing it in equals adding it)
  const unsigned R0 = P.T_mid > P.r0_bits ? 16u << P.T_mid : 0u, R1 = P.T_mid > P.r0_bits ? 0u : 16u << P.r0_bits;
  const unsigned regions_end = (16u << P.r0_bits) + (16u << P.T_mid);
  uint2 *tab1 = reinterpret_cast<uint2 *>(smem + regions_end);
  uint2 *tab2 = tab1 + (1 << (P.st[0].m_bits - 4));
  long *offtab = reinterpret_cast<long *>(tab2 + (KB2 > 0 ? 1 << (P.st[1].m_bits - 4) : 0));

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;

  // ---- copy phases: thread handles elements e = tid + 256 * i of a tile
  unsigned in_lane = 0, out_lane = 0;
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    if ((tid >> b) & 1) {
      if (b < P.T_in) in_lane += (unsigned)P.in_stride[b] * 16u;
      if (b < P.T_out) out_lane += (unsigned)P.out_stride[b] * 16u;
    }
  }
  long in_hi[4], out_hi[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    in_hi[b] = 8 + b < P.T_in ? P.in_stride[8 + b] * 16 : 0;
    out_hi[b] = 8 + b < P.T_out ? P.out_stride[8 + b] * 16 : 0;
  }
  const unsigned tid16 = tid * 16;
  const int n_in_iters = 1 << (P.T_in - 8);
  const int n_out_iters = P.T_out >= 8 ? 1 << (P.T_out - 8) : 1;
  const bool out_active = P.T_out >= 8 || tid < (1 << P.T_out);

  fill_msub_table128(P.st[0], nullptr, tab1, tid);
  if (KB2 > 0) fill_msub_table128(P.st[1], &P.st[0], tab2, tid);
  const unsigned tab1_a = regions_end, tab2_a = tab1_a + (8u << (P.st[0].m_bits - 4));
  const Stage128<KB1> L1 = stage_const128<KB1>(P.st[0], nullptr, j, g, wave, tab1_a, R0, R1);
  const Stage128<KB2e> L2 = stage_const128<KB2e>(P.st[KB2 > 0 ? 1 : 0], &P.st[0], j, g, wave, tab2_a, R1, R0);
  const ArtnStage *zout = &P.st[KB2 > 0 ? 1 : 0];
  const unsigned tid16_out = swz16(tid16, zout);
  unsigned out_i_swz[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) out_i_swz[i] = swz16(i * (ARTN_WG_THREADS * 16), zout);
  const OffTab OT = build_offset_table(P, offtab, tid);
  double W1[S1], W2[S2];
  long prev_b1 = -1, prev_b2 = -1;
  __syncthreads(); // tables are in LDS

  // software pipeline over tiles, as artn_k_bits: stages(t) -> result region to registers -> refill R0 with tile t + 1
  // (loads issued one iteration ago) -> stores(t) -> issue loads(t + 2)
  constexpr int NV = 8;
  f32x4 v[NV];
  const bool pf_half = n_in_iters == NV / 2 && n_out_iters <= 8;
  const bool prefetch = (n_in_iters == NV || pf_half) && n_out_iters <= 8;
  TileOff off = {0, 0, 0, 0}, noff = {0, 0, 0, 0};
  long t0 = blockIdx.x, G = gridDim.x, n_tiles = P.n_tiles;
  if ((G & 7) == 0) t0 = (long)(blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3); // XCD-aware: see artn_k_bits
  if (P.blocked) {
    const long per = (P.n_tiles + gridDim.x - 1) / gridDim.x;
    t0 = per * blockIdx.x;
    G = 1;
    n_tiles = t0 + per < P.n_tiles ? t0 + per : P.n_tiles;
  }
  if (t0 < n_tiles) {
    off = tile_offsets<false>(P, OT, t0);
    copy_in_sync(reinterpret_cast<const char *>(A + off.a), in_hi, in_lane, R0, tid16, n_in_iters);
    if (t0 + G < n_tiles) {
      noff = tile_offsets<false>(P, OT, t0 + G);
      if (pf_half) issue_loads<NV, false, 0, NV / 2>(v, reinterpret_cast<const char *>(A + noff.a), in_hi, in_lane);
      else if (prefetch) issue_loads<NV, false>(v, reinterpret_cast<const char *>(A + noff.a), in_hi, in_lane);
    }
  }
  __syncthreads();

  for (long tile = t0; tile < n_tiles; tile += G) {
    if (off.b1 != prev_b1) {
      prev_b1 = off.b1;
      load_w128<KB1>(W1, reinterpret_cast<const char *>(B1 + off.b1), L1);
    }
    if (KB2 > 0 && off.b2 != prev_b2) {
      prev_b2 = off.b2;
      load_w128<KB2e>(W2, reinterpret_cast<const char *>(B2 + off.b2), L2);
    }
    const long next = tile + G, next2 = tile + 2 * G;
    TileOff n2off = noff;
    if (next2 < n_tiles) n2off = next_offsets<false>(P, OT, noff, next, G);

    // ---- stage 1: R0 -> R1, fused stage 2: R1 -> R0
    run_stage128<KB1>(L1, W1, g);
    __syncthreads();
    unsigned outr = R1;
    if (KB2 > 0) {
      run_stage128<KB2e>(L2, W2, g);
      __syncthreads();
      outr = R0;
    }

    unsigned lo_in = in_lane, lo_out = out_lane, t16 = tid16, t16o = tid16_out;
    OPAQUE_V(lo_in);
    OPAQUE_V(lo_out);
    OPAQUE_V(t16);
    OPAQUE_V(t16o);
    char *Cbase = reinterpret_cast<char *>(C + off.c);
    if (prefetch) {
      f32x4 x[8];
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (i < n_out_iters && out_active) x[i] = lds_read16(outr + (t16o ^ out_i_swz[i]));
      if (KB2 > 0) __syncthreads(); // fused: the result sat in R0, which is refilled next
      if (next < n_tiles) {
        if (pf_half) store_lds<NV, NV / 2>(v, R0, t16);
        else store_lds(v, R0, t16);
      }
      if constexpr (ACC) { // (into the registers the refill has just freed)
        f32x4 c[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (i < n_out_iters && out_active) {
            long o = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b)
              if ((i >> b) & 1) o += out_hi[b];
            c[i] = *reinterpret_cast<const f32x4 *>(Cbase + o + lo_out);
          }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (i < n_out_iters && out_active) x[i] = add_c128(x[i], c[i]);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (i < n_out_iters && out_active) {
          long o = 0;
#pragma unroll
          for (int b = 0; b < 4; ++b)
            if ((i >> b) & 1) o += out_hi[b];
          __builtin_nontemporal_store(x[i], reinterpret_cast<f32x4 *>(Cbase + o + lo_out));
        }
      }
      if (next2 < n_tiles) {
        if (pf_half) issue_loads<NV, false, 0, NV / 2>(v, reinterpret_cast<const char *>(A + n2off.a), in_hi, lo_in);
        else issue_loads<NV, false>(v, reinterpret_cast<const char *>(A + n2off.a), in_hi, lo_in);
      }
    } else {
      // other tile sizes: stream out, then load the next tile
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (i < n_out_iters && out_active) {
          long o = 0;
#pragma unroll
          for (int b = 0; b < 4; ++b)
            if ((i >> b) & 1) o += out_hi[b];
          f32x4 xv = lds_read16(outr + (t16o ^ out_i_swz[i]));
          if constexpr (ACC) xv = add_c128(xv, *reinterpret_cast<const f32x4 *>(Cbase + o + lo_out));
          *reinterpret_cast<f32x4 *>(Cbase + o + lo_out) = xv;
        }
      }
      if (KB2 > 0) __syncthreads();
      if (next < n_tiles) copy_in_sync(reinterpret_cast<const char *>(A + noff.a), in_hi, lo_in, R0, t16, n_in_iters);
    }
    __syncthreads(); // R0 holds the next tile; every wave is done with the result region
    off = noff;
    noff = n2off;
  }
}
