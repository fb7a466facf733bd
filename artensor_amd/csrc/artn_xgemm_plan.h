// artn_xgemm_plan.h -- launch plan of the EXTENT-based two-operand GEMM (artn_k_xgemm, artn_xgemm_kernel.h).
//
// The reference contracts tensor networks of any bond dimension: torch.einsum at
// /root/reference/artensor/contraction.py:70 takes whatever `bond_dims` the AbstractTensorNetwork was built with
// (/root/reference/artensor/tensor_network.py:4-30; the toy networks of /root/reference/tests/test_core.py:11,24).
// The bit planners of artn_plan.h need power-of-two extents (a label of extent 2^p is p address bits); a label of
// extent 3, 5, 6 ... used to send its step to the strided kernel.  Here a step is a GEMM over three FLATTENED mixed-radix
// indices,
//     C[h, m, n] = sum_k A[h, m, k] B[h, k, n]        m = free labels of A, n = free labels of B, k = contracted, h = batch,
// whose element offsets are separable: off_A(m, k) = offAm(m) + offAk(k), and so on for B and C.  A tile is 128 consecutive
// values of m x 32 nb consecutive values of n (no padding except in the last tile of a row / column: extents need not
// divide anything), the contracted index is walked in chunks of 16 values, zero-padded to an even count only at the end of
// a group.  Offsets come from small per-level tables built in LDS once per workgroup (levels 0 and 1: the innermost labels,
// at most 256 combinations each) and a mixed-radix decode of what is left (once per tile and row).
//
// Pure C++ (no HIP): included by artn_plan.h, compiled into libartn_hip.so and into the CPU plan emulator.
#ifndef ARTN_XGEMM_PLAN_H
#define ARTN_XGEMM_PLAN_H

#define ARTN_XG_MAXL 20   /* labels per flattened index */
#define ARTN_XG_MAXH 8    /* batch labels */
#define ARTN_XG_TM 128    /* rows (values of m) per tile: 4 waves x 32 */
#define ARTN_XG_KC 16     /* contracted values per LDS chunk */
#define ARTN_XG_LEVEL 256 /* entries of one level table */
#define ARTN_XG_KTAB (ARTN_XG_LEVEL + ARTN_XG_KC) /* the k tables are padded by one chunk: no clamping in the copy loop */
#define ARTN_XG_FLUSH 4096 /* contracted values per fp32 partial sum (as ARTN_GEMM_FLUSH_LOG2) */
#define ARTN_XROW_MAX 48      /* artn_k_xrow: contracted values, and columns, of the small operand */
#define ARTN_XROW_MIN_ROWS (1 << 15)
#define ARTN_XROW_L2_MAX 4096 /* entries of the third level of its row-offset tables (what the two 256-entry levels leave of the row index) */

// One flattened index: labels innermost first, each with its extent and its element stride in the two tensors that
// carry it (m: A and C; n: B and C; k: A and B).
struct ArtnXSide {
  int32_t n_lab;
  int32_t n0, n1;  // labels of level 0 / level 1 (prefixes whose extents multiply to at most ARTN_XG_LEVEL)
  int32_t L0, L1;  // product of the extents of level 0 / level 1
  int32_t pad_;
  int64_t total;   // product of all extents
  int32_t ext[ARTN_XG_MAXL];
  int64_t s0[ARTN_XG_MAXL], s1[ARTN_XG_MAXL];
};

struct ArtnXGemmPlan {
  ArtnXSide m, n, k;
  int32_t n_h;                 // batch labels (carried by A, B and C): one tile belongs to one value of them
  int32_t h_ext[ARTN_XG_MAXH];
  int64_t h_sA[ARTN_XG_MAXH], h_sB[ARTN_XG_MAXH], h_sC[ARTN_XG_MAXH];
  int32_t nb;                  // 32-column MFMA blocks per tile (1..3): the tile is 128 x 32 nb
  int32_t trans;               // 1: C's fastest label is a label of n -- the MFMA roles of the operands are swapped so that the lanes
                               //    of a store run along n (accumulator register <-> row of m); 0: lanes run along m
  int32_t amode, bmode;        // copy lanes of an operand run along its free index (0) or along k (1): whichever its fastest label is
  int32_t swapped;             // 1: the kernel's first operand is the caller's B
  int32_t cpg;                 // chunks per group of k.L0 contracted values: ceil(k.L0 / kc)
  int32_t flush_chunks;        // partial sums leave the registers every this many chunks (read-add-write of C); 0: never
  int32_t prio;                // 1: the workgroup in the odd wave slots runs its MFMA loops at s_setprio 1
  int32_t kc;                  // contracted values per chunk: 16, or 8 (few contracted values, nb = 1: four workgroups per CU)
  int32_t pc;                  // 1: artn_k_xgemm_pc -- one 8-wave workgroup per CU, four consumer and four producer waves (kc = 16)
  int32_t c128;                // 1: complex128 operands -- artn_k_xgemm128 (16-byte elements, kc = 8, nb = 1, f64 MFMA)
  int32_t rowmode;             // 1 (round 6): artn_k_xrow, 2: artn_k_xrow64 (a lane per row) -- the row-streaming form (artn_xrow_kernel.h): at most 48 contracted values and 48
                               //    columns, the small operand in registers, rows straight into the MFMA operand registers, no LDS staging
  uint32_t row_bytes_a, row_bytes_c; // rowmode: bytes spanned by the kernel's first operand and by the result (buffer range checks)
  int32_t col0;                // first column of this launch (0; the tail launch: 32 nb tiles_n of the main one)
  int32_t tail_nb;             // > 0 (round 6): the columns behind the tiles_n FULL column tiles -- fewer than nb blocks of 32 -- run as a
                               //    second launch of the instantiation with tail_nb blocks per tile (artn_xg_tail_plan), so that a narrow last
                               //    tile does not multiply blocks that hold no column (216 columns: 2 tiles of 96 + 1 of 32, was 4 of 64)
  int32_t tail_grid, tail_lds; // workgroups and LDS bytes of that launch
  int64_t k_groups;            // k.total / k.L0
  int64_t tiles_m, tiles_n, n_tiles; // n_tiles = tiles_m * tiles_n * prod(h_ext); tile index = (h, tile of m, tile of n), n fastest
};

// LDS layout shared by the kernel, the launcher and the emulator (byte offsets).
static inline int artn_xg_pitch_a() { return ARTN_XG_TM + 2; }          // elements between two contracted values of the A image
static inline int artn_xg_pitch_b(int nb) { return 32 * nb + 2; }       // ... of the B image
static inline int artn_xg_stage_bytes(int nb, int kc) { return kc * (artn_xg_pitch_a() + artn_xg_pitch_b(nb)) * 8; }
static inline int artn_xg_level_bytes() { return 8 * ARTN_XG_LEVEL * 4 + 2 * ARTN_XG_KTAB * 4; } // mA0 mC0 mA1 mC1 nB0 nC0 nB1 nC1, kA kB
static inline int artn_xg_tiletab_bytes() { return 4 * ARTN_XG_TM * 4; }   // rowA rowC colB colC of one tile
static inline int artn_xg_lds_bytes(int nb, int kc) { return 2 * artn_xg_stage_bytes(nb, kc) + artn_xg_level_bytes() + 2 * artn_xg_tiletab_bytes(); }
// complex128 (artn_k_xgemm128): 16-byte elements, chunks of 8 contracted values, nb = 1
static inline int artn_xg128_lds_bytes(int nb) { return 2 * 8 * (artn_xg_pitch_a() + artn_xg_pitch_b(nb)) * 16 + artn_xg_level_bytes() + 2 * artn_xg_tiletab_bytes(); }
static inline int artn_xg_pc_lds_bytes(int nb) { return 2 * artn_xg_stage_bytes(nb, ARTN_XG_KC) + artn_xg_level_bytes() + 8 * 1024; } // four row + four column table sets

// artn_k_xrow: MFMA steps of four contracted values, column blocks of 16, prefetch distance in 16-row blocks, waves per SIMD
// -- kernel, launcher, planner, emulator
static inline constexpr int artn_xrow_steps(int64_t k_total) { return (int)((k_total + 3) / 4); }
static inline constexpr int artn_xrow_nbk(int64_t n_total) { return (int)((n_total + 15) / 16); }
#ifdef ARTN_XROW_DEEP /* development builds: two blocks ahead for the largest fragments too */
static inline constexpr int artn_xrow_depth(int S) { return S <= 2 ? 4 : (S <= 4 ? 3 : (S <= 8 ? 2 : 1)); }
#else
static inline constexpr int artn_xrow_depth(int S) { return S <= 2 ? 4 : (S <= 4 ? 3 : (S <= 6 ? 2 : 1)); }
#endif
static inline constexpr int artn_xrow_waves(int S, int NBK) { // (what hipcc's register counts of the 36 instantiations allow: 48 ... 210)
  return NBK == 1 ? (S <= 2 ? 8 : (S <= 3 ? 6 : (S <= 6 ? 5 : (S <= 8 ? 6 : (S <= 9 ? 5 : 4)))))
       : NBK == 2 ? (S <= 1 ? 6 : (S <= 2 ? 5 : (S <= 8 ? 4 : 3)))
                  : (S <= 1 ? 6 : (S <= 3 ? 4 : (S <= 8 ? 3 : 2)));
}
static inline int artn_xrow_lds_bytes(int64_t L2) { return 4096 + 8 * (int)L2; } // three levels of (A, C) byte offsets of a row
// artn_k_xrow64 (rowmode 2: a lane per row, 64-row superblocks): S <= 8, NBK <= 2; waves per SIMD by hipcc's register counts
static inline constexpr int artn_xrow64_waves(int S, int NBK) {
  return NBK == 1 ? (S <= 3 ? 4 : (S <= 7 ? 3 : 2)) : (S <= 1 ? 4 : (S <= 7 ? 2 : 1));
}

// The tail launch of a plan with tail_nb > 0: the same step restricted to the columns behind the full tiles, one column tile wide.
static inline ArtnXGemmPlan artn_xg_tail_plan(const ArtnXGemmPlan &P) {
  ArtnXGemmPlan T = P;
  T.col0 = (int32_t)(P.tiles_n * 32 * P.nb);
  T.nb = P.tail_nb;
  T.n_tiles = P.n_tiles / P.tiles_n; // tiles_m x batch values
  T.tiles_n = 1;
  T.tail_nb = 0;
  T.tail_grid = T.tail_lds = 0;
  return T;
}

// Mixed-radix decode of `idx` over labels [first, first + count) of a side: the two element offsets.
#if defined(__HIPCC__)
#define ARTN_XG_HD __host__ __device__ __forceinline__
#else
#define ARTN_XG_HD static inline
#endif
ARTN_XG_HD void artn_xg_decode(const ArtnXSide &S, int first, int count, uint32_t idx, uint32_t &o0, uint32_t &o1) {
  uint32_t a = 0, b = 0;
  for (int i = first; i < first + count; ++i) {
    const uint32_t e = (uint32_t)S.ext[i];
    const uint32_t q = idx / e, d = idx - q * e;
    a += d * (uint32_t)S.s0[i];
    b += d * (uint32_t)S.s1[i];
    idx = q;
  }
  o0 = a;
  o1 = b;
}

// artn_k_xrow: a row's position in the three levels of the row index (level 0 fastest) and its advance by a fixed stride
// without a division: `d` is the stride's own position (d.i0 < L0, d.i1 < L1); i2 is left to grow past its level (rows past the
// end: the kernel clamps the table read and stores nothing).
struct ArtnXRowPos { uint32_t i0, i1, i2; };
ARTN_XG_HD void artn_xrow_place(uint32_t m, uint32_t L0, uint32_t L1, ArtnXRowPos &p) {
  const uint32_t q0 = m / L0;
  p.i0 = m - q0 * L0;
  p.i2 = q0 / L1;
  p.i1 = q0 - p.i2 * L1;
}
ARTN_XG_HD void artn_xrow_advance(ArtnXRowPos &p, const ArtnXRowPos &d, uint32_t L0, uint32_t L1) {
  p.i0 += d.i0;
  uint32_t c = p.i0 >= L0 ? 1u : 0u;
  p.i0 -= c ? L0 : 0u;
  p.i1 += d.i1 + c;
  c = p.i1 >= L1 ? 1u : 0u;
  p.i1 -= c ? L1 : 0u;
  p.i2 += d.i2 + c;
}

#endif
