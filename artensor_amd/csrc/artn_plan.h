// artn_plan.h -- host-side lowering of pairwise contraction steps (ArtnStepDesc, the
// label-list form of the einsum strings built at /root/reference/artensor/contraction.py:13-20)
// into the launch plan consumed by the gfx950 kernels in artn_kernels.hip.
//
// Pure C++ (no HIP): it is compiled into libartn_hip.so and, separately, into the
// CPU-only plan emulator the tests use to check the index algebra without a GPU.
//
// Vocabulary.  Every label whose extent is a power of two is split into *bits* (extent-2
// axes); a contraction over all-dims-2 circuit tensors is then a permutation of address
// bits around a small complex GEMM:
//   K bits  -- carried by A and B, not C (contracted)
//   M bits  -- carried by A and C only   (free bits of the big "state" operand)
//   N bits  -- carried by B and C only   (free bits of the small operand)
//   H axes  -- carried by all three      (batch: the sparse path's shared row label)
// A *tile* is the set {all K bits} u {M_t: a subset of M bits} of A, staged in LDS by one
// workgroup; a *stage* multiplies it by the small operand and scatters the result tile
// {M_t} u {N_t} into a second LDS region.  A plan has one stage (one reference step) or
// two (two consecutive steps on the same state tensor fused into ONE pass over HBM: the
// second stage contracts bits of the first stage's result while it is still in LDS).
// Everything not inside the tile is an *outer* axis enumerated by the tile index.
#ifndef ARTN_PLAN_H
#define ARTN_PLAN_H

#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>

#include "artn.h"
#include "artn_xgemm_plan.h"

#define ARTN_MAX_OUTER 36
#define ARTN_TILE_BITS_MAX 13 /* 2^13 complex64 = 64 KiB per LDS region */
#define ARTN_TILE_BITS_TARGET 12
#define ARTN_LDS_BUDGET (128 * 1024)
#define ARTN_WG_THREADS 256

struct ArtnOuterDim {
  int64_t ext;
  int64_t sA, sB1, sB2, sC; /* element strides; 0 where the operand does not carry the axis */
  int32_t log2ext;          /* >= 0 for powers of two, -1 otherwise */
  int32_t pad_;
};

// One MFMA stage: LDS region `in` (tile-local bit order of its input tile) -> region `out`.
struct ArtnStage {
  int32_t k, nt;           // contracted bits, N bits produced inside the tile
  int32_t wn_log2;         // waves along n' tiles (16 complex n each): max(0, nt - 4)
  int32_t m_bits;          // free input-tile bits = 5 lane bits + sub-tile bits
  int32_t lane_in_pos[5], lane_out_pos[5]; // MFMA column bit j -> tile-local bit (in / out)
  int32_t msub_in_pos[9], msub_out_pos[9]; // sub-tile bit      -> tile-local bit (in / out)
  int32_t k_in_pos[8];                     // K bit i (kc bit i) -> tile-local input bit
  int32_t n_out_pos[6];                    // N_t bit i          -> tile-local output bit
  int64_t k_b_stride[8];                   // K bit i   -> small-operand element stride
  int64_t n_b_stride[6];                   // N_t bit i -> small-operand element stride
  // XOR swizzle of this stage's OUTPUT region: element-offset bit swz_dst[i] ^= bit swz_src[i].
  // The 16 lanes of one ds_write_b64 group differ in MFMA column bits 0..3; where those sit
  // above the 128-byte bank window (tile-local position >= 4) they are folded into a free
  // position 1..3 so the group spreads over the banks (at most 2-way conflicts remain).
  int32_t swz_n, swz_src[4], swz_dst[4]; // (complex128: units of 16 bytes, targets 0..3, up to four entries)
  int32_t m3; // 1: three real products per complex product (5 contracted bits, 32+ columns: a wave owns 32-column
              // sub-tiles; wn_log2 = nt - 5; lane half h carries column bit 2); 2: the 4-bit stage of such a launch, on
              // 16 x 16 x 4 blocks (lane group l >> 4 carries contracted bits 0, 1 and column bits 2, 3)
};

// Launch plan of the LDS-tiled bit-permuted complex GEMM (kernel argument, POD).
struct ArtnBitsPlan {
  int32_t n_stages;           // 1, 2 (fused pair) or 3 (fused triple: artn_k_bits3)
  int32_t T_in, T_mid, T_out; // log2 elements: input tile, tile between the (first two) stages, output tile
  int32_t T_mid2;             // three stages: the tile between the second and the third (region 0 again)
  int32_t r0_bits;            // LDS region 0 holds 2^r0_bits elements (region 1 follows it)
  int32_t r1_bits;            // region 1 holds 2^r1_bits elements (= T_mid for one or two stages)
  int32_t run_in, run_out;    // tile-local bits [0,run) are global bits [0,run)
  int32_t n_outer;
  int32_t stage_prio;         // 1: one of the two co-resident workgroups runs its MFMA stages at s_setprio 2
  int32_t blocked;            // 1: a workgroup takes a contiguous range of tiles instead of a grid-stride sequence
  int64_t n_tiles;
  int32_t m3;                 // 1: every stage with 5 contracted bits runs the 3M arithmetic (ArtnStage::m3): the M3 instantiation
  int32_t c128;               // 1: complex128 elements -- artn_k_bits128 (16-column sub-tiles: 4 lane bits, f64 MFMA)
  int32_t ksplit;             // 1: 7-8 contracted bits, one 32 x 16 block per tile: the four waves split the chain
  int32_t split;              // MFMA arithmetic: 0 fp32; 3 fp32-grade from three bf16 pieces; 1 plain bf16 operands
  int32_t wide8;              // 1: fused pair of 2^12-element tiles run by artn_k_wide (one 8-wave workgroup per CU, every stage 3M on
  int32_t accumulate;         //    16 x 16 x 4 blocks, tiles by LDS-DMA); the plan itself is the one artn_k_bits would run
                              // accumulate = 1: C += result (artn_contract_acc / artn_contract2_acc: the store phase reads the
                              // accumulator's chunk and adds -- every result element belongs to exactly one lane of one tile; set
                              // by the entry point on plans for which bits_can_accumulate() holds
  int32_t narrow3;            // 1: single step with 5 or 6 contracted bits and at most 4 result bits in the tile: three products on
  int32_t pad9_;              //    16 x 16 x 4 blocks (ArtnStage::m3 = 2; artn_k_bits<..., N3>) instead of four on 32 x 32 blocks of
                              //    which at most 16 rows are results; 2: the second stage of a fused pair, likewise
  int64_t in_stride[ARTN_TILE_BITS_MAX];  // tile-local input bit  -> A element stride
  int64_t out_stride[ARTN_TILE_BITS_MAX]; // tile-local output bit -> C element stride
  ArtnStage st[3];
  ArtnOuterDim outer[ARTN_MAX_OUTER];
  // fused row gather (artn_contract_gather, read by the GATHER instantiations only): along outer axis
  // `gather_dim` operand A is read at row rows_a[x] and the small operand at rows_b[x] instead of x
  // (device pointers, nullptr = x); indices outside [0, src_rows) read row 0 and set *gather_err
  int32_t gather_dim; // -1: none
  int32_t nt_loads;   // 1: every A tile is read once, in full 128-byte runs, by a big launch: non-temporal loads
  const int64_t *rows_a;
  const int64_t *rows_b;
  int64_t src_rows_a, src_rows_b;
  int32_t *gather_err;
};

// Plan of the strided fallback (one thread per C element).
#define ARTN_GEN_MAX_OUT 64
#define ARTN_GEN_MAX_RED 32
struct ArtnGenericPlan {
  int32_t n_out, n_red;
  int64_t out_numel, red_numel;
  // output axes, fastest (stride_c == 1) first; adjacent axes that are contiguous in A
  // and B are merged, so these hold far fewer entries than the label count
  int64_t out_ext[ARTN_GEN_MAX_OUT], out_sA[ARTN_GEN_MAX_OUT], out_sB[ARTN_GEN_MAX_OUT];
  // reduction axes, innermost first
  int64_t red_ext[ARTN_GEN_MAX_RED], red_sA[ARTN_GEN_MAX_RED], red_sB[ARTN_GEN_MAX_RED];
};

// Launch plan of the two-operand LDS GEMM (artn_k_gemm): both operands are staged through LDS in
// chunks of 2^kc contracted values, the remaining contracted bits are looped over inside the
// kernel with the accumulators in registers, the result tile leaves through LDS in C order.
// For steps whose contracted set or second operand is too big for the state-streaming kernel above
// (7+ contracted bits; big x big steps of sliced circuits and random networks).
#define ARTN_GEMM_MAX_KO 40
#define ARTN_GEMM_KC 4           /* contracted bits per LDS chunk */
#define ARTN_GEMM_KC_TALL 6      /* ... of a 32 x 32 tile (image rows 2^5 elements apart) */
#define ARTN_GEMM_PITCH_TALL_LOG2 5
#define ARTN_GEMM_FLUSH_LOG2 12   /* fp32: partial sums leave the registers every 2^12 contracted values */
#define ARTN_GEMM_EPI_BITS 13    /* the result tile leaves in passes of 2^13 elements (64 KiB) */
#define ARTN_GEMM_PITCH_LOG2 7   /* rows of both LDS images are 2^7 elements apart whatever mt / nt: every
                                    LDS read of the MFMA loop has a compile-time offset */
struct ArtnGemmPlan {
  int32_t mt, nt, kc;        // tile: free bits of the first / second operand, contracted bits per chunk
  int32_t n_ko;              // contracted bits looped over inside the kernel (2^n_ko chunks per tile)
  int32_t swapped;           // 1: the kernel's first operand is the caller's B (it has the free bits to tile)
  int32_t mb_log2, nb_log2;  // 32-row / 16-column MFMA blocks per wave
  int32_t wm_log2, wn_log2;  // waves along m / along n (idle waves when the sum is < 2)
  int32_t split;             // 0 fp32 MFMA; 1 bf16 operands (ARTN_C64_BF16)
  int32_t n_outer;
  int32_t blocked;           // (always 0; kept so the tile-offset helpers serve both plans)
  int64_t n_tiles;
  // global -> LDS: bit b of a chunk's element index, elements ordered by stride in the operand
  // (bit 0 = the operand's stride-1 bit: a thread moves elements 2c, 2c+1 with one 16-byte load)
  int32_t ta_bits, tb_bits;  // mt + kc, nt + kc
  int64_t a_stride[12], b_stride[12]; // -> element stride in the operand
  int32_t a_lds[12], b_lds[12];       // -> byte offset in the LDS image (fp32: [kc][2^7 rows] x 8 B; bf16: [kc >> 2][2^7 rows][kc & 3] x 4 B)
  int64_t ko_sA[ARTN_GEMM_MAX_KO], ko_sB[ARTN_GEMM_MAX_KO]; // looped contracted bit -> element strides
  // epilogue: the result tile in C order
  int32_t tc_bits;           // mt + nt
  int32_t m3;                // 1: three real products per complex product (blocks of 32 columns; nb_log2 counts those)
  int64_t out_stride[14];    // C-tile-local bit -> C element stride
  int32_t m_pos[8], n_pos[8]; // m_local / n_local bit -> C-tile-local position
  int32_t swz_n, swz_src[4], swz_dst[4]; // XOR swizzle of the LDS result image (as ArtnStage::swz_*)
  int32_t wk_log2;           // tiles with fewer than 4 MFMA blocks: 2^wk_log2 waves share a block and split each chunk's
                             // contracted values between them (partial blocks are added up in the LDS result image)
  int32_t pitch_log2;        // rows of the fp32 operand images are 2^pitch_log2 elements apart: 7, or 5 with kc = 6 (32 x 32 tiles)
  int32_t pad1_[1];
  ArtnOuterDim outer[ARTN_MAX_OUTER];
  // (unused by this kernel; present so tile_offsets<> compiles for both plan types)
  int32_t gather_dim;
  int32_t pad2_;
  const int64_t *rows_a;
  const int64_t *rows_b;
  int64_t src_rows_a, src_rows_b;
  int32_t *gather_err;
};

// Launch plan of the packed-operand GEMM (artn_k_pack_bf16 + artn_k_pgemm under ARTN_C64_BF16; artn_k_pack_f32 + artn_k_pgemm3m
// for plain ARTN_C64 since round 3, tuning().packed = 2): big x big steps with
// many contracted bits (BASELINE configs[4]: 2^30 x 2^29 elements over 15 contracted bits).  One pass per operand
// rounds it to bfloat16 and writes it in the order the GEMM's LDS images have -- [tile][chunk][kc >> 2][row][kc & 3],
// 4 bytes per complex element -- into a caller-supplied workspace; the GEMM then moves half the bytes per operand
// element, as contiguous 16-byte lanes straight into LDS (LDS-DMA), and its 256 x 128 tiles re-read each operand
// element a quarter as often as the 128 x 64 tiles of artn_k_gemm.
#define ARTN_PG_MT 8 /* 256 rows of the first operand per tile  */
#define ARTN_PG_NT 7 /* 128 rows of the second operand per tile */
#define ARTN_PG_KC 5 /* 32 contracted values per chunk          */
#define ARTN_PG_EPI_BITS 13
struct ArtnPackSide { // one operand: element strides of its bits
  int64_t row[8];     // tile-row bit i
  int64_t kc[ARTN_PG_KC]; // chunk bit q (q = 0, 1: inside a 16-byte lane; q = 2..4: the plane index)
  int64_t ko[ARTN_GEMM_MAX_KO]; // looped contracted bit
  int64_t to[32];     // tile-outer bit
  int32_t n_row, n_to;
};
struct ArtnPackPlan {
  int32_t arith;      // 0: bfloat16 operands (4-byte packed elements, chunks of 2^5); 1: fp32, 3M arithmetic (8-byte elements, chunks of 2^4)
  int32_t kc_bits;    // contracted bits per chunk
  int32_t swapped;    // 1: the kernel's first operand is the caller's B
  int32_t n_ko;       // 2^n_ko chunks per tile
  int32_t n_mo, n_no; // tile-outer bits of the first / second operand
  int64_t n_tiles;
  ArtnPackSide a, b;
  int64_t c_mo[32], c_no[32]; // C element strides of the tile-outer bits
  int64_t out_stride[16];     // C-tile-local bit (tile bits ordered by C stride) -> C element stride
  int32_t m_pos[8], n_pos[8]; // m_local / n_local bit -> C-tile-local position
  int32_t swz_n, swz_src[4], swz_dst[4];
  int32_t flush_chunks; // fp32: partial sums leave the registers every this many chunks (2^12 contracted values); 0: never
};

struct ArtnPlan {
  int kernel; // ARTN_KERNEL_*
  ArtnPackPlan pack;
  int n_cu;   // compute units the plan was made for
  ArtnBitsPlan bits;
  ArtnGemmPlan gemm;
  ArtnXGemmPlan xg;
  ArtnGenericPlan gen;
  ArtnStepInfo info;
  int64_t stage1_repeats = 1; // fused pairs: visits of one input tile = values of the second step's result bits outside the tile
  std::string why_generic;
};

namespace artn {

// Development knobs (environment: ARTN_WG_PER_CU, ARTN_TILE_TARGET, ARTN_RUN_MAX, ARTN_SWIZZLE, ARTN_STAGE_PRIO,
// ARTN_SPLIT, ARTN_NT); the defaults are what ships.
struct Tuning {
  int wg_per_cu = 2;  // persistent workgroups per CU (grid = CUs * this), capped by LDS
  int tile_target = ARTN_TILE_BITS_TARGET;
  int run_max = 4;    // longest contiguous run (log2 elements) the tile is forced to keep
  int swizzle = 1;    // XOR-swizzle stage output regions against LDS bank conflicts
  int stage_prio = 3; // 1: asymmetric MFMA-stage priority between the two workgroups of a CU; 3: and every workgroup's copy
                      // phases at raised priority (with 3M stages +1 % on n30, A/B in one session: 57.7 -> 57.1 ms; 2: copy only)
  int split = 0;      // complex64 chains: 0 fp32 MFMA, 3 fp32-grade split-bf16 MFMA
  int nt = 1;         // non-temporal loads of A tiles that are read once
  int xg_tail = 1;    // extent GEMM: the columns behind the full column tiles as a second, narrower launch (ARTN_XG_TAIL=0: one padded launch)
  int xrow64 = 1;     // ... its 64-row shape (artn_k_xrow64; ARTN_XROW64=0: the 16-row shape only)
  int xrow = 1;       // the row-streaming form of the extent GEMM (artn_k_xrow; ARTN_XROW=0: artn_k_xgemm for those steps too)
  int gemm = 1;       // two-operand LDS GEMM: 0 never, 1 for 7+ contracted bits or a big second operand, 2 whenever it fits
  int gemm_3m = 1;    // GEMM kernel, fp32, tiles with 32+ columns: three real products per complex product
  int gemm_tall = 1;  // GEMM kernel, fp32, 32 x 32 tiles: chunks of 2^6 contracted values
  int packed = 2;     // packed-operand GEMM (ArtnPackPlan): 1 reduced-precision mode only (2^9+ contracted values),
                      // 2 also complex64 arithmetic (3M on fp32 MFMA, 2^10+ contracted values), 0 never
  int bits128 = 1;         // complex128 on the state-streaming kernel: 0 never, 1 fused pairs (+ singles the GEMM declines), 2 singles first
  int gemm_deep = 2;       // GEMM kernel, fp32 3M, one block column per wave: operand loads two chunks ahead (artn_k_gemm_deep); 2: also 64-row waves
  int idle_to_gemm = 1;    // single steps whose state-streaming tile would leave waves idle go to the GEMM kernel
  int gather_gemm = 2;     // row-gather steps with 7+ contracted bits on the GEMM kernel: 1 when the second operand has 5+ free bits, 2 always, 0 never
  int fuse_max_rereads = 4;   // fused pairs: how often the first stage may be repeated per input tile (second-step result bits outside the
                              // tile).  Measured (tools/ab_env.sh, ms per slice, limit none / 8 / 4 / 2): n53 m20 76.0 / 76.4 / 75.0 / 75.1;
                              // n53, rand2, rand4, n30 x 10 000 within noise
  int grow_nt = 1;         // single growth steps: result bits beyond the contracted count taken into the tile (0 or 1)
  int fuse_66 = 0;         // fused pairs of two 6-bit steps (complex64): 1 allows them
  int m3_frag = 96;        // 3M in fused pairs up to this many fragment registers (80: not in 5+6 / 6+5 pairs)
  int fuse3_frag = 80;     // triples: fragment registers of the three stages (ARTN_FUSE3_FRAG); fuse3_run: shortest input run (log2 elements)
  int fuse3_run = 4;
  int fuse3 = 1;           // three consecutive steps on one tensor in one pass where they fit a 2^12 tile (artn_k_bits3): ARTN_FUSE3
  int pgemm16 = 1;         // reduced-precision packed GEMM: 0 v_mfma_f32_32x32x16_bf16, 1 16x16x32 (three chunk buffers), 2 16x16x32 on a ring of six half-chunk slots: ARTN_PGEMM16
  int packed_min_k = 11;   // complex64 arithmetic: contracted bits from which the packed-operand GEMM is used.  8 in round 3 (tools/ab_packk.sh:
                           // +1 % on n53 m20 and the D = 4 network then); since artn_k_gemm reads its operands two chunks ahead it wins up
                           // to 2^10 contracted values -- the D = 4 network's 2^10-deep step 4.21 -> 3.93 ms, the leg 107.7 -> 114.0
                           // TFLOP/s, n53 m20 91.5 -> 92.2 (A/B in one session) -- and the packed form keeps the 2^15-deep step of the
                           // big-batch slice (180 against 131 TFLOP/s)
  int packed_min_ai = 160; // ... and the FLOP per byte of the step it needs (64 until round 4: the 2^20 x 2^8 x 2^8 step of an n53 m14 slice --
                           // 128 FLOP per byte, a quarter of its time in the packing passes -- takes 4.65 ms packed and 4.25 on
                           // artn_k_gemm; the 2^8- and 2^10-deep steps of the D = 4 random network (204-205) stay packed: 1.00 against 1.08 ms)
  int xgemm = 1;      // steps with a label whose extent is not a power of two on the extent-based GEMM (artn_k_xgemm); 0: strided kernel (ARTN_XGEMM)
  int narrow3 = 1;    // single steps with 5-6 contracted bits and <= 4 result bits in the tile on 16 x 16 x 4 blocks, 3M (ArtnBitsPlan::narrow3)
  int wide = 2;       // fused pairs of 2^12-element tiles on artn_k_wide (ArtnBitsPlan::wide8): 0 never; 1 all of them (loses:
                      // 56.5 ms on n30 against 53.5); 2 (default) the pairs with 11+ contracted bits -- 5+6, 6+5, 6+6 -- whose
                      // fragments artn_k_bits cannot hold next to three accumulators (it runs them as four-product chains, or not
                      // at all): the 5+6 pair of n30 x 10 000 7.50 -> 6.05 ms
  int wide_min_tiles = 0; // ... for launches of at least this many tiles (0: 8 per CU)
  int alt = 2;        // big launches of the state-streaming kernel: 1: one 8-wave workgroup per CU, two groups alternating
                      // between MFMA stages and copy phases (artn_k_alt); 0: two independent workgroups per CU (artn_k_bits);
                      // 2: artn_k_alt where a tile's OUTPUT runs are shorter than a 128-byte line (its stores, slow
                      // there, hide under the other group's stages: 5.15 -> 4.86 ms on the n30 pair that writes 64-byte
                      // runs), artn_k_bits elsewhere (two waves per SIMD in the stages hide each other's LDS latencies:
                      // measured equal or faster wherever the copies are cheap; A/B in one session)
  int bits_3m = 2;    // state-streaming kernel, fp32 stages with 5/6 contracted bits and 32+ columns: the same
                      // (1: not in fused pairs that hold a 6-bit stage, 0: never)
};
// Product builds read THREE planner switches from the environment -- ARTN_WIDE, ARTN_WIDE_MIN_TILES (the tests force
// artn_k_wide onto every fused pair through them) and ARTN_XGEMM (0: steps with odd extents back on the strided kernel,
// the A/B of the round-5 tests).  Everything else is the switch of one concluded A/B measurement (DESIGN.md sections 4-7)
// and exists only in development builds (`make dev`: -DARTN_DEV_SWITCHES), where tools/ab_env.sh can flip it.
static inline Tuning &tuning() {
  static Tuning t = [] {
    Tuning x;
    if (const char *e = getenv("ARTN_WIDE")) x.wide = std::min(2, std::max(0, atoi(e)));
    if (const char *e = getenv("ARTN_WIDE_MIN_TILES")) x.wide_min_tiles = atoi(e);
    if (const char *e = getenv("ARTN_XGEMM")) x.xgemm = atoi(e) != 0;
    if (const char *e = getenv("ARTN_XROW")) x.xrow = atoi(e);
    if (const char *e = getenv("ARTN_XROW64")) x.xrow64 = atoi(e) != 0;
    if (const char *e = getenv("ARTN_XG_TAIL")) x.xg_tail = atoi(e) != 0;
#ifdef ARTN_DEV_SWITCHES
    if (const char *e = getenv("ARTN_WG_PER_CU")) x.wg_per_cu = std::max(1, atoi(e));
    if (const char *e = getenv("ARTN_TILE_TARGET")) x.tile_target = std::min(ARTN_TILE_BITS_MAX, std::max(9, atoi(e)));
    if (const char *e = getenv("ARTN_RUN_MAX")) x.run_max = std::min(6, std::max(1, atoi(e)));
    if (const char *e = getenv("ARTN_SWIZZLE")) x.swizzle = atoi(e) != 0;
    if (const char *e = getenv("ARTN_STAGE_PRIO")) {
      const int v = (int)strtol(e, nullptr, 0);   // (>= 0x100: one level per phase, see set_prio_rt in artn_kernels.hip)
      x.stage_prio = v >= 0x100 ? (v & 0x17f) : std::min(4, std::max(0, v));
    }
    if (const char *e = getenv("ARTN_NT")) x.nt = atoi(e) != 0;
    if (const char *e = getenv("ARTN_GEMM")) x.gemm = std::min(2, std::max(0, atoi(e)));
    if (const char *e = getenv("ARTN_GEMM_3M")) x.gemm_3m = atoi(e) != 0;
    if (const char *e = getenv("ARTN_GEMM_TALL")) x.gemm_tall = atoi(e) != 0;
    if (const char *e = getenv("ARTN_M3_FRAG")) x.m3_frag = atoi(e);
    if (const char *e = getenv("ARTN_FUSE_66")) x.fuse_66 = atoi(e) != 0;
    if (const char *e = getenv("ARTN_GROW_NT")) x.grow_nt = atoi(e) != 0;
    if (const char *e = getenv("ARTN_FUSE_MAX_REREADS")) x.fuse_max_rereads = std::max(1, atoi(e));
    if (const char *e = getenv("ARTN_GATHER_GEMM")) x.gather_gemm = atoi(e);
    if (const char *e = getenv("ARTN_IDLE_TO_GEMM")) x.idle_to_gemm = atoi(e) != 0;
    if (const char *e = getenv("ARTN_GEMM_DEEP")) x.gemm_deep = atoi(e);
    if (const char *e = getenv("ARTN_BITS128")) x.bits128 = atoi(e);
    if (const char *e = getenv("ARTN_FUSE3")) x.fuse3 = atoi(e) != 0;
    if (const char *e = getenv("ARTN_FUSE3_FRAG")) x.fuse3_frag = atoi(e);
    if (const char *e = getenv("ARTN_FUSE3_RUN")) x.fuse3_run = atoi(e);
    if (const char *e = getenv("ARTN_PGEMM16")) x.pgemm16 = std::min(2, std::max(0, atoi(e)));
    if (const char *e = getenv("ARTN_PACKED_MIN_K")) x.packed_min_k = std::max(6, atoi(e));
    if (const char *e = getenv("ARTN_PACKED_MIN_AI")) x.packed_min_ai = std::max(1, atoi(e));
    if (const char *e = getenv("ARTN_PACKED")) x.packed = std::min(2, std::max(0, atoi(e)));
    if (const char *e = getenv("ARTN_ALT")) x.alt = std::min(2, std::max(0, atoi(e)));
    if (const char *e = getenv("ARTN_NARROW3")) x.narrow3 = atoi(e);
    if (const char *e = getenv("ARTN_BITS_3M")) x.bits_3m = std::min(2, std::max(0, atoi(e)));
#ifdef ARTN_DEV_SPLIT3
    if (const char *e = getenv("ARTN_SPLIT")) { int v = atoi(e); x.split = (v == 3 || v == 1) ? v : 0; }
#endif
#endif
    return x;
  }();
  return t;
}

static inline int ilog2_exact(int64_t v) {
  if (v <= 0 || (v & (v - 1))) return -1;
  int l = 0;
  while ((int64_t(1) << l) < v) ++l;
  return l;
}

// Returns 0 or a negative ARTN_E_* with `err` set.
static inline int validate(const ArtnStepDesc *d, std::string &err) {
  if (!d) { err = "null descriptor"; return ARTN_E_INVALID; }
  if (d->dtype != ARTN_C64 && d->dtype != ARTN_C128 && d->dtype != ARTN_C64_BF16) { err = "unknown dtype"; return ARTN_E_UNSUPPORTED; }
  if (d->n_labels < 0 || d->n_labels > ARTN_MAX_LABELS) { err = "n_labels out of range"; return ARTN_E_INVALID; }
  for (int l = 0; l < d->n_labels; ++l) {
    if (d->extent[l] < 1) { err = "label extent < 1"; return ARTN_E_INVALID; }
    bool a = d->stride_a[l] >= 0, b = d->stride_b[l] >= 0;
    if (!a && !b) { err = "label carried by neither input operand"; return ARTN_E_INVALID; }
  }
  // C must be dense row-major over its labels: sorted by stride, stride_i = prod of faster extents
  std::vector<std::pair<int64_t, int64_t>> cs;
  for (int l = 0; l < d->n_labels; ++l)
    if (d->stride_c[l] >= 0 && d->extent[l] > 1) cs.push_back({d->stride_c[l], d->extent[l]});
  std::sort(cs.begin(), cs.end());
  int64_t expect = 1;
  for (auto &p : cs) {
    if (p.first != expect) { err = "C is not dense row-major over its labels"; return ARTN_E_INVALID; }
    expect *= p.second;
  }
  return ARTN_OK;
}

static inline bool make_generic(const ArtnStepDesc *d, ArtnPlan &p, std::string &err) {
  ArtnGenericPlan &g = p.gen;
  memset(&g, 0, sizeof(g));
  std::vector<int> outl, redl;
  for (int l = 0; l < d->n_labels; ++l) {
    if (d->extent[l] == 1) continue;
    (d->stride_c[l] >= 0 ? outl : redl).push_back(l);
  }
  std::sort(outl.begin(), outl.end(), [&](int x, int y) { return d->stride_c[x] < d->stride_c[y]; });
  g.out_numel = 1;
  for (int l : outl) {
    int64_t e = d->extent[l];
    int64_t sa = d->stride_a[l] >= 0 ? d->stride_a[l] : 0, sb = d->stride_b[l] >= 0 ? d->stride_b[l] : 0;
    g.out_numel *= e;
    if (g.n_out > 0) { // C is dense, so only A and B decide whether two axes fuse
      int q = g.n_out - 1;
      if (sa == g.out_sA[q] * g.out_ext[q] && sb == g.out_sB[q] * g.out_ext[q]) { g.out_ext[q] *= e; continue; }
    }
    if (g.n_out >= ARTN_GEN_MAX_OUT) { err = "more than 64 unfusable output axes"; return false; }
    g.out_ext[g.n_out] = e; g.out_sA[g.n_out] = sa; g.out_sB[g.n_out] = sb;
    ++g.n_out;
  }
  // innermost reduction axis = the one with the smallest A stride (best locality)
  std::sort(redl.begin(), redl.end(), [&](int x, int y) {
    int64_t sx = d->stride_a[x] >= 0 ? d->stride_a[x] : d->stride_b[x];
    int64_t sy = d->stride_a[y] >= 0 ? d->stride_a[y] : d->stride_b[y];
    return sx < sy;
  });
  g.red_numel = 1;
  for (int l : redl) {
    if (g.n_red >= ARTN_GEN_MAX_RED) { err = "more than 32 reduction axes"; return false; }
    g.red_ext[g.n_red] = d->extent[l];
    g.red_sA[g.n_red] = d->stride_a[l] >= 0 ? d->stride_a[l] : 0;
    g.red_sB[g.n_red] = d->stride_b[l] >= 0 ? d->stride_b[l] : 0;
    g.red_numel *= d->extent[l];
    ++g.n_red;
  }
  p.kernel = ARTN_KERNEL_GENERIC;
  p.info.kernel = ARTN_KERNEL_GENERIC;
  int64_t blocks = (g.out_numel + ARTN_WG_THREADS - 1) / ARTN_WG_THREADS;
  p.info.grid = (int32_t)std::min<int64_t>(blocks, 1 << 20);
  p.info.n_tiles = blocks;
  return true;
}

// ----------------------------------------------------------------------------------------
// bit-GEMM planner (one step, or two fused steps d1: A,B1->C1 and d2: C1,B2->C2)
// ----------------------------------------------------------------------------------------
// An axis of the (possibly fused) problem.  Strides are element strides, -1 = absent.
// For a single step sB2 is absent and sC (the final output stride) == sC1.
struct Axis {
  int64_t ext;
  int64_t sA, sB1, sC1, sB2, sC;
  bool bit;
  bool gathered = false; // the label whose rows are gathered: always one whole outer axis
  bool k1() const { return sA >= 0 && sB1 >= 0 && sC1 < 0; }
  bool m1() const { return sA >= 0 && sB1 < 0 && sC1 >= 0; }
  bool n1() const { return sA < 0 && sB1 >= 0 && sC1 >= 0; }
  bool h1() const { return sA >= 0 && sB1 >= 0 && sC1 >= 0; }
  bool k2() const { return sC1 >= 0 && sB2 >= 0 && sC < 0; } // contracted by stage 2
  bool n2() const { return sC1 < 0 && sB2 >= 0 && sC >= 0; } // produced by stage 2
};

static inline void expand_axes(const ArtnStepDesc *d, std::vector<Axis> &ax, int gather_label = -1) {
  for (int l = 0; l < d->n_labels; ++l) {
    int64_t e = d->extent[l];
    if (e == 1 && l != gather_label) continue;
    int lg = l == gather_label ? -1 : ilog2_exact(e);
    int parts = lg > 0 ? lg : 1;
    for (int jb = 0; jb < parts; ++jb) {
      Axis a;
      a.bit = lg > 0;
      a.ext = lg > 0 ? 2 : e;
      int sh = lg > 0 ? jb : 0;
      a.sA = d->stride_a[l] >= 0 ? d->stride_a[l] << sh : -1;
      a.sB1 = d->stride_b[l] >= 0 ? d->stride_b[l] << sh : -1;
      a.sC1 = d->stride_c[l] >= 0 ? d->stride_c[l] << sh : -1;
      a.sB2 = -1;
      a.sC = a.sC1;
      a.gathered = l == gather_label;
      ax.push_back(a);
    }
  }
}

// Fuse the axes of d2 (whose operand A is d1's C) into `ax`.  Returns false if the two
// descriptors do not chain bit for bit.
static inline bool chain_axes(const ArtnStepDesc *d2, std::vector<Axis> &ax, std::string &why) {
  std::vector<Axis> ax2;
  expand_axes(d2, ax2); // here sA = position in C1, sB1 = stride in B2, sC1 = stride in C2
  for (auto &a : ax) a.sC = -1;
  std::vector<bool> used(ax2.size(), false);
  for (auto &a : ax) {
    if (a.sC1 < 0) continue;
    bool found = false;
    for (size_t i = 0; i < ax2.size(); ++i) {
      if (used[i] || ax2[i].sA != a.sC1) continue;
      if (ax2[i].ext != a.ext) { why = "fused steps disagree on an axis extent"; return false; }
      a.sB2 = ax2[i].sB1;
      a.sC = ax2[i].sC1;
      used[i] = true;
      found = true;
      break;
    }
    if (!found) { why = "second step does not carry every axis of the first result"; return false; }
  }
  for (size_t i = 0; i < ax2.size(); ++i) {
    if (used[i]) continue;
    if (ax2[i].sA >= 0) { why = "second step carries an axis the first result lacks"; return false; }
    Axis a = ax2[i];
    a.sA = -1; a.sB1 = -1; a.sC1 = -1;
    a.sB2 = ax2[i].sB1;
    a.sC = ax2[i].sC1;
    ax.push_back(a);
  }
  return true;
}

static inline bool make_bits(const ArtnStepDesc *d1, const ArtnStepDesc *d2, ArtnPlan &p, int n_cu,
                             int64_t min_tiles, int gather_label = -1) {
  auto c64 = [](int dt) { return dt == ARTN_C64 || dt == ARTN_C64_BF16; };
  // complex128 (artn_k_bits128): 16-byte elements -- half the elements per LDS region, one element per copy lane,
  // sub-tiles of 16 columns (4 lane bits), blocks of 8 result columns (at most 4 blocks: nt <= 5)
  const bool c128 = d1->dtype == ARTN_C128;
  if ((!c64(d1->dtype) && !c128) || (d2 && d2->dtype != d1->dtype)) { p.why_generic = "dtype is not complex64 / complex128"; return false; }
  if (c128 && (gather_label >= 0 || !tuning().bits128)) { p.why_generic = "complex128: no row gather on the state-streaming kernel"; return false; }
  const int esz = c128 ? 16 : 8, lb = c128 ? 4 : 5, pass_bits = c128 ? 8 : 9, shrink = c128 ? 1 : 0;
  const bool fused = d2 != nullptr;
  std::vector<Axis> ax;
  expand_axes(d1, ax, gather_label);
  if (fused && gather_label >= 0) { p.why_generic = "row gather in a fused pair"; return false; }
  if (fused && !chain_axes(d2, ax, p.why_generic)) return false;

  // ---- classify
  std::vector<int> K1, M1, N1, K2, N2, O; // O: outer-only axes (batch / non power-of-two free axes)
  for (int i = 0; i < (int)ax.size(); ++i) {
    const Axis &a = ax[i];
    if (a.sA >= 0 || a.sB1 >= 0) { // lives in stage 1
      if (a.k1()) {
        if (!a.bit) { p.why_generic = "contracted label with a non power-of-two extent"; return false; }
        K1.push_back(i);
      } else if (a.m1() || a.n1()) {
        if (fused && a.sB2 >= 0 && a.sC >= 0) { // batch axis of the second step: an outer axis with a B2 stride
          O.push_back(i);
          continue;
        }
        if (fused && a.k2()) {
          if (!a.bit) { p.why_generic = "contracted label with a non power-of-two extent"; return false; }
          K2.push_back(i);
          (a.m1() ? M1 : N1).push_back(i);
        } else if (fused && a.sC < 0) {
          p.why_generic = "label summed out of a single operand";
          return false;
        } else if (a.bit) {
          (a.m1() ? M1 : N1).push_back(i);
        } else {
          O.push_back(i);
        }
      } else if (a.h1()) {
        // batch axis of the first step; in a fused pair it must survive the second step (as a
        // free or again as a batch axis): a batch axis the second step contracts cannot be outer
        if (fused && a.sC < 0) { p.why_generic = "batch axis of the first step contracted by the second"; return false; }
        O.push_back(i);
      } else {
        p.why_generic = "label summed out of a single operand";
        return false;
      }
    } else { // only in stage 2's small operand
      if (!a.n2()) { p.why_generic = "label summed out of a single operand"; return false; }
      if (a.bit) N2.push_back(i); else O.push_back(i);
    }
  }
  const int k1 = (int)K1.size(), k2 = (int)K2.size();
  // up to 6 contracted bits run as one MFMA chain with the small operand in registers; a 7th
  // and 8th are looped over inside the stage (fragments reloaded per value, single steps only)
  if (k1 < 1 || k1 > (fused || c128 ? 6 : 8)) { p.why_generic = "contracted bit count outside 1..8 (1..6 when fused)"; return false; }
  if (fused && (k2 < 1 || k2 > 6)) { p.why_generic = "contracted bit count outside 1..6 (second step)"; return false; }
  auto byA = [&](int x, int y) { return ax[x].sA < ax[y].sA; };
  auto byC1 = [&](int x, int y) { return ax[x].sC1 < ax[y].sC1; };
  auto byC = [&](int x, int y) { return ax[x].sC < ax[y].sC; };
  std::sort(K1.begin(), K1.end(), byA);
  std::sort(M1.begin(), M1.end(), byA);
  std::sort(N1.begin(), N1.end(), byC1);
  std::sort(K2.begin(), K2.end(), byC1);
  std::sort(N2.begin(), N2.end(), byC);
  auto in_set = [](const std::vector<int> &v, int x) { return std::find(v.begin(), v.end(), x) != v.end(); };

  // contiguous run at the bottom of A (over K1 u M1 bits) and of the final C (bits that reach it)
  auto run_len = [&](bool in_side) {
    int r = 0;
    for (; r < tuning().run_max - shrink; ++r) {
      bool found = false;
      for (int i = 0; i < (int)ax.size() && !found; ++i) {
        const Axis &a = ax[i];
        if (!a.bit || in_set(O, i)) continue;
        int64_t s = in_side ? a.sA : a.sC;
        if (s == (int64_t(1) << r)) found = true;
      }
      if (!found) break;
    }
    return r;
  };
  int run_in = run_len(true), run_out = run_len(false);
  if (!c128 && (run_in < 1 || run_out < 1)) { p.why_generic = "no contiguous 16-byte run at the bottom of A or C"; return false; }

  // ---- choose the tile
  // Forced members: every M bit inside the input/output runs (16-byte lanes in >= 2^run x 8 B
  // contiguous pieces), every bit the second stage contracts.  A pass whose forced bits push a
  // tile to 2^13 elements needs 128 KiB of LDS and runs one workgroup per CU, which costs
  // more than shorter runs do: so run lengths are shortened (down to 32 B) until the tile
  // fits the 2^12 target; only if that is impossible is the first LDS-feasible choice taken.
  const int n1 = (int)N1.size(), n2 = (int)N2.size();
  std::vector<int> Mt, N1t, N2t;
  int T_in = 0, T_mid = 0, T_out = 0;
  auto try_runs = [&](int rin_bits, int rout_bits, bool need_target) {
    Mt.clear(); N1t.clear(); N2t.clear();
    const int64_t rin = int64_t(1) << rin_bits, rout = int64_t(1) << rout_bits;
    for (int i : M1)
      if (ax[i].sA < rin || (ax[i].sC >= 0 && ax[i].sC < rout) || in_set(K2, i)) Mt.push_back(i);
    for (int i : N1)
      if ((ax[i].sC >= 0 && ax[i].sC < rout) || in_set(K2, i)) N1t.push_back(i);
    for (int i : N2)
      if (ax[i].sC < rout) N2t.push_back(i);
    // grow N tiles towards full 16-column MFMA tiles (lowest result positions first)
    const int n_cap = c128 ? 5 : 6, n_lo = c128 ? 3 : 4;
    // (grow_nt: a single step that doubles its tensor takes one more result bit into the tile -- a 2^11-element input
    //  tile for a 2^12-element output tile, every input tile read once -- instead of visiting every input tile twice)
    const int grow = (!fused && n1 > k1) ? tuning().grow_nt : 0;
    int n1_target = std::min(n1, std::min(n_cap, std::max(std::min(k1 + grow, n_cap), n_lo)));
    for (int i : N1) { if ((int)N1t.size() >= n1_target) break; if (!in_set(N1t, i)) N1t.push_back(i); }
    int n2_target = std::min(n2, std::min(n_cap, std::max(k2, n_lo)));
    for (int i : N2) { if ((int)N2t.size() >= n2_target) break; if (!in_set(N2t, i)) N2t.push_back(i); }
    if ((int)N1t.size() > n_cap || (int)N2t.size() > n_cap) return false;
    auto sizes = [&](int mt) {
      T_in = k1 + mt;
      T_mid = mt + (int)N1t.size();
      T_out = fused ? T_mid - k2 + (int)N2t.size() : T_mid;
    };
    auto fits = [&](int mt) {
      sizes(mt);
      // a fused pair only pays with two workgroups per CU (2^12-element regions): with 2^13 it
      // measured slower than the two steps one after the other
      const int tmax = (fused ? ARTN_TILE_BITS_MAX - 1 : ARTN_TILE_BITS_MAX) - shrink;
      if (T_in > tmax || T_mid > tmax || T_out > tmax) return false;
      int r0 = fused ? std::max(T_in, T_out) : T_in;
      return ((long long)esz << r0) + ((long long)esz << T_mid) <= ARTN_LDS_BUDGET;
    };
    if (!fits((int)Mt.size())) return false;
    // 7-8 contracted bits: the kernel instantiation for them prefetches 2^13-element tiles
    const int target = (k1 > 6 ? ARTN_TILE_BITS_MAX : tuning().tile_target) - shrink;
    if (need_target && std::max(T_in, std::max(T_mid, T_out)) > target && (int)Mt.size() >= 5) return false;
    // grow M_t towards the target tile size (lowest A positions first), within the LDS budget
    for (int i : M1) {
      if (in_set(Mt, i)) continue;
      sizes((int)Mt.size());
      int biggest = std::max(T_in, std::max(T_mid, T_out));
      const bool enough = (int)Mt.size() >= 5 && T_in >= pass_bits && T_out >= pass_bits; // prefer one full copy pass per tile
      if ((enough && biggest >= target) || !fits((int)Mt.size() + 1)) break;
      Mt.push_back(i);
    }
    sizes((int)Mt.size());
    return true;
  };
  {
    bool done = false;
    const int r_in0 = run_in, r_out0 = run_out;
    for (int pass = 0; pass < 2 && !done; ++pass) {      // pass 0: must fit the target tile
      for (int cut = 0; cut <= 6 && !done; ++cut) {      // total bits shaved off the two runs
        for (int co = (cut + 1) / 2; co >= 0 && !done; --co) {
          const int ci = cut - co;                       // shave the output run first
          const int ri = r_in0 - ci, ro = r_out0 - co;
          const int r_floor = c128 ? 1 : 2, r_min = c128 ? 0 : 1; // (runs of 32 bytes at least, where the labels allow)
          if (ri < std::min(r_in0, r_floor) || ro < std::min(r_out0, r_floor) || ri < r_min || ro < r_min) continue;
          if (try_runs(ri, ro, pass == 0)) { run_in = ri; run_out = ro; done = true; }
        }
      }
    }
    if (!done) { p.why_generic = "forced tile bits exceed the LDS tile"; return false; }
  }
  const int mt = (int)Mt.size(), nt1 = (int)N1t.size(), nt2 = (int)N2t.size();
  if (mt < 5) { p.why_generic = "too few free A bits for a tile"; return false; }
  const int m2 = T_mid - k2; // free bits of stage 2's input tile
  if (fused && m2 < 5) { p.why_generic = "too few free bits for the second stage"; return false; }
  if (mt - lb > 9 || (fused && m2 - lb > 9)) { p.why_generic = "too many sub-tile bits"; return false; }

  ArtnBitsPlan &b = p.bits;
  memset(&b, 0, sizeof(b));
  b.n_stages = fused ? 2 : 1;
  b.T_in = T_in; b.T_mid = T_mid; b.T_out = T_out;
  b.r0_bits = fused ? std::max(T_in, T_out) : T_in;
  b.r1_bits = T_mid; b.T_mid2 = 0;
  b.run_in = run_in; b.run_out = run_out;
  b.stage_prio = tuning().stage_prio;
  b.c128 = c128 ? 1 : 0;

  // ---- tile-local orders: input (by A stride), mid (by C1 stride), output (by final C stride)
  std::vector<int> tin(K1), tmid(Mt), tout;
  tin.insert(tin.end(), Mt.begin(), Mt.end());
  tmid.insert(tmid.end(), N1t.begin(), N1t.end());
  std::sort(tin.begin(), tin.end(), byA);
  std::sort(tmid.begin(), tmid.end(), byC1);
  if (fused) {
    for (int i : tmid) if (!in_set(K2, i)) tout.push_back(i);
    tout.insert(tout.end(), N2t.begin(), N2t.end());
    std::sort(tout.begin(), tout.end(), byC);
  } else {
    tout = tmid;
  }
  auto pos = [](const std::vector<int> &v, int axis) { return (int)(std::find(v.begin(), v.end(), axis) - v.begin()); };
  for (int i = 0; i < T_in; ++i) b.in_stride[i] = ax[tin[i]].sA;
  for (int i = 0; i < T_out; ++i) b.out_stride[i] = ax[tout[i]].sC;
  for (int i = 0; i < run_in; ++i)
    if (b.in_stride[i] != (int64_t(1) << i)) { p.why_generic = "internal: input run broken"; return false; }
  for (int i = 0; i < run_out; ++i)
    if (b.out_stride[i] != (int64_t(1) << i)) { p.why_generic = "internal: output run broken"; return false; }

  // (not for the split-bf16 arithmetic, whose chains are built differently)
  const bool use_3m = tuning().bits_3m && d1->dtype == ARTN_C64 && tuning().split == 0;
  auto fill_stage = [&](ArtnStage &s, const std::vector<int> &Kx, const std::vector<int> &Mx,
                        const std::vector<int> &Nx, const std::vector<int> &tile_in,
                        const std::vector<int> &tile_out, bool second) {
    s.k = (int)Kx.size();
    s.nt = (int)Nx.size();
    s.wn_log2 = std::max(0, s.nt - (c128 ? 3 : 4));
    s.m3 = (use_3m && (s.k == 5 || s.k == 6) && s.nt >= 5 && gather_label < 0 && (int)K1.size() <= 6) ? 1 : 0;
    if (s.m3) s.wn_log2 = s.nt - 5;
    s.m_bits = (int)Mx.size();
    std::vector<int> Ms(Mx);
    std::sort(Ms.begin(), Ms.end(), [&](int x, int y) { return pos(tile_in, x) < pos(tile_in, y); });
    for (int i = 0; i < lb; ++i) { s.lane_in_pos[i] = pos(tile_in, Ms[i]); s.lane_out_pos[i] = pos(tile_out, Ms[i]); }
    for (int i = lb; i < s.m_bits; ++i) { s.msub_in_pos[i - lb] = pos(tile_in, Ms[i]); s.msub_out_pos[i - lb] = pos(tile_out, Ms[i]); }
    for (int i = 0; i < s.k; ++i) { s.k_in_pos[i] = pos(tile_in, Kx[i]); s.k_b_stride[i] = second ? ax[Kx[i]].sB2 : ax[Kx[i]].sB1; }
    for (int i = 0; i < s.nt; ++i) { s.n_out_pos[i] = pos(tile_out, Nx[i]); s.n_b_stride[i] = second ? ax[Nx[i]].sB2 : ax[Nx[i]].sB1; }
    // output-region swizzle
    s.swz_n = 0;
    bool taken[4] = {!c128, false, false, false}; // complex64: position 0 (the 8-byte half) is never a target
    for (int i = 0; i < 4; ++i) if (s.lane_out_pos[i] < 4) taken[s.lane_out_pos[i]] = true;
    for (int i = 0; i < 4 && tuning().swizzle; ++i) {
      if (s.lane_out_pos[i] < 4) continue;
      int f = -1;
      for (int c = 0; c < 4; ++c) if (!taken[c]) { f = c; break; }
      if (f < 0) break;
      taken[f] = true;
      s.swz_src[s.swz_n] = s.lane_out_pos[i];
      s.swz_dst[s.swz_n] = f;
      ++s.swz_n;
    }
  };
  fill_stage(b.st[0], K1, Mt, N1t, tin, tmid, false);
  if (fused) {
    std::vector<int> M2t;
    for (int i : tmid) if (!in_set(K2, i)) M2t.push_back(i);
    fill_stage(b.st[1], K2, M2t, N2t, tmid, tout, true);
  }
  // the 3M kernels are separate instantiations in which EVERY 5- or 6-bit stage runs three products: all or
  // none, and only while the fragments of both stages fit the register file next to three accumulators
  {
    bool any = false, all = true;
    int frag = 0;
    // (the SECOND stage of a pair that keeps at most 4 of its result bits in the tile -- a pair that shrinks its tensor -- counts as
    //  three-product capable: it runs on 16 x 16 x 4 blocks, ArtnBitsPlan::narrow3 = 2, below)
    const bool narrow2 = fused && use_3m && tuning().narrow3 && gather_label < 0 && (b.st[1].k == 5 || b.st[1].k == 6) && b.st[1].nt <= 4 &&
                         b.st[0].k >= 2 && b.st[0].k <= 6 && (int)K1.size() <= 6 && !(b.T_in == 12 && b.T_out == 12);
    for (int q = 0; q < b.n_stages; ++q) {
      const bool wide = b.st[q].k == 5 || b.st[q].k == 6;
      const bool ok3 = b.st[q].m3 || (q == 1 && narrow2);
      if (wide) { any = any || ok3; all = all && ok3; }
      frag += (q == 1 && narrow2) ? 3 << (b.st[q].k - 2) : 2 << (std::min(b.st[q].k, 6) - 1);
    }
    if (frag > tuning().m3_frag) all = false; // fragments of both stages next to three accumulators: 6+4 (80) fits, 6+5 (96) spills 24-36 registers
    // (a fused pair with a 6-bit 3M stage compiles with 8 spilled registers and still wins: 6+4 pairs of n30
    //  6.98 -> 5.98 ms; ARTN_BITS_3M=1 excludes them)
    if (b.n_stages == 2 && (b.st[0].k == 6 || b.st[1].k == 6) && tuning().bits_3m < 2) all = false;
    b.m3 = (any && all) ? 1 : 0;
    if (!b.m3)
      for (int q = 0; q < b.n_stages; ++q)
        if (b.st[q].m3) { b.st[q].m3 = 0; b.st[q].wn_log2 = std::max(0, b.st[q].nt - 4); } // (never complex128: no 3M there)
    // the 2- to 4-bit stage of a 3M pair runs three products too, on 16 x 16 x 4 blocks (ArtnStage::m3 = 2: the M3 instantiations
    // treat every such stage that way; waves split column blocks of 16 as in the 4M chains)
    if (b.m3)
      for (int q = 0; q < b.n_stages; ++q)
        if (b.st[q].k >= 2 && b.st[q].k <= 4) b.st[q].m3 = 2;
    if (b.m3 && narrow2) { // (artn_k_bits<KB1, KB2, ..., M3, ..., N3 = 2>)
      b.narrow3 = 2;
      b.st[1].m3 = 2;
      b.st[1].wn_log2 = 0;
    }
    // a single step with 5-6 contracted bits that brings at most 4 result bits into the tile (the steps that SHRINK their
    // tensor): on 32 x 32 blocks at most 16 of 32 rows are results and the stage runs four products; the 16 x 16 x 4 form
    // (the stage of artn_k_wide on four waves) halves the rows and runs three -- n53's 2^30 -> 2^27 step is MFMA-bound on
    // the wasted rows otherwise
    if (!fused && use_3m && tuning().narrow3 && gather_label < 0 && (b.st[0].k == 5 || b.st[0].k == 6) && (int)K1.size() <= 6 &&
        b.st[0].nt <= 4 && !b.st[0].m3) {
      b.narrow3 = 1;
      b.st[0].m3 = 2;
      b.st[0].wn_log2 = 0;
    }
  }

  // ---- outer axes: N-outer fastest (tiles sharing an A tile run together), then M-outer by A
  //      stride, then batch / generic axes
  std::vector<int> outer;
  p.stage1_repeats = 1;
  for (int i : N2) if (!in_set(N2t, i)) { outer.push_back(i); p.stage1_repeats *= ax[i].ext; }
  for (int i : N1) if (!in_set(N1t, i)) outer.push_back(i);
  for (int i : M1) if (!in_set(Mt, i)) outer.push_back(i);
  std::sort(O.begin(), O.end(), [&](int x, int y) {
    auto key = [&](int z) { return ax[z].sA >= 0 ? ax[z].sA : (ax[z].sB1 >= 0 ? ax[z].sB1 : ax[z].sB2); };
    return key(x) < key(y);
  });
  outer.insert(outer.end(), O.begin(), O.end());
  b.n_tiles = 1;
  b.gather_dim = -1;
  b.nt_loads = 0;
  b.rows_a = b.rows_b = nullptr;
  b.src_rows_a = b.src_rows_b = 0;
  b.gather_err = nullptr;
  int64_t a_rereads = 1, stage1_reruns = 1;
  for (int i : outer) {
    const Axis &a = ax[i];
    ArtnOuterDim od;
    od.ext = a.ext;
    od.sA = a.sA >= 0 ? a.sA : 0;
    od.sB1 = a.sB1 >= 0 ? a.sB1 : 0;
    od.sB2 = a.sB2 >= 0 ? a.sB2 : 0;
    od.sC = a.sC >= 0 ? a.sC : 0;
    od.log2ext = a.gathered ? -1 : ilog2_exact(a.ext); // the gathered axis is decoded the slow way, on its own
    od.pad_ = 0;
    b.n_tiles *= a.ext;
    if (a.sA < 0) a_rereads *= a.ext;
    if (a.sA < 0 && a.sB1 < 0) stage1_reruns *= a.ext; // an outer result label of the SECOND step: stage 1 runs again per value
    if (b.n_outer > 0) { // merge with the previous dim when both are powers of two and contiguous everywhere
      ArtnOuterDim &pr = b.outer[b.n_outer - 1];
      auto okf = [&](int64_t ps, int64_t ns) { return (ps == 0 && ns == 0) || (ps != 0 && ns == ps * pr.ext); };
      if (pr.log2ext >= 0 && od.log2ext >= 0 && okf(pr.sA, od.sA) && okf(pr.sB1, od.sB1) && okf(pr.sB2, od.sB2) &&
          okf(pr.sC, od.sC) && pr.log2ext + od.log2ext < 31) {
        pr.ext *= od.ext;
        pr.log2ext += od.log2ext;
        continue;
      }
    }
    if (b.n_outer >= ARTN_MAX_OUTER) { p.why_generic = "too many outer axes"; return false; }
    if (a.gathered) b.gather_dim = b.n_outer;
    b.outer[b.n_outer++] = od;
  }

  // A batch axis (carried by A, B and C: the shared row label of the sparse path) makes the small
  // operand a function of the tile.  Grid-stride order then changes it at every tile of a
  // workgroup (a reload of up to 256 fragment registers per lane from global memory); in
  // contiguous ranges the batch axes, which are the slowest tile digits, change once per row.
  if (gather_label >= 0 && b.gather_dim < 0) { p.why_generic = "gathered label is not an outer axis"; return false; }
  b.nt_loads = (!c128 && a_rereads == 1 && run_in >= 4 && b.n_tiles >= (1 << 14) && tuning().nt) ? 1 : 0;
  // (not when the small operand also has outer free bits: those are the fastest tile digits, in
  //  grid-stride order a workgroup keeps its value of them -- and its fragments -- while a
  //  contiguous range would step through them tile by tile)
  b.blocked = 0;
  for (int i = 0; i < b.n_outer; ++i)
    if (((b.outer[i].sA != 0 && b.outer[i].sB1 != 0) || (b.outer[i].sB2 != 0 && b.outer[i].sC != 0 && b.outer[i].sA != 0)) &&
        a_rereads == 1)
      b.blocked = 1;

  // ---- envelope checks
  // the copy phases move 16 bytes (two elements) per lane and need every thread busy
  if (b.T_in < pass_bits || b.T_out < 5 - shrink) { p.why_generic = "tile smaller than one copy pass"; return false; }
  for (int i = 1; i < b.T_in && !c128; ++i) if (b.in_stride[i] & 1) { p.why_generic = "odd A stride"; return false; }
  for (int i = 1; i < b.T_out && !c128; ++i) if (b.out_stride[i] & 1) { p.why_generic = "odd C stride"; return false; }
  for (int i = 0; i < b.n_outer && !c128; ++i)
    if ((b.outer[i].sA & 1) || (b.outer[i].sC & 1)) { p.why_generic = "odd outer stride"; return false; }
  {
    // per-lane byte offsets inside the kernel are 32-bit: copy chunks span tile bits 1..8,
    // the small operands are addressed by their N_t / K bits
    int64_t si = 0, so = 0;
    for (int i = 1 - shrink; i <= 8 - shrink; ++i) { if (i < b.T_in) si += b.in_stride[i]; if (i < b.T_out) so += b.out_stride[i]; }
    const int64_t lim = (int64_t(1) << (28 - shrink)) - 1; // elements: * 8 B (16 B) < 2^31
    // (the small operands are addressed with 64-bit offsets: a "small" operand of a split-K
    // step can itself be gigabytes)
    if (si > lim || so > lim) { p.why_generic = "lane offsets exceed 32 bits"; return false; }
  }
  if (b.n_tiles < min_tiles) { p.why_generic = "too few tiles to fill the chip"; return false; }

  p.kernel = ARTN_KERNEL_BITS_MFMA;
  ArtnStepInfo &f = p.info;
  f.kernel = ARTN_KERNEL_BITS_MFMA;
  f.k_bits = k1; f.m_tile_bits = mt; f.n_tile_bits = nt1;
  f.tile_in_bits = b.T_in; f.tile_out_bits = b.T_out;
  f.run_in_bits = run_in; f.run_out_bits = run_out;
  // two tile regions + the sub-tile offset tables of both stages (8 bytes per sub-tile)
  {
    int pow2_bits = 0;
    for (int i = 0; i < b.n_outer && b.outer[i].log2ext >= 0; ++i) pow2_bits += b.outer[i].log2ext;
    if (pow2_bits > 32) { p.why_generic = "more than 2^32 tiles"; return false; }
    const int64_t off_tab = 512LL * 8 + 32 * 32; // tile-offset nibble tables (8 x 16 x 4 longs) + grid-stride deltas
    // 7-8 contracted bits with a single 32 x 16 result block per tile (few free bits on both
    // sides: the chunk steps of the sparse path) would keep one wave busy; the four waves take a
    // quarter of the chain each and three partial blocks (4 KiB each) are summed through LDS
    b.ksplit = (k1 > 6 && !fused && mt == 5 && b.st[0].wn_log2 == 0) ? 1 : 0;
    // arithmetic of the chains: ARTN_C64_BF16 asks for plain bf16 operands; complex64 uses the
    // split (three bf16 pieces) when tuning().split says so; 7-8 contracted bits stay fp32
    b.split = d1->dtype == ARTN_C64_BF16 ? 1 : ((k1 > 6 || c128) ? 0 : tuning().split);
    f.lds_bytes = (int32_t)(((long long)esz << b.r0_bits) + ((long long)esz << b.T_mid) + (8LL << (mt - lb)) + (fused ? (8LL << (m2 - lb)) : 0) + off_tab +
                            (b.ksplit ? 3 * 4096 + 16 : 0));
  }
  f.n_tiles = b.n_tiles;
  f.a_rereads = a_rereads;
  f.stage1_reruns = fused ? (int32_t)std::min<int64_t>(stage1_reruns, INT32_MAX) : 1;
  f.k2_bits = k2; f.n2_tile_bits = nt2; f.tile_mid_bits = b.T_mid;
  int wg_per_cu = std::max(1, std::min(tuning().wg_per_cu, (160 * 1024) / f.lds_bytes));
  f.grid = (int32_t)std::min<int64_t>(b.n_tiles, (int64_t)n_cu * wg_per_cu);
  return true;
}


// ----------------------------------------------------------------------------------------
// two-operand LDS GEMM planner (artn_k_gemm)
// ----------------------------------------------------------------------------------------
// Bits of one step: K (contracted), M (first operand and C), N (second operand and C); everything
// else (batch axes, non power-of-two free axes) is enumerated by the tile index.  A workgroup owns a
// C tile of 2^mt x 2^nt elements and walks the contracted index in chunks of 2^kc: per chunk the
// [2^kc][2^mt] piece of the first operand and the [2^kc][2^nt] piece of the second are copied to LDS
// (16-byte lanes, runs as long as the low address bits of each operand allow), every wave multiplies
// its 32-row x 16-column MFMA blocks, accumulators stay in registers across all chunks.
static inline bool make_gemm(const ArtnStepDesc *d, ArtnPlan &p, int n_cu, int64_t min_tiles, bool only_if_preferred,
                             int use_3m = -1 /* -1: as tuning() says */, int gather_label = -1) {
  if (d->dtype != ARTN_C64 && d->dtype != ARTN_C64_BF16) { p.why_generic = "dtype is not complex64"; return false; }
  if (gather_label >= 0 && d->dtype != ARTN_C64) { p.why_generic = "row gather: complex64 arithmetic only"; return false; }
  std::vector<Axis> ax;
  expand_axes(d, ax, gather_label);
  std::vector<int> K, M, N, O;
  for (int i = 0; i < (int)ax.size(); ++i) {
    const Axis &a = ax[i];
    if (a.k1()) {
      if (!a.bit) { p.why_generic = "contracted label with a non power-of-two extent"; return false; }
      K.push_back(i);
    } else if (a.m1()) {
      (a.bit ? M : O).push_back(i);
    } else if (a.n1()) {
      (a.bit ? N : O).push_back(i);
    } else if (a.h1()) {
      O.push_back(i);
    } else {
      p.why_generic = "label summed out of a single operand";
      return false;
    }
  }
  const int k = (int)K.size();
  const bool bf16 = d->dtype == ARTN_C64_BF16;
  int kc = bf16 ? ARTN_GEMM_KC + 1 : ARTN_GEMM_KC; // bf16 MFMAs take 8 contracted values each: chunks of 32
  int pitch = ARTN_GEMM_PITCH_LOG2;
  if (k < kc) { p.why_generic = "fewer contracted bits than one LDS chunk of the GEMM kernel"; return false; }
  if (k - kc > ARTN_GEMM_MAX_KO) { p.why_generic = "too many contracted bits"; return false; }
  if (only_if_preferred) {
    int64_t nb = 1;
    for (int i = 0; i < (int)ax.size(); ++i) if (ax[i].sB1 >= 0 && (ax[i].k1() || ax[i].n1())) nb *= ax[i].ext;
    int64_t na = 1;
    for (int i = 0; i < (int)ax.size(); ++i) if (ax[i].sA >= 0 && (ax[i].k1() || ax[i].m1())) na *= ax[i].ext;
    const bool big_second = std::min(na, nb) >= (int64_t(1) << 15);
    // measured on n53 / sparse-state steps: with 7-8 contracted bits the GEMM kernel wins once both
    // operands bring 5+ free bits (32-column blocks: three real products per complex product; 89 against
    // 81 TFLOP/s on a 2^30 x 2^12 step); with fewer, or with up to 6 contracted bits, one pass of the
    // state-streaming kernel with the second operand in registers is faster (107 against 96 at 6 bits)
    const int free_small = (int)std::min(M.size(), N.size());
    if (!(k > 8 || (k > 6 && free_small >= 5) || big_second)) { p.why_generic = "state-streaming kernel preferred"; return false; }
  }
  // the first operand supplies the 32-row MFMA blocks: it needs 5 free bits
  const bool swapped = M.size() < 5 && N.size() >= 5;
  if (swapped) {
    for (auto &a : ax) std::swap(a.sA, a.sB1);
    std::swap(M, N);
  }
  const int m = (int)M.size(), n = (int)N.size();
  if (m < 5) { p.why_generic = "too few free bits for an MFMA tile"; return false; }

  auto in_set = [](const std::vector<int> &v, int x) { return std::find(v.begin(), v.end(), x) != v.end(); };
  auto has_unit = [&](const std::vector<int> &s1, const std::vector<int> &s2, int which) {
    for (const std::vector<int> *s : {&s1, &s2})
      for (int i : *s) {
        const int64_t st = which == 0 ? ax[i].sA : which == 1 ? ax[i].sB1 : ax[i].sC;
        if (st == 1) return true;
      }
    return false;
  };
  if (!has_unit(K, M, 0) || !has_unit(K, N, 1) || !has_unit(M, N, 2)) {
    p.why_generic = "no contiguous 16-byte run at the bottom of an operand";
    return false;
  }
  // ---- forced tile members: the bits inside the contiguous runs of A, B and C (2^r elements)
  int ra = 4, rb = 4, rc = 4;
  std::vector<int> Kf, Mf, Nf;
  for (;;) {
    Kf.clear(); Mf.clear(); Nf.clear();
    const int64_t la = int64_t(1) << ra, lb = int64_t(1) << rb, lc = int64_t(1) << rc;
    for (int i : K) if (ax[i].sA < la || ax[i].sB1 < lb) Kf.push_back(i);
    for (int i : M) if (ax[i].sA < la || ax[i].sC < lc) Mf.push_back(i);
    for (int i : N) if (ax[i].sB1 < lb || ax[i].sC < lc) Nf.push_back(i);
    int *shrink = nullptr;
    if ((int)Kf.size() > kc) shrink = rb >= ra ? &rb : &ra;
    else if ((int)Mf.size() > 7) shrink = rc >= ra ? &rc : &ra;
    else if ((int)Nf.size() > 7) shrink = rc >= rb ? &rc : &rb;
    if (!shrink) break;
    if (*shrink <= 1) { p.why_generic = "forced tile bits exceed the GEMM tile"; return false; }
    --*shrink;
  }
  auto byA = [&](int x, int y) { return ax[x].sA < ax[y].sA; };
  auto byB = [&](int x, int y) { return ax[x].sB1 < ax[y].sB1; };
  auto byC = [&](int x, int y) { return ax[x].sC < ax[y].sC; };
  std::sort(K.begin(), K.end(), byA);
  std::sort(M.begin(), M.end(), byA);
  std::sort(N.begin(), N.end(), byB);
  int64_t outer_ext = 1;
  for (int i : O) outer_ext *= ax[i].ext;
  // (bf16: 128 x 64 tiles at most -- twice the chunk depth has to fit the same prefetch registers; so does
  //  the 3M arithmetic, whose 32 x 32 blocks carry three accumulators)
  const bool want_m3 = !bf16 && (use_3m < 0 ? tuning().gemm_3m != 0 : use_3m != 0) && n >= 5;
  int mt = std::min(m, 7), nt = std::min(n, (bf16 || want_m3) ? 6 : 7);
  if ((int)Nf.size() > nt) {
    if (bf16) { p.why_generic = "forced tile bits exceed the bf16 GEMM tile"; return false; }
    nt = std::min(n, 7); // (3M gives way)
  }
  // too few tiles to fill the chip: smaller tiles (down to what the runs force)
  while (((int64_t(1) << (m - mt + n - nt)) * outer_ext) < n_cu) {
    if (mt >= nt + 1 && mt > 5 && mt > (int)Mf.size()) --mt;
    else if (nt > 4 && nt > (int)Nf.size()) --nt;
    else if (mt > 5 && mt > (int)Mf.size()) --mt;
    else break;
  }
  // tiles of one 32 x 32 block (closing steps: two big tensors down to a few amplitudes, 2^16 contracted values per
  // tile): the same 16 KiB images hold 2^6 contracted values of 2^5 rows instead of 2^4 of 2^7 -- a quarter of the
  // chunk iterations, each of them a latency chain (loads one chunk ahead), for the same bytes
  if (!bf16 && mt <= 5 && nt <= 5 && k >= ARTN_GEMM_KC_TALL && (int)Kf.size() <= ARTN_GEMM_KC_TALL && tuning().gemm_tall &&
      gather_label < 0) { // (no row-gather instantiation of the tall-chunk kernel)
    kc = ARTN_GEMM_KC_TALL;
    pitch = ARTN_GEMM_PITCH_TALL_LOG2;
  }
  std::vector<int> Kc(Kf), Mt(Mf), Nt(Nf);
  for (int i : K) { if ((int)Kc.size() >= kc) break; if (!in_set(Kc, i)) Kc.push_back(i); }
  for (int i : M) { if ((int)Mt.size() >= mt) break; if (!in_set(Mt, i)) Mt.push_back(i); }
  for (int i : N) { if ((int)Nt.size() >= nt) break; if (!in_set(Nt, i)) Nt.push_back(i); }
  mt = (int)Mt.size(); nt = (int)Nt.size();
  if (mt < 5) { p.why_generic = "too few free bits for an MFMA tile"; return false; }

  ArtnGemmPlan &g = p.gemm;
  memset(&g, 0, sizeof(g));
  g.mt = mt; g.nt = nt; g.kc = kc; g.n_ko = k - kc; g.swapped = swapped ? 1 : 0;
  g.pitch_log2 = pitch;
  g.split = bf16 ? 1 : 0;
  g.gather_dim = -1;
  // ---- waves: blocks per wave from the allowed instantiations (1,1) (1,2) (2,2) (1,4) (2,4); 3M: (1,1) (1,2) (2,1)
  g.m3 = (want_m3 && nt >= 5 && nt <= 6) ? 1 : 0;
  if (g.m3) {
    const int mbl = mt - 5, nbl = nt - 5;
    int best = 1 << 30;
    for (int wm = 0; wm <= std::min(2, mbl); ++wm) {
      const int wn = std::min(2 - wm, nbl);
      const int MB = mbl - wm, NB = nbl - wn;
      if (MB + NB > 1) continue;
      const int cost = (1 << MB) + (1 << NB) + 8 * (2 - wm - wn);
      if (cost < best) { best = cost; g.wm_log2 = wm; g.wn_log2 = wn; g.mb_log2 = MB; g.nb_log2 = NB; }
    }
    if (best == (1 << 30)) g.m3 = 0;
  }
  if (!g.m3) {
    const int mbl = mt - 5, nbl = std::max(nt - 4, 0);
    int best = 1 << 30;
    for (int wm = 0; wm <= std::min(2, mbl); ++wm) {
      const int wn = std::min(2 - wm, nbl);
      const int MB = mbl - wm, NB = nbl - wn;
      const bool allowed = (MB == 0 && NB <= 2) || (MB == 1 && (NB == 1 || NB == 2));
      if (!allowed) continue;
      const int cost = (1 << MB) + (1 << NB) + 8 * (2 - wm - wn);
      if (cost < best) { best = cost; g.wm_log2 = wm; g.wn_log2 = wn; g.mb_log2 = MB; g.nb_log2 = NB; }
    }
    if (best == (1 << 30)) { p.why_generic = "internal: no wave layout for the GEMM tile"; return false; }
  }
  g.wk_log2 = 2 - g.wm_log2 - g.wn_log2;
  // ---- LDS images: [kc value][m_local] and [kc value][n_local].  Local bit order = stride order in the
  //      operand: the copy lanes (lowest-stride bits of a chunk) then write neighbouring rows and
  //      chunk values -- ordered by C stride instead, a bf16 step ran 8x slower on LDS bank conflicts
  std::sort(Kc.begin(), Kc.end(), byA);
  std::sort(Mt.begin(), Mt.end(), byA);
  std::sort(Nt.begin(), Nt.end(), byB);
  auto pos = [](const std::vector<int> &v, int axis) { return (int)(std::find(v.begin(), v.end(), axis) - v.begin()); };
  std::vector<int> tA(Kc), tB(Kc), tC(Mt);
  tA.insert(tA.end(), Mt.begin(), Mt.end());
  tB.insert(tB.end(), Nt.begin(), Nt.end());
  tC.insert(tC.end(), Nt.begin(), Nt.end());
  std::sort(tA.begin(), tA.end(), byA);
  std::sort(tB.begin(), tB.end(), byB);
  std::sort(tC.begin(), tC.end(), byC);
  g.ta_bits = mt + kc; g.tb_bits = nt + kc; g.tc_bits = mt + nt;
  // byte offset of (row bit i | chunk bit q) in an image: fp32 [kc][128 rows] x 8 B; bf16 [kc >> 2][128 rows][kc & 3] x 4 B
  auto lds_row = [&](int i) { return bf16 ? 16 << i : 8 << i; };
  auto lds_kc = [&](int q) { return bf16 ? (q < 2 ? 4 << q : (16 << ARTN_GEMM_PITCH_LOG2) << (q - 2)) : (8 << pitch) << q; };
  for (int b = 0; b < g.ta_bits; ++b) {
    g.a_stride[b] = ax[tA[b]].sA;
    g.a_lds[b] = in_set(Mt, tA[b]) ? lds_row(pos(Mt, tA[b])) : lds_kc(pos(Kc, tA[b]));
  }
  for (int b = 0; b < g.tb_bits; ++b) {
    g.b_stride[b] = ax[tB[b]].sB1;
    g.b_lds[b] = in_set(Nt, tB[b]) ? lds_row(pos(Nt, tB[b])) : lds_kc(pos(Kc, tB[b]));
  }
  if (g.a_stride[0] != 1 || g.b_stride[0] != 1) { p.why_generic = "internal: operand run broken"; return false; }
  for (int b = 0; b < g.tc_bits; ++b) g.out_stride[b] = ax[tC[b]].sC;
  if (g.tc_bits > 0 && g.out_stride[0] != 1) { p.why_generic = "internal: output run broken"; return false; }
  for (int i = 0; i < mt; ++i) g.m_pos[i] = pos(tC, Mt[i]);
  for (int i = 0; i < nt; ++i) g.n_pos[i] = pos(tC, Nt[i]);
  // looped contracted bits, lowest A stride first
  {
    int q = 0;
    for (int i : K) if (!in_set(Kc, i)) { g.ko_sA[q] = ax[i].sA; g.ko_sB[q] = ax[i].sB1; ++q; }
  }
  // result-image swizzle: the 16 lanes of a ds_write_b64 group differ in m_local bits 0..3
  {
    bool taken[4] = {true, false, false, false};
    for (int i = 0; i < 4; ++i) if (g.m_pos[i] < 4) taken[g.m_pos[i]] = true;
    // n_local bit 0 is written by the same lane in two instructions, it may share a window position;
    // positions held by other low tile bits stay usable as XOR targets (the map stays a bijection)
    for (int i = 0; i < 4 && tuning().swizzle; ++i) {
      if (g.m_pos[i] < 4) continue;
      int f = -1;
      for (int c = 1; c < 4; ++c) if (!taken[c]) { f = c; break; }
      if (f < 0) break;
      taken[f] = true;
      g.swz_src[g.swz_n] = g.m_pos[i];
      g.swz_dst[g.swz_n] = f;
      ++g.swz_n;
    }
  }
  // ---- outer axes: N-outer fastest (tiles sharing an A row block run together), M-outer, the rest
  std::vector<int> outer;
  for (int i : N) if (!in_set(Nt, i)) outer.push_back(i);
  for (int i : M) if (!in_set(Mt, i)) outer.push_back(i);
  std::sort(O.begin(), O.end(), [&](int x, int y) {
    auto key = [&](int z) { return ax[z].sA >= 0 ? ax[z].sA : ax[z].sB1; };
    return key(x) < key(y);
  });
  outer.insert(outer.end(), O.begin(), O.end());
  g.n_tiles = 1;
  int64_t a_rereads = 1;
  for (int i : outer) {
    const Axis &a = ax[i];
    ArtnOuterDim od;
    od.ext = a.ext;
    od.sA = a.sA >= 0 ? a.sA : 0;
    od.sB1 = a.sB1 >= 0 ? a.sB1 : 0;
    od.sB2 = 0;
    od.sC = a.sC >= 0 ? a.sC : 0;
    od.log2ext = a.gathered ? -1 : ilog2_exact(a.ext); // the gathered axis is decoded the slow way, on its own
    od.pad_ = 0;
    g.n_tiles *= a.ext;
    if (a.sA < 0) a_rereads *= a.ext;
    if (g.n_outer > 0) {
      ArtnOuterDim &pr = g.outer[g.n_outer - 1];
      auto okf = [&](int64_t ps, int64_t ns) { return (ps == 0 && ns == 0) || (ps != 0 && ns == ps * pr.ext); };
      if (pr.log2ext >= 0 && od.log2ext >= 0 && okf(pr.sA, od.sA) && okf(pr.sB1, od.sB1) && okf(pr.sC, od.sC) &&
          pr.log2ext + od.log2ext < 31) {
        pr.ext *= od.ext;
        pr.log2ext += od.log2ext;
        continue;
      }
    }
    if (g.n_outer >= ARTN_MAX_OUTER) { p.why_generic = "too many outer axes"; return false; }
    if (a.gathered) g.gather_dim = g.n_outer;
    g.outer[g.n_outer++] = od;
  }
  if (gather_label >= 0 && g.gather_dim < 0) { p.why_generic = "gathered label is not an outer axis"; return false; }
  // ---- envelope: 16-byte lanes, 32-bit per-lane offsets
  for (int b = 1; b < g.ta_bits; ++b) if (g.a_stride[b] & 1) { p.why_generic = "odd A stride"; return false; }
  for (int b = 1; b < g.tb_bits; ++b) if (g.b_stride[b] & 1) { p.why_generic = "odd B stride"; return false; }
  for (int b = 1; b < g.tc_bits; ++b) if (g.out_stride[b] & 1) { p.why_generic = "odd C stride"; return false; }
  for (int i = 0; i < g.n_outer; ++i)
    if ((g.outer[i].sA & 1) || (g.outer[i].sB1 & 1) || (g.outer[i].sC & 1)) { p.why_generic = "odd outer stride"; return false; }
  for (int q = 0; q < g.n_ko; ++q) if ((g.ko_sA[q] & 1) || (g.ko_sB[q] & 1)) { p.why_generic = "odd contracted stride"; return false; }
  {
    int64_t sa = 0, sb = 0, sc = 0;
    for (int b = 1; b <= 8; ++b) {
      if (b < g.ta_bits) sa += g.a_stride[b];
      if (b < g.tb_bits) sb += g.b_stride[b];
      if (b < g.tc_bits) sc += g.out_stride[b];
    }
    const int64_t lim = (int64_t(1) << 28) - 1;
    if (sa > lim || sb > lim || sc > lim) { p.why_generic = "lane offsets exceed 32 bits"; return false; }
  }
  {
    int pow2_bits = 0;
    for (int i = 0; i < g.n_outer && g.outer[i].log2ext >= 0; ++i) pow2_bits += g.outer[i].log2ext;
    if (pow2_bits > 32) { p.why_generic = "more than 2^32 tiles"; return false; }
  }
  if (g.n_tiles < min_tiles && !(k > 8)) { p.why_generic = "too few tiles to fill the chip"; return false; }

  p.kernel = ARTN_KERNEL_GEMM_MFMA;
  ArtnStepInfo &f = p.info;
  f.kernel = ARTN_KERNEL_GEMM_MFMA;
  f.k_bits = k; f.m_tile_bits = mt; f.n_tile_bits = nt;
  f.tile_in_bits = g.ta_bits; f.tile_out_bits = g.tc_bits;
  f.run_in_bits = ra; f.run_out_bits = rc;
  const int64_t stage = 2 * (8LL << (ARTN_GEMM_PITCH_LOG2 + ARTN_GEMM_KC));
  const int64_t epi = 8LL << std::min(g.tc_bits, ARTN_GEMM_EPI_BITS);
  f.lds_bytes = (int32_t)(std::max(2 * stage, epi) + 512LL * 8 + 32 * 32 + 16LL * ARTN_GEMM_MAX_KO);
  f.n_tiles = g.n_tiles;
  f.a_rereads = a_rereads;
  const int wg_per_cu = std::max(1, std::min(2, (160 * 1024) / f.lds_bytes));
  f.grid = (int32_t)std::min<int64_t>(g.n_tiles, (int64_t)n_cu * wg_per_cu);
  return true;
}


// ----------------------------------------------------------------------------------------
// complex128 on v_mfma_f64_16x16x4_f64 (artn_k_gemm128): the same two-operand LDS GEMM with 16-byte
// elements -- one element per copy lane, chunks of 2^3 contracted values, MFMA blocks of 16 rows (m) x 8
// complex columns (n), two row blocks x 2^nb column blocks per wave.  ArtnGemmPlan is shared: split = 2.
// ----------------------------------------------------------------------------------------
#define ARTN_GEMM128_KC 3
#define ARTN_GEMM128_EPI_BITS 12 /* result passes of 2^12 elements (64 KiB) */
static inline bool make_gemm128(const ArtnStepDesc *d, ArtnPlan &p, int n_cu, int64_t min_tiles, int gather_label = -1) {
  if (d->dtype != ARTN_C128) { p.why_generic = "dtype is not complex128"; return false; }
  std::vector<Axis> ax;
  expand_axes(d, ax, gather_label);
  std::vector<int> K, M, N, O;
  for (int i = 0; i < (int)ax.size(); ++i) {
    const Axis &a = ax[i];
    if (a.k1()) {
      if (!a.bit) { p.why_generic = "contracted label with a non power-of-two extent"; return false; }
      K.push_back(i);
    } else if (a.m1()) (a.bit ? M : O).push_back(i);
    else if (a.n1()) (a.bit ? N : O).push_back(i);
    else if (a.h1()) O.push_back(i);
    else { p.why_generic = "label summed out of a single operand"; return false; }
  }
  const int k = (int)K.size(), kc = ARTN_GEMM128_KC;
  if (k < kc) { p.why_generic = "fewer contracted bits than one LDS chunk of the complex128 GEMM kernel"; return false; }
  if (k - kc > ARTN_GEMM_MAX_KO) { p.why_generic = "too many contracted bits"; return false; }
  const bool swapped = M.size() < 5 && N.size() >= 5;
  if (swapped) {
    for (auto &a : ax) std::swap(a.sA, a.sB1);
    std::swap(M, N);
  }
  const int m = (int)M.size(), n = (int)N.size();
  if (m < 5) { p.why_generic = "too few free bits for an MFMA tile"; return false; }
  auto in_set = [](const std::vector<int> &v, int x) { return std::find(v.begin(), v.end(), x) != v.end(); };
  auto byA = [&](int x, int y) { return ax[x].sA < ax[y].sA; };
  auto byB = [&](int x, int y) { return ax[x].sB1 < ax[y].sB1; };
  auto byC = [&](int x, int y) { return ax[x].sC < ax[y].sC; };
  std::sort(K.begin(), K.end(), byA);
  std::sort(M.begin(), M.end(), byA);
  std::sort(N.begin(), N.end(), byB);
  // wave grid / blocks: mt = 5 + wm, nt = 3 + nb + wn (wm + wn <= 2): the biggest tile the free bits allow
  int best_size = -1, wm_b = 0, wn_b = 0, nb_b = 0;
  for (int wm = 0; wm <= 2; ++wm)
    for (int wn = 0; wm + wn <= 2; ++wn)
      for (int nb = 0; nb <= 2; ++nb) {
        const int mt = 5 + wm, nt = 3 + nb + wn;
        if (mt > m || (nt > n && !(wn == 0 && nb == 0))) continue;
        const int size = mt + std::min(nt, n) + (wm + wn == 2 ? 1 : 0);
        if (size > best_size) { best_size = size; wm_b = wm; wn_b = wn; nb_b = nb; }
      }
  int mt = 5 + wm_b, nt = std::min(3 + nb_b + wn_b, n);
  // contiguous runs: the low-stride bits of A, B and C belong in the tile (2^r elements of 16 bytes)
  int ra = 3, rb = 3, rc = 3;
  std::vector<int> Kf, Mf, Nf;
  for (;;) {
    Kf.clear(); Mf.clear(); Nf.clear();
    const int64_t la = int64_t(1) << ra, lb = int64_t(1) << rb, lc = int64_t(1) << rc;
    for (int i : K) if (ax[i].sA < la || ax[i].sB1 < lb) Kf.push_back(i);
    for (int i : M) if (ax[i].sA < la || ax[i].sC < lc) Mf.push_back(i);
    for (int i : N) if (ax[i].sB1 < lb || ax[i].sC < lc) Nf.push_back(i);
    int *shrink = nullptr;
    if ((int)Kf.size() > kc) shrink = rb >= ra ? &rb : &ra;
    else if ((int)Mf.size() > mt) shrink = rc >= ra ? &rc : &ra;
    else if ((int)Nf.size() > nt) shrink = rc >= rb ? &rc : &rb;
    if (!shrink) break;
    if (*shrink <= 0) { p.why_generic = "forced tile bits exceed the complex128 GEMM tile"; return false; }
    --*shrink;
  }
  std::vector<int> Kc(Kf), Mt(Mf), Nt(Nf);
  for (int i : K) { if ((int)Kc.size() >= kc) break; if (!in_set(Kc, i)) Kc.push_back(i); }
  for (int i : M) { if ((int)Mt.size() >= mt) break; if (!in_set(Mt, i)) Mt.push_back(i); }
  for (int i : N) { if ((int)Nt.size() >= nt) break; if (!in_set(Nt, i)) Nt.push_back(i); }
  if ((int)Mt.size() != mt || (int)Nt.size() != nt) { p.why_generic = "internal: complex128 tile"; return false; }
  ArtnGemmPlan &g = p.gemm;
  memset(&g, 0, sizeof(g));
  g.mt = mt; g.nt = nt; g.kc = kc; g.n_ko = k - kc; g.swapped = swapped ? 1 : 0;
  g.split = 2; g.gather_dim = -1;
  g.wm_log2 = wm_b; g.wn_log2 = wn_b; g.mb_log2 = 1; g.nb_log2 = nb_b;
  std::sort(Kc.begin(), Kc.end(), byA);
  std::sort(Mt.begin(), Mt.end(), byA);
  std::sort(Nt.begin(), Nt.end(), byB);
  auto pos = [](const std::vector<int> &v, int axis) { return (int)(std::find(v.begin(), v.end(), axis) - v.begin()); };
  std::vector<int> tA(Kc), tB(Kc), tC(Mt);
  tA.insert(tA.end(), Mt.begin(), Mt.end());
  tB.insert(tB.end(), Nt.begin(), Nt.end());
  tC.insert(tC.end(), Nt.begin(), Nt.end());
  std::sort(tA.begin(), tA.end(), byA);
  std::sort(tB.begin(), tB.end(), byB);
  std::sort(tC.begin(), tC.end(), byC);
  g.ta_bits = mt + kc; g.tb_bits = nt + kc; g.tc_bits = mt + nt;
  auto lds_row = [&](int i) { return 16 << i; };
  auto lds_kc = [&](int q) { return (16 << ARTN_GEMM_PITCH_LOG2) << q; };
  for (int b = 0; b < g.ta_bits; ++b) { g.a_stride[b] = ax[tA[b]].sA; g.a_lds[b] = in_set(Mt, tA[b]) ? lds_row(pos(Mt, tA[b])) : lds_kc(pos(Kc, tA[b])); }
  for (int b = 0; b < g.tb_bits; ++b) { g.b_stride[b] = ax[tB[b]].sB1; g.b_lds[b] = in_set(Nt, tB[b]) ? lds_row(pos(Nt, tB[b])) : lds_kc(pos(Kc, tB[b])); }
  for (int b = 0; b < g.tc_bits; ++b) g.out_stride[b] = ax[tC[b]].sC;
  for (int i = 0; i < mt; ++i) g.m_pos[i] = pos(tC, Mt[i]);
  for (int i = 0; i < nt; ++i) g.n_pos[i] = pos(tC, Nt[i]);
  { int q = 0; for (int i : K) if (!in_set(Kc, i)) { g.ko_sA[q] = ax[i].sA; g.ko_sB[q] = ax[i].sB1; ++q; } }
  // result-image swizzle: the 16 lanes of a ds_write_b64 group are the 16 rows of an MFMA block (m_local bits 0..3);
  // elements are 16 bytes: fold row bits that sit above the 128-byte window (position >= 3) into free positions 0..2
  {
    bool taken[3] = {false, false, false};
    for (int i = 0; i < 4; ++i) if (g.m_pos[i] < 3) taken[g.m_pos[i]] = true;
    for (int i = 0; i < 4 && tuning().swizzle; ++i) {
      if (g.m_pos[i] < 3) continue;
      int f = -1;
      for (int c = 0; c < 3; ++c) if (!taken[c]) { f = c; break; }
      if (f < 0) break;
      taken[f] = true;
      g.swz_src[g.swz_n] = g.m_pos[i]; g.swz_dst[g.swz_n] = f; ++g.swz_n;
    }
  }
  std::vector<int> outer;
  for (int i : N) if (!in_set(Nt, i)) outer.push_back(i);
  for (int i : M) if (!in_set(Mt, i)) outer.push_back(i);
  std::sort(O.begin(), O.end(), [&](int x, int y) {
    auto key = [&](int z) { return ax[z].sA >= 0 ? ax[z].sA : ax[z].sB1; };
    return key(x) < key(y);
  });
  outer.insert(outer.end(), O.begin(), O.end());
  g.n_tiles = 1;
  int64_t a_rereads = 1;
  for (int i : outer) {
    const Axis &a = ax[i];
    ArtnOuterDim od;
    od.ext = a.ext; od.sA = a.sA >= 0 ? a.sA : 0; od.sB1 = a.sB1 >= 0 ? a.sB1 : 0; od.sB2 = 0; od.sC = a.sC >= 0 ? a.sC : 0;
    od.log2ext = a.gathered ? -1 : ilog2_exact(a.ext); od.pad_ = 0; // (the gathered axis is decoded the slow way, on its own)
    g.n_tiles *= a.ext;
    if (a.sA < 0) a_rereads *= a.ext;
    if (g.n_outer > 0) {
      ArtnOuterDim &pr = g.outer[g.n_outer - 1];
      auto okf = [&](int64_t ps, int64_t ns) { return (ps == 0 && ns == 0) || (ps != 0 && ns == ps * pr.ext); };
      if (pr.log2ext >= 0 && od.log2ext >= 0 && okf(pr.sA, od.sA) && okf(pr.sB1, od.sB1) && okf(pr.sC, od.sC) && pr.log2ext + od.log2ext < 31) {
        pr.ext *= od.ext; pr.log2ext += od.log2ext;
        continue;
      }
    }
    if (g.n_outer >= ARTN_MAX_OUTER) { p.why_generic = "too many outer axes"; return false; }
    if (a.gathered) g.gather_dim = g.n_outer;
    g.outer[g.n_outer++] = od;
  }
  if (gather_label >= 0 && g.gather_dim < 0) { p.why_generic = "gathered label is not an outer axis"; return false; }
  {
    int64_t sa = 0, sb = 0, sc = 0;
    for (int b = 0; b < 8; ++b) {
      if (b < g.ta_bits) sa += g.a_stride[b];
      if (b < g.tb_bits) sb += g.b_stride[b];
      if (b < g.tc_bits) sc += g.out_stride[b];
    }
    const int64_t lim = (int64_t(1) << 27) - 1; // elements of 16 bytes: < 2^31 bytes
    if (sa > lim || sb > lim || sc > lim) { p.why_generic = "lane offsets exceed 32 bits"; return false; }
    int pow2_bits = 0;
    for (int i = 0; i < g.n_outer && g.outer[i].log2ext >= 0; ++i) pow2_bits += g.outer[i].log2ext;
    if (pow2_bits > 32) { p.why_generic = "more than 2^32 tiles"; return false; }
  }
  if (g.n_tiles < min_tiles && k <= 8) { p.why_generic = "too few tiles to fill the chip"; return false; }
  p.kernel = ARTN_KERNEL_GEMM_MFMA;
  ArtnStepInfo &f = p.info;
  f.kernel = ARTN_KERNEL_GEMM_MFMA;
  f.k_bits = k; f.m_tile_bits = mt; f.n_tile_bits = nt;
  f.tile_in_bits = g.ta_bits; f.tile_out_bits = g.tc_bits;
  f.run_in_bits = ra; f.run_out_bits = rc;
  const int64_t stage = 2 * (16LL << (ARTN_GEMM_PITCH_LOG2 + kc));
  const int64_t epi = 16LL << std::min(g.tc_bits, ARTN_GEMM128_EPI_BITS);
  f.lds_bytes = (int32_t)(std::max(2 * stage, epi) + 512LL * 8 + 32 * 32 + 16LL * ARTN_GEMM_MAX_KO);
  f.n_tiles = g.n_tiles;
  f.a_rereads = a_rereads;
  const int wg_per_cu = std::max(1, std::min(2, (160 * 1024) / f.lds_bytes));
  f.grid = (int32_t)std::min<int64_t>(g.n_tiles, (int64_t)n_cu * wg_per_cu);
  return true;
}

// ----------------------------------------------------------------------------------------
// packed-operand GEMM (ARTN_C64_BF16, and ARTN_C64 in 3M fp32): see ArtnPackPlan
// ----------------------------------------------------------------------------------------
static inline bool make_pgemm(const ArtnStepDesc *d, ArtnPlan &p, int n_cu, bool any_intensity = false) {
  // (any_intensity: the plan emulator replays the mechanics of small steps the routing rule below would send elsewhere)
  if (d->dtype != ARTN_C64_BF16 && d->dtype != ARTN_C64) { p.why_generic = "packed GEMM: complex64 only"; return false; }
  const bool bf = d->dtype == ARTN_C64_BF16;
  const int KC = bf ? ARTN_PG_KC : ARTN_PG_KC - 1; // the stage holds the same bytes either way
  std::vector<Axis> ax;
  expand_axes(d, ax);
  std::vector<int> K, M, N;
  for (int i = 0; i < (int)ax.size(); ++i) {
    const Axis &a = ax[i];
    if (!a.bit) { p.why_generic = "packed GEMM: non power-of-two extent"; return false; }
    if (a.k1()) K.push_back(i);
    else if (a.m1()) M.push_back(i);
    else if (a.n1()) N.push_back(i);
    else { p.why_generic = "packed GEMM: batch label or label summed out of one operand"; return false; }
  }
  const bool swapped = M.size() < N.size(); // the operand with more free bits supplies the 256-row side
  if (swapped) {
    for (auto &a : ax) std::swap(a.sA, a.sB1);
    std::swap(M, N);
  }
  const int k = (int)K.size(), m = (int)M.size(), n = (int)N.size();
  // worth two packing passes: 2^9+ contracted values (fp32, whose MFMAs are 16 times slower per FLOP and whose packed copy
  // saves no bytes: 2^10+), and enough tiles for every CU
  if (k < (bf ? 9 : (any_intensity ? 8 : tuning().packed_min_k)) || m < ARTN_PG_MT || n < ARTN_PG_NT) { p.why_generic = "packed GEMM: too few contracted or free bits"; return false; }
  if (!bf && !any_intensity) { // the packing passes move every operand element twice more: only where the GEMM itself is far from memory-bound
    const double flops = 8.0 * (double)(int64_t(1) << (m + n)) * (double)(int64_t(1) << k);
    const double bytes = 8.0 * ((double)(int64_t(1) << (m + k)) + (double)(int64_t(1) << (n + k)) + (double)(int64_t(1) << (m + n)));
    if (flops / bytes < tuning().packed_min_ai) { p.why_generic = "packed GEMM: too few FLOP per byte to pay for the packing passes"; return false; }
  }
  if (k - KC > ARTN_GEMM_MAX_KO || m - ARTN_PG_MT > 32 || n - ARTN_PG_NT > 32) { p.why_generic = "packed GEMM: too many bits"; return false; }
  if ((int64_t(1) << (m - ARTN_PG_MT + n - ARTN_PG_NT)) < n_cu) { p.why_generic = "packed GEMM: too few tiles to fill the chip"; return false; }
  auto byA = [&](int x, int y) { return ax[x].sA < ax[y].sA; };
  auto byB = [&](int x, int y) { return ax[x].sB1 < ax[y].sB1; };
  auto byC = [&](int x, int y) { return ax[x].sC < ax[y].sC; };
  auto in_set = [](const std::vector<int> &v, int x) { return std::find(v.begin(), v.end(), x) != v.end(); };
  auto pos = [](const std::vector<int> &v, int axis) { return (int)(std::find(v.begin(), v.end(), axis) - v.begin()); };
  // chunk bits: the contracted bits with the lowest strides in the first (bigger) operand, so that its packing pass
  // reads neighbouring elements; tile rows: the free bits with the lowest C strides first (coalesced result stores),
  // topped up by lowest operand stride
  std::sort(K.begin(), K.end(), byA);
  std::vector<int> Kc(K.begin(), K.begin() + KC);
  auto pick_rows = [&](std::vector<int> &F, int want, auto byOp) {
    std::vector<int> byc(F), out;
    std::sort(byc.begin(), byc.end(), byC);
    for (int i : byc) { if ((int)out.size() >= 4 || (int)out.size() >= want) break; if (ax[i].sC < 16) out.push_back(i); } // the bits inside a 128-byte run of C
    std::vector<int> byo(F);
    std::sort(byo.begin(), byo.end(), byOp);
    for (int i : byo) { if ((int)out.size() >= want) break; if (!in_set(out, i)) out.push_back(i); }
    std::sort(out.begin(), out.end(), byOp);
    return out;
  };
  std::vector<int> Mt = pick_rows(M, ARTN_PG_MT, byA), Nt = pick_rows(N, ARTN_PG_NT, byB);
  ArtnPackPlan &g = p.pack;
  memset(&g, 0, sizeof(g));
  g.arith = bf ? 0 : 1;
  g.kc_bits = KC;
  g.flush_chunks = bf ? 0 : 1 << (ARTN_GEMM_FLUSH_LOG2 - KC);
  g.swapped = swapped ? 1 : 0;
  g.n_ko = k - KC;
  g.n_mo = m - ARTN_PG_MT;
  g.n_no = n - ARTN_PG_NT;
  g.n_tiles = int64_t(1) << (g.n_mo + g.n_no);
  g.a.n_row = ARTN_PG_MT; g.b.n_row = ARTN_PG_NT; g.a.n_to = g.n_mo; g.b.n_to = g.n_no;
  for (int i = 0; i < ARTN_PG_MT; ++i) g.a.row[i] = ax[Mt[i]].sA;
  for (int i = 0; i < ARTN_PG_NT; ++i) g.b.row[i] = ax[Nt[i]].sB1;
  for (int q = 0; q < KC; ++q) { g.a.kc[q] = ax[Kc[q]].sA; g.b.kc[q] = ax[Kc[q]].sB1; }
  { int q = 0; for (int i : K) if (!in_set(Kc, i)) { g.a.ko[q] = ax[i].sA; g.b.ko[q] = ax[i].sB1; ++q; } }
  { int q = 0; for (int i : M) if (!in_set(Mt, i)) { g.a.to[q] = ax[i].sA; g.c_mo[q] = ax[i].sC; ++q; } }
  { int q = 0; for (int i : N) if (!in_set(Nt, i)) { g.b.to[q] = ax[i].sB1; g.c_no[q] = ax[i].sC; ++q; } }
  std::vector<int> tC(Mt);
  tC.insert(tC.end(), Nt.begin(), Nt.end());
  std::sort(tC.begin(), tC.end(), byC);
  for (int b = 0; b < ARTN_PG_MT + ARTN_PG_NT; ++b) g.out_stride[b] = ax[tC[b]].sC;
  if (g.out_stride[0] != 1) { p.why_generic = "packed GEMM: no 16-byte run at the bottom of the result"; return false; }
  for (int b = 1; b < ARTN_PG_MT + ARTN_PG_NT; ++b) if (g.out_stride[b] & 1) { p.why_generic = "packed GEMM: odd C stride"; return false; }
  for (int q = 0; q < g.n_mo; ++q) if (g.c_mo[q] & 1) { p.why_generic = "packed GEMM: odd C stride"; return false; }
  for (int q = 0; q < g.n_no; ++q) if (g.c_no[q] & 1) { p.why_generic = "packed GEMM: odd C stride"; return false; }
  {
    int64_t sc = 0;
    for (int b = 1; b <= 9; ++b) sc += g.out_stride[b];
    if (sc > (int64_t(1) << 28) - 1) { p.why_generic = "packed GEMM: lane offsets exceed 32 bits"; return false; }
  }
  for (int i = 0; i < ARTN_PG_MT; ++i) g.m_pos[i] = pos(tC, Mt[i]);
  for (int i = 0; i < ARTN_PG_NT; ++i) g.n_pos[i] = pos(tC, Nt[i]);
  { // result-image swizzle, as in make_gemm: the 16 lanes of a ds_write_b64 group differ in m_local bits 0..3
    bool taken[4] = {true, false, false, false};
    for (int i = 0; i < 4; ++i) if (g.m_pos[i] < 4) taken[g.m_pos[i]] = true;
    for (int i = 0; i < 4 && tuning().swizzle; ++i) {
      if (g.m_pos[i] < 4) continue;
      int f = -1;
      for (int c = 1; c < 4; ++c) if (!taken[c]) { f = c; break; }
      if (f < 0) break;
      taken[f] = true;
      g.swz_src[g.swz_n] = g.m_pos[i]; g.swz_dst[g.swz_n] = f; ++g.swz_n;
    }
  }
  p.kernel = ARTN_KERNEL_PGEMM;
  ArtnStepInfo &f = p.info;
  f.kernel = ARTN_KERNEL_PGEMM;
  f.k_bits = k; f.m_tile_bits = ARTN_PG_MT; f.n_tile_bits = ARTN_PG_NT;
  f.tile_in_bits = ARTN_PG_MT + KC; f.tile_out_bits = ARTN_PG_MT + ARTN_PG_NT;
  f.run_in_bits = 0; f.run_out_bits = 0;
  f.lds_bytes = 3 * ((4 << (ARTN_PG_MT + ARTN_PG_KC)) + (4 << (ARTN_PG_NT + ARTN_PG_KC))); // three chunk buffers
  f.n_tiles = g.n_tiles;
  f.a_rereads = int64_t(1) << g.n_no;
  f.grid = (int32_t)std::min<int64_t>(g.n_tiles, (int64_t)n_cu);
  f.workspace_bytes = (bf ? 4 : 8) * ((int64_t(1) << (m + k)) + (int64_t(1) << (n + k)));
  return true;
}

// ----------------------------------------------------------------------------------------
// extent-based GEMM planner (artn_k_xgemm, artn_xgemm_plan.h): complex64, any extents and strides.
// Takes what the bit planners decline because a label's extent is not a power of two -- networks of bond
// dimension 3, 5, 6 ... (reference tensor_network.py:4-30 accepts any bond_dims; the einsum at contraction.py:70
// has no such restriction).
// ----------------------------------------------------------------------------------------
// launch geometry of the row-streaming forms (ArtnXGemmPlan::rowmode 1: 16-row blocks, 2: 64-row superblocks), dealt round-robin to
// the waves of n_cu x (waves per SIMD the instantiation's registers allow) workgroups
static inline void xrow_fill_info(const ArtnXGemmPlan &x, ArtnStepInfo &I, int n_cu) {
  const int S = artn_xrow_steps(x.k.total), NBK = artn_xrow_nbk(x.n.total);
  const int rows = x.rowmode == 2 ? 64 : 16;
  I.m_tile_bits = x.rowmode == 2 ? 6 : 4;
  I.lds_bytes = artn_xrow_lds_bytes(x.m.total / ((int64_t)x.m.L0 * x.m.L1));
  I.n_tiles = (x.m.total + rows - 1) / rows;
  I.grid = (int32_t)std::min<int64_t>((I.n_tiles + 3) / 4, (int64_t)n_cu * (x.rowmode == 2 ? artn_xrow64_waves(S, NBK) : artn_xrow_waves(S, NBK)));
  I.a_rereads = 1;
}

static inline bool make_xgemm(const ArtnStepDesc *d, ArtnPlan &p, int n_cu, int64_t min_tiles) {
  const bool c128 = d->dtype == ARTN_C128; // (artn_k_xgemm128: the same plan with 16-byte elements, chunks of 8, one or two column blocks)
  ArtnXGemmPlan &x = p.xg;
  memset(&x, 0, sizeof(x));
  struct Lab { int64_t e, sA, sB, sC; };
  std::vector<Lab> M, N, K, H;
  int64_t spanA = 1, spanB = 1, spanC = 1;
  double work = 8.0;
  for (int l = 0; l < d->n_labels; ++l) {
    const int64_t e = d->extent[l];
    if (e == 1) continue;
    const bool a = d->stride_a[l] >= 0, b = d->stride_b[l] >= 0, c = d->stride_c[l] >= 0;
    Lab L = {e, a ? d->stride_a[l] : 0, b ? d->stride_b[l] : 0, c ? d->stride_c[l] : 0};
    if (a) spanA += (e - 1) * L.sA;
    if (b) spanB += (e - 1) * L.sB;
    if (c) spanC += (e - 1) * L.sC;
    work *= (double)e;
    if (a && b && !c) K.push_back(L);
    else if (a && !b && c) M.push_back(L);
    else if (!a && b && c) N.push_back(L);
    else if (a && b && c) H.push_back(L);
    else { p.why_generic = "extent GEMM: label summed out of a single operand"; return false; }
    if (e >= (int64_t(1) << 31)) { p.why_generic = "extent GEMM: extent beyond 2^31"; return false; }
  }
  const int64_t lim = int64_t(1) << 31; // flattened indices, tiles and chunks are 31-bit in the kernel
  // element offsets are UNSIGNED 32-bit in the kernel (tables, decode, sums; widened before the scaling by the element size):
  // a tensor may span up to 2^32 elements -- 32 GiB of complex64, e.g. 3^20 = 2^31.7 -- (round 6: the limit used to be 2^31)
  const int64_t span_lim = int64_t(1) << 32;
  if (spanA >= span_lim || spanB >= span_lim || spanC >= span_lim) { p.why_generic = "extent GEMM: a tensor of 2^32 elements or more"; return false; }
  auto prod = [](const std::vector<Lab> &v) { int64_t t = 1; for (auto &l : v) t *= l.e; return t; };
  if (min_tiles > 1 && work < (double)(int64_t(1) << 24)) { p.why_generic = "extent GEMM: too little work for a tiled launch"; return false; }
  // the first operand supplies the 128-row tiles: it is the one with more free values
  x.swapped = prod(N) > prod(M) ? 1 : 0;
  if (x.swapped) {
    std::swap(M, N);
    for (auto *v : {&M, &N, &K, &H})
      for (auto &l : *v) std::swap(l.sA, l.sB);
  }
  if ((int)M.size() > ARTN_XG_MAXL || (int)N.size() > ARTN_XG_MAXL || (int)K.size() > ARTN_XG_MAXL || (int)H.size() > ARTN_XG_MAXH) {
    p.why_generic = "extent GEMM: too many labels on one side";
    return false;
  }
  // which label is the fastest of each tensor decides how its copy lanes run
  auto fastest_is = [&](const std::vector<Lab> &cand, int which) { // which: 0 A, 1 B, 2 C
    int64_t best = -1;
    for (auto *v : {&M, &N, &K, &H})
      for (auto &l : *v) {
        const bool has = which == 0 ? (v != &N) : which == 1 ? (v != &M) : (v != &K);
        const int64_t s = which == 0 ? l.sA : which == 1 ? l.sB : l.sC;
        if (has && (best < 0 || s < best)) best = s;
      }
    for (auto &l : cand) {
      const int64_t s = which == 0 ? l.sA : which == 1 ? l.sB : l.sC;
      if (s == best) return true;
    }
    return false;
  };
  x.amode = fastest_is(K, 0) ? 1 : 0;
  x.bmode = fastest_is(K, 1) ? 1 : 0;
  x.trans = fastest_is(N, 2) ? 1 : 0;
  x.prio = 0; // (measured: 6.72 / 3.06 ms without, 6.87 / 3.14 ms with, on the two biggest steps of the bond-dimension-3 network)
#ifdef ARTN_DEV_SWITCHES // (A/B of the copy modes: no difference on any measured layout, tools/xgemm_modes.py)
  if (const char *e = getenv("ARTN_XG_AMODE")) x.amode = atoi(e) != 0;
  if (const char *e = getenv("ARTN_XG_BMODE")) x.bmode = atoi(e) != 0;
  if (const char *e = getenv("ARTN_XG_PRIO")) x.prio = atoi(e);
#endif
  // label order inside each flattened index (innermost first): that of the tensor whose copy lanes run along it
  if (x.amode == 0) std::sort(M.begin(), M.end(), [](const Lab &u, const Lab &v) { return u.sA < v.sA; });
  else std::sort(M.begin(), M.end(), [](const Lab &u, const Lab &v) { return u.sC < v.sC; });
  if (x.bmode == 0) std::sort(N.begin(), N.end(), [](const Lab &u, const Lab &v) { return u.sB < v.sB; });
  else std::sort(N.begin(), N.end(), [](const Lab &u, const Lab &v) { return u.sC < v.sC; });
  if (x.amode == 0 && x.bmode == 1) std::sort(K.begin(), K.end(), [](const Lab &u, const Lab &v) { return u.sB < v.sB; });
  else std::sort(K.begin(), K.end(), [](const Lab &u, const Lab &v) { return u.sA < v.sA; });
  auto fill = [&](ArtnXSide &S, const std::vector<Lab> &v, int kind) { // kind 0: m (A, C), 1: n (B, C), 2: k (A, B)
    S.n_lab = (int32_t)v.size();
    S.total = 1;
    for (int i = 0; i < S.n_lab; ++i) {
      S.ext[i] = (int32_t)v[i].e;
      S.s0[i] = kind == 1 ? v[i].sB : v[i].sA;
      S.s1[i] = kind == 2 ? v[i].sB : v[i].sC;
      S.total *= v[i].e;
    }
    S.L0 = S.L1 = 1;
    S.n0 = S.n1 = 0;
    while (S.n0 < S.n_lab && (int64_t)S.L0 * S.ext[S.n0] <= ARTN_XG_LEVEL) S.L0 *= S.ext[S.n0++];
    if (kind != 2)
      while (S.n0 + S.n1 < S.n_lab && (int64_t)S.L1 * S.ext[S.n0 + S.n1] <= ARTN_XG_LEVEL) S.L1 *= S.ext[S.n0 + S.n1++];
  };
  fill(x.m, M, 0);
  fill(x.n, N, 1);
  fill(x.k, K, 2);
  if (x.k.n_lab > 0 && x.k.n0 == 0) { p.why_generic = "extent GEMM: innermost contracted label longer than a level table"; return false; }
  if (x.m.total >= lim || x.n.total >= lim || x.k.total >= lim) { p.why_generic = "extent GEMM: a flattened index beyond 2^31"; return false; }
  x.n_h = (int32_t)H.size();
  int64_t hprod = 1;
  for (int i = 0; i < x.n_h; ++i) {
    x.h_ext[i] = (int32_t)H[i].e;
    x.h_sA[i] = H[i].sA; x.h_sB[i] = H[i].sB; x.h_sC[i] = H[i].sC;
    hprod *= H[i].e;
  }
  // 32-column blocks per tile (1..3).  A tile costs a fixed part -- staging 128 rows of the first operand, tables, barriers --
  // plus its blocks' MFMAs: time ~ tiles x a + blocks x b with a = 0.95 b (from the measured 73 / 96 / 108 TFLOP/s of one / two /
  // three blocks per tile on long contractions).  Round 6: columns that do not fill a last tile run as a SECOND launch with
  // just the blocks they need (tail_nb, artn_xg_tail_plan) -- 216 columns are 2 tiles of 3 blocks + 1 tile of 1 (7 blocks in 3
  // tiles; was 4 tiles of 2 blocks, 8 in 4: the width that wasted the fewest columns), 243 columns 2 x 3 + 1 x 2 (was 4 x 2).
  // Ties go to the narrower tile.  (ARTN_XG_TAIL=0: one launch, the last tile padded, as before.)
  const int64_t blocks_n = (x.n.total + 31) / 32;
  auto pick_nb = [&](bool tail) {
    double best = 0;
    int pick = 1;
    for (int nb = 1; nb <= (c128 ? 1 : 3) && nb <= blocks_n; ++nb) { // (complex128: one block -- two need 128 accumulator registers and spill)
      const int64_t tiles = (blocks_n + nb - 1) / nb;
      const double cost = 0.95 * (double)tiles + (double)(tail ? blocks_n : tiles * nb);
      if (nb == 1 || cost < best - 1e-9) { best = cost; pick = nb; }
    }
    return pick;
  };
  x.nb = pick_nb(false);
  bool with_tail = false;
  if (tuning().xg_tail && !c128) {
    // (two launches are two tails: a step of 244 tiles -- one round of the 512 resident workgroups -- went from 0.26 to 0.49 ms as
    //  122 + 61 tiles; the split is for launches of many rounds: 8 760 tiles 7.2 -> 6.9 ms, 49 824 tiles 3.06 -> 2.88 ms)
    const int nb_t = pick_nb(true);
    const int64_t full_cols = blocks_n / nb_t;
    const int64_t tiles_main = ((x.m.total + ARTN_XG_TM - 1) / ARTN_XG_TM) * full_cols * hprod;
    if (blocks_n % nb_t != 0 && full_cols >= 1 && tiles_main >= (int64_t)8 * n_cu) { x.nb = nb_t; with_tail = true; }
  }
#ifdef ARTN_DEV_SWITCHES
  if (const char *e = getenv("ARTN_XG_NB")) { const int v = atoi(e); if (v >= 1 && v <= 3) x.nb = v; }
#endif
  // chunks of 8 where a tile is a handful of contracted values x at most 32 columns: those steps are latency chains per tile,
  // and the small chunk lets four workgroups share a CU (artn_k_xgemm<1, *, 8>)
  x.kc = (x.nb == 1 && x.k.total <= 32) ? 8 : ARTN_XG_KC;
#ifdef ARTN_DEV_SWITCHES
  if (const char *e = getenv("ARTN_XG_KC")) { const int v = atoi(e); if ((v == 8 && x.nb == 1) || v == 16) x.kc = v; }
#endif
  x.c128 = c128 ? 1 : 0;
  if (c128) { x.kc = 8; x.trans = 0; }
  x.pc = 0; // (artn_k_xgemm_pc, the producer / consumer form: correct, measured slower -- 7.5 against 6.7 ms on the biggest step of the
            //  bond-dimension-3 network -- and compiled into development builds only)
#if defined(ARTN_DEV_SWITCHES) && defined(ARTN_DEV_XGPC)
  if (const char *e = getenv("ARTN_XG_PC")) x.pc = (atoi(e) != 0 && x.kc == ARTN_XG_KC && !c128) ? 1 : 0;
#endif
  x.cpg = (x.k.L0 + x.kc - 1) / x.kc;
  x.k_groups = x.k.total / x.k.L0;
  const int64_t chunks = x.k_groups * x.cpg;
  if (chunks >= lim) { p.why_generic = "extent GEMM: too many chunks per tile"; return false; }
  // (complex128: f64 accumulators need no periodic flush -- the interval exists to bound fp32 rounding growth)
  x.flush_chunks = (!c128 && chunks > ARTN_XG_FLUSH / x.kc) ? ARTN_XG_FLUSH / x.kc : 0;
  x.tiles_m = (x.m.total + ARTN_XG_TM - 1) / ARTN_XG_TM;
  x.tiles_n = (x.n.total + 32 * x.nb - 1) / (32 * x.nb);
  x.col0 = 0;
  x.tail_nb = 0;
  if (with_tail && !x.pc) { // the last column tile would be narrower: its own launch
    x.tail_nb = (int32_t)(blocks_n % x.nb);
    x.tiles_n -= 1;
  }
  x.n_tiles = x.tiles_m * x.tiles_n * hprod;
  if (x.n_tiles >= lim) { p.why_generic = "extent GEMM: too many tiles"; return false; }
  // the row-streaming form (artn_k_xrow): a handful of contracted values into a handful of columns on very many rows, the lanes
  // of a store along rows; buffer instructions with 32-bit byte offsets: tensors below 4 GiB
  {
    const int64_t span_first = x.swapped ? spanB : spanA, row_lim = (int64_t(1) << 29) - 2;
    const int64_t l2 = x.m.total / ((int64_t)x.m.L0 * x.m.L1);
    // (the 16-row shape, measured on the bond-dimension-3 network, gpurun_out/s_r6t: 9 -> 9 on 3^16 rows 2.30 -> 1.70 ms, 27 -> 27 on
    //  3^13 rows 0.38 -> 0.22 ms, but 27 -> 27 on 3^15 rows only ties, 1.76-1.79 against 1.74-1.77 ms -- 84 MFMAs per 16 rows at
    //  four waves per SIMD: blocks of more than 16 x 16 stay with artn_k_xgemm from 2^22 rows on; ARTN_XROW=2: wherever it fits)
    const bool pays = (x.k.total <= 16 && x.n.total <= 16) || x.m.total < (int64_t(1) << 22) || tuning().xrow == 2;
    const bool fits = !c128 && x.n_h == 0 && x.trans == 0 && x.k.total <= ARTN_XROW_MAX && x.n.total <= ARTN_XROW_MAX &&
                      x.m.total >= ARTN_XROW_MIN_ROWS && l2 <= ARTN_XROW_L2_MAX && span_first <= row_lim && spanC <= row_lim && tuning().xrow;
    // rowmode 2 (artn_k_xrow64): a lane per row, 64-row superblocks -- up to 32 contracted values and 32 columns; 4.5-4.9 TB/s on the
    // steps above (tools/probes/xrow64_probe.hip), so it is taken wherever it fits, the 16-row shape (rowmode 1) where that pays
    const bool wide = fits && tuning().xrow64 && artn_xrow_steps(x.k.total) <= 8 && artn_xrow_nbk(x.n.total) <= 2;
    x.rowmode = wide ? 2 : ((fits && pays) ? 1 : 0);
    x.row_bytes_a = x.rowmode ? (uint32_t)(8 * span_first) : 0;
    x.row_bytes_c = x.rowmode ? (uint32_t)(8 * spanC) : 0;
  }
  p.kernel = ARTN_KERNEL_XGEMM;
  ArtnStepInfo &I = p.info;
  I.kernel = ARTN_KERNEL_XGEMM;
  I.m_tile_bits = 7;
  I.n_tile_bits = 5;
  I.k_bits = 4;
  I.lds_bytes = c128 ? artn_xg128_lds_bytes(x.nb) : (x.pc ? artn_xg_pc_lds_bytes(x.nb) : artn_xg_lds_bytes(x.nb, x.kc));
  I.n_tiles = x.n_tiles;
  I.grid = (int32_t)std::min<int64_t>(x.n_tiles, (int64_t)n_cu * (c128 ? 2 : (x.pc ? 1 : (x.kc == 8 ? 4 : 2))));
  I.a_rereads = x.tiles_n;
  if (x.tail_nb) { // (the tail launch: chunks as in the main one, two workgroups per CU)
    const int64_t tail_tiles = x.tiles_m * hprod;
    x.tail_grid = (int32_t)std::min<int64_t>(tail_tiles, (int64_t)n_cu * 2);
    x.tail_lds = artn_xg_lds_bytes(x.tail_nb, x.kc);
    I.n_tiles += tail_tiles;
    I.a_rereads += 1;
  }
  if (x.rowmode) xrow_fill_info(x, I, n_cu);
  return true;
}

static inline void step_cost(const ArtnStepDesc *d, double &flops, double &na, double &nb, double &nc) {
  double prod = 1;
  na = nb = nc = 1;
  for (int l = 0; l < d->n_labels; ++l) {
    prod *= (double)d->extent[l];
    if (d->stride_a[l] >= 0) na *= (double)d->extent[l];
    if (d->stride_b[l] >= 0) nb *= (double)d->extent[l];
    if (d->stride_c[l] >= 0) nc *= (double)d->extent[l];
  }
  flops = 8.0 * prod;
}

// min_tiles: below this many LDS tiles the strided kernel is used instead (a handful of
// workgroups cannot fill 256 CUs; such steps are launch-latency bound either way).
static inline int make_plan(const ArtnStepDesc *d, ArtnPlan &p, std::string &err, int n_cu = 256,
                            bool allow_bits = true, int64_t min_tiles = 32, int gather_label = -1, bool allow_gemm = true,
                            bool allow_packed = false, bool allow_xgemm = true) {
  int rc = validate(d, err);
  if (rc) return rc;
  memset(&p.info, 0, sizeof(p.info));
  p.n_cu = n_cu;
  bool ok = false;
  // (needs a workspace: only where the caller can supply one -- artn_contract_query reports its size, artn_contract_ws takes it)
  if (allow_packed && allow_bits && allow_gemm && gather_label < 0 && tuning().gemm && tuning().packed &&
      (d->dtype == ARTN_C64_BF16 || (d->dtype == ARTN_C64 && tuning().packed >= 2)))
    ok = make_pgemm(d, p, n_cu);
  allow_gemm = allow_gemm && !ok;
  if (d->dtype == ARTN_C128 && allow_bits && gather_label < 0 && tuning().bits128 >= 2)
    ok = make_bits(d, nullptr, p, n_cu, min_tiles); // (development: the state-streaming kernel for single steps too)
  if (!ok && d->dtype == ARTN_C128 && allow_bits && allow_gemm && gather_label < 0 && tuning().gemm)
    ok = make_gemm128(d, p, n_cu, min_tiles); // complex128: the f64 MFMA GEMM kernel, else the state-streaming one, else strided
  allow_gemm = allow_gemm && allow_bits && gather_label < 0 && tuning().gemm && d->dtype != ARTN_C128 && !ok;
  if (allow_gemm) ok = make_gemm(d, p, n_cu, min_tiles, tuning().gemm < 2);
  // row gather (artn_contract_gather): chunk steps with 7-8 contracted bits whose second operand brings 5+ free bits run
  // on the GEMM kernel (3M arithmetic, two workgroups per CU) -- the state-streaming kernel's big-K instantiation runs
  // one workgroup per CU with 4M chains: 60 against 100+ TFLOP/s on the chunk steps of the n30 x 100 scheme
  if (!ok && gather_label >= 0 && allow_bits && d->dtype == ARTN_C64 && tuning().gemm && tuning().gather_gemm) {
    int kk = 0, nn = 0;
    for (int l = 0; l < d->n_labels; ++l) {
      if (l == gather_label || d->extent[l] < 2 || ilog2_exact(d->extent[l]) < 0) continue;
      const int lg = ilog2_exact(d->extent[l]);
      if (d->stride_a[l] >= 0 && d->stride_b[l] >= 0 && d->stride_c[l] < 0) kk += lg;
      if (d->stride_a[l] < 0 && d->stride_b[l] >= 0 && d->stride_c[l] >= 0) nn += lg;
    }
    if (kk >= 7 && (nn >= 5 || tuning().gather_gemm >= 2)) ok = make_gemm(d, p, n_cu, min_tiles, false, -1, gather_label);
  }
  // complex128 row gather: the f64 GEMM kernel (3+ contracted bits); there is no gathering state-streaming kernel in complex128
  if (!ok && gather_label >= 0 && allow_bits && d->dtype == ARTN_C128 && tuning().gemm)
    ok = make_gemm128(d, p, n_cu, min_tiles, gather_label);
  if (!ok) ok = allow_bits && (d->dtype != ARTN_C128 || gather_label < 0) && make_bits(d, nullptr, p, n_cu, min_tiles, gather_label);
  // A state-streaming tile with fewer than four (sub-tile, column block) units leaves waves without work -- and the busy
  // waves of the two workgroups of a CU sit on the same SIMDs: a 6-bit step of the n53 slices with 16 result columns and
  // 64-row tiles ran two of four matrix pipes (1.26 ms; the GEMM kernel, whose four waves share a tile's rows: 0.82 ms).
  if (ok && allow_gemm && p.kernel == ARTN_KERNEL_BITS_MFMA && p.bits.st[0].k <= 6 && p.bits.st[0].k >= 5 &&
      p.bits.st[0].m_bits - 5 + p.bits.st[0].wn_log2 < 2 && tuning().idle_to_gemm) {
    ArtnPlan q;
    memset(&q.info, 0, sizeof(q.info));
    q.n_cu = n_cu;
    if (make_gemm(d, q, n_cu, min_tiles, false)) {
      q.why_generic = p.why_generic;
      p = q;
    }
  }
  if (!ok && allow_gemm) { // what the state-streaming kernel declines
    const std::string why = p.why_generic;
    ok = make_gemm(d, p, n_cu, min_tiles, false);
    if (!ok) p.why_generic = why + "; GEMM kernel: " + p.why_generic;
  }
  if (!ok && gather_label >= 0) { err = "row gather needs the tiled kernel: " + p.why_generic; return ARTN_E_UNSUPPORTED; }
  // what every bit planner declined (in practice: a label whose extent is not a power of two): the extent-based GEMM
  if (!ok && allow_bits && allow_xgemm && tuning().xgemm) { // (complex64: artn_k_xgemm; complex128: artn_k_xgemm128)
    const std::string why = p.why_generic;
    ok = make_xgemm(d, p, n_cu, min_tiles);
    if (!ok) p.why_generic = why + "; " + p.why_generic;
  }
  if (!ok && !make_generic(d, p, err)) return ARTN_E_UNSUPPORTED;
  double na, nb, nc;
  step_cost(d, p.info.flops, na, nb, nc);
  p.info.bytes = (d->dtype == ARTN_C128 ? 16.0 : 8.0) * (na + nb + nc);
  // what the matrix pipe executes (3M: three real products per complex product) and in which arithmetic
  if (p.kernel == ARTN_KERNEL_BITS_MFMA) {
    const bool m3 = p.bits.st[0].m3 != 0;
    p.info.arith = p.bits.c128 ? 3 : (p.bits.split == 1 ? 2 : (m3 ? 1 : 0));
    p.info.mfma_flops = p.info.flops * (m3 ? 0.75 : 1.0);
  } else if (p.kernel == ARTN_KERNEL_PGEMM) {
    p.info.arith = p.pack.arith == 0 ? 2 : 1;
    p.info.mfma_flops = p.info.flops * (p.pack.arith == 0 ? 1.0 : 0.75);
  } else if (p.kernel == ARTN_KERNEL_GEMM_MFMA) {
    p.info.arith = d->dtype == ARTN_C128 ? 3 : (p.gemm.split == 1 ? 2 : (p.gemm.m3 ? 1 : 0));
    p.info.mfma_flops = p.info.flops * (p.gemm.m3 ? 0.75 : 1.0);
  } else if (p.kernel == ARTN_KERNEL_XGEMM) {
    p.info.arith = d->dtype == ARTN_C128 ? 3 : 1; // complex64: three real products per complex product; complex128: four, on the f64 MFMA
    p.info.mfma_flops = p.info.flops * (d->dtype == ARTN_C128 ? 1.0 : 0.75);
  } else {
    p.info.arith = -1;
    p.info.mfma_flops = 0.0;
  }
  return ARTN_OK;
}

// artn_k_wide runs a fused pair when all three tiles are 2^12 complex64 elements (so every stage brings as many bits as it
// contracts and 8 waves have two 16 x 16 blocks each), 3..6 contracted bits per stage, fp32 arithmetic, no row gather, and
// the launch is big enough for one workgroup per CU to matter.
static inline bool wide_eligible(const ArtnPlan &p) {
  const ArtnBitsPlan &b = p.bits;
  if (!tuning().wide || b.c128 || b.split != 0 || b.n_stages != 2 || b.gather_dim >= 0) return false;
  if (tuning().wide == 2 && b.st[0].k + b.st[1].k < 11) return false;
  if (b.T_in != ARTN_TILE_BITS_TARGET || b.T_mid != ARTN_TILE_BITS_TARGET || b.T_out != ARTN_TILE_BITS_TARGET) return false;
  for (int q = 0; q < 2; ++q) {
    const ArtnStage &s = b.st[q];
    if (s.k < 3 || s.k > 6 || s.nt != s.k || s.m_bits != ARTN_TILE_BITS_TARGET - s.k) return false;
  }
  if (b.n_tiles < (tuning().wide_min_tiles > 0 ? (int64_t)tuning().wide_min_tiles : (int64_t)p.n_cu * 8)) return false;
  int64_t si = 0, so = 0; // per-lane byte offsets of the copies span tile bits 1..9 and are 32-bit
  for (int i = 1; i <= 9; ++i) { si += b.in_stride[i]; so += b.out_stride[i]; }
  if (si > (int64_t(1) << 28) - 1 || so > (int64_t(1) << 28) - 1) return false;
  return true;
}

// C += result in the store phase of artn_k_bits (reference simulation.py:114, `collect_tensor += tensor_contraction(...)`, without
// writing the slice's result and reading it back): the FULL instantiations of the complex64 kernel (input and output tiles of
// 2^12 elements, fp32 chains, no row gather).
static inline bool bits_can_accumulate(const ArtnPlan &p) {
  if (p.kernel != ARTN_KERNEL_BITS_MFMA) return false;
  const ArtnBitsPlan &b = p.bits;
  if (b.wide8 || b.n_stages > 2 || b.gather_dim >= 0 || b.st[0].k > 6 || b.split != 0) return false;
  if (b.c128) return true; // (artn_k_bits128<*, *, true>: every tile shape; its own instantiations)
  if (b.T_in != 12 || b.T_out != 12) return false;
  // ... and the launch must be one of the FULL instantiations (launch_bits_k2): 3M pairs / steps, or non-temporal loads with
  // 3+ contracted bits per stage
  const int k1 = b.st[0].k, k2 = b.n_stages == 2 ? b.st[1].k : 0;
  const bool wide_stage = k1 == 5 || k1 == 6 || k2 == 5 || k2 == 6;
  if (wide_stage && !((k1 >= 5 && k2 >= 5) && k1 + k2 > 11) && b.m3) return true;
  return k1 >= 3 && (k2 == 0 || k2 >= 3) && b.nt_loads != 0;
}

// Two consecutive steps on the same big operand, d2's A being d1's C, in ONE pass.
// Fails with ARTN_E_UNSUPPORTED (err says why) when the pair does not fit one LDS tile;
// the caller then runs the two steps one after the other.
static inline int make_plan_fused(const ArtnStepDesc *d1, const ArtnStepDesc *d2, ArtnPlan &p, std::string &err,
                                  int n_cu = 256, int64_t min_tiles = 32) {
  int rc = validate(d1, err);
  if (!rc) rc = validate(d2, err);
  if (rc) return rc;
  memset(&p.info, 0, sizeof(p.info));
  p.n_cu = n_cu;
  if (!make_bits(d1, d2, p, n_cu, min_tiles)) { err = "not fusable: " + p.why_generic; return ARTN_E_UNSUPPORTED; }
  // a fused pair whose second step has result bits outside the tile visits every input tile once per value of those
  // bits -- and runs its FIRST stage again each time
  // 6 + 6 contracted bits: 128 fragment registers next to the accumulators -- the pair runs four-product chains with 480 bytes of
  // scratch per lane (65 TFLOP/s); two single 3M steps are faster although they move the tensor twice
  // (artn_k_wide holds them -- 16 x 16 x 4 blocks need 3 x 16 fragment registers per stage: wide_eligible() below)
  if (p.bits.st[0].k == 6 && p.bits.st[1].k == 6 && !p.bits.c128 && !tuning().fuse_66 && !wide_eligible(p)) { err = "not fusable: 6 + 6 contracted bits spill"; return ARTN_E_UNSUPPORTED; }
  if (p.stage1_repeats > tuning().fuse_max_rereads) { err = "not fusable: the first stage would run " + std::to_string(p.stage1_repeats) + " times per input tile"; return ARTN_E_UNSUPPORTED; }
  double f1, f2, a1, b1, c1, a2, b2, c2;
  step_cost(d1, f1, a1, b1, c1);
  step_cost(d2, f2, a2, b2, c2);
  p.info.flops = f1 + f2;
  p.info.bytes = (d1->dtype == ARTN_C128 ? 16.0 : 8.0) * (a1 + b1 + b2 + c2); // the intermediate C1 never touches HBM
  p.info.arith = p.bits.c128 ? 3 : (p.bits.split == 1 ? 2 : ((p.bits.st[0].m3 || p.bits.st[1].m3) ? 1 : 0));
  p.info.mfma_flops = f1 * (p.bits.st[0].m3 ? 0.75 : 1.0) + f2 * (p.bits.st[1].m3 ? 0.75 : 1.0);
  if (wide_eligible(p)) {
    // artn_k_wide: the same plan, one 512-thread workgroup per CU, a ring of three input regions + the middle one
    ArtnBitsPlan &b = p.bits;
    b.wide8 = 1;
    p.info.grid = (int32_t)std::min<int64_t>(b.n_tiles, (int64_t)n_cu);
    p.info.lds_bytes = (int32_t)(4 * (8 << ARTN_TILE_BITS_TARGET) + (8 << (b.st[0].m_bits - 5)) + (8 << (b.st[1].m_bits - 5)) + 512 * 8 + 32 * 32);
    p.info.arith = 1;
    p.info.mfma_flops = 0.75 * (f1 + f2);
  }
  return ARTN_OK;
}


// ----------------------------------------------------------------------------------------
// THREE consecutive steps on the same first operand in one pass over HBM (artn_k_bits3; round 4).
//
// The state tensor is operand 0 of every step that touches it (reference contraction.py:41-46), and consecutive gates of a
// circuit act on overlapping qubits: most contracted bits of the second and third step are bits the step before has just
// produced, and those live inside the tile by construction.  A triple therefore needs only
//     K1  u  (old bits among K2, K3)  u  the low run bits of A  u  the old bits among the low run bits of C3
// in the 2^12-element input tile -- tools/fusion_depth.py counts how often that holds on the committed schemes (n30 m14:
// three triples, 13 -> 11 passes over the 2^30-amplitude state).  Scope (everything else is left to pairs and single steps):
// complex64, power-of-two extents, no batch label, every result bit of every step inside the tile, all four tiles 2^12
// elements (each step brings as many bits as it contracts: the rank-preserving steps of a full-amplitude run), 3..5
// contracted bits per step, small operands whose fragments fit the register file next to the accumulators.
// Stage s reads region (s - 1) & 1 and writes region s & 1; the result leaves from region 1.
// ----------------------------------------------------------------------------------------
struct ChainBit {
  int born = 0, dead = 0;        // stage whose small operand brings the bit (0: a bit of A) / contracts it (0: it reaches C3)
  int64_t sBn = -1, sBk = -1;    // element stride in the small operand that brings it / that contracts it
  int64_t pos[4] = {-1, -1, -1, -1}; // element stride in A (pos[0]) and in the results C1, C2, C3; -1 where the bit does not exist
  bool seen = false;
};

static inline bool chain3_bits(const ArtnStepDesc *const d[3], std::vector<ChainBit> &bits, std::string &why) {
  for (int s = 1; s <= 3; ++s) {
    const ArtnStepDesc *ds = d[s - 1];
    for (auto &b : bits) b.seen = false;
    for (int l = 0; l < ds->n_labels; ++l) {
      const int64_t e = ds->extent[l];
      if (e == 1) continue;
      const int lg = ilog2_exact(e);
      if (lg < 0) { why = "triple: non power-of-two extent"; return false; }
      const int64_t sa = ds->stride_a[l], sb = ds->stride_b[l], sc = ds->stride_c[l];
      for (int jb = 0; jb < lg; ++jb) {
        if (sa >= 0) {
          int idx = -1;
          if (s == 1) {
            ChainBit nb;
            nb.pos[0] = sa << jb;
            bits.push_back(nb);
            idx = (int)bits.size() - 1;
          } else {
            for (int i = 0; i < (int)bits.size(); ++i)
              if (bits[i].dead == 0 && !bits[i].seen && bits[i].pos[s - 1] == (sa << jb)) { idx = i; break; }
            if (idx < 0) { why = "triple: a step does not carry the axes of the result before it"; return false; }
          }
          ChainBit &b = bits[idx];
          b.seen = true;
          if (sb >= 0 && sc < 0) { b.dead = s; b.sBk = sb << jb; }
          else if (sb < 0 && sc >= 0) b.pos[s] = sc << jb;
          else { why = "triple: batch label or label summed out of one operand"; return false; }
        } else {
          if (!(sb >= 0 && sc >= 0)) { why = "triple: label summed out of one operand"; return false; }
          ChainBit nb;
          nb.born = s; nb.sBn = sb << jb; nb.pos[s] = sc << jb; nb.seen = true;
          bits.push_back(nb);
        }
      }
    }
    for (const auto &b : bits)
      if (!b.seen && b.dead == 0 && b.born < s) { why = "triple: a step does not carry every axis of the result before it"; return false; }
  }
  return true;
}

static inline bool make_bits3(const ArtnStepDesc *const d[3], ArtnPlan &p, int n_cu, int64_t min_tiles = 1 << 14) {
  for (int s = 0; s < 3; ++s)
    if (d[s]->dtype != ARTN_C64) { p.why_generic = "triple: complex64 arithmetic only"; return false; }
  if (!tuning().fuse3 || tuning().split != 0 || !tuning().nt) { p.why_generic = "triple: switched off"; return false; }
  std::vector<ChainBit> bits;
  if (!chain3_bits(d, bits, p.why_generic)) return false;
  const int T = ARTN_TILE_BITS_TARGET;
  std::vector<int> K[4], N[4];
  for (int i = 0; i < (int)bits.size(); ++i) {
    if (bits[i].dead) K[bits[i].dead].push_back(i);
    if (bits[i].born) N[bits[i].born].push_back(i);
  }
  int frag = 0;
  bool wide = false;
  for (int s = 1; s <= 3; ++s) {
    const int k = (int)K[s].size(), n = (int)N[s].size();
    if (k < 3 || k > 5) { p.why_generic = "triple: 3..5 contracted bits per step"; return false; }
    if (n != k) { p.why_generic = "triple: a step changes the size of its tensor"; return false; }
    frag += 2 << (k - 1);
    wide = wide || k == 5;
  }
  // (fragments of three stages next to three accumulators and the prefetched tile: 80 registers -- 5+5+4 -- is the most that
  //  compiles without spills)
  if (frag > std::min(tuning().m3_frag, tuning().fuse3_frag)) { p.why_generic = "triple: small-operand fragments exceed the register budget"; return false; }
  auto in_set = [](const std::vector<int> &v, int x) { return std::find(v.begin(), v.end(), x) != v.end(); };
  // ---- the input tile: forced old bits, then the lowest free A bits up to 2^T
  std::vector<int> old_free; // bits of A that reach C3
  for (int i = 0; i < (int)bits.size(); ++i) if (bits[i].born == 0 && bits[i].dead == 0) old_free.push_back(i);
  std::sort(old_free.begin(), old_free.end(), [&](int x, int y) { return bits[x].pos[0] < bits[y].pos[0]; });
  auto run_len = [&](int which) { // contiguous run at the bottom of A (which = 0) / of C3 (which = 3), over bits alive there
    int r = 0;
    for (; r < tuning().run_max; ++r) {
      bool found = false;
      for (const auto &b : bits) if (b.pos[which] == (int64_t(1) << r)) { found = true; break; }
      if (!found) break;
    }
    return r;
  };
  const int r_in0 = run_len(0), r_out0 = run_len(3);
  if (r_in0 < 1 || r_out0 < 1) { p.why_generic = "triple: no contiguous 16-byte run at the bottom of A or C"; return false; }
  std::vector<int> told;
  int run_in = 0, run_out = 0;
  bool done = false;
  for (int cut = 0; cut <= 4 && !done; ++cut) {       // total bits shaved off the two runs (128-byte runs first)
    for (int co = (cut + 1) / 2; co >= 0 && !done; --co) {
      const int ri = r_in0 - (cut - co), ro = r_out0 - co;
      if (ri < std::min(r_in0, 2) || ro < std::min(r_out0, 2) || ri < 1 || ro < 1) continue;
      std::vector<int> t;
      for (int i = 0; i < (int)bits.size(); ++i) {
        const ChainBit &b = bits[i];
        if (b.born != 0) continue;
        if (b.dead != 0 || b.pos[0] < (int64_t(1) << ri) || (b.pos[3] >= 0 && b.pos[3] < (int64_t(1) << ro))) t.push_back(i);
      }
      if ((int)t.size() > T) continue;
      for (int i : old_free) { if ((int)t.size() >= T) break; if (!in_set(t, i)) t.push_back(i); }
      if ((int)t.size() != T) continue;                 // (a tensor of fewer than 2^12 elements is not worth a triple)
      told = t; run_in = ri; run_out = ro; done = true;
    }
  }
  if (!done) { p.why_generic = "triple: forced tile bits exceed the 2^12 tile"; return false; }
  if (run_in < tuning().fuse3_run) { p.why_generic = "triple: input runs shorter than 128 bytes"; return false; } // (the NT instantiations only)

  // ---- tile-local orders of the four tiles
  std::vector<int> tile[4];
  tile[0] = told;
  std::sort(tile[0].begin(), tile[0].end(), [&](int x, int y) { return bits[x].pos[0] < bits[y].pos[0]; });
  for (int s = 1; s <= 3; ++s) {
    for (int i : tile[s - 1]) if (bits[i].dead != s) tile[s].push_back(i);
    tile[s].insert(tile[s].end(), N[s].begin(), N[s].end());
    std::sort(tile[s].begin(), tile[s].end(), [&](int x, int y) { return bits[x].pos[s] < bits[y].pos[s]; });
    if ((int)tile[s].size() != T) { p.why_generic = "internal: triple tile size"; return false; }
  }
  ArtnBitsPlan &b = p.bits;
  memset(&b, 0, sizeof(b));
  b.n_stages = 3;
  b.T_in = b.T_mid = b.T_mid2 = b.T_out = T;
  b.r0_bits = b.r1_bits = T;
  b.run_in = run_in; b.run_out = run_out;
  b.stage_prio = tuning().stage_prio;
  auto pos = [](const std::vector<int> &v, int x) { return (int)(std::find(v.begin(), v.end(), x) - v.begin()); };
  for (int i = 0; i < T; ++i) { b.in_stride[i] = bits[tile[0][i]].pos[0]; b.out_stride[i] = bits[tile[3][i]].pos[3]; }
  for (int i = 0; i < run_in; ++i) if (b.in_stride[i] != (int64_t(1) << i)) { p.why_generic = "internal: input run broken"; return false; }
  for (int i = 0; i < run_out; ++i) if (b.out_stride[i] != (int64_t(1) << i)) { p.why_generic = "internal: output run broken"; return false; }
  const bool use_3m = tuning().bits_3m != 0 && wide;
  for (int s = 1; s <= 3; ++s) {
    ArtnStage &st = b.st[s - 1];
    const std::vector<int> &tin = tile[s - 1], &tout = tile[s];
    std::vector<int> Kx(K[s]), Nx(N[s]), Mx;
    std::sort(Kx.begin(), Kx.end(), [&](int x, int y) { return pos(tin, x) < pos(tin, y); });
    std::sort(Nx.begin(), Nx.end(), [&](int x, int y) { return pos(tout, x) < pos(tout, y); });
    for (int i : tin) if (bits[i].dead != s) Mx.push_back(i); // (in tile-input order)
    st.k = (int)Kx.size();
    st.nt = (int)Nx.size();
    st.wn_log2 = std::max(0, st.nt - 4);
    st.m3 = (use_3m && st.k == 5 && st.nt >= 5) ? 1 : 0;
    if (st.m3) st.wn_log2 = st.nt - 5;
    st.m_bits = (int)Mx.size();
    for (int i = 0; i < 5; ++i) { st.lane_in_pos[i] = pos(tin, Mx[i]); st.lane_out_pos[i] = pos(tout, Mx[i]); }
    for (int i = 5; i < st.m_bits; ++i) { st.msub_in_pos[i - 5] = pos(tin, Mx[i]); st.msub_out_pos[i - 5] = pos(tout, Mx[i]); }
    for (int i = 0; i < st.k; ++i) { st.k_in_pos[i] = pos(tin, Kx[i]); st.k_b_stride[i] = bits[Kx[i]].sBk; }
    for (int i = 0; i < st.nt; ++i) { st.n_out_pos[i] = pos(tout, Nx[i]); st.n_b_stride[i] = bits[Nx[i]].sBn; }
    st.swz_n = 0;
    bool taken[4] = {true, false, false, false};
    for (int i = 0; i < 4; ++i) if (st.lane_out_pos[i] < 4) taken[st.lane_out_pos[i]] = true;
    for (int i = 0; i < 4 && tuning().swizzle; ++i) {
      if (st.lane_out_pos[i] < 4) continue;
      int f = -1;
      for (int c = 0; c < 4; ++c) if (!taken[c]) { f = c; break; }
      if (f < 0) break;
      taken[f] = true;
      st.swz_src[st.swz_n] = st.lane_out_pos[i];
      st.swz_dst[st.swz_n] = f;
      ++st.swz_n;
    }
  }
  // 3M: every 5-bit stage or none (the M3 instantiation), the narrow stages of such a launch on 16 x 16 x 4 blocks
  {
    bool any = false, all = true;
    for (int q = 0; q < 3; ++q) if (b.st[q].k == 5) { any = any || b.st[q].m3; all = all && b.st[q].m3; }
    b.m3 = (any && all) ? 1 : 0;
    for (int q = 0; q < 3; ++q) {
      if (!b.m3 && b.st[q].m3) { b.st[q].m3 = 0; b.st[q].wn_log2 = std::max(0, b.st[q].nt - 4); }
      if (b.m3 && b.st[q].k >= 2 && b.st[q].k <= 4) b.st[q].m3 = 2;
    }
    if (wide && !b.m3) { p.why_generic = "triple: a 5-bit stage without the 3M arithmetic"; return false; } // (four-product chains of three stages spill)
  }
  // ---- outer axes: the free bits of A outside the tile, by A stride
  b.n_tiles = 1;
  b.gather_dim = -1;
  for (int i : old_free) {
    if (in_set(told, i)) continue;
    ArtnOuterDim od;
    od.ext = 2; od.sA = bits[i].pos[0]; od.sB1 = 0; od.sB2 = 0; od.sC = bits[i].pos[3]; od.log2ext = 1; od.pad_ = 0;
    b.n_tiles *= 2;
    if (b.n_outer > 0) {
      ArtnOuterDim &pr = b.outer[b.n_outer - 1];
      if (od.sA == pr.sA * pr.ext && od.sC == pr.sC * pr.ext && pr.log2ext + 1 < 31) { pr.ext *= 2; pr.log2ext += 1; continue; }
    }
    if (b.n_outer >= ARTN_MAX_OUTER) { p.why_generic = "too many outer axes"; return false; }
    b.outer[b.n_outer++] = od;
  }
  if (b.n_tiles < min_tiles) { p.why_generic = "triple: fewer than 2^14 tiles"; return false; } // (the big launches only: the NT instantiations)
  b.nt_loads = run_in >= 4 ? 1 : 0; // (shorter runs: two tiles share a 128-byte line, which must stay in L2 for the second)
  b.blocked = 0;
  for (int i = 1; i < T; ++i) if ((b.in_stride[i] & 1) || (b.out_stride[i] & 1)) { p.why_generic = "odd stride"; return false; }
  for (int i = 0; i < b.n_outer; ++i) if ((b.outer[i].sA & 1) || (b.outer[i].sC & 1)) { p.why_generic = "odd outer stride"; return false; }
  {
    int64_t si = 0, so = 0;
    for (int i = 1; i <= 8; ++i) { si += b.in_stride[i]; so += b.out_stride[i]; }
    const int64_t lim = (int64_t(1) << 28) - 1;
    if (si > lim || so > lim) { p.why_generic = "lane offsets exceed 32 bits"; return false; }
    int pow2_bits = 0;
    for (int i = 0; i < b.n_outer; ++i) pow2_bits += b.outer[i].log2ext;
    if (pow2_bits > 32) { p.why_generic = "more than 2^32 tiles"; return false; }
  }
  p.kernel = ARTN_KERNEL_BITS_MFMA;
  ArtnStepInfo &f = p.info;
  f.kernel = ARTN_KERNEL_BITS_MFMA;
  f.k_bits = b.st[0].k; f.m_tile_bits = b.st[0].m_bits; f.n_tile_bits = b.st[0].nt;
  f.k2_bits = b.st[1].k; f.n2_tile_bits = b.st[1].nt; f.k3_bits = b.st[2].k;
  f.tile_in_bits = T; f.tile_mid_bits = T; f.tile_out_bits = T;
  f.run_in_bits = run_in; f.run_out_bits = run_out;
  int64_t tabs = 0;
  for (int q = 0; q < 3; ++q) tabs += 8LL << (b.st[q].m_bits - 5);
  f.lds_bytes = (int32_t)((16LL << T) + tabs + 512LL * 8 + 32 * 32);
  f.n_tiles = b.n_tiles;
  f.a_rereads = 1;
  const int wg_per_cu = std::max(1, std::min(tuning().wg_per_cu, (160 * 1024) / f.lds_bytes));
  f.grid = (int32_t)std::min<int64_t>(b.n_tiles, (int64_t)n_cu * wg_per_cu);
  return true;
}

static inline int make_plan_fused3(const ArtnStepDesc *d1, const ArtnStepDesc *d2, const ArtnStepDesc *d3, ArtnPlan &p,
                                   std::string &err, int n_cu = 256, int64_t min_tiles = 1 << 14) {
  const ArtnStepDesc *const d[3] = {d1, d2, d3};
  for (int s = 0; s < 3; ++s)
    if (int rc = validate(d[s], err)) return rc;
  memset(&p.info, 0, sizeof(p.info));
  p.n_cu = n_cu;
  if (!make_bits3(d, p, n_cu, min_tiles)) { err = "not fusable: " + p.why_generic; return ARTN_E_UNSUPPORTED; }
  p.info.flops = 0.0;
  p.info.mfma_flops = 0.0;
  double a_in = 0, c_out = 0, small = 0;
  for (int s = 0; s < 3; ++s) {
    double f, na, nb, nc;
    step_cost(d[s], f, na, nb, nc);
    p.info.flops += f;
    p.info.mfma_flops += f * (p.bits.st[s].m3 ? 0.75 : 1.0);
    small += nb;
    if (s == 0) a_in = na;
    if (s == 2) c_out = nc;
  }
  p.info.bytes = 8.0 * (a_in + small + c_out); // neither intermediate touches HBM
  p.info.arith = p.bits.m3 ? 1 : 0;
  return ARTN_OK;
}

} // namespace artn
#endif
