// artn_kernels.hip -- gfx950 (MI355X, CDNA4) kernels and the C ABI of libartn_hip.so.
//
// What runs here replaces what torch.einsum does underneath the reference's executors
// (/root/reference/artensor/contraction.py:70 and :147-190): instead of
// permute -> contiguous copy -> bmm -> permuted view, one kernel reads A once, writes C
// once and does the bit permutation on chip:
//
//   artn_k_bits<KB,PM>   LDS-tiled bit-permuted complex GEMM on v_mfma_f32_32x32x2_f32.
//                        A workgroup stages a 2^T_in-element tile of A (all K bits + the
//                        tile's M bits) in LDS with 16-byte coalesced runs, each wave
//                        multiplies 32-column sub-tiles by the small operand held in
//                        registers, results are transposed in place through the same LDS
//                        and leave as 16-byte coalesced runs of C.
//   artn_k_generic<>     strided fallback: one thread per C element (any extents).
//   artn_k_gather_rows   row gather of the sparse-state path.
//   artn_k_axpy          slice accumulation.
//   artn_k_absmax/scale  running renormalisation (scientific_notation).
//
// Complex arithmetic on a real MFMA without wasted FLOPs: interleaved complex64 A is a
// real [M x 2K] matrix; the small operand is expanded on the fly to the real
// [2K x 2N] block matrix [[re, im], [-im, re]]; C comes out as interleaved complex64.
// 8*M*K*N real FLOP, exactly the 8 FLOP per complex multiply-add the metric counts.
#include <hip/hip_runtime.h>
#include <atomic>
#include <map>
#include <stdio.h>
#include <stdlib.h>
#include <mutex>
#include <string>
#include <type_traits>

#include "artn_plan.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ----------------------------------------------------------------------------------------
// error plumbing
// ----------------------------------------------------------------------------------------
static thread_local std::string g_err;
static thread_local std::string g_note; // why the last planned step fell back to the strided kernel
static int fail(int code, const std::string &msg) {
  g_err = msg;
  return code;
}
#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess)                                                                  \
      return fail(ARTN_E_LAUNCH, std::string(#expr) + ": " + hipGetErrorString(e_));       \
  } while (0)

// ----------------------------------------------------------------------------------------
// bit-permuted complex GEMM, one or two stages per pass
// ----------------------------------------------------------------------------------------
// Lane roles inside one v_mfma_f32_32x32x2_f32 (D[i][j] += sum_kk Aop[i][kk] * Bop[kk][j]):
//   i = real output column n' = 2*n_local + (0: re, 1: im)   -> Aop lane l: [i = l&31][kk = l>>5]
//   j = tile column m (32 elements of the input tile)          -> Bop lane l: [kk = l>>5][j = l&31]
//   kk = h = l>>5 selects complex K index kc = 2*s + h; the re and im parts of that input
//   element are fed by two consecutive MFMAs (phase p = 0, 1), so one 8-byte LDS read
//   serves two MFMAs.
// Accumulator (guide section 3): lane l holds column j = l&31 and rows
//   i = (r&3) + 8*(r>>2) + 4*(l>>5), r = 0..15  =>  n_local = (r&3)/2 + 2*h + 4*(r>>2),
//   (acc[4q+2b], acc[4q+2b+1]) = (re, im) of n_local = b + 2h + 4q.
//
// LDS holds two regions.  Copy-in fills R0; stage 1 reads R0 and scatters its result tile
// into R1; a fused second stage reads R1 and scatters into R0; copy-out streams the last
// region written.  Stages never write the region they read, so a wave needs no barrier
// between its MFMA chain and its scatter, and sub-tiles are a run-time loop (one 16-register
// accumulator whatever the tile size).
//
// Register discipline: everything that is the same for all lanes lives in SGPRs and is
// recomputed from the kernel argument per tile; per-lane state is a handful of 32-bit
// offsets.  OPAQUE_V() stops the compiler from hoisting per-chunk address arithmetic out
// of the tile loop (that hoisting, not the algorithm, used to cost >100 VGPRs).
#define OPAQUE_V(x) asm volatile("" : "+v"(x))

// Diagnostic build only (make stamps): per-phase cycle sums of every wave, never in the product.
#ifdef ARTN_STAMPS
#define ARTN_N_STAMPS 12
__device__ unsigned long long artn_stamp_buf[4096 * ARTN_N_STAMPS];
#define STAMP_DECL unsigned long long st_prev = __builtin_amdgcn_s_memtime(), st_acc[ARTN_N_STAMPS] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define STAMP(i)                                                   \
  do {                                                             \
    __builtin_amdgcn_sched_barrier(0);                             \
    unsigned long long now_ = __builtin_amdgcn_s_memtime();        \
    __builtin_amdgcn_s_waitcnt(0xC07F);                            \
    st_acc[i] += now_ - st_prev;                                   \
    st_prev = now_;                                                \
    __builtin_amdgcn_sched_barrier(0);                             \
  } while (0)
#define STAMP_FLUSH                                                                   \
  if ((threadIdx.x & 63) == 0 && blockIdx.x * 4 + (threadIdx.x >> 6) < 4096)          \
    for (int q_ = 0; q_ < ARTN_N_STAMPS; ++q_)                                        \
      artn_stamp_buf[(blockIdx.x * 4 + (threadIdx.x >> 6)) * ARTN_N_STAMPS + q_] = st_acc[q_];
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_FLUSH
#endif
// Cheap phase marks (make phases): thread 0 of each workgroup records s_memrealtime at up to 8
// points of iterations 20..21 plus its HW_ID / XCC_ID.
#if defined(ARTN_STAMPS) || defined(ARTN_PHASES)
__device__ unsigned long long artn_phase_buf[1024 * 20];
#define PHASE_MARK(k)                                                                                  \
  if (threadIdx.x == 0 && blockIdx.x < 1024) {                                                         \
    const long it_ = (tile - t0) / G;                                                                  \
    if (it_ == 0 && (k) == 0) {                                                                        \
      artn_phase_buf[blockIdx.x * 20 + 0] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);     \
      artn_phase_buf[blockIdx.x * 20 + 1] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);    \
    }                                                                                                  \
    if (it_ >= 20 && it_ < 22) artn_phase_buf[blockIdx.x * 20 + 2 + 9 * (it_ - 20) + (k)] = __builtin_amdgcn_s_memrealtime(); \
  }
#define PROG_MARK(k)                                                                  \
  if (threadIdx.x == 0 && blockIdx.x == 0 && (k) < 512) {                             \
    artn_phase_buf[(k)] = __builtin_amdgcn_s_memrealtime();                           \
    artn_phase_buf[512 + (k)] = __builtin_readcyclecounter(); /* shader clock */      \
  }
#define PROG_FINE(Lrel, k, val_)                                                       \
  if (threadIdx.x == 0 && blockIdx.x == 0 && (Lrel) < 40) {                           \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::"s"(val_) : "memory");            \
    artn_phase_buf[100 + 8 * (Lrel) + (k)] = __builtin_amdgcn_s_memrealtime();       \
  }
// artn_k_alt: lane 0 of wave 0 of each group, blocks < 128, halves 40..43: shader clock at up to 8 points of a half
#define ALT_MARK(k)                                                                                                       \
  if (lane == 0 && w4 == 0 && blockIdx.x < 64 && half >= 40 && half < 44) {                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    artn_phase_buf[(((blockIdx.x * 2 + grp) * 4) + (half - 40)) * 8 + (k)] = __builtin_amdgcn_s_memtime();               \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
  }
#define SMARK(k)                                                                    \
  if (L.mark_base >= 0 && lane == 0) {                                                \
    __builtin_amdgcn_sched_barrier(0);                                                \
    artn_phase_buf[L.mark_base + (k)] = __builtin_amdgcn_s_memtime();                 \
    __builtin_amdgcn_sched_barrier(0);                                                \
  }
#else
#define SMARK(k)
#define ALT_MARK(k)
#define PHASE_MARK(k)
#define PROG_MARK(k)
#define PROG_FINE(Lrel, k, val_)
#endif

// Tile index -> element offsets of the tile in A, B1, B2, C.
// Reading the plan's outer-axis table with scalar loads inside the tile loop costs thousands
// of cycles per tile (dependent s_load latency).  The outer axes that are powers of two
// come first, so over their bits the map tile -> offset is bit-linear: at kernel start the
// workgroup tabulates it per 4-bit nibble of the tile index in LDS (16 entries x 4 offsets
// per nibble); per tile a handful of independent uniform-address LDS reads and adds replace
// the scalar loop.  Non power-of-two axes (batch rows of the sparse path) sit above those
// bits and are decoded from the plan the slow way.
struct TileOff {
  long a, b1, b2, c;
};
struct OffTab {
  const long *tab;   // LDS: [nibble index][16][4]
  const long *delta; // LDS: [c][4] = offsets(t + G) - offsets(t) when (t / G) ends in c one-bits (G a power of two)
  int n_nib;         // nibbles covering the power-of-two prefix
  int pow2_bits;     // bits of that prefix
  int first_generic;
  int g_log2;        // log2 of the grid size if it is a power of two inside the prefix, else -1
};
template <typename PlanT>
__device__ __forceinline__ OffTab build_offset_table(const PlanT &P, long *tab, int tid, int stride = 0) {
  OffTab T;
  int bits = 0, d = 0;
  for (; d < P.n_outer && P.outer[d].log2ext >= 0; ++d) bits += P.outer[d].log2ext;
  T.tab = tab;
  T.pow2_bits = bits;
  T.n_nib = (bits + 3) >> 2;
  T.first_generic = d;
  if (tid < 16 * T.n_nib) {
    const int nibble = tid >> 4, val = tid & 15;
    long a = 0, b1 = 0, b2 = 0, c = 0;
    for (int b = 0; b < 4; ++b) {
      const int bit = 4 * nibble + b;
      if (!((val >> b) & 1) || bit >= bits) continue;
      int lo = 0;
      for (int e = 0; e < T.first_generic; ++e) { // which axis owns tile bit `bit`
        const int lg = P.outer[e].log2ext;
        if (bit < lo + lg) {
          const int r = bit - lo;
          a += P.outer[e].sA << r;
          b1 += P.outer[e].sB1 << r;
          b2 += P.outer[e].sB2 << r;
          c += P.outer[e].sC << r;
          break;
        }
        lo += lg;
      }
    }
    long *e = tab + (long)tid * 4;
    e[0] = a; e[1] = b1; e[2] = b2; e[3] = c;
  }
  // Grid-stride increments: with G = 2^g, tile + G flips the trailing ones of (tile >> g) and
  // sets the next bit, so the offset difference depends only on the number c of trailing ones.
  long *dl = tab + 8 * 16 * 4;
  T.delta = dl;
  const int G = stride ? stride : (P.blocked ? 1 : gridDim.x); // distance between consecutive tiles of a workgroup
  T.g_log2 = -1;
  if ((G & (G - 1)) == 0 && T.first_generic == P.n_outer) {
    int g = 0;
    while ((1 << g) < G) ++g;
    if (g <= bits) T.g_log2 = g;
  }
  if (T.g_log2 >= 0 && tid >= 128 && tid < 128 + 32) {
    const int cnum = tid - 128; // number of trailing ones
    long d[4] = {0, 0, 0, 0};
    for (int b = 0; b <= cnum; ++b) {
      const int bit = T.g_log2 + b;
      if (bit >= bits) break;
      int lo = 0;
      for (int e = 0; e < T.first_generic; ++e) {
        const int lg = P.outer[e].log2ext;
        if (bit < lo + lg) {
          const int r = bit - lo;
          const long sgn = b == cnum ? 1 : -1;
          d[0] += sgn * (P.outer[e].sA << r);
          d[1] += sgn * (P.outer[e].sB1 << r);
          d[2] += sgn * (P.outer[e].sB2 << r);
          d[3] += sgn * (P.outer[e].sC << r);
          break;
        }
        lo += lg;
      }
    }
    long *e = dl + cnum * 4;
    e[0] = d[0]; e[1] = d[1]; e[2] = d[2]; e[3] = d[3];
  }
  return T;
}
__device__ __forceinline__ long uniform64(long x) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)x);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long)x >> 32));
  return (long)(((unsigned long)hi << 32) | lo);
}
// (gc: the contribution of the non power-of-two axes of the last tile looked up.  A workgroup that walks a contiguous
//  range of tiles -- every launch with batch rows -- stays on one batch row for 2^pow2_bits tiles: without the cache
//  each tile paid two 64-bit divisions and, with a row gather, two dependent loads of row indices before its first
//  LDS read could be waited for: 1.2 us of a 5.8 us tile period in the chunked steps of the n30 x 10 000 scheme)
struct GenCache {
  long r, a, b1, b2, c;
};
template <bool GATHER = false, typename PlanT = ArtnBitsPlan>
__device__ __forceinline__ TileOff tile_offsets(const PlanT &P, const OffTab &T, long tile, GenCache *gc = nullptr) {
  long a = 0, b1 = 0, b2 = 0, c = 0;
#pragma unroll
  for (int n = 0; n < 8; ++n) {
    if (n < T.n_nib) {
      const long *e = T.tab + ((n << 4) + (int)((tile >> (4 * n)) & 15)) * 4;
      a += e[0]; b1 += e[1]; b2 += e[2]; c += e[3];
    }
  }
  TileOff t = {uniform64(a), uniform64(b1), uniform64(b2), uniform64(c)};
  if (T.first_generic < P.n_outer) { // rare: batch axes with arbitrary extents
    long r = tile >> T.pow2_bits;
    if (gc && gc->r == r) {
      t.a += gc->a; t.b1 += gc->b1; t.b2 += gc->b2; t.c += gc->c;
      return t;
    }
    const TileOff base = t;
    const long r_key = r;
    for (int d = T.first_generic; d < P.n_outer; ++d) {
      const long ext = P.outer[d].ext;
      const long x = r % ext;
      r /= ext;
      long xa = x, xb = x;
      if constexpr (GATHER) { // fused row gather: the operands are read at rows_a[x] / rows_b[x]
        if (d == P.gather_dim) {
          if (P.rows_a) {
            xa = P.rows_a[x];
            if (xa < 0 || xa >= P.src_rows_a) { xa = 0; if (P.gather_err) *P.gather_err = 1; }
          }
          if (P.rows_b) {
            xb = P.rows_b[x];
            if (xb < 0 || xb >= P.src_rows_b) { xb = 0; if (P.gather_err) *P.gather_err = 1; }
          }
        }
      }
      t.a += xa * P.outer[d].sA;
      t.b1 += xb * P.outer[d].sB1;
      t.b2 += x * P.outer[d].sB2;
      t.c += x * P.outer[d].sC;
    }
    if (gc) {
      gc->r = r_key;
      gc->a = uniform64(t.a - base.a); gc->b1 = uniform64(t.b1 - base.b1); gc->b2 = uniform64(t.b2 - base.b2); gc->c = uniform64(t.c - base.c);
    }
  }
  return t;
}

// offsets(tile + G) from offsets(tile): one 32-byte LDS lookup when the grid is a power of two
template <bool GATHER = false, typename PlanT = ArtnBitsPlan>
__device__ __forceinline__ TileOff next_offsets(const PlanT &P, const OffTab &T, const TileOff &cur, long tile,
                                                long G, GenCache *gc = nullptr) {
  if (T.g_log2 < 0) return tile_offsets<GATHER>(P, T, tile + G, gc);
  const unsigned hi = (unsigned)(tile >> T.g_log2);
  const int c = __builtin_ctz(~hi);
  const long *e = T.delta + c * 4;
  TileOff t = {uniform64(cur.a + e[0]), uniform64(cur.b1 + e[1]), uniform64(cur.b2 + e[2]), uniform64(cur.c + e[3])};
  return t;
}

// LDS accesses by 32-bit byte address.  artn_k_bits has no static __shared__ data, so its
// dynamic LDS segment starts at byte 0 of the workgroup's allocation (checked once at kernel
// start): addresses are plain integers, no base register and no add per access.
typedef float v2f_t __attribute__((ext_vector_type(2)));
typedef v2f_t __attribute__((address_space(3))) lds_v2f_t;
typedef f32x4 __attribute__((address_space(3))) lds_v4f_t;
typedef unsigned u2_t __attribute__((ext_vector_type(2)));
typedef u2_t __attribute__((address_space(3))) lds_u2_t;
typedef __attribute__((address_space(3))) unsigned char lds_byte_t;
__device__ __forceinline__ v2f_t lds_read8(unsigned a) { return *(lds_v2f_t *)(unsigned long)a; }
__device__ __forceinline__ void lds_write8(unsigned a, v2f_t v) { *(lds_v2f_t *)(unsigned long)a = v; }
__device__ __forceinline__ f32x4 lds_read16(unsigned a) { return *(lds_v4f_t *)(unsigned long)a; }
__device__ __forceinline__ void lds_write16(unsigned a, f32x4 v) { *(lds_v4f_t *)(unsigned long)a = v; }
__device__ __forceinline__ u2_t lds_read_u2(unsigned a) { return *(lds_u2_t *)(unsigned long)a; }
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double __attribute__((address_space(3))) lds_f64_t;
__device__ __forceinline__ double lds_read_f64(unsigned a) { return *(lds_f64_t *)(unsigned long)a; }
__device__ __forceinline__ void lds_write_f64(unsigned a, double v) { *(lds_f64_t *)(unsigned long)a = v; }

// Copy-in.  Full-size tiles (2^12 elements: exactly 8 x 16 B per thread) are software
// pipelined: issue_loads puts the 8 chunks of tile t+1 in flight (uniform 64-bit base in SGPRs
// + one 32-bit per-lane byte offset) while tile t is computed, store_lds writes them to LDS
// linearly.  Every issued load is consumed: a load whose result is never used leaves the
// compiler a pending write to guard, and it does so with vmcnt waits in the middle of the MFMA
// chain that drain the whole prefetch.  Other tile sizes take the plain loop of copy_in_sync.
// hi[b] = byte stride of tile-local bit 9+b (0 beyond the tile).
__device__ __forceinline__ long chunk_off(const long (&hi)[4], int i) {
  long off = 0;
#pragma unroll
  for (int b = 0; b < 4; ++b)
    if ((i >> b) & 1) off += hi[b];
  return off;
}
__device__ __forceinline__ f32x4 load_chunk(const char *__restrict__ Abase, long off, unsigned lane_off) {
#ifdef ARTN_ABLATE_MEM
  f32x4 r = f32x4{1.f, 2.f, 3.f, 4.f};
  asm volatile("" : "+v"(r) : "s"(Abase), "v"(lane_off));
  return r;
#else
  return *reinterpret_cast<const f32x4 *>(Abase + off + lane_off);
#endif
}
// NT: every A tile is read exactly once by the launch, in full 128-byte runs: non-temporal loads
// keep the stream from displacing lines that will be used again.  (A compile-time choice: behind
// a run-time branch the compiler merges the two loads and drops the hint.)
template <int NV, bool NT = false, int U0 = 0, int U1 = NV>
__device__ __forceinline__ void issue_loads(f32x4 (&v)[NV], const char *__restrict__ Abase, const long (&hi)[4],
                                            unsigned lane_off) {
#pragma unroll
  for (int u = U0; u < U1; ++u) {
#ifndef ARTN_ABLATE_MEM
    if constexpr (NT) {
      v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(Abase + chunk_off(hi, u) + lane_off));
      continue;
    }
#endif
    v[u] = load_chunk(Abase, chunk_off(hi, u), lane_off);
  }
}
template <int NV, int U1 = NV>
__device__ __forceinline__ void store_lds(const f32x4 (&v)[NV], unsigned ldsb, unsigned tid16) {
#pragma unroll
  for (int u = 0; u < U1; ++u) lds_write16(ldsb + tid16 + u * (ARTN_WG_THREADS * 16), v[u]);
}
__device__ __forceinline__ void copy_in_sync(const char *__restrict__ Abase, const long (&hi)[4], unsigned lane_off,
                                             unsigned ldsb, unsigned tid16, int n_iters) {
  for (int i = 0; i < n_iters; ++i)
    lds_write16(ldsb + tid16 + i * (ARTN_WG_THREADS * 16), load_chunk(Abase, chunk_off(hi, i), lane_off));
}

// XOR swizzle of an LDS region (ArtnStage::swz_*), applied to byte offsets.  It is linear over
// XOR and every LDS address below is a sum of disjoint bit fields, so each field is swizzled
// once at kernel start and the fields are combined with XOR: no per-access cost.
__device__ __forceinline__ unsigned swz(unsigned byte_off, const ArtnStage *z) {
  if (z) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (i < z->swz_n && ((byte_off >> (z->swz_src[i] + 3)) & 1)) byte_off ^= 8u << z->swz_dst[i];
  }
  return byte_off;
}

// Per-wave / per-lane constants of one stage, computed once per kernel.
template <int KB>
struct StageConst {
  unsigned lane_in, lane_out;         // per-lane byte offsets: LDS input tile, LDS output tile
  long lane_b;                        // per-lane byte offset into the small operand (only used when fragments are reloaded)
  bool w_valid;
  unsigned kin[KB > 1 ? KB : 2];      // byte offset of K bit b in the LDS input tile
  long kb[KB > 1 ? KB : 2];           // byte stride of K bit b in the small operand
  unsigned o0, o2, o3;                // byte offsets of N bits 0, 2, 3 in the LDS output tile
  unsigned o1, o4;                    // 3M stages: N bits 1 and 4 (accumulator row = (r & 3) + 4h + 8(r >> 2))
  bool m3;                            // three real products per complex product (ArtnStage::m3)
  int k_hi;                           // contracted bits beyond the chain (0..2), looped over
  unsigned kin_hi[2];
  long kb_hi[2];
  int nt_eff, wm, wm_count, msubs;
  unsigned msub_tab;                  // LDS byte address of the table: sub-tile -> (input, output) byte offsets
  unsigned mo0x, mo0y, mo1x, mo1y;    // the table entries of this wave's first two sub-tiles (wm, wm + wm_count): the first
                                      // operand reads of a stage do not wait for a table read
  unsigned hin, hout;                 // 3M on 16 x 16 x 4 blocks (4-bit stages of the 3M instantiations): byte offsets of column
                                      // bit 4 of a 32-column sub-tile (its two 16-column halves) in the input / output tile
  int ksplit_wave;                    // >= 0: the waves split the chain of a big-K tile; this wave's share
  unsigned ksplit_scratch;            // LDS byte address of the 3 x 4 KiB partial blocks
#if defined(ARTN_PHASES)
  int mark_base;                      // diagnostics: slot in artn_phase_buf for the marks inside the stage, or -1
#endif
};
// zin: stage whose output region this stage reads (nullptr: the unswizzled copy-in region).
// hb: the contracted bit carried by the lane half h (0 for the fp32 chain: kc = 2s + h; 2 for the
// split-bf16 chain, whose MFMA takes 8 complex kc per instruction: kc = 8g + 4h + u).
template <int KB, bool M3 = false>
__device__ __forceinline__ StageConst<KB> stage_const(const ArtnStage &st, const ArtnStage *zin, int j, int h, int wave,
                                                      unsigned tab, unsigned in_base, unsigned out_base, int hb = 0) {
  StageConst<KB> L;
  const int wn = wave & ((1 << st.wn_log2) - 1);
  L.wm = wave >> st.wn_log2;
  L.wm_count = 4 >> st.wn_log2;
  L.msubs = 1 << (st.m_bits - 5);
  L.nt_eff = st.nt < 4 ? st.nt : 4;
  L.msub_tab = tab;
  {
    unsigned e[2][2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int m = L.wm + q * L.wm_count;
      unsigned oi = 0, oo = 0;
#pragma unroll
      for (int b = 0; b < 9; ++b) {
        if (b < st.m_bits - 5 && ((m >> b) & 1)) {
          oi += 8u << st.msub_in_pos[b];
          oo += 8u << st.msub_out_pos[b];
        }
      }
      e[q][0] = swz(oi, zin);
      e[q][1] = swz(oo, &st);
    }
    L.mo0x = e[0][0]; L.mo0y = e[0][1]; L.mo1x = e[1][0]; L.mo1y = e[1][1];
  }
  L.ksplit_wave = -1;
  L.ksplit_scratch = 0;
#if defined(ARTN_PHASES)
  L.mark_base = -1;
#endif
  L.lane_in = (unsigned)h << (st.k_in_pos[hb] + 3);
  L.lane_out = 0;
#pragma unroll
  for (int b = 0; b < 5; ++b) {
    if ((j >> b) & 1) {
      L.lane_in += 8u << st.lane_in_pos[b];
      L.lane_out += 8u << st.lane_out_pos[b];
    }
  }
  L.m3 = M3 && (KB == 5 || KB == 6); // (the planner launches the M3 instantiation only when every such stage asks for it)
  if (L.m3) L.lane_out += (unsigned)h << (st.n_out_pos[2] + 3);
  else if (st.nt > 1) L.lane_out += (unsigned)h << (st.n_out_pos[1] + 3);
  const int nloc = L.m3 ? j : j >> 1;
  L.w_valid = L.m3 || (nloc >> L.nt_eff) == 0;
  L.lane_b = (long)h * st.k_b_stride[hb] * 8;
#pragma unroll
  for (int b = 0; b < 5; ++b)
    if (b < (L.m3 ? 5 : L.nt_eff) && ((nloc >> b) & 1)) L.lane_b += st.n_b_stride[b] * 8;
  const int nb0 = L.m3 ? 5 : 4; // first column bit dealt to the waves
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    if (b < st.wn_log2 && ((wn >> b) & 1) && nb0 + b < 6) {
      L.lane_out += 8u << st.n_out_pos[nb0 + b];
      L.lane_b += st.n_b_stride[nb0 + b] * 8;
    }
  }
#pragma unroll
  for (int b = 0; b < KB; ++b) {
    L.kin[b] = swz(8u << st.k_in_pos[b], zin);
    L.kb[b] = st.k_b_stride[b] * 8;
  }
  L.k_hi = st.k > KB ? st.k - KB : 0;
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    L.kin_hi[b] = b < L.k_hi ? swz(8u << st.k_in_pos[KB + b], zin) : 0u;
    L.kb_hi[b] = b < L.k_hi ? st.k_b_stride[KB + b] * 8 : 0;
  }
  L.o0 = st.nt > 0 ? swz(8u << st.n_out_pos[0], &st) : 0;
  L.o2 = st.nt > 2 ? swz(8u << st.n_out_pos[2], &st) : 0;
  L.o3 = st.nt > 3 ? swz(8u << st.n_out_pos[3], &st) : 0;
  L.o1 = st.nt > 1 ? swz(8u << st.n_out_pos[1], &st) : 0;
  L.o4 = st.nt > 4 ? swz(8u << st.n_out_pos[4], &st) : 0;
  L.hin = L.hout = 0;
  if constexpr (M3 && KB >= 2 && KB <= 4) {
    // 2- to 4-bit stage of a 3M instantiation: blocks of 16 complex columns n x 16 tile columns m x 4 contracted values on
    // v_mfma_f32_16x16x4_f32 (A[l & 15][l >> 4], B[l >> 4][l & 15]; D: column l & 15, rows 4 (l >> 4) + r): lane (jj, g) reads
    // x[kc = 4 s + g][m = 16 half + jj], the W lane (n, g) holds w[kc = 4 s + g][n], accumulator register r is column
    // n = 4 g + r; three products per complex product as in the 32-column stages, a quarter fewer MFMA cycles than 4M
    const int ln = j | (h << 5), jj = ln & 15, g = ln >> 4;
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
      if (b < st.wn_log2 && ((wn >> b) & 1)) {
        lo += 8u << st.n_out_pos[4 + b];
        lb += st.n_b_stride[4 + b] * 8;
      }
    }
    L.w_valid = (jj >> L.nt_eff) == 0;
    L.lane_in = li;
    L.lane_out = lo;
    L.lane_b = lb;
    L.hin = swz(8u << st.lane_in_pos[4], zin);
    L.hout = swz(8u << st.lane_out_pos[4], &st);
  }
  // region bases are multiples of the region size: XOR-ing them in equals adding them
  L.lane_in = swz(L.lane_in, zin) ^ in_base;
  L.lane_out = swz(L.lane_out, &st) ^ out_base;
  return L;
}
// Fill the LDS sub-tile table of a stage (all threads cooperate; caller barriers).
__device__ __forceinline__ void fill_msub_table(const ArtnStage &st, const ArtnStage *zin, uint2 *tab, int tid) {
  const int msubs = 1 << (st.m_bits - 5);
  for (int m = tid; m < msubs; m += ARTN_WG_THREADS) {
    unsigned oi = 0, oo = 0;
    for (int b = 0; b < st.m_bits - 5; ++b) {
      if ((m >> b) & 1) {
        oi += 8u << st.msub_in_pos[b];
        oo += 8u << st.msub_out_pos[b];
      }
    }
    tab[m] = make_uint2(swz(oi, zin), swz(oo, &st));
  }
}

// Small-operand fragments of this lane: W[n' = lane&31][(kc = 2s + h, p)], p = 0 (x re) / 1 (x im).
template <int KB>
__device__ __forceinline__ void load_w(float (&W0)[1 << (KB - 1)], float (&W1)[1 << (KB - 1)],
                                       const char *__restrict__ Bbase, const StageConst<KB> &L, int ro) {
  // (callers add the byte offset of looped-over contracted bits to Bbase)
  constexpr int S = 1 << (KB - 1);
#pragma unroll
  for (int s = 0; s < S; ++s) {
    long ko = 0;
#pragma unroll
    for (int b = 1; b < KB; ++b)
      if ((s >> (b - 1)) & 1) ko += L.kb[b];
    float2 bv = make_float2(0.f, 0.f);
    if (L.w_valid) bv = *reinterpret_cast<const float2 *>(Bbase + ko + L.lane_b);
    W0[s] = ro ? bv.y : bv.x;
    W1[s] = ro ? bv.x : -bv.y;
  }
  // Consume the fragments HERE: the reload is conditional, and if the compiler is left to
  // place the s_waitcnt for these loads at their first use (inside the MFMA chain) it emits an
  // unconditional vmcnt(0) there, which drains the next tile's prefetched loads every tile.
#pragma unroll
  for (int s = 0; s < S; ++s) asm volatile("" : "+v"(W0[s]), "+v"(W1[s]));
}

// 3M stages: row n = lane&31 of the small operand, (re, im, re + im) of B[kc = 2s + h][n]
template <int KB>
__device__ __forceinline__ void load_w3(float (&W0)[1 << (KB - 1)], float (&W1)[1 << (KB - 1)], float (&W2)[1],
                                        const char *__restrict__ Bbase, const StageConst<KB> &L) {
  constexpr int S = 1 << (KB - 1);
#pragma unroll
  for (int s = 0; s < S; ++s) {
    long ko = 0;
#pragma unroll
    for (int b = 1; b < KB; ++b)
      if ((s >> (b - 1)) & 1) ko += L.kb[b];
    const float2 bv = *reinterpret_cast<const float2 *>(Bbase + ko + L.lane_b);
    W0[s] = bv.x;
    W1[s] = bv.y;
  }
  (void)W2;
#pragma unroll
  for (int s = 0; s < S; ++s) asm volatile("" : "+v"(W0[s]), "+v"(W1[s])); // see load_w
}

// Split-bf16 arithmetic.  v_mfma_f32_32x32x16_bf16 runs at 16x the rate of the fp32 MFMA.  An
// fp32 number is the exact sum of three bf16 pieces (8 significand bits each), so the product of
// two is the sum of nine piece products; the six with piece indices i + j <= 2 carry everything
// above 2^-23 of the product -- fp32-grade results from 6 bf16 MFMAs per 8 complex kc instead of
// 16 fp32 MFMAs (NP = 3).  NP = 1 keeps only the leading pieces: plain bf16 operands, fp32
// accumulation (the reduced-precision path of BASELINE configs[4]).
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
// 2- to 4-bit stage of a 3M instantiation (16 x 16 x 4 blocks): W0[s] / W1[s] = re / im of w[kc = 4 s + g][n = lane & 15],
// s < 2^(KB - 2)
template <int KB>
__device__ __forceinline__ void load_w4m3(float (&W0)[1 << (KB - 1)], float (&W1)[1 << (KB - 1)], const char *__restrict__ Bbase,
                                          const StageConst<KB> &L) {
  static_assert(KB >= 2 && KB <= 4, "16 x 16 x 4 blocks: 2- to 4-bit stages");
  constexpr int NST = 1 << (KB - 2);
#pragma unroll
  for (int s = 0; s < NST; ++s) {
    long ko = 0;
#pragma unroll
    for (int b = 2; b < KB; ++b)
      if ((s >> (b - 2)) & 1) ko += L.kb[b];
    float2 bv = make_float2(0.f, 0.f);
    if (L.w_valid) bv = *reinterpret_cast<const float2 *>(Bbase + ko + L.lane_b);
    W0[s] = bv.x;
    W1[s] = bv.y;
  }
#pragma unroll
  for (int s = NST; s < (1 << (KB - 1)); ++s) { W0[s] = 0.f; W1[s] = 0.f; }
#pragma unroll
  for (int s = 0; s < NST; ++s) asm volatile("" : "+v"(W0[s]), "+v"(W1[s])); // see load_w
}

__device__ __forceinline__ unsigned pack_bf16(float a, float b) { // RNE, a in the low half
  v2f_t v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
// (a, b) -> NP packed pieces; piece p+1 splits what piece p left over (exact subtraction).
// The first piece is rounded to nearest even, so what it leaves has no preferred sign and the
// dropped piece products do not add up to a bias; the later pieces are cut off (their top 16
// bits: one v_perm_b32 per pair), which is exact for the last one -- what the first two pieces
// leave of an fp32 number has at most 8 significant bits.
template <int NP>
__device__ __forceinline__ void split_pair(float a, float b, unsigned (&piece)[NP]) {
  unsigned d = pack_bf16(a, b);
  piece[0] = d;
#pragma unroll
  for (int p = 1; p < NP; ++p) {
    a -= __builtin_bit_cast(float, p == 1 ? d << 16 : __builtin_bit_cast(unsigned, a) & 0xffff0000u);
    b -= __builtin_bit_cast(float, p == 1 ? d & 0xffff0000u : __builtin_bit_cast(unsigned, b) & 0xffff0000u);
    d = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, a), 0x07060302u);
    piece[p] = d;
  }
}
// Small-operand fragments for the split chain: WS[p][g] = 8 bf16 of piece p for the lane's row
// n' = lane&31 and the 4 complex kc = 8g + 4h + u: element 2u multiplies re(a), 2u+1 im(a).
template <int KB, int NP>
__device__ __forceinline__ void load_w_split(u32x4_t (&WS)[NP][KB >= 3 ? 1 << (KB - 3) : 1],
                                             const char *__restrict__ Bbase, const StageConst<KB> &L, int ro) {
  constexpr int G = KB >= 3 ? 1 << (KB - 3) : 1;
#pragma unroll
  for (int g = 0; g < G; ++g) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      long ko = 0;
      if (u & 1) ko += L.kb[0];
      if (u & 2) ko += L.kb[KB > 1 ? 1 : 0];
#pragma unroll
      for (int b = 3; b < KB; ++b)
        if ((g >> (b - 3)) & 1) ko += L.kb[b];
      float2 bv = make_float2(0.f, 0.f);
      if (L.w_valid) bv = *reinterpret_cast<const float2 *>(Bbase + ko + L.lane_b);
      unsigned piece[NP];
      split_pair<NP>(ro ? bv.y : bv.x, ro ? bv.x : -bv.y, piece);
#pragma unroll
      for (int p = 0; p < NP; ++p) WS[p][g][u] = piece[p];
    }
  }
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int g = 0; g < G; ++g) asm volatile("" : "+v"(WS[p][g])); // see load_w
}

// One plane of bf16 fragments (NP = 1) for the 7-8 bit kernel: row[g] = 8 bf16 for kc = 8g + 4h + u
template <int KB>
__device__ __forceinline__ void load_w_plane(u32x4_t (&row)[KB >= 3 ? 1 << (KB - 3) : 1], const char *__restrict__ Bbase,
                                             const StageConst<KB> &L, int ro) {
  constexpr int G = KB >= 3 ? 1 << (KB - 3) : 1;
#pragma unroll
  for (int g = 0; g < G; ++g) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      long ko = 0;
      if (u & 1) ko += L.kb[0];
      if (u & 2) ko += L.kb[KB > 1 ? 1 : 0];
#pragma unroll
      for (int b = 3; b < KB; ++b)
        if ((g >> (b - 3)) & 1) ko += L.kb[b];
      float2 bv = make_float2(0.f, 0.f);
      if (L.w_valid) bv = *reinterpret_cast<const float2 *>(Bbase + ko + L.lane_b);
      row[g][u] = pack_bf16(ro ? bv.y : bv.x, ro ? bv.x : -bv.y);
    }
  }
#pragma unroll
  for (int g = 0; g < G; ++g) asm volatile("" : "+v"(row[g])); // see load_w
}

// One stage on this wave's sub-tiles.  Per sub-tile: a chain of 2^KB MFMAs over the
// contracted bits, then the scatter of the 32 x 16 complex result into the output region.
// The chain must never wait on LDS and the scatter must not sit between two chains, so the
// stage is software-pipelined over "units" of CH <= 16 chain steps:
//   * the operands of unit u+1 (CH x ds_read_b64) are issued before the MFMAs of unit u,
//     into the other half of a ping-pong register buffer;
//   * accumulators ping-pong too: the scatter of sub-tile i is issued after the first MFMA
//     pair of sub-tile i+1, so the LDS writes drain under that chain.
template <int KB, bool BIGK, int NP = 0, bool M3 = false>
struct StageRun {
  static constexpr int S = 1 << (KB - 1);
  static constexpr int CH = S < 16 ? S : 16; // chain steps per unit
  static constexpr int UPS = S / CH;         // units per sub-tile (1, or 2 for KB = 6)
  static constexpr bool SPLIT = NP > 0 && KB >= 3;
  static constexpr int G = KB >= 3 ? 1 << (KB - 3) : 1;
  // split fragments: [piece][group]; the 7-8 bit kernel runs NP = 1 only and uses the first index
  // for the looped-over value instead
  static constexpr int WSD = SPLIT ? (BIGK ? 4 : NP) : 1;
  static_assert(!(BIGK && NP > 1), "the 7-8 bit kernel has no three-piece split");
  static constexpr bool CAN3M = M3 && (KB == 5 || KB == 6) && !BIGK && NP == 0;
  static constexpr bool K4M3 = M3 && KB >= 2 && KB <= 4 && !BIGK && NP == 0; // 3M on 16 x 16 x 4 blocks (stage_const)
  static constexpr int NST4 = KB >= 2 && KB <= 4 ? 1 << (KB - 2) : 1;         // ... chain steps of 4 contracted values
  const StageConst<KB> &L;
  float (&W0)[S];
  float (&W1)[S];
  float (&W2)[1]; // (unused: the 3M sum fragment is formed on the fly)
  int h, lane;
  // 7-8 contracted bits (BIGK): fragments for every value of the looped-over bits, all in
  // registers (the instantiation runs one wave per SIMD, so 512 VGPRs are available)
  float (&WH0)[BIGK ? 3 : 1][S];
  float (&WH1)[BIGK ? 3 : 1][S];
  u32x4_t (&WS)[WSD][G]; // split chain: bf16 pieces of the small operand

  // LDS offset of read s of a chain: fp32 chain kc = 2s + h; split chain kc = 8g + 4h + u, s = 4g + u
  __device__ __forceinline__ unsigned ko(int s) const {
    unsigned k = 0;
    if constexpr (SPLIT) {
      if (s & 1) k ^= L.kin[0];
      if (s & 2) k ^= L.kin[1];
#pragma unroll
      for (int b = 3; b < KB; ++b)
        if ((s >> (b - 1)) & 1) k ^= L.kin[b];
    } else {
#pragma unroll
      for (int b = 1; b < KB; ++b)
        if ((s >> (b - 1)) & 1) k ^= L.kin[b];
    }
    return k;
  }
  // operands of one unit: steps [base, base + CH) of the sub-tile at LDS offset li
  template <int BASE>
  __device__ __forceinline__ void load_unit(v2f_t (&buf)[CH], unsigned li) const {
#pragma unroll
    for (int s = 0; s < CH; ++s) buf[s] = lds_read8(li ^ ko(BASE + s));
  }
  __device__ __forceinline__ void scatter(const f32x16 &acc, unsigned lo) const {
    if (L.nt_eff == 4) { // full 16-column tile: no per-lane predicate (the common case, n >= 4)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int b0 = 0; b0 < 2; ++b0) {
          const unsigned o = lo ^ (b0 ? L.o0 : 0u) ^ ((q & 1) ? L.o2 : 0u) ^ ((q >> 1) ? L.o3 : 0u);
          lds_write8(o, v2f_t{acc[4 * q + 2 * b0], acc[4 * q + 2 * b0 + 1]});
        }
      }
      return;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int b0 = 0; b0 < 2; ++b0) {
        const int nl = b0 + 2 * h + 4 * (q & 1) + 8 * (q >> 1);
        if ((nl >> L.nt_eff) == 0) {
          const unsigned o = lo ^ (b0 ? L.o0 : 0u) ^ ((q & 1) ? L.o2 : 0u) ^ ((q >> 1) ? L.o3 : 0u);
          lds_write8(o, v2f_t{acc[4 * q + 2 * b0], acc[4 * q + 2 * b0 + 1]});
        }
      }
    }
  }
  // MFMAs of one unit; after the first pair, the pending scatter of the previous sub-tile
  template <int BASE>
  __device__ __forceinline__ void chain_unit(f32x16 &acc, const v2f_t (&buf)[CH], bool pending, const f32x16 &pacc,
                                             unsigned plo) const {
    if constexpr (SPLIT) {
      // Software pipeline over the groups of 8 complex kc: the pieces of group g+1 are split
      // (VALU) while the NT piece products of group g run on the matrix pipe.  Issue is in
      // order, so the two streams must alternate in the instruction stream itself: after each
      // MFMA comes the split of one element of the next group, fenced by sched_barrier so the
      // compiler keeps that order.
      constexpr int NPe = SPLIT ? NP : 1;
      constexpr int NG = CH / 4;
      constexpr int NT = NPe * (NPe + 1) / 2; // piece products kept: i + j <= NP - 1
      u32x4_t a_cur[NPe], a_nxt[NPe];
      auto split_elem = [&](int gg, int u, u32x4_t (&a)[NPe]) {
        unsigned piece[NPe];
        split_pair<NPe>(buf[4 * gg + u].x, buf[4 * gg + u].y, piece);
#pragma unroll
        for (int p = 0; p < NPe; ++p) a[p][u] = piece[p];
      };
#pragma unroll
      for (int u = 0; u < 4; ++u) split_elem(0, u, a_cur);
#pragma unroll
      for (int gg = 0; gg < NG; ++gg) {
        const int g = BASE / 4 + gg;
        int t = 0;
        // smallest terms first
#pragma unroll
        for (int sum = NPe - 1; sum >= 0; --sum) {
#pragma unroll
          for (int i = 0; i <= sum; ++i) {
#ifdef ARTN_ABLATE_MFMA
            asm volatile("" ::"v"(a_cur[sum - i]), "v"(WS[i][g]));
#else
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, WS[i][g]),
                                                          __builtin_bit_cast(bf16x8_t, a_cur[sum - i]), acc, 0, 0, 0);
#endif
            if (gg + 1 < NG) {
              // NT >= 4 products: one element per product; fewer (NP = 1): all four after the only one
              if (NT >= 4) {
                if (t < 4) split_elem(gg + 1, t, a_nxt);
              } else {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                  if (u * NT / 4 == t) split_elem(gg + 1, u, a_nxt);
              }
              __builtin_amdgcn_sched_barrier(0);
            }
            ++t;
          }
        }
        if (gg + 1 < NG) {
#pragma unroll
          for (int p = 0; p < NPe; ++p) a_cur[p] = a_nxt[p];
        }
        if (gg == 0 && BASE == 0 && pending) scatter(pacc, plo);
      }
      return;
    }
#pragma unroll
    for (int s = 0; s < CH; ++s) {
#ifdef ARTN_ABLATE_MFMA
      asm volatile("" ::"v"(buf[s].x), "v"(buf[s].y), "v"(W0[BASE + s]), "v"(W1[BASE + s]));
#else
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(W0[BASE + s], buf[s].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(W1[BASE + s], buf[s].y, acc, 0, 0, 0);
#endif
      if (s == 0 && BASE == 0 && pending) scatter(pacc, plo);
    }
  }
  __device__ __forceinline__ void zero(f32x16 &acc) const {
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  }

  // More than 6 contracted bits: the extra one or two are looped over.  The tile holds all K
  // bits, so a wave has at most two sub-tiles; accumulators stay in registers across the loop,
  // fragments of value 0 live in W0/W1 and of values 1..3 in WH0/WH1, operands ping-pong.
  template <int HI>
  __device__ __forceinline__ const float (&w0(void) const)[S] {
    if constexpr (HI == 0) return W0; else return WH0[(BIGK ? HI - 1 : 0)];
  }
  template <int HI>
  __device__ __forceinline__ const float (&w1(void) const)[S] {
    if constexpr (HI == 0) return W1; else return WH1[(BIGK ? HI - 1 : 0)];
  }
  template <int HI, int BASE>
  __device__ __forceinline__ void chain_hi(f32x16 &acc, const v2f_t (&buf)[CH]) const {
    if constexpr (SPLIT) { // bf16 operands (NP = 1): one MFMA per group of 8 complex kc
#pragma unroll
      for (int gg = 0; gg < CH / 4; ++gg) {
        u32x4_t a;
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = pack_bf16(buf[4 * gg + u].x, buf[4 * gg + u].y);
#ifdef ARTN_ABLATE_MFMA
        asm volatile("" ::"v"(a), "v"(WS[BIGK ? HI : 0][BASE / 4 + gg]));
#else
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, WS[BIGK ? HI : 0][BASE / 4 + gg]),
                                                      __builtin_bit_cast(bf16x8_t, a), acc, 0, 0, 0);
#endif
      }
      return;
    }
    const float(&a0)[S] = w0<HI>();
    const float(&a1)[S] = w1<HI>();
#pragma unroll
    for (int s = 0; s < CH; ++s) {
#ifdef ARTN_ABLATE_MFMA
      asm volatile("" ::"v"(buf[s].x), "v"(buf[s].y), "v"(a0[BASE + s]), "v"(a1[BASE + s]));
#else
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[BASE + s], buf[s].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[BASE + s], buf[s].y, acc, 0, 0, 0);
#endif
    }
  }
  __device__ __forceinline__ unsigned kin_of(int hi) const {
    unsigned kin = 0;
#pragma unroll
    for (int b = 0; b < 2; ++b)
      if ((hi >> b) & 1) kin ^= L.kin_hi[b];
    return kin;
  }
  // one sub-tile: units (hi, half) in sequence, operands of the next unit loaded before the
  // MFMAs of the current one
  template <int HI>
  __device__ __forceinline__ void sub_hi(f32x16 &acc, v2f_t (&bA)[CH], v2f_t (&bB)[CH], unsigned li, int n_hi) const {
    if (HI >= n_hi) return;
    const unsigned base = li ^ kin_of(HI);
    if (UPS == 2) {
      load_unit<(UPS == 2 ? CH : 0)>(bB, base);
      chain_hi<HI, 0>(acc, bA);
      if (HI + 1 < n_hi) load_unit<0>(bA, li ^ kin_of(HI + 1));
      chain_hi<HI, (UPS == 2 ? CH : 0)>(acc, bB);
    } else {
      // (KB = 6 always has two units; kept for completeness)
      chain_hi<HI, 0>(acc, bA);
      if (HI + 1 < n_hi) load_unit<0>(bA, li ^ kin_of(HI + 1));
    }
  }
  // One 32 x 16 result block per tile: wave w runs the chain segment of looped-over value w
  // (its fragments sit in W0/W1), waves 1..3 park their partial block in LDS, wave 0 adds
  // them in order and scatters.  Every wave reaches the barrier.
  __device__ __forceinline__ void run_ksplit(int lane) const {
    const int n_hi = 1 << L.k_hi, w = L.ksplit_wave;
    const u2_t mo0 = lds_read_u2(L.msub_tab);
    f32x16 acc0;
    zero(acc0);
    if (w < n_hi) {
      v2f_t bA[CH], bB[CH];
      const unsigned base = L.lane_in ^ mo0.x ^ kin_of(w);
      load_unit<0>(bA, base);
      if (UPS == 2) {
        load_unit<(UPS == 2 ? CH : 0)>(bB, base);
        chain_hi<0, 0>(acc0, bA);
        chain_hi<0, (UPS == 2 ? CH : 0)>(acc0, bB);
      } else {
        chain_hi<0, 0>(acc0, bA);
      }
      if (w > 0) {
        const unsigned dst = L.ksplit_scratch + (unsigned)(w - 1) * 4096u + (unsigned)lane * 16u;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          lds_write16(dst + q * 1024u, f32x4{acc0[4 * q], acc0[4 * q + 1], acc0[4 * q + 2], acc0[4 * q + 3]});
      }
    }
    __syncthreads();
    if (w == 0) {
      for (int o = 1; o < n_hi; ++o) {
        const unsigned src = L.ksplit_scratch + (unsigned)(o - 1) * 4096u + (unsigned)lane * 16u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 t = lds_read16(src + q * 1024u);
          acc0[4 * q] += t[0]; acc0[4 * q + 1] += t[1]; acc0[4 * q + 2] += t[2]; acc0[4 * q + 3] += t[3];
        }
      }
      scatter(acc0, L.lane_out ^ mo0.y);
    }
  }
  __device__ __forceinline__ void run_big_k() const {
    const int m0 = L.wm, m1 = L.wm + L.wm_count;
    const bool has0 = m0 < L.msubs, has1 = m1 < L.msubs;
    if (!has0) return;
    const u2_t mo0 = lds_read_u2(L.msub_tab + m0 * 8);
    u2_t mo1 = mo0;
    if (has1) mo1 = lds_read_u2(L.msub_tab + m1 * 8);
    const int n_hi = 1 << L.k_hi;
    v2f_t bA[CH], bB[CH];
    f32x16 acc0, acc1;
    zero(acc0);
    {
      const unsigned li = L.lane_in ^ mo0.x;
      load_unit<0>(bA, li);
      sub_hi<0>(acc0, bA, bB, li, n_hi);
      sub_hi<1>(acc0, bA, bB, li, n_hi);
      sub_hi<2>(acc0, bA, bB, li, n_hi);
      sub_hi<3>(acc0, bA, bB, li, n_hi);
    }
    if (has1) {
      zero(acc1);
      const unsigned li = L.lane_in ^ mo1.x;
      load_unit<0>(bA, li);
      sub_hi<0>(acc1, bA, bB, li, n_hi);
      scatter(acc0, L.lane_out ^ mo0.y); // drains under the second sub-tile's chains
      sub_hi<1>(acc1, bA, bB, li, n_hi);
      sub_hi<2>(acc1, bA, bB, li, n_hi);
      sub_hi<3>(acc1, bA, bB, li, n_hi);
      scatter(acc1, L.lane_out ^ mo1.y);
    } else {
      scatter(acc0, L.lane_out ^ mo0.y);
    }
  }

  // 3M stage (ArtnStage::m3): the complex product from THREE real products -- T1 = A_re B_re, T2 = A_im B_im,
  // T3 = (A_re + A_im)(B_re + B_im); C_re = T1 - T2, C_im = T3 - T1 - T2 -- on MFMA blocks of 32 rows (columns m
  // of the tile) x 32 complex columns n: three accumulators, 3 MFMAs per pair of contracted values where the
  // 4M chains of two waves need 4 for the same 32 x 32 outputs.  One wave owns a whole 32-column sub-tile.
  __device__ __forceinline__ void scatter3(const f32x16 &t1, const f32x16 &t2, const f32x16 &t3, unsigned lo) const {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const unsigned o = lo ^ ((r & 1) ? L.o0 : 0u) ^ ((r & 2) ? L.o1 : 0u) ^ ((r & 4) ? L.o3 : 0u) ^ ((r & 8) ? L.o4 : 0u);
      lds_write8(o, v2f_t{t1[r] - t2[r], t3[r] - t1[r] - t2[r]});
    }
  }
  // operands in units of U3 chain steps (2 VGPRs each), ping-pong: the next unit -- of this sub-tile or the first of
  // the next one -- is read under the MFMAs of the current one.  (6 contracted bits: units of 2, which is what lets
  // the fused 6+4 instantiation fit the register file without spilling a prefetched chunk -- the spill made the
  // load-issue phase wait for HBM; 6 MFMAs = 384 cycles still cover an LDS read)
#ifndef ARTN_U3_6
#define ARTN_U3_6 2
#endif
  static constexpr int U3 = KB == 6 ? ARTN_U3_6 : 8;
  static constexpr int NU3 = S / U3 > 0 ? S / U3 : 1;
  template <int BASE>
  __device__ __forceinline__ void load_u3(v2f_t (&buf)[U3], unsigned li) const {
#pragma unroll
    for (int s = 0; s < U3; ++s) buf[s] = lds_read8(li ^ ko(BASE + s));
  }
  template <int BASE>
  __device__ __forceinline__ void chain3(f32x16 &t1, f32x16 &t2, f32x16 &t3, const v2f_t (&buf)[U3]) const {
#pragma unroll
    for (int s = 0; s < U3; ++s) {
      const float xs = buf[s].x + buf[s].y;
      // (re + im of the small operand is one v_add per MFMA triple: cheaper than 16-32 more fragment registers.  For the
      //  6-bit stage hipcc hoisted all 32 sums out of the TILE loop -- they are loop-invariant -- into 32 registers, and under
      //  that pressure its scheduler sank every operand read next to its first use: ds_read, s_waitcnt lgkmcnt(0), MFMA,
      //  one exposed LDS round trip per triple (found in the ISA in round 6).  A volatile asm is not hoisted.)
      float ws;
      if constexpr (KB == 6) asm volatile("v_add_f32 %0, %1, %2\n\ts_nop 1" : "=v"(ws) : "v"(W0[BASE + s]), "v"(W1[BASE + s]));
      else ws = W0[BASE + s] + W1[BASE + s];
#ifdef ARTN_ABLATE_MFMA
      asm volatile("" ::"v"(buf[s].x), "v"(buf[s].y), "v"(xs), "v"(W0[BASE + s]), "v"(W1[BASE + s]), "v"(ws));
#else
      t1 = __builtin_amdgcn_mfma_f32_32x32x2f32(W0[BASE + s], buf[s].x, t1, 0, 0, 0);
      t2 = __builtin_amdgcn_mfma_f32_32x32x2f32(W1[BASE + s], buf[s].y, t2, 0, 0, 0);
      t3 = __builtin_amdgcn_mfma_f32_32x32x2f32(ws, xs, t3, 0, 0, 0);
#endif
    }
  }
  template <int UI>
  __device__ __forceinline__ void units3(f32x16 &t1, f32x16 &t2, f32x16 &t3, v2f_t (&b)[2][U3], unsigned li, bool more,
                                         unsigned li_next) const {
    if constexpr (UI < NU3) {
      if constexpr (UI + 1 < NU3) load_u3<(UI + 1 < NU3 ? (UI + 1) * U3 : 0)>(b[(UI + 1) & 1], li);
      else if (more) load_u3<0>(b[0], li_next); // (NU3 is even: the next sub-tile starts in b[0] again)
      chain3<UI * U3>(t1, t2, t3, b[UI & 1]);
      units3<UI + 1>(t1, t2, t3, b, li, more, li_next);
    }
  }
  __device__ __forceinline__ void run3() const {
    static_assert(NU3 % 2 == 0 || !CAN3M, "ping-pong parity");
    int msub = L.wm;
    if (msub >= L.msubs) return;
    v2f_t b[2][U3];
    SMARK(0);
    u2_t mo = u2_t{L.mo0x, L.mo0y};
    load_u3<0>(b[0], L.lane_in ^ mo.x);
#if defined(ARTN_PHASES)
    if (L.mark_base >= 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#endif
    SMARK(1);
    bool first = true;
    for (;;) {
      const unsigned li = L.lane_in ^ mo.x, lo = L.lane_out ^ mo.y;
      const int nmsub = msub + L.wm_count;
      const bool more = nmsub < L.msubs;
      u2_t mo_n = mo;
      if (more) mo_n = first ? u2_t{L.mo1x, L.mo1y} : lds_read_u2(L.msub_tab + nmsub * 8);
      first = false;
      f32x16 t1, t2, t3;
      zero(t1); zero(t2); zero(t3);
      units3<0>(t1, t2, t3, b, li, more, L.lane_in ^ mo_n.x);
      SMARK(2);
      scatter3(t1, t2, t3, lo);
      SMARK(4);
      if (!more) return;
      msub = nmsub;
      mo = mo_n;
    }
  }

  // 2- to 4-bit stage of a 3M instantiation: per 32-column sub-tile two halves of 16 columns x 2^(KB - 2) chain steps x 3
  // products on v_mfma_f32_16x16x4_f32 (4 bits: 24 MFMAs of 32 cycles = 768 cycles; the 4M chain: 16 of 64 = 1 024); the
  // operands of the next sub-tile are read under the MFMAs of this one
  __device__ __forceinline__ void run4() const {
    int msub = L.wm;
    if (msub >= L.msubs) return;
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    v2f_t x[2][2][NST4]; // [buffer][half][step]
    auto kx = [&](int s) {
      unsigned k = 0;
#pragma unroll
      for (int b = 2; b < KB; ++b)
        if ((s >> (b - 2)) & 1) k ^= L.kin[b];
      return k;
    };
    auto load = [&](v2f_t (&b)[2][NST4], unsigned li) {
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int s = 0; s < NST4; ++s) b[hf][s] = lds_read8(li ^ (hf ? L.hin : 0u) ^ kx(s));
    };
    const int g = lane >> 4;
    u2_t mo = u2_t{L.mo0x, L.mo0y};
    load(x[0], L.lane_in ^ mo.x);
    bool first = true;
    auto body = [&](v2f_t (&cur)[2][NST4], v2f_t (&nxt)[2][NST4]) -> bool {
      const unsigned lo = L.lane_out ^ mo.y;
      const int nmsub = msub + L.wm_count;
      const bool more = nmsub < L.msubs;
      u2_t mo_n = mo;
      if (more) mo_n = first ? u2_t{L.mo1x, L.mo1y} : lds_read_u2(L.msub_tab + nmsub * 8);
      first = false;
      if (more) load(nxt, L.lane_in ^ mo_n.x);
      f32x4_t t[2][3];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int q = 0; q < 3; ++q) t[hf][q] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < NST4; ++s) {
        const float ws = W0[s] + W1[s];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const float xs = cur[hf][s].x + cur[hf][s].y;
#ifdef ARTN_ABLATE_MFMA
          asm volatile("" ::"v"(cur[hf][s].x), "v"(cur[hf][s].y), "v"(xs), "v"(W0[s]), "v"(W1[s]), "v"(ws));
#else
          t[hf][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(W0[s], cur[hf][s].x, t[hf][0], 0, 0, 0);
          t[hf][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(W1[s], cur[hf][s].y, t[hf][1], 0, 0, 0);
          t[hf][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(ws, xs, t[hf][2], 0, 0, 0);
#endif
        }
      }
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const unsigned o = lo ^ (hf ? L.hout : 0u) ^ ((r & 1) ? L.o0 : 0u) ^ ((r & 2) ? L.o1 : 0u);
          if (L.nt_eff == 4 || ((4 * g + r) >> L.nt_eff) == 0)
            lds_write8(o, v2f_t{t[hf][0][r] - t[hf][1][r], t[hf][2][r] - t[hf][0][r] - t[hf][1][r]});
        }
      if (!more) return false;
      msub = nmsub;
      mo = mo_n;
      return true;
    };
    for (;;) {
      if (!body(x[0], x[1])) return;
      if (!body(x[1], x[0])) return;
    }
  }

  __device__ __forceinline__ void run() const {
    if constexpr (CAN3M) {
      run3();
      return;
    }
    if constexpr (K4M3) {
      run4();
      return;
    }
    if constexpr (BIGK) { // 7 or 8 contracted bits: only instantiated for the single-stage KB = 6 kernel
      if (L.ksplit_wave >= 0) { run_ksplit(lane); return; }
      if (L.k_hi > 0) { run_big_k(); return; }
    }
    int msub = L.wm;
    if (msub >= L.msubs) return;
    v2f_t bA[CH], bB[CH];
    f32x16 acc0, acc1;
    SMARK(0);
    u2_t mo = u2_t{L.mo0x, L.mo0y};
    load_unit<0>(bA, L.lane_in ^ mo.x);
#if defined(ARTN_PHASES)
    if (L.mark_base >= 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#endif
    SMARK(1);
    bool pending = false, first = true;
    unsigned plo = 0;
    // two sub-tiles per trip so buffers and accumulators have static names
    for (;;) {
      // ---- even sub-tile: operands in bA (first unit), result in acc0
      {
        const unsigned li = L.lane_in ^ mo.x, lo = L.lane_out ^ mo.y;
        const int nmsub = msub + L.wm_count;
        const bool more = nmsub < L.msubs;
        u2_t mo_n = mo;
        if (more) mo_n = first ? u2_t{L.mo1x, L.mo1y} : lds_read_u2(L.msub_tab + nmsub * 8);
        first = false;
        zero(acc0);
        if (UPS == 2) {
          load_unit<(UPS == 2 ? CH : 0)>(bB, li);
          chain_unit<0>(acc0, bA, pending, acc1, plo);
          if (more) load_unit<0>(bA, L.lane_in ^ mo_n.x);
          chain_unit<(UPS == 2 ? CH : 0)>(acc0, bB, false, acc1, plo);
        } else {
          if (more) load_unit<0>(bB, L.lane_in ^ mo_n.x);
          chain_unit<0>(acc0, bA, pending, acc1, plo);
        }
        pending = true;
        plo = lo;
        SMARK(2);
        if (!more) { scatter(acc0, plo); SMARK(4); return; }
        msub = nmsub;
        mo = mo_n;
      }
      // ---- odd sub-tile: operands in bB (UPS == 1) or bA (UPS == 2), result in acc1
      {
        const unsigned li = L.lane_in ^ mo.x, lo = L.lane_out ^ mo.y;
        const int nmsub = msub + L.wm_count;
        const bool more = nmsub < L.msubs;
        u2_t mo_n = mo;
        if (more) mo_n = lds_read_u2(L.msub_tab + nmsub * 8);
        zero(acc1);
        if (UPS == 2) {
          load_unit<(UPS == 2 ? CH : 0)>(bB, li);
          chain_unit<0>(acc1, bA, pending, acc0, plo);
          if (more) load_unit<0>(bA, L.lane_in ^ mo_n.x);
          chain_unit<(UPS == 2 ? CH : 0)>(acc1, bB, false, acc0, plo);
        } else {
          if (more) load_unit<0>(bA, L.lane_in ^ mo_n.x);
          chain_unit<0>(acc1, bB, pending, acc0, plo);
        }
        plo = lo;
        SMARK(3);
        if (!more) { scatter(acc1, plo); SMARK(4); return; }
        msub = nmsub;
        mo = mo_n;
      }
    }
  }
};

template <int KB, bool BIGK, int NP, bool M3>
__device__ __forceinline__ void run_stage(const StageConst<KB> &L, float (&W0)[1 << (KB - 1)],
                                          float (&W1)[1 << (KB - 1)], float (&W2)[1],
                                          int h, int lane,
                                          float (&WH0)[BIGK ? 3 : 1][1 << (KB - 1)],
                                          float (&WH1)[BIGK ? 3 : 1][1 << (KB - 1)],
                                          u32x4_t (&WS)[(NP > 0 && KB >= 3) ? (BIGK ? 4 : NP) : 1][KB >= 3 ? 1 << (KB - 3) : 1]) {
  StageRun<KB, BIGK, NP, M3> r{L, W0, W1, W2, h, lane, WH0, WH1, WS};
  r.run();
}

#include "artn_wide_kernel.h"

// s_setprio takes an immediate: a run-time level goes through a scalar switch (the priority experiments of round 6:
// ArtnBitsPlan::stage_prio >= 0x100 packs one level per phase -- bits 0-1 stage 1, 2-3 stage 2, 4-5 the copy phases,
// bit 6: only the workgroup in the odd wave slots raises its stages)
__device__ __forceinline__ void set_prio_rt(int v) {
  switch (v & 3) {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
  }
}

// KB2 == 0: single stage.  BIGK: 7 or 8 contracted bits (KB1 = 6 of them in the chain).
// NP: 0 = fp32 MFMA chains; 3 / 1 = split-bf16 chains (stages with >= 3 contracted bits).
// GATHER: row indices on one outer axis (artn_contract_gather; single stage, fp32 chains).
// NT: non-temporal loads of the A tiles (see issue_loads).
// M3: every stage with 5 contracted bits runs the 3M arithmetic (StageRun::run3; fp32 chains only).
// FULL: input and output tiles are 2^12 elements (every big launch): the copy loops and their predicates are
// compile-time constants -- a third of the scalar instructions and most of the branches of the tile loop go.
// N3: single step with 5-6 contracted bits and at most 4 result bits in the tile (ArtnBitsPlan::narrow3): the stage of
// artn_k_wide on this kernel's four waves -- 16 x 16 x 4 blocks, three products -- instead of four-product 32 x 32 chains
// of which at most 16 rows are results.
// (N3 = 2: the second stage of a fused pair instead -- a pair that shrinks its tensor; an M3 instantiation)
template <int KB1, int KB2, bool BIGK, int NP = 0, bool GATHER = false, bool NT = false, bool M3 = false, bool FULL = false, int N3 = 0>
__global__ __launch_bounds__(ARTN_WG_THREADS, (BIGK ? 1 : 2)) void artn_k_bits(const float2 *__restrict__ A,
                                                                  const float2 *__restrict__ B1,
                                                                  const float2 *__restrict__ B2,
                                                                  float2 *__restrict__ C, const ArtnBitsPlan P) {
  constexpr int S1 = 1 << (KB1 - 1);
  constexpr int KB2e = KB2 > 0 ? KB2 : 1;
  constexpr int S2 = 1 << (KB2e - 1);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if ((unsigned)(unsigned long)(lds_byte_t *)smem != 0) __builtin_trap(); // see lds_read8
  // the larger region first: each region base is then a multiple of that region's size, so
  // XOR-ing the base into an in-region offset equals adding it
  const unsigned R0 = P.T_mid > P.r0_bits ? 8u << P.T_mid : 0u, R1 = P.T_mid > P.r0_bits ? 0u : 8u << P.r0_bits;
  const unsigned regions_end = (8u << P.r0_bits) + (8u << P.T_mid);
  uint2 *tab1 = reinterpret_cast<uint2 *>(smem + regions_end);
  uint2 *tab2 = tab1 + (1 << (P.st[0].m_bits - 5));
  long *offtab = reinterpret_cast<long *>(tab2 + (KB2 > 0 ? 1 << (P.st[1].m_bits - 5) : 0));

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5, ro = j & 1;

  // ---- copy phases: thread handles 16-byte chunks c = tid + 256*i (tile-local elements 2c, 2c+1);
  //      per-lane byte offsets fit 32 bits (checked by the planner)
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
  const int n_in_iters = FULL ? 8 : 1 << (P.T_in - 9);
  // output tiles smaller than one copy pass (2^9 elements): one pass, upper threads idle
  const int n_out_iters = FULL ? 8 : (P.T_out >= 9 ? 1 << (P.T_out - 9) : 1);
  const bool out_active = FULL || P.T_out >= 9 || tid < (1 << (P.T_out - 1));

  // ---- per-stage constants, sub-tile tables, outer-axis digits
  fill_msub_table(P.st[0], nullptr, tab1, tid);
  if (KB2 > 0) fill_msub_table(P.st[1], &P.st[0], tab2, tid);
  const unsigned tab1_a = regions_end, tab2_a = tab1_a + (8u << (P.st[0].m_bits - 5));
  constexpr bool SP1 = NP > 0 && KB1 >= 3, SP2 = NP > 0 && KB2 >= 3;
  constexpr int G1 = KB1 >= 3 ? 1 << (KB1 - 3) : 1, G2 = KB2e >= 3 ? 1 << (KB2e - 3) : 1;
  StageConst<KB1> L1 = stage_const<KB1, M3>(P.st[0], nullptr, j, h, wave, tab1_a, R0, R1, SP1 ? 2 : 0);
  if (BIGK && P.ksplit) {
    L1.ksplit_wave = wave;
    L1.ksplit_scratch = (regions_end + (8u << (P.st[0].m_bits - 5)) + (512u * 8u + 32u * 32u) + 15u) & ~15u;
  }
  const StageConst<KB2e> L2 = stage_const<KB2e, M3>(P.st[KB2 > 0 ? 1 : 0], &P.st[0], j, h, wave, tab2_a, R1, R0, SP2 ? 2 : 0);
  static_assert(N3 != 1 || (KB2 == 0 && (KB1 == 5 || KB1 == 6) && !BIGK && NP == 0 && !GATHER && !M3 && !FULL), "narrow 3M: single 5-6 bit steps");
  static_assert(N3 != 2 || ((KB2 == 5 || KB2 == 6) && M3 && !BIGK && NP == 0 && !GATHER && !FULL), "narrow 3M: second stage of a 3M pair");
  constexpr int KBN = N3 == 1 ? KB1 : (N3 == 2 ? KB2e : 2), NSTN = 1 << (KBN - 2);
  WideConst<KBN> LN;
  if constexpr (N3 == 1) LN = wide_const<KBN, 4>(P.st[0], nullptr, lane, wave, tab1_a);
  if constexpr (N3 == 2) LN = wide_const<KBN, 4>(P.st[1], &P.st[0], lane, wave, tab2_a);
  float WN0[NSTN], WN1[NSTN], WN2[NSTN];
  // copy-out reads the last stage's (swizzled) output region
  const ArtnStage *zout = &P.st[KB2 > 0 ? 1 : 0];
  const unsigned tid16_out = swz(tid16, zout);
  unsigned out_i_swz[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) out_i_swz[i] = swz(i * (ARTN_WG_THREADS * 16), zout);
  const OffTab OT = build_offset_table(P, offtab, tid);
  float W10[S1], W11[S1], W20[S2], W21[S2]; // (whichever of the fp32 / split fragment sets a stage does not use is dead)
  constexpr bool C31 = M3 && (KB1 == 5 || KB1 == 6) && !BIGK && NP == 0, C32 = M3 && (KB2 == 5 || KB2 == 6) && NP == 0;
  float W12[1], W22[1]; // (unused)
  u32x4_t WS1[SP1 ? (BIGK ? 4 : NP) : 1][G1], WS2[SP2 ? NP : 1][G2];
  float WH0[BIGK ? 3 : 1][S1], WH1[BIGK ? 3 : 1][S1], WD0[1][S2], WD1[1][S2]; // BIGK: fragments of looped-over values 1..3
  long prev_b1 = -1, prev_b2 = -1;
  __syncthreads(); // tables are in LDS

  // Software pipeline over tiles.  Order of one iteration (tile t is already in R0):
  //   stages(t) -> result region to registers x[] -> refill R0 with tile t+1 (loads issued one
  //   iteration ago) -> stores(t) from x[] -> issue loads(t+2).
  // The refill waits on loads that are followed in the (in-order) VMEM queue by nothing, so
  // its s_waitcnt never waits for stores issued after them; the stores have a whole stage
  // phase to drain before the next refill looks at the counter.
  // (the 7-8 bit instantiation runs one wave per SIMD and prefetches 2^13-element tiles)
  constexpr int NV = BIGK ? 16 : 8;
  f32x4 v[NV];
  // (2^11-element input tiles -- steps that double their tensor -- prefetch too: half the chunks)
  const bool pf_half = !FULL && !BIGK && n_in_iters == NV / 2 && n_out_iters <= 8;
  const bool prefetch = FULL || ((n_in_iters == NV || pf_half) && n_out_iters <= 8);
  TileOff off = {0, 0, 0, 0}, noff = {0, 0, 0, 0};
  // tiles of this workgroup: t0, t0 + G, ... < n_tiles (grid-stride), or one contiguous range
  long t0 = blockIdx.x, G = gridDim.x, n_tiles = P.n_tiles;
  // Workgroups are dealt round-robin to the 8 XCDs (each with its own L2): give each XCD a
  // contiguous eighth of every grid-stride period instead of every eighth tile, so that tiles
  // which share an A tile (outer bits of the small operand are the fastest tile digits) and
  // neighbouring lines meet in one L2 (-0.8 % on the n30 contraction, A/B in one session).
  if ((G & 7) == 0) t0 = (long)(blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  if (P.blocked) {
    const long per = (P.n_tiles + gridDim.x - 1) / gridDim.x;
    t0 = per * blockIdx.x;
    G = 1;
    n_tiles = t0 + per < P.n_tiles ? t0 + per : P.n_tiles;
  }
  GenCache gcache = {-1, 0, 0, 0, 0};
  // (the row-gather instantiations only: ten more live scalar registers in the tile loop of the others are not free;
  //  grid-stride sequences change the batch row at every tile)
  GenCache *const gc = (GATHER && P.blocked) ? &gcache : nullptr;
  if (t0 < n_tiles) {
    off = tile_offsets<GATHER>(P, OT, t0, gc);
    copy_in_sync(reinterpret_cast<const char *>(A + off.a), in_hi, in_lane, R0, tid16, n_in_iters);
    if (t0 + G < n_tiles) {
      noff = tile_offsets<GATHER>(P, OT, t0 + G, gc);
      if (pf_half) issue_loads<NV, NT, 0, NV / 2>(v, reinterpret_cast<const char *>(A + noff.a), in_hi, in_lane);
      else if (prefetch) issue_loads<NV, NT>(v, reinterpret_cast<const char *>(A + noff.a), in_hi, in_lane);
    }
  }
  __syncthreads();

  // Two workgroups share a CU and run the same phase sequence.  Left alone they settle into
  // near-lockstep (tools/phases.py: stage phases 67 % of the period, start offset 0.2 of it
  // whatever the start-up stagger -- lockstep is an attractor): both fight for the matrix
  // pipe in their stage phases and both leave it idle in their copy phases.  Raising the
  // issue priority of ONE of the two (the one whose waves sit in the odd wave slots) during
  // its MFMA stages only breaks the symmetry: its chains never wait, the other workgroup's
  // chains fill the pipe while it copies, and the pair locks into alternation (offset 0.48).
  // (mode 4, development builds: EVERY workgroup raises its stages and leaves its copy phases at 0 -- a workgroup in a stage
  //  starves its partner's copy phase, which is what holds tools/probes/tri_probe.hip in anti-phase)
  const bool stage_prio = ((P.stage_prio == 1 || P.stage_prio == 3) && (__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 1)) || P.stage_prio == 4; // HW_ID.wave_id bit 0
  // experiment (ARTN_STAGE_PRIO=2/3): the COPY phases of every workgroup run at raised priority instead, so their few
  // instructions never queue behind the co-resident workgroup's MFMA stream (3: on top of the asymmetric stage priority)
  const bool copy_prio = P.stage_prio == 2 || P.stage_prio == 3;
  const bool prio_rt = P.stage_prio >= 0x100;
  const bool prio_mine = !((P.stage_prio >> 6) & 1) || (__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 1);
  const int prio_s1 = prio_mine ? P.stage_prio & 3 : 0, prio_s2 = prio_mine ? (P.stage_prio >> 2) & 3 : 0, prio_cp = (P.stage_prio >> 4) & 3;
  STAMP_DECL
  bool half_pending = false;
  for (long tile = t0; tile < n_tiles; tile += G) {
    if (off.b1 != prev_b1) {
      prev_b1 = off.b1;
      const char *Bb = reinterpret_cast<const char *>(B1 + off.b1);
      long kb0 = 0; // ksplit: this wave's share of the looped-over bits, in W10/W11
      if (BIGK && L1.ksplit_wave > 0) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
          if ((L1.ksplit_wave >> b) & 1) kb0 += L1.kb_hi[b];
      }
      if constexpr (N3 == 1) wide_load_w<KBN>(WN0, WN1, WN2, Bb, LN);
      else if constexpr (SP1 && BIGK) load_w_plane<KB1>(WS1[0], Bb + kb0, L1, ro);
      else if constexpr (SP1) load_w_split<KB1, (SP1 ? NP : 1)>(WS1, Bb, L1, ro);
      else if constexpr (C31) load_w3<KB1>(W10, W11, W12, Bb, L1);
      else if constexpr (M3 && KB1 >= 2 && KB1 <= 4 && !BIGK && NP == 0) load_w4m3<KB1>(W10, W11, Bb, L1);
      else load_w<KB1>(W10, W11, Bb + kb0, L1, ro);
      if constexpr (BIGK) { // fragments of the looped-over contracted bits' values 1..3
#pragma unroll
        for (int hi = 1; hi < 4; ++hi) {
          if (L1.ksplit_wave >= 0) break;
          if (hi < (1 << L1.k_hi)) {
            long kbo = 0;
#pragma unroll
            for (int b = 0; b < 2; ++b)
              if ((hi >> b) & 1) kbo += L1.kb_hi[b];
            if constexpr (SP1) load_w_plane<KB1>(WS1[SP1 ? hi : 0], Bb + kbo, L1, ro);
            else load_w<KB1>(WH0[hi - 1], WH1[hi - 1], Bb + kbo, L1, ro);
          }
        }
      }
    }
    if (KB2 > 0 && off.b2 != prev_b2) {
      prev_b2 = off.b2;
      if constexpr (N3 == 2) wide_load_w<KBN>(WN0, WN1, WN2, reinterpret_cast<const char *>(B2 + off.b2), LN);
      else if constexpr (SP2) load_w_split<KB2e, (SP2 ? NP : 1)>(WS2, reinterpret_cast<const char *>(B2 + off.b2), L2, ro);
      else if constexpr (C32) load_w3<KB2e>(W20, W21, W22, reinterpret_cast<const char *>(B2 + off.b2), L2);
      else if constexpr (M3 && KB2 >= 2 && KB2 <= 4 && NP == 0) load_w4m3<KB2e>(W20, W21, reinterpret_cast<const char *>(B2 + off.b2), L2);
      else load_w<KB2e>(W20, W21, reinterpret_cast<const char *>(B2 + off.b2), L2, ro);
    }
    const long next = tile + G, next2 = tile + 2 * G;
    TileOff n2off = noff;
    if (next2 < n_tiles) n2off = next_offsets<GATHER>(P, OT, noff, next, G, gc);
    STAMP(0); // W reload, offsets of the tile after next
    PHASE_MARK(0);

    // ---- stage 1: R0 -> R1, fused stage 2: R1 -> R0
    if (stage_prio) __builtin_amdgcn_s_setprio(2);
    if (prio_rt) set_prio_rt(prio_s1);
    if constexpr (N3 == 1) {
      WideStage<KBN> sn{LN, WN0, WN1, WN2, R0, R1, -1, 0, 0};
      WideNoFill nf;
      sn.run(nf);
    } else {
      run_stage<KB1, BIGK, NP, M3>(L1, W10, W11, W12, h, lane, WH0, WH1, WS1);
    }
    if (stage_prio && KB2 == 0) __builtin_amdgcn_s_setprio(0);
    PHASE_MARK(1);
    STAMP(5);
    if (FULL && half_pending) { // the second half of the next tile's loads (see below)
      unsigned li2 = in_lane;
      OPAQUE_V(li2);
      issue_loads<NV, NT, NV / 2, NV>(v, reinterpret_cast<const char *>(A + noff.a), in_hi, li2);
    }
    __syncthreads();
    unsigned outr = R1;
    if (KB2 > 0) {
      if (prio_rt) set_prio_rt(prio_s2);
      STAMP(6);
      if constexpr (N3 == 2) {
        WideStage<KBN> sn{LN, WN0, WN1, WN2, R1, R0, -1, 0, 0};
        WideNoFill nf;
        sn.run(nf);
      } else {
        run_stage<KB2e, false, NP, M3>(L2, W20, W21, W22, h, lane, WD0, WD1, WS2);
      }
      if (stage_prio) __builtin_amdgcn_s_setprio(0);
      STAMP(5);
      __syncthreads();
      outr = R0;
    }
    STAMP(6); // barriers after the stages
    PHASE_MARK(2);
    if (copy_prio) __builtin_amdgcn_s_setprio(3);
    if (prio_rt) set_prio_rt(prio_cp);

    unsigned lo_in = in_lane, lo_out = out_lane, t16 = tid16, t16o = tid16_out;
    OPAQUE_V(lo_in);
    OPAQUE_V(lo_out);
    OPAQUE_V(t16);
    OPAQUE_V(t16o);
    char *Cbase = reinterpret_cast<char *>(C + off.c);
    if (prefetch) {
      // result tile -> registers
      f32x4 x[8];
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (i < n_out_iters && out_active) x[i] = lds_read16(outr + (t16o ^ out_i_swz[i]));
      if (KB2 > 0) __syncthreads(); // fused: the result sat in R0, which is refilled next
      PHASE_MARK(3);
      STAMP(1);
      // refill R0 with the next tile (its loads were issued one iteration ago)
      if (next < n_tiles) {
        if (pf_half) store_lds<NV, NV / 2>(v, R0, t16);
        else store_lds(v, R0, t16);
      }
      PHASE_MARK(4);
      STAMP(2);
      // C += result (ArtnBitsPlan::accumulate: the slice loop's `collect += ...` in the store phase of a slice's last launch):
      // the accumulator's chunks are read into registers the refill has just freed and added before the stores.  (Each
      // element belongs to exactly one lane of one tile: plain loads and stores.  global_atomic_add_f32 instead -- four per
      // chunk, no registers -- ran the launch at a FIFTH of the speed: 42.4 ms per n30_sliced3 slice against 22.4.)
      // (FULL instantiations only -- bits_can_accumulate(): behind the run-time tile sizes of the others the conditional loads
      //  cost every launch 40 %, accumulating or not)
      if (!FULL && P.accumulate) __builtin_trap();
#ifndef ARTN_ABLATE_MEM
      if (FULL && P.accumulate) {
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
          if (i < n_out_iters && out_active) x[i] += c[i];
      }
#endif
      // stores of this tile, then the loads of the tile after next
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (i < n_out_iters && out_active) {
          long o = 0;
#pragma unroll
          for (int b = 0; b < 4; ++b)
            if ((i >> b) & 1) o += out_hi[b];
#ifdef ARTN_ABLATE_MEM
          asm volatile("" ::"v"(x[i]), "s"(Cbase), "v"(lo_out));
#else
          // (results are not read again by this launch)
          __builtin_nontemporal_store(x[i], reinterpret_cast<f32x4 *>(Cbase + o + lo_out));
#endif
        }
      }
      PHASE_MARK(5);
      STAMP(7);
      // (FULL: four of the eight chunks here, four after the next tile's first stage -- a workgroup's 32 KiB request
      //  burst in two halves half a tile period apart: +0.5 % on n30, A/B in one session; all eight late, or a second
      //  tile in flight, lose)
      if (FULL) {
        half_pending = next2 < n_tiles;
        if (next2 < n_tiles) issue_loads<NV, NT, 0, NV / 2>(v, reinterpret_cast<const char *>(A + n2off.a), in_hi, lo_in);
      } else if (next2 < n_tiles) {
        if (pf_half) issue_loads<NV, NT, 0, NV / 2>(v, reinterpret_cast<const char *>(A + n2off.a), in_hi, lo_in);
        else issue_loads<NV, NT>(v, reinterpret_cast<const char *>(A + n2off.a), in_hi, lo_in);
      }
      PHASE_MARK(6);
      STAMP(4);
    } else {
      // big tiles (2^13): no register prefetch; stream out, then load the next tile
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (i < n_out_iters && out_active) {
          long o = 0;
#pragma unroll
          for (int b = 0; b < 4; ++b)
            if ((i >> b) & 1) o += out_hi[b];
          const f32x4 xx = lds_read16(outr + (t16o ^ out_i_swz[i]));
#ifdef ARTN_ABLATE_MEM
          asm volatile("" ::"v"(xx), "s"(Cbase), "v"(lo_out));
#else
          *reinterpret_cast<f32x4 *>(Cbase + o + lo_out) = xx;
#endif
        }
      }
      if (KB2 > 0) __syncthreads();
      if (next < n_tiles) copy_in_sync(reinterpret_cast<const char *>(A + noff.a), in_hi, lo_in, R0, t16, n_in_iters);
    }
    if (copy_prio) __builtin_amdgcn_s_setprio(0);
    __syncthreads(); // R0 holds the next tile; every wave is done with the result region
    PHASE_MARK(7);
    STAMP(3);
    off = noff;
    noff = n2off;
  }
  STAMP_FLUSH
}

// ----------------------------------------------------------------------------------------
// artn_k_alt -- the same tiles, stages and plans as artn_k_bits, for the big launches (2^12-element tiles, fp32
// chains, no row gather), with the overlap of MFMA stages and copy phases ENFORCED instead of hoped for.
//
// artn_k_bits puts two independent 4-wave workgroups on a CU and relies on one being in its copy phases while the
// other runs its MFMA stages.  Measured (tools/diag_r03.sh, round 3) they do not: two workgroups per CU are only
// 1.14-1.24 x as fast as one, and their phase offset is 0.2 of a period.  Sharing the matrix pipe is processor
// sharing -- whichever workgroup is behind speeds up as soon as the other leaves its stage, so lockstep is an
// attractor -- and in lockstep both fight for the pipe in their stages and both leave it idle while they copy.
//
// Here ONE workgroup of 8 waves per CU holds two groups of 4 waves (one wave of each group per SIMD), each group
// with its own pair of LDS regions and its own grid-stride tile sequence, and the groups alternate roles every
// half period, separated by workgroup barriers:
//
//     half h, group g with (h ^ g) even:  stage 1 | barrier | stage 2            | barrier     (matrix pipe)
//              the other group:           result -> registers -> global stores | barrier | refill R0 with its
//                                         next tile (loads issued a period ago), issue the loads of the tile after | barrier
//
// The four barriers of a period are all the synchronisation either group needs (stage 1 -> stage 2, result reads
// before the refill, refill before stage 1), each SIMD's matrix pipe serves one chain at a time, and a group's copy
// phases have the whole length of the other group's stages to hide in.
template <int KB1, int KB2, bool NT, bool M3>
__global__ __launch_bounds__(2 * ARTN_WG_THREADS, 1) void artn_k_alt(const float2 *__restrict__ A, const float2 *__restrict__ B1,
                                                                     const float2 *__restrict__ B2, float2 *__restrict__ C,
                                                                     const ArtnBitsPlan P) {
  constexpr int S1 = 1 << (KB1 - 1);
  constexpr int KB2e = KB2 > 0 ? KB2 : 1;
  constexpr int S2 = 1 << (KB2e - 1);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if ((unsigned)(unsigned long)(lds_byte_t *)smem != 0) __builtin_trap(); // see lds_read8
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, w4 = wave & 3, tg = tid & (ARTN_WG_THREADS - 1);
  const int j = lane & 31, h = lane >> 5, ro = j & 1;
  // LDS: group g owns [g * 64 KiB, +64 KiB): R0 (input tile, result of a fused pair) and R1; tables behind both
  const unsigned R0 = (unsigned)grp << 16, R1 = R0 + (8u << 12);
  const unsigned regions_end = 2u << 16;
  uint2 *tab1 = reinterpret_cast<uint2 *>(smem + regions_end);
  uint2 *tab2 = tab1 + (1 << (P.st[0].m_bits - 5));
  long *offtab = reinterpret_cast<long *>(tab2 + (KB2 > 0 ? 1 << (P.st[1].m_bits - 5) : 0));

  unsigned in_lane = 0, out_lane = 0;
#pragma unroll
  for (int b = 1; b <= 8; ++b) {
    if ((tg >> (b - 1)) & 1) {
      in_lane += (unsigned)P.in_stride[b] * 8u;
      out_lane += (unsigned)P.out_stride[b] * 8u;
    }
  }
  long in_hi[4], out_hi[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    in_hi[b] = b < 3 ? P.in_stride[9 + b] * 8 : 0;
    out_hi[b] = b < 3 ? P.out_stride[9 + b] * 8 : 0;
  }
  const unsigned tid16 = tg * 16;

  fill_msub_table(P.st[0], nullptr, tab1, tid);
  if (KB2 > 0) fill_msub_table(P.st[1], &P.st[0], tab2, tid);
  const unsigned tab1_a = regions_end, tab2_a = tab1_a + (8u << (P.st[0].m_bits - 5));
  const StageConst<KB1> L1 = stage_const<KB1, M3>(P.st[0], nullptr, j, h, w4, tab1_a, R0, R1, 0);
  const StageConst<KB2e> L2 = stage_const<KB2e, M3>(P.st[KB2 > 0 ? 1 : 0], &P.st[0], j, h, w4, tab2_a, R1, R0, 0);
  const ArtnStage *zout = &P.st[KB2 > 0 ? 1 : 0];
  const unsigned tid16_out = swz(tid16, zout);
  unsigned out_i_swz[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) out_i_swz[i] = swz(i * (ARTN_WG_THREADS * 16), zout);
  // tiles: virtual workgroup vb = block + group * grid of a grid of VG = 2 * grid; XCD-aware start as in artn_k_bits
  const long VG = 2L * gridDim.x, n_tiles = P.n_tiles;
  const OffTab OT = build_offset_table(P, offtab, tid, (int)VG);
  float W10[S1], W11[S1], W20[S2], W21[S2];
  constexpr bool C31 = M3 && (KB1 == 5 || KB1 == 6), C32 = M3 && (KB2 == 5 || KB2 == 6);
  float W12[1], W22[1];
  u32x4_t WS1[1][KB1 >= 3 ? 1 << (KB1 - 3) : 1], WS2[1][KB2e >= 3 ? 1 << (KB2e - 3) : 1]; // (unused: fp32 chains only)
  float WH0[1][S1], WH1[1][S1], WD0[1][S2], WD1[1][S2];
  long prev_b1 = -1, prev_b2 = -1;
  __syncthreads();

  const long vb = (long)blockIdx.x + (long)grp * gridDim.x;
  const long t0 = (VG & 7) == 0 ? (vb & 7) * (VG >> 3) + (vb >> 3) : vb;
  const long t_max = (n_tiles + VG - 1) / VG; // tiles of the busiest virtual workgroup
  f32x4 v[8];
  // ct: the tile this group computes next (in R0 once refilled); rt: the tile whose loads are in flight in v[]
  // (loff: its offsets).  Loads and refills are UNCONDITIONAL -- past its last tile a group simply loads its last tile
  // again -- so that v[] has one definition per half: a conditional load leaves the compiler a merge of old and new
  // v[], which it resolves with register copies guarded by vmcnt waits in the middle of the copy phase.
  long ct = t0, rt = t0;
  const long t_last = n_tiles - 1;
  TileOff coff = tile_offsets(P, OT, t0 < n_tiles ? t0 : t_last), loff = coff;
  long o_c = 0;          // C offset of the tile whose result waits for its copy-out
  bool have_result = false;
  if (grp == 0) { // group 0 computes first: its first tile goes straight to LDS
    copy_in_sync(reinterpret_cast<const char *>(A + coff.a), in_hi, in_lane, R0, tid16, 8);
    rt = t0 + VG;
    if (rt < n_tiles) loff = tile_offsets(P, OT, rt);
  }               // group 1 starts in the copy role: its first tile arrives through v[] like every later one
  issue_loads<8, NT>(v, reinterpret_cast<const char *>(A + loff.a), in_hi, in_lane);
  __syncthreads();

  // Both groups run the SAME straight-line sequence compute half, copy half, ... (one definition of v[] per period:
  // nothing for the compiler to merge or copy); group 1 is half a period behind because it enters through one extra
  // copy half (which refills R0 with its first tile), and group 0 leaves through two extra barriers.
  auto compute_half = [&](long half) {
    const bool work = ct < n_tiles;
    ALT_MARK(0);
    if (work) {
      if (coff.b1 != prev_b1) {
        prev_b1 = coff.b1;
        const char *Bb = reinterpret_cast<const char *>(B1 + coff.b1);
        if constexpr (C31) load_w3<KB1>(W10, W11, W12, Bb, L1);
        else if constexpr (M3 && KB1 >= 2 && KB1 <= 4) load_w4m3<KB1>(W10, W11, Bb, L1);
        else load_w<KB1>(W10, W11, Bb, L1, ro);
      }
      if (KB2 > 0 && coff.b2 != prev_b2) {
        prev_b2 = coff.b2;
        if constexpr (C32) load_w3<KB2e>(W20, W21, W22, reinterpret_cast<const char *>(B2 + coff.b2), L2);
        else if constexpr (M3 && KB2 >= 2 && KB2 <= 4) load_w4m3<KB2e>(W20, W21, reinterpret_cast<const char *>(B2 + coff.b2), L2);
        else load_w<KB2e>(W20, W21, reinterpret_cast<const char *>(B2 + coff.b2), L2, ro);
      }
#if defined(ARTN_PHASES)
      StageConst<KB1> L1m = L1;
      L1m.mark_base = (w4 == 0 && blockIdx.x < 64 && half >= 40 && half < 44) ? 4096 + ((((int)blockIdx.x * 2 + grp) * 4 + (int)(half - 40)) * 2 + 0) * 8 : -1;
      run_stage<KB1, false, 0, M3>(L1m, W10, W11, W12, h, lane, WH0, WH1, WS1);
#else
      run_stage<KB1, false, 0, M3>(L1, W10, W11, W12, h, lane, WH0, WH1, WS1);
#endif
    }
    ALT_MARK(1);
    __syncthreads();
    ALT_MARK(2);
#if defined(ARTN_PHASES)
    if (KB2 > 0 && work) {
      StageConst<KB2e> L2m = L2;
      L2m.mark_base = (w4 == 0 && blockIdx.x < 64 && half >= 40 && half < 44) ? 4096 + ((((int)blockIdx.x * 2 + grp) * 4 + (int)(half - 40)) * 2 + 1) * 8 : -1;
      run_stage<KB2e, false, 0, M3>(L2m, W20, W21, W22, h, lane, WD0, WD1, WS2);
    }
#else
    if (KB2 > 0 && work) run_stage<KB2e, false, 0, M3>(L2, W20, W21, W22, h, lane, WD0, WD1, WS2);
#endif
    ALT_MARK(3);
    __syncthreads();
    ALT_MARK(4);
    if (work) {
      have_result = true;
      o_c = coff.c;
      ct += VG;
    }
  };
  auto copy_half = [&](long half) {
    // result of the tile computed in the previous half -> registers -> global; next tile -> R0
    unsigned lo_in = in_lane, lo_out = out_lane, t16 = tid16, t16o = tid16_out;
    OPAQUE_V(lo_in);
    OPAQUE_V(lo_out);
    OPAQUE_V(t16);
    OPAQUE_V(t16o);
    // (order as in artn_k_bits: the refill waits on loads that nothing younger follows in the in-order VMEM queue;
    //  stores first would put a vmcnt(0) -- a wait for the stores themselves -- in front of the refill)
    f32x4 x[8];
    const bool had = have_result;
    ALT_MARK(0);
    if (had) {
      const unsigned outr = KB2 > 0 ? R0 : R1;
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = lds_read16(outr + (t16o ^ out_i_swz[i]));
    }
    ALT_MARK(1);
    __syncthreads(); // every wave of the group has its part of the result in registers: R0 may be refilled
    ALT_MARK(2);
    store_lds(v, R0, t16);
    ALT_MARK(5);
    coff = loff; // (rt == ct: the tile now in R0 is the one this group computes next, if there is one)
    if (had) {
      char *Cbase = reinterpret_cast<char *>(C + o_c);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        long o = 0;
#pragma unroll
        for (int b = 0; b < 3; ++b)
          if ((i >> b) & 1) o += out_hi[b];
#ifdef ARTN_ABLATE_MEM
        asm volatile("" ::"v"(x[i]), "s"(Cbase), "v"(lo_out));
#else
        __builtin_nontemporal_store(x[i], reinterpret_cast<f32x4 *>(Cbase + o + lo_out));
#endif
      }
      have_result = false;
    }
    ALT_MARK(6);
    if (rt + VG < n_tiles) loff = next_offsets(P, OT, loff, rt, VG);
    issue_loads<8, NT>(v, reinterpret_cast<const char *>(A + loff.a), in_hi, lo_in);
    rt += VG;
    ALT_MARK(3);
    __syncthreads();
    ALT_MARK(4);
  };
  if (grp == 1) copy_half(-1);
  for (long it = 0; it < t_max; ++it) {
    compute_half(2 * it + grp);
    copy_half(2 * it + 1 + grp);
  }
  if (grp == 0) {
    __syncthreads();
    __syncthreads();
  }
}

// Translation units.  The product library is linked from eight objects compiled from THIS file (make -j: the
// ~100 artn_k_bits instantiations dominate the build): -DARTN_TU_BITS=K emits only artn_k_bits<K, *> behind
// artn_launch_bits_kK(), -DARTN_TU_B128 only artn_k_bits128<*, *> behind artn_launch_bits128(), -DARTN_TU_MAIN
// everything else and calls those; with none of the macros (diagnostic and development builds) the file is one
// translation unit as before.
#if defined(ARTN_TU_B128A) && !defined(ARTN_TU_B128)
#define ARTN_TU_B128 1 /* (-DARTN_TU_B128A: the accumulating instantiations artn_k_bits128<*, *, true> behind artn_launch_bits128_acc()) */
#endif
#if defined(ARTN_TU_BITS) || defined(ARTN_TU_B128) || defined(ARTN_TU_BITS3) || defined(ARTN_TU_WIDE)
#define ARTN_TU_PART 1
#endif
// (-DARTN_TU_WIDE: only artn_k_wide<*, *> behind artn_launch_wide(); artn_wide_kernel.h itself is included above artn_k_bits,
//  which borrows its stage)
// (-DARTN_TU_BITS3=K: only artn_k_bits3<K, *, *> behind artn_launch_bits3_kK())
#if defined(ARTN_TU_BITS3) || (defined(ARTN_DEV_BITS3) && !defined(ARTN_TU_PART) && !defined(ARTN_TU_MAIN))
#include "artn_bits3_kernel.h"
#endif
#if defined(ARTN_TU_B128) || (!defined(ARTN_TU_PART) && !defined(ARTN_TU_MAIN))
#include "artn_bits128_kernel.h"
#endif
#ifndef ARTN_TU_PART
#include "artn_gemm_kernel.h"
#include "artn_gemm128_kernel.h"
#include "artn_pgemm_kernel.h"
#include "artn_xgemm_kernel.h"
#include "artn_xgemm128_kernel.h"
#include "artn_xrow_kernel.h"
#ifdef ARTN_DEV_XGPC
#include "artn_xgemm_pc_kernel.h"
#endif

// ----------------------------------------------------------------------------------------
// strided fallback: one thread per C element
// ----------------------------------------------------------------------------------------
// (mixed-radix digits of a flat index -> element offsets in both operands.  The small closing steps of a circuit slice
//  spend their time HERE, not in memory: 64-bit `%` and `/` per label and term were 1.6 us per term -- a 256-result step of
//  256-term sums took 412 us, 2 % of an n53 m14 slice.  Indices below 2^31 are decoded in 32 bits, extents that are
//  powers of two -- every label of a circuit -- with a mask and a shift.)
template <typename I>
__device__ __forceinline__ void gen_decode(I r, int n, const int64_t *ext, const int64_t *sA, const int64_t *sB, long &oa, long &ob) {
  oa = ob = 0;
  for (int d = 0; d < n; ++d) {
    const I e = (I)ext[d];
    I x;
    if ((e & (e - 1)) == 0) {
      x = r & (e - 1);
      r >>= __builtin_ctzll((unsigned long long)e);
    } else {
      x = r % e;
      r /= e;
    }
    oa += (long)x * sA[d];
    ob += (long)x * sB[d];
  }
}
template <typename T2, typename T, typename I>
__device__ __forceinline__ void gen_terms(const T2 *__restrict__ A, const T2 *__restrict__ B, const ArtnGenericPlan &G, long oa, long ob,
                                          I q0, I q1, I stride, T &re, T &im) {
  // (four terms per trip: their loads are in flight together)
  I q = q0;
  for (; q + 3 * stride < q1; q += 4 * stride) {
    T2 a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      long ka, kb;
      gen_decode<I>(q + (I)u * stride, G.n_red, G.red_ext, G.red_sA, G.red_sB, ka, kb);
      a[u] = A[oa + ka];
      b[u] = B[ob + kb];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      re += a[u].x * b[u].x - a[u].y * b[u].y;
      im += a[u].x * b[u].y + a[u].y * b[u].x;
    }
  }
  for (; q < q1; q += stride) {
    long ka, kb;
    gen_decode<I>(q, G.n_red, G.red_ext, G.red_sA, G.red_sB, ka, kb);
    const T2 a = A[oa + ka], b = B[ob + kb];
    re += a.x * b.x - a.y * b.y;
    im += a.x * b.y + a.y * b.x;
  }
}
template <typename T2, typename T>
__global__ __launch_bounds__(ARTN_WG_THREADS) void artn_k_generic(const T2 *__restrict__ A,
                                                                  const T2 *__restrict__ B,
                                                                  T2 *__restrict__ C,
                                                                  const ArtnGenericPlan G) {
  const bool narrow = G.out_numel < (1l << 31) && G.red_numel < (1l << 29);
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < G.out_numel;
       idx += (long)gridDim.x * blockDim.x) {
    long oa, ob;
    T re = 0, im = 0;
    if (narrow) {
      gen_decode<unsigned>((unsigned)idx, G.n_out, G.out_ext, G.out_sA, G.out_sB, oa, ob);
      gen_terms<T2, T, unsigned>(A, B, G, oa, ob, 0u, (unsigned)G.red_numel, 1u, re, im);
    } else {
      gen_decode<long>(idx, G.n_out, G.out_ext, G.out_sA, G.out_sB, oa, ob);
      gen_terms<T2, T, long>(A, B, G, oa, ob, 0l, G.red_numel, 1l, re, im);
    }
    T2 o;
    o.x = re;
    o.y = im;
    C[idx] = o;
  }
}


// The same with ONE WORKGROUP per output element: closing steps contract thousands of values into a handful of results (the
// last step of a closed network is a dot product: one output, 1 296 terms on the bond-dimension-6 network -- 1.06 ms on a
// single thread of the kernel above, a sixth of that network's contraction).  The 256 threads stride over the terms; their
// partial sums are added in a fixed order (a tree over LDS), so the result does not depend on scheduling.
template <typename T2, typename T>
__global__ __launch_bounds__(ARTN_WG_THREADS) void artn_k_generic_red(const T2 *__restrict__ A, const T2 *__restrict__ B, T2 *__restrict__ C,
                                                                     const ArtnGenericPlan G) {
  __shared__ T part_re[ARTN_WG_THREADS], part_im[ARTN_WG_THREADS];
  const bool narrow = G.out_numel < (1l << 31) && G.red_numel < (1l << 29);
  for (long idx = blockIdx.x; idx < G.out_numel; idx += gridDim.x) {
    long oa, ob;
    T re = 0, im = 0;
    if (narrow) {
      gen_decode<unsigned>((unsigned)idx, G.n_out, G.out_ext, G.out_sA, G.out_sB, oa, ob);
      gen_terms<T2, T, unsigned>(A, B, G, oa, ob, threadIdx.x, (unsigned)G.red_numel, (unsigned)ARTN_WG_THREADS, re, im);
    } else {
      gen_decode<long>(idx, G.n_out, G.out_ext, G.out_sA, G.out_sB, oa, ob);
      gen_terms<T2, T, long>(A, B, G, oa, ob, (long)threadIdx.x, G.red_numel, (long)ARTN_WG_THREADS, re, im);
    }
    part_re[threadIdx.x] = re;
    part_im[threadIdx.x] = im;
    __syncthreads();
    for (int s = ARTN_WG_THREADS / 2; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) {
        part_re[threadIdx.x] += part_re[threadIdx.x + s];
        part_im[threadIdx.x] += part_im[threadIdx.x + s];
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      T2 o;
      o.x = part_re[0];
      o.y = part_im[0];
      C[idx] = o;
    }
    __syncthreads();
  }
}

// ----------------------------------------------------------------------------------------
// small-step programs: the launch-latency tail of a scheme in one launch
// ----------------------------------------------------------------------------------------
#define ARTN_PROG_MAX_OUT 24
#define ARTN_PROG_MAX_RED 12
#define ARTN_PROG_MAX_REDN 2048
#define ARTN_PROG_RED_ENTRIES 2048              /* reduction-offset tables of one group: 16 KiB of LDS */
#define ARTN_PROG_FAST_MAX 64                   /* matrix-core steps of one group: their bit-stride tables sit in LDS (12 KiB) */
#define ARTN_PROG_BITS_BYTES (ARTN_PROG_FAST_MAX * 48 * 4)
#define ARTN_PROG_ARENA_BYTES (128 * 1024)      /* LDS arena of one group: operands and results of its steps */
#define ARTN_PROG_LDS_BYTES (ARTN_PROG_RED_ENTRIES * 8 + ARTN_PROG_BITS_BYTES + ARTN_PROG_ARENA_BYTES)
#define ARTN_PROG_PRELOAD_MAX 4096              /* external operands up to this many elements are copied into the arena */
#define ARTN_PROG_MAGIC 0x41525032              /* "ARP2" */
#define ARTN_PROG_TASK_ELEMS 128                /* output elements of a wave task (two per lane) */
struct ArtnProgStep {
  int32_t n_out, n_red, out_numel, red_numel;
  int32_t a_numel, b_numel; // elements of the (dense) operands
  int64_t loc_a, loc_b, loc_c; // >= 0: workspace byte offset; < 0: external pointer -(loc + 1)
  int64_t tab_off;             // image byte offset of the step's reduction table: (ka, kb) element offsets per term
  int32_t lds_a, lds_b, lds_c; // byte offset in the group's LDS arena, -1: not there
  int32_t pre_a, pre_b;        // 1: this record copies its external operand into the arena before the first level
  int32_t to_ws;               // 1: the result is (also) written to the workspace
  int32_t red_base;            // first entry of the step's reduction-offset table in LDS (steps that are not `fast`)
  int32_t level;
  int32_t fast;                // 1: matrix-core step (see prog_mfma_task): wave tasks are 32 x 16 blocks, not 128 elements
  int32_t n_mbits, n_nbits;    // fast: output bits carried by the first / by the second operand
  int32_t fast_index;          // fast: which 192-byte slot of the group's LDS bit-stride area
  int32_t mbit_sA[14], mbit_sC[14]; // fast: first-operand bit b -> element stride in the first operand / in the result
  int32_t nbit_sB[10], nbit_sC[10]; // fast: second-operand bit b -> element stride in the second operand / in the result
  // output axes, fastest in the FIRST OPERAND first (axes it does not carry last): the 64 lanes of a wave task then
  // read neighbouring elements of it for every reduction term (enumerated in C order, the lanes of one ds_read hit one
  // LDS bank 32 deep on the transposing steps of a circuit: 8.5 us per level of n12); the result is scattered instead,
  // one write per element against up to 2^11 reads
  int32_t out_ext[ARTN_PROG_MAX_OUT], out_lg[ARTN_PROG_MAX_OUT], out_sA[ARTN_PROG_MAX_OUT], out_sB[ARTN_PROG_MAX_OUT], out_sC[ARTN_PROG_MAX_OUT];
  int32_t red_ext[ARTN_PROG_MAX_RED], red_lg[ARTN_PROG_MAX_RED], red_sA[ARTN_PROG_MAX_RED], red_sB[ARTN_PROG_MAX_RED];
};
// Image = header, groups, levels, wave tasks, records (byte offsets in the header; one device buffer).
struct ArtnProgHeader {
  int32_t magic, n_groups, n_steps, n_levels, n_wtasks, pad_[3];
  int64_t off_groups, off_levels, off_wtasks, off_records, off_tables, pad2_;
};
struct ArtnProgGroup { int32_t step_begin, step_end, level_begin, level_end; };
struct ArtnProgLevel { int32_t wt_begin, wt_count, first_step, pad_; }; // first_step: record of the level's first task (prefetch hint)
struct ArtnProgWTask { int32_t step, first; }; // 64 consecutive output elements of one step
struct ArtnExtPtrs {
  const void *p[ARTN_PROGRAM_MAX_EXT];
};
// One workgroup (16 waves) per group of steps.  A group's steps are sorted into LEVELS of its dependency tree
// (level 1 = both operands external, level n = an operand made at level n - 1): the steps of a level are
// independent, their output elements are dealt to the waves in slices of 64 (a wave task: one step, so the record
// is wave-uniform and read with scalar loads), one workgroup barrier per level -- n12: 19 barriers for 68 steps.
// Intermediates live in an LDS arena laid out by the host (first fit, freed after the consumer's level), small
// external operands are copied there first, the reduction offsets of every step are tabulated once up front;
// only what a later launch reads (to_ws) or what the arena cannot hold goes through the workspace.
// NE output elements per lane (a wave task is 64 * NE consecutive output elements): sum over the reduction table of
// A[oa[e] + ka] * B[ob[e] + kb].  AL / BL: the operand sits in the LDS arena (byte address a_lds / b_lds, read with
// ds_read_b64) or behind a global pointer.  Eight terms per trip: their table entries, then their 16 * NE operand
// loads, are in flight together -- the loop is a chain of load latencies, not of arithmetic.
// (element type of a program: complex64 -- the general path below plus the matrix-core steps -- or complex128, general path
//  only, 16-byte elements through ds_read_b128 / ds_write_b128, four terms in flight instead of eight: 128 registers per lane)
typedef double v2d_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) v2d_t lds_v2d_t;
template <typename T> struct ProgElem;
template <> struct ProgElem<float> {
  typedef v2f_t V2;
  typedef float2 G2;
  static constexpr unsigned ESZ = 8u;
  static constexpr int U = 8;
  static __device__ __forceinline__ V2 ldsr(unsigned a) { return lds_read8(a); }
  static __device__ __forceinline__ void ldsw(unsigned a, V2 v) { lds_write8(a, v); }
};
template <> struct ProgElem<double> {
  typedef v2d_t V2;
  typedef double2 G2;
  static constexpr unsigned ESZ = 16u;
  static constexpr int U = 4;
  static __device__ __forceinline__ V2 ldsr(unsigned a) { return *(lds_v2d_t *)(unsigned long)a; }
  static __device__ __forceinline__ void ldsw(unsigned a, V2 v) { *(lds_v2d_t *)(unsigned long)a = v; }
};
template <bool AL, bool BL, int NE, typename T>
__device__ __forceinline__ void prog_reduce(const typename ProgElem<T>::G2 *__restrict__ Ag, unsigned a_lds,
                                            const typename ProgElem<T>::G2 *__restrict__ Bg, unsigned b_lds,
                                            const int (&oa)[NE], const int (&ob)[NE], unsigned red_addr, int red_numel, T (&re)[NE],
                                            T (&im)[NE]) {
  typedef ProgElem<T> E;
  typedef typename E::V2 V2;
  constexpr int U = E::U;
  auto ldA = [&](int i) -> V2 {
    if constexpr (AL) return E::ldsr(a_lds + E::ESZ * (unsigned)i);
    else { const typename E::G2 v = Ag[i]; return V2{v.x, v.y}; }
  };
  auto ldB = [&](int i) -> V2 {
    if constexpr (BL) return E::ldsr(b_lds + E::ESZ * (unsigned)i);
    else { const typename E::G2 v = Bg[i]; return V2{v.x, v.y}; }
  };
#pragma unroll
  for (int e = 0; e < NE; ++e) re[e] = im[e] = (T)0;
  int q = 0;
  for (; q + U <= red_numel; q += U) {
    u2_t t[U];
    V2 a[NE][U], b[NE][U];
#pragma unroll
    for (int u = 0; u < U; ++u) t[u] = lds_read_u2(red_addr + 8u * (unsigned)(q + u));
#pragma unroll
    for (int e = 0; e < NE; ++e)
#pragma unroll
      for (int u = 0; u < U; ++u) { a[e][u] = ldA(oa[e] + (int)t[u].x); b[e][u] = ldB(ob[e] + (int)t[u].y); }
#pragma unroll
    for (int e = 0; e < NE; ++e)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        re[e] += a[e][u].x * b[e][u].x - a[e][u].y * b[e][u].y;
        im[e] += a[e][u].x * b[e][u].y + a[e][u].y * b[e][u].x;
      }
  }
  for (; q < red_numel; ++q) {
    const u2_t t = lds_read_u2(red_addr + 8u * (unsigned)q);
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const V2 a = ldA(oa[e] + (int)t.x), b = ldB(ob[e] + (int)t.y);
      re[e] += a.x * b.x - a.y * b.y;
      im[e] += a.x * b.y + a.y * b.x;
    }
  }
}
// `fast` steps -- the stem of a circuit scheme: a 2^12-element tensor absorbing one small tensor per step, 2^12 x 2^k
// complex multiply-adds each, which one CU's vector ALU does in 7-17 us (8 wave-instructions per multiply-add: n12
// spent 130 of its 176 us there) -- run on the matrix cores instead.  Every extent is a power of two, no output bit
// belongs to both operands, 5+ output bits belong to the first operand: a wave task is a block of 32 first-operand
// rows m x 16 second-operand columns n, computed like a sub-tile of artn_k_bits: interleaved complex64 times the
// real block form of the small operand on v_mfma_f32_32x32x2_f32 (lane roles: see artn_k_bits).  Offsets are
// bit-linear: the host tabulates a stride per output bit (mbit_* / nbit_*); the reduction table is the step's LDS
// table, operands come from the arena or from global memory.
template <bool AL, bool BL>
__device__ __forceinline__ void prog_mfma_task(int n_mbits, int n_nbits, int task, int lane, const float2 *__restrict__ Ag, unsigned a_lds,
                                               const float2 *__restrict__ Bg, unsigned b_lds, unsigned tab, int red_numel,
                                               unsigned c_lds, bool c_in_lds, float2 *__restrict__ Cg, unsigned bits, int dbg = 1000) {
  auto ldA = [&](int i) -> v2f_t {
    if constexpr (AL) return lds_read8(a_lds + 8u * (unsigned)i);
    else { const float2 v = Ag[i]; return v2f_t{v.x, v.y}; }
  };
  auto ldB = [&](int i) -> v2f_t {
    if constexpr (BL) return lds_read8(b_lds + 8u * (unsigned)i);
    else { const float2 v = Bg[i]; return v2f_t{v.x, v.y}; }
  };
  const int msub = task & ((1 << (n_mbits - 5)) - 1), ntile = task >> (n_mbits - 5);
  const int j = lane & 31, h = lane >> 5, ro = j & 1;
  // (48 scalar loads from the record, each waited for where the predicated adds below use it, were 3 of the 5 us of
  //  a task: the tables are copied to LDS once per launch and read here with twelve 16-byte broadcast reads)
  int bt[48];
#pragma unroll
  for (int v = 0; v < 12; ++v) {
    typedef int i32x4_t __attribute__((ext_vector_type(4)));
    const i32x4_t q4 = *(__attribute__((address_space(3))) i32x4_t *)(unsigned long)(bits + 16u * (unsigned)v); // (integers: read as integers)
#pragma unroll
    for (int c = 0; c < 4; ++c) bt[4 * v + c] = q4[c];
  }
  const int *sa = bt, *sc = bt + 14, *sb = bt + 28, *sn = bt + 38;
  // X side: tile column j = first-operand row msub * 32 + j
  const int midx = msub * 32 + j;
  int oa = 0, ocm = 0;
#pragma unroll
  for (int b = 0; b < 14; ++b)
    if (b < n_mbits && ((midx >> b) & 1)) { oa += sa[b]; ocm += sc[b]; }
  // W side: MFMA row i = j = 2 * n_in_block + ro
  const int nidx = ntile * 16 + (j >> 1);
  const bool w_valid = nidx < (1 << n_nbits);
  int ob = 0;
#pragma unroll
  for (int b = 0; b < 10; ++b)
    if (b < n_nbits && ((nidx >> b) & 1)) ob += sb[b];
  // result: accumulator register r of lane (j, h) is column n = ntile * 16 + ((r >> 1) & 1) + 2 h + 4 (r >> 2), part r & 1
  const int nbase = ntile * 16 + 2 * h;
  int ocn = 0;
#pragma unroll
  for (int b = 1; b < 10; ++b)
    if (b < n_nbits && ((nbase >> b) & 1)) ocn += sn[b];
  const int c0 = n_nbits > 0 ? sn[0] : 0, c2 = n_nbits > 2 ? sn[2] : 0, c3 = n_nbits > 3 ? sn[3] : 0;
  PROG_FINE(dbg, 4, oa + ocm + ob + ocn + c0 + c2 + c3);
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const int n_steps = red_numel >> 1; // contracted values 2 s + h
  int s0 = 0;
  // (batches of 8, then 4, MFMA pairs: the table entries of a batch, then its 16 operand reads, are in flight together
  //  -- per batch the chain pays two LDS round trips whatever its length)
  for (; s0 + 8 <= n_steps; s0 += 8) {
    u2_t t[8];
    v2f_t x[8], w[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = lds_read_u2(tab + 8u * (unsigned)(2 * (s0 + u) + h));
#pragma unroll
    for (int u = 0; u < 8; ++u) { x[u] = ldA(oa + (int)t[u].x); w[u] = ldB(ob + (int)t[u].y); }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float w0 = w_valid ? (ro ? w[u].y : w[u].x) : 0.f, w1 = w_valid ? (ro ? w[u].x : -w[u].y) : 0.f;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w0, x[u].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w1, x[u].y, acc, 0, 0, 0);
    }
  }
  for (; s0 + 4 <= n_steps; s0 += 4) {
    u2_t t[4];
    v2f_t x[4], w[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) t[u] = lds_read_u2(tab + 8u * (unsigned)(2 * (s0 + u) + h));
#pragma unroll
    for (int u = 0; u < 4; ++u) { x[u] = ldA(oa + (int)t[u].x); w[u] = ldB(ob + (int)t[u].y); }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float w0 = w_valid ? (ro ? w[u].y : w[u].x) : 0.f, w1 = w_valid ? (ro ? w[u].x : -w[u].y) : 0.f;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w0, x[u].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w1, x[u].y, acc, 0, 0, 0);
    }
  }
  for (; s0 < n_steps; ++s0) {
    const u2_t t = lds_read_u2(tab + 8u * (unsigned)(2 * s0 + h));
    const v2f_t x = ldA(oa + (int)t.x), w = ldB(ob + (int)t.y);
    const float w0 = w_valid ? (ro ? w.y : w.x) : 0.f, w1 = w_valid ? (ro ? w.x : -w.y) : 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w0, x.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w1, x.y, acc, 0, 0, 0);
  }
  PROG_FINE(dbg, 5, (int)acc[0]);
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int b0 = 0; b0 < 2; ++b0) {
      const int n = nbase + b0 + 4 * q;
      if (n < (1 << n_nbits)) {
        const int oc = ocm + ocn + (b0 ? c0 : 0) + ((q & 1) ? c2 : 0) + ((q >> 1) ? c3 : 0);
        const v2f_t val = {acc[4 * q + 2 * b0], acc[4 * q + 2 * b0 + 1]};
        if (c_in_lds) lds_write8(c_lds + 8u * (unsigned)oc, val);
        if (Cg) Cg[oc] = make_float2(val.x, val.y);
      }
    }
}
template <typename T>
__global__ __launch_bounds__(1024) void artn_k_program(const char *__restrict__ image, const ArtnExtPtrs ext, char *__restrict__ ws) {
  typedef ProgElem<T> E;
  typedef typename E::G2 G2;
  extern __shared__ __attribute__((aligned(16))) unsigned char prog_smem[];
  int2 *red_all = reinterpret_cast<int2 *>(prog_smem);
  unsigned char *arena = prog_smem + ARTN_PROG_RED_ENTRIES * 8 + ARTN_PROG_BITS_BYTES;
  int *bits_all = reinterpret_cast<int *>(prog_smem + ARTN_PROG_RED_ENTRIES * 8);
  // LDS byte addresses: reduction tables, bit-stride tables of the matrix-core steps, arena
  const unsigned red_lds = (unsigned)(unsigned long)(lds_byte_t *)prog_smem, bits_lds = red_lds + ARTN_PROG_RED_ENTRIES * 8,
                 arena_lds = bits_lds + ARTN_PROG_BITS_BYTES;
  const ArtnProgHeader *H = reinterpret_cast<const ArtnProgHeader *>(image);
  const ArtnProgGroup G = reinterpret_cast<const ArtnProgGroup *>(image + H->off_groups)[blockIdx.x];
  const ArtnProgLevel *levels = reinterpret_cast<const ArtnProgLevel *>(image + H->off_levels);
  const ArtnProgWTask *wtasks = reinterpret_cast<const ArtnProgWTask *>(image + H->off_wtasks);
  const ArtnProgStep *recs = reinterpret_cast<const ArtnProgStep *>(image + H->off_records);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  PROG_MARK(0);
  // ---- reduction-offset tables and the external operands that live in the arena: one wave per record
  for (int s = G.step_begin + wave; s < G.step_end; s += 16) {
    const ArtnProgStep &R = recs[s];
    const int n_red = R.n_red, red_numel = R.red_numel, red_base = R.red_base;
    int e4[4], l4[4], a4[4], b4[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) { e4[d] = R.red_ext[d]; l4[d] = R.red_lg[d]; a4[d] = R.red_sA[d]; b4[d] = R.red_sB[d]; }
    for (int q = lane; q < red_numel; q += 64) {
      int rr = q, ka = 0, kb = 0;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        if (d < n_red) {
          const int e = e4[d], lg = l4[d];
          const int x = lg >= 0 ? rr & (e - 1) : rr % e;
          rr = lg >= 0 ? rr >> lg : rr / e;
          ka += x * a4[d];
          kb += x * b4[d];
        }
      }
      for (int d = 4; d < n_red; ++d) {
        const int e = R.red_ext[d], lg = R.red_lg[d];
        const int x = lg >= 0 ? rr & (e - 1) : rr % e;
        rr = lg >= 0 ? rr >> lg : rr / e;
        ka += x * R.red_sA[d];
        kb += x * R.red_sB[d];
      }
      red_all[red_base + q] = make_int2(ka, kb);
    }
    if (R.fast && lane < 48) // (mbit_sA, mbit_sC, nbit_sB, nbit_sC are contiguous: 48 ints)
      bits_all[R.fast_index * 48 + lane] = (reinterpret_cast<const int *>(&R) + offsetof(ArtnProgStep, mbit_sA) / 4)[lane];
    if (R.pre_a) {
      const G2 *src = reinterpret_cast<const G2 *>(ext.p[-(R.loc_a + 1)]);
      G2 *dst = reinterpret_cast<G2 *>(arena + R.lds_a);
      for (int e = lane; e < R.a_numel; e += 64) dst[e] = src[e];
    }
    if (R.pre_b) {
      const G2 *src = reinterpret_cast<const G2 *>(ext.p[-(R.loc_b + 1)]);
      G2 *dst = reinterpret_cast<G2 *>(arena + R.lds_b);
      for (int e = lane; e < R.b_numel; e += 64) dst[e] = src[e];
    }
  }
  __syncthreads();
  PROG_MARK(1);
  // Record fields come through scalar loads from global memory, ~0.3 us each when they miss the scalar cache: a wave
  // takes a CONTIGUOUS range of the level's tasks (mostly one step: its record is read once and kept), every field is
  // requested up front -- the first 8 output axes as five 8-dword loads -- and the next task's table entry is fetched
  // before the current task computes.  A task is 128 output elements, two per lane: half as many trips through the
  // load-latency chain of the reduction.  (Dependent loads inside the decode loop, one element per lane, tasks dealt
  // round-robin: the 19 levels of n12 took 290 us.)
  // (the next level's table entry is requested a level ahead; pulling the lines it points to -- first task, that
  //  task's record -- into the scalar cache before the barrier cost more than it saved: 152 -> 183 us on n12)
  constexpr int NE = ARTN_PROG_TASK_ELEMS / 64;
  ArtnProgLevel lv_next = levels[G.level_begin];
  for (int L = G.level_begin; L < G.level_end; ++L) {
    const ArtnProgLevel lv = lv_next;
    if (L + 1 < G.level_end) lv_next = levels[L + 1];
    const int per = (lv.wt_count + 15) >> 4, t_begin = wave * per, t_end = t_begin + per < lv.wt_count ? t_begin + per : lv.wt_count;
    ArtnProgWTask W = {0, 0};
    if (t_begin < t_end) W = wtasks[lv.wt_begin + t_begin];
    int cur = -1, cur_dims = -1;
    int n_out = 0, out_numel = 0, red_numel = 0, red_base = 0, lds_a = -1, lds_b = -1, lds_c = -1, to_ws = 0, fast = 0, fast_index = 0;
    int n_mbits = 0, n_nbits = 0;
    long loc_a = 0, loc_b = 0, loc_c = 0;
    int e8[8], l8[8], a8[8], b8[8], c8[8];
    for (int t = t_begin; t < t_end; ++t) {
      if (t == t_begin) PROG_FINE(L - G.level_begin, 0, lv.wt_count);
      if (t == t_begin) PROG_FINE(L - G.level_begin, 1, W.step);
      ArtnProgWTask Wn = W;
      if (t + 1 < t_end) Wn = wtasks[lv.wt_begin + t + 1];
      const ArtnProgStep &R = recs[W.step];
      if (W.step != cur) { // (one batch of scalar loads: the head of the record)
        cur = W.step;
        n_out = R.n_out; out_numel = R.out_numel; red_numel = R.red_numel; red_base = R.red_base;
        lds_a = R.lds_a; lds_b = R.lds_b; lds_c = R.lds_c; to_ws = R.to_ws; fast = R.fast; fast_index = R.fast_index;
        n_mbits = R.n_mbits; n_nbits = R.n_nbits;
        loc_a = R.loc_a; loc_b = R.loc_b; loc_c = R.loc_c;
      }
      if (t == t_begin) PROG_FINE(L - G.level_begin, 2, n_out + (int)loc_c);
      // (an operand is in the arena, in the workspace or behind an external pointer)
      const G2 *A = nullptr, *B = nullptr;
      if (lds_a < 0) A = reinterpret_cast<const G2 *>(loc_a >= 0 ? ws + loc_a : (const char *)ext.p[-(loc_a + 1)]);
      if (lds_b < 0) B = reinterpret_cast<const G2 *>(loc_b >= 0 ? ws + loc_b : (const char *)ext.p[-(loc_b + 1)]);
      const unsigned al = arena_lds + (unsigned)lds_a, bl = arena_lds + (unsigned)lds_b, rt = red_lds + 8u * (unsigned)red_base;
      if constexpr (std::is_same<T, float>::value) if (fast) {
        float2 *Cg = to_ws ? reinterpret_cast<float2 *>(ws + loc_c) : nullptr;
        const unsigned cl = arena_lds + (unsigned)lds_c, bits = bits_lds + 192u * (unsigned)fast_index;
        const int dbg = t == t_begin ? L - G.level_begin : 1000;
        if (lds_a >= 0 && lds_b >= 0) prog_mfma_task<true, true>(n_mbits, n_nbits, W.first, lane, A, al, B, bl, rt, red_numel, cl, lds_c >= 0, Cg, bits, dbg);
        else if (lds_a >= 0) prog_mfma_task<true, false>(n_mbits, n_nbits, W.first, lane, A, al, B, bl, rt, red_numel, cl, lds_c >= 0, Cg, bits, dbg);
        else if (lds_b >= 0) prog_mfma_task<false, true>(n_mbits, n_nbits, W.first, lane, A, al, B, bl, rt, red_numel, cl, lds_c >= 0, Cg, bits, dbg);
        else prog_mfma_task<false, false>(n_mbits, n_nbits, W.first, lane, A, al, B, bl, rt, red_numel, cl, lds_c >= 0, Cg, bits, dbg);
        if (t == t_begin) PROG_FINE(L - G.level_begin, 3, W.first);
        W = Wn;
        continue;
      }
      if (W.step != cur_dims) { // (the general path's axes: five 8-dword loads, kept while the wave stays on this step)
        cur_dims = W.step;
#pragma unroll
        for (int d = 0; d < 8; ++d) { e8[d] = R.out_ext[d]; l8[d] = R.out_lg[d]; a8[d] = R.out_sA[d]; b8[d] = R.out_sB[d]; c8[d] = R.out_sC[d]; }
      }
      int oa[NE], ob[NE], oc[NE];
      bool live[NE];
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const int idx = W.first + 64 * e + lane;
        live[e] = idx < out_numel;
        int r = live[e] ? idx : 0; // (lanes past the end compute element 0 and store nothing)
        oa[e] = ob[e] = oc[e] = 0;
#pragma unroll
        for (int d = 0; d < 8; ++d) {
          if (d < n_out) {
            const int ex = e8[d], lg = l8[d];
            const int x = lg >= 0 ? r & (ex - 1) : r % ex;
            r = lg >= 0 ? r >> lg : r / ex;
            oa[e] += x * a8[d];
            ob[e] += x * b8[d];
            oc[e] += x * c8[d];
          }
        }
        for (int d = 8; d < n_out; ++d) {
          const int ex = R.out_ext[d], lg = R.out_lg[d];
          const int x = lg >= 0 ? r & (ex - 1) : r % ex;
          r = lg >= 0 ? r >> lg : r / ex;
          oa[e] += x * R.out_sA[d];
          ob[e] += x * R.out_sB[d];
          oc[e] += x * R.out_sC[d];
        }
      }
      T re[NE], im[NE];
      if (lds_a >= 0 && lds_b >= 0) prog_reduce<true, true, NE, T>(A, al, B, bl, oa, ob, rt, red_numel, re, im);
      else if (lds_a >= 0) prog_reduce<true, false, NE, T>(A, al, B, bl, oa, ob, rt, red_numel, re, im);
      else if (lds_b >= 0) prog_reduce<false, true, NE, T>(A, al, B, bl, oa, ob, rt, red_numel, re, im);
      else prog_reduce<false, false, NE, T>(A, al, B, bl, oa, ob, rt, red_numel, re, im);
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        if (live[e]) {
          if (lds_c >= 0) E::ldsw(arena_lds + (unsigned)lds_c + E::ESZ * (unsigned)oc[e], typename E::V2{re[e], im[e]});
          if (to_ws) {
            G2 o;
            o.x = re[e];
            o.y = im[e];
            reinterpret_cast<G2 *>(ws + loc_c)[oc[e]] = o;
          }
        }
      }
      W = Wn;
    }
    PROG_MARK(2 + 2 * (L - G.level_begin));
    __threadfence_block();
    __syncthreads();
    PROG_MARK(3 + 2 * (L - G.level_begin));
  }
}

// ----------------------------------------------------------------------------------------
// row gather / slice accumulate / renormalise
// ----------------------------------------------------------------------------------------
template <typename V>
__global__ __launch_bounds__(ARTN_WG_THREADS) void artn_k_gather_rows(const V *__restrict__ src,
                                                                      const int64_t *__restrict__ idx,
                                                                      V *__restrict__ dst, long nrows,
                                                                      long row_vecs, long src_rows,
                                                                      int *err_flag) {
  const long total = nrows * row_vecs;
  for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += (long)gridDim.x * blockDim.x) {
    const long r = c / row_vecs, col = c - r * row_vecs;
    const long s = idx[r];
    V v;
    if (s >= 0 && s < src_rows) {
      v = src[s * row_vecs + col];
    } else {
      memset(&v, 0, sizeof(V));
      if (err_flag && col == 0) atomicOr(err_flag, 1);
    }
    dst[c] = v;
  }
}

__global__ __launch_bounds__(ARTN_WG_THREADS) void artn_k_axpy4(float4 *__restrict__ acc,
                                                                const float4 *__restrict__ x, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float4 a = acc[i];
    const float4 b = x[i];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    acc[i] = a;
  }
}
__global__ void artn_k_axpy1(float *__restrict__ acc, const float *__restrict__ x, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    acc[i] += x[i];
}

__global__ __launch_bounds__(ARTN_WG_THREADS) void artn_k_absmax(const float2 *__restrict__ x, long n,
                                                                 unsigned int *out_bits) {
  float m = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float2 v = x[i];
    m = fmaxf(m, hypotf(v.x, v.y));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  __shared__ float part[ARTN_WG_THREADS / 64];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < ARTN_WG_THREADS / 64; ++w) m = fmaxf(m, part[w]);
    atomicMax(out_bits, __float_as_uint(m)); // non-negative floats order like their bit patterns
  }
}
__global__ __launch_bounds__(ARTN_WG_THREADS) void artn_k_divide(float2 *__restrict__ x, long n,
                                                                 const float *__restrict__ denom) {
  const float d = *denom;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float2 v = x[i];
    v.x /= d;
    v.y /= d;
    x[i] = v;
  }
}
// the same in complex128 (the reference renormalises whatever dtype it runs in, contraction.py:197-200)
__global__ __launch_bounds__(ARTN_WG_THREADS) void artn_k_absmax128(const double2 *__restrict__ x, long n,
                                                                    unsigned long long *out_bits) {
  double m = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const double2 v = x[i];
    m = fmax(m, hypot(v.x, v.y));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
  __shared__ double part[ARTN_WG_THREADS / 64];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < ARTN_WG_THREADS / 64; ++w) m = fmax(m, part[w]);
    atomicMax(out_bits, (unsigned long long)__double_as_longlong(m)); // non-negative doubles order like their bit patterns
  }
}
__global__ __launch_bounds__(ARTN_WG_THREADS) void artn_k_divide128(double2 *__restrict__ x, long n,
                                                                    const double *__restrict__ denom) {
  const double d = *denom;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    double2 v = x[i];
    v.x /= d;
    v.y /= d;
    x[i] = v;
  }
}

#endif // !ARTN_TU_PART

// ----------------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------------
static int g_ndev = -1, g_ncu = 256;
static std::once_flag g_once;
static void probe_devices() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
  int good = 0;
  for (int d = 0; d < n; ++d) {
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, d) != hipSuccess) continue;
    if (strncmp(pr.gcnArchName, "gfx950", 6) == 0) {
      ++good;
      g_ncu = pr.multiProcessorCount;
    }
  }
  (void)hipGetLastError();
  g_ndev = good;
}

static bool env_flag(const char *name) {
  const char *v = getenv(name);
  return v && v[0] && v[0] != '0';
}

// Kernels that need more than 64 KiB of dynamic LDS must say so once per device; repeating
// the call per launch is needless host work and is not welcome during stream capture.
template <auto Kern>
static hipError_t ensure_lds(size_t lds) {
  static std::atomic<int> have[16];
  if (lds <= 64 * 1024) return hipSuccess;
  int dev = 0;
  if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
  std::atomic<int> &h = have[dev & 15];
  if ((int)lds <= h.load(std::memory_order_relaxed)) return hipSuccess;
  hipError_t e = hipFuncSetAttribute((const void *)Kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e == hipSuccess) h.store((int)lds, std::memory_order_relaxed);
  return e;
}

// (-DARTN_TU_HALF=0 / 1 with -DARTN_TU_BITS=K: only the second-stage counts 0..3 / 4..6 of that family -- the families of 5 and
//  6 contracted bits took a minute each to compile and were the long pole of `make -j8`)
#ifndef ARTN_TU_HALF
#define ARTN_TU_HALF -1
#endif
#define ARTN_K2_LO (ARTN_TU_HALF != 1)
#define ARTN_K2_HI (ARTN_TU_HALF != 0)
template <int KB1>
static hipError_t launch_bits_k2(const ArtnPlan &p, const float2 *A, const float2 *B1, const float2 *B2, float2 *C,
                                 hipStream_t st) {
  dim3 grid(p.info.grid), block(ARTN_WG_THREADS);
  const size_t lds = (size_t)p.info.lds_bytes;
  const int k2 = p.bits.n_stages == 2 ? p.bits.st[1].k : 0;
  const int split = p.bits.split;
  const bool full = p.bits.T_in == 12 && p.bits.T_out == 12; // (the FULL instantiations)
#define ARTN_LAUNCH_NP(K2, NPV)                                                                           \
  {                                                                                                       \
    auto kern = artn_k_bits<KB1, K2, false, NPV>;                                                         \
    if (hipError_t e = ensure_lds<artn_k_bits<KB1, K2, false, NPV>>(lds); e != hipSuccess) return e;      \
    hipLaunchKernelGGL(kern, grid, block, lds, st, A, B1, B2, C, p.bits);                                 \
  }
  // split-bf16 instantiations exist only where a stage has the >= 3 contracted bits they need -- and only in development
  // builds (-DARTN_DEV_SPLIT3): fp32-grade results from three bfloat16 pieces, parity-green, no faster on any workload
  // (DESIGN.md 4.1); the product never plans split = 3
#ifdef ARTN_DEV_SPLIT3
#define ARTN_SPLIT3_CASE(K2) if (split == 3) { ARTN_LAUNCH_NP(K2, 3) break; }
#else
#define ARTN_SPLIT3_CASE(K2) if (split == 3) return hipErrorInvalidValue;
#endif
#define ARTN_LAUNCH(K2)                                                                                   \
  case K2: {                                                                                              \
    if constexpr (KB1 >= 3 || K2 >= 3) {                                                                  \
      ARTN_SPLIT3_CASE(K2)                                                                                \
      if constexpr (KB1 >= 3 && (K2 == 0 || K2 >= 3)) {                                                   \
        if (split == 1 && p.bits.nt_loads) { /* bf16 operands: every big launch is HBM-bound */           \
          auto kern = artn_k_bits<KB1, K2, false, 1, false, true>;                                        \
          if (hipError_t e = ensure_lds<artn_k_bits<KB1, K2, false, 1, false, true>>(lds); e != hipSuccess) return e; \
          hipLaunchKernelGGL(kern, grid, block, lds, st, A, B1, B2, C, p.bits);                           \
          break;                                                                                          \
        }                                                                                                 \
      }                                                                                                   \
      if (split == 1) { ARTN_LAUNCH_NP(K2, 1) break; }                                                    \
    }                                                                                                     \
    if constexpr ((KB1 == 5 || KB1 == 6 || K2 == 5 || K2 == 6) && !((KB1 >= 5 && K2 >= 5) && KB1 + K2 > 11)) {      \
      if (p.bits.m3 && p.bits.nt_loads && full) { /* three real products per complex product in the 5-bit stages */ \
        auto kern = artn_k_bits<KB1, K2, false, 0, false, true, true, true>;                              \
        if (hipError_t e = ensure_lds<artn_k_bits<KB1, K2, false, 0, false, true, true, true>>(lds); e != hipSuccess) return e; \
        hipLaunchKernelGGL(kern, grid, block, lds, st, A, B1, B2, C, p.bits);                             \
        break;                                                                                            \
      }                                                                                                   \
      if (p.bits.m3 && full) {                                                                            \
        auto kern = artn_k_bits<KB1, K2, false, 0, false, false, true, true>;                             \
        if (hipError_t e = ensure_lds<artn_k_bits<KB1, K2, false, 0, false, false, true, true>>(lds); e != hipSuccess) return e; \
        hipLaunchKernelGGL(kern, grid, block, lds, st, A, B1, B2, C, p.bits);                             \
        break;                                                                                            \
      }                                                                                                   \
      if (p.bits.m3 && p.bits.nt_loads) {                                                                 \
        auto kern = artn_k_bits<KB1, K2, false, 0, false, true, true>;                                    \
        if (hipError_t e = ensure_lds<artn_k_bits<KB1, K2, false, 0, false, true, true>>(lds); e != hipSuccess) return e; \
        hipLaunchKernelGGL(kern, grid, block, lds, st, A, B1, B2, C, p.bits);                             \
        break;                                                                                            \
      }                                                                                                   \
      if (p.bits.m3) {                                                                                    \
        auto kern = artn_k_bits<KB1, K2, false, 0, false, false, true>;                                   \
        if (hipError_t e = ensure_lds<artn_k_bits<KB1, K2, false, 0, false, false, true>>(lds); e != hipSuccess) return e; \
        hipLaunchKernelGGL(kern, grid, block, lds, st, A, B1, B2, C, p.bits);                             \
        break;                                                                                            \
      }                                                                                                   \
    }                                                                                                     \
    if constexpr (KB1 >= 3 && (K2 == 0 || K2 >= 3)) {                                                     \
      if (p.bits.nt_loads && full) { /* the big steps: non-temporal loads of A, 2^12-element tiles */     \
        auto kern = artn_k_bits<KB1, K2, false, 0, false, true, false, true>;                             \
        if (hipError_t e = ensure_lds<artn_k_bits<KB1, K2, false, 0, false, true, false, true>>(lds); e != hipSuccess) return e; \
        hipLaunchKernelGGL(kern, grid, block, lds, st, A, B1, B2, C, p.bits);                             \
        break;                                                                                            \
      }                                                                                                   \
      if (p.bits.nt_loads) { /* non-temporal loads of A */                                                \
        auto kern = artn_k_bits<KB1, K2, false, 0, false, true>;                                          \
        if (hipError_t e = ensure_lds<artn_k_bits<KB1, K2, false, 0, false, true>>(lds); e != hipSuccess) return e; \
        hipLaunchKernelGGL(kern, grid, block, lds, st, A, B1, B2, C, p.bits);                             \
        break;                                                                                            \
      }                                                                                                   \
    }                                                                                                     \
    ARTN_LAUNCH_NP(K2, 0)                                                                                 \
    break;                                                                                                \
  }
  // the big launches (2^12-element tiles, fp32 chains, grid-stride tiles): one 8-wave workgroup per CU whose two
  // groups alternate between the MFMA stages and the copy phases (artn_k_alt)
  if constexpr (KB1 >= 3) {
    if (full && p.bits.T_mid == 12 && split == 0 && p.bits.gather_dim < 0 && !p.bits.blocked && p.bits.st[0].k <= 6 && !p.bits.accumulate &&
        (artn::tuning().alt == 1 || (artn::tuning().alt == 2 && p.bits.run_out < 4)) &&
        p.bits.n_tiles >= 64 && (k2 == 0 || k2 >= 3)) {
      const long half_tiles = (long)((p.bits.n_tiles + 1) / 2);
      dim3 agrid((unsigned)std::min<long>((long)p.n_cu, half_tiles)), ablock(2 * ARTN_WG_THREADS);
      const size_t alds = lds + 65536;
#define ARTN_ALT_GO(K2, NTV, M3V)                                                                           \
  {                                                                                                       \
    auto kern = artn_k_alt<KB1, K2, NTV, M3V>;                                                            \
    if (hipError_t e = ensure_lds<artn_k_alt<KB1, K2, NTV, M3V>>(alds); e != hipSuccess) return e;        \
    hipLaunchKernelGGL(kern, agrid, ablock, alds, st, A, B1, B2, C, p.bits);                              \
    return hipGetLastError();                                                                             \
  }
#define ARTN_ALT_CASE(K2)                                                                                 \
  case K2: {                                                                                              \
    if constexpr ((KB1 == 5 || KB1 == 6 || K2 == 5 || K2 == 6) && !((KB1 >= 5 && K2 >= 5) && KB1 + K2 > 11)) { \
      if (p.bits.m3) {                                                                                    \
        if (p.bits.nt_loads) ARTN_ALT_GO(K2, true, true) else ARTN_ALT_GO(K2, false, true)                \
      }                                                                                                   \
    }                                                                                                     \
    if (!p.bits.m3) {                                                                                     \
      if (p.bits.nt_loads) ARTN_ALT_GO(K2, true, false) else ARTN_ALT_GO(K2, false, false)                \
    }                                                                                                     \
    break;                                                                                                \
  }
      switch (k2) {
#if ARTN_K2_LO
        ARTN_ALT_CASE(0)
        ARTN_ALT_CASE(3)
#endif
#if ARTN_K2_HI
        ARTN_ALT_CASE(4)
        ARTN_ALT_CASE(5)
        ARTN_ALT_CASE(6)
#endif
        default: break;
      }
#undef ARTN_ALT_CASE
#undef ARTN_ALT_GO
    }
  }
#if ARTN_K2_LO
  if constexpr (KB1 == 5 || KB1 == 6) { // single steps that keep at most 4 result bits in the tile: 16 x 16 x 4 blocks, three products
    if (p.bits.narrow3 == 1 && k2 == 0 && split == 0 && p.bits.gather_dim < 0 && p.bits.st[0].k <= 6 && !full) {
      if (p.bits.nt_loads) {
        auto kern = artn_k_bits<KB1, 0, false, 0, false, true, false, false, 1>;
        if (hipError_t e = ensure_lds<artn_k_bits<KB1, 0, false, 0, false, true, false, false, 1>>(lds); e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, grid, block, lds, st, A, B1, B2, C, p.bits);
      } else {
        auto kern = artn_k_bits<KB1, 0, false, 0, false, false, false, false, 1>;
        if (hipError_t e = ensure_lds<artn_k_bits<KB1, 0, false, 0, false, false, false, false, 1>>(lds); e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, grid, block, lds, st, A, B1, B2, C, p.bits);
      }
      return hipGetLastError();
    }
  }
#endif
#if ARTN_K2_HI
  if constexpr (KB1 >= 2) { // 3M pairs whose SECOND stage keeps at most 4 result bits in the tile (ArtnBitsPlan::narrow3 = 2)
    if (p.bits.narrow3 == 2 && p.bits.m3 && (k2 == 5 || k2 == 6) && split == 0 && p.bits.gather_dim < 0 && p.bits.st[0].k <= 6 && !full) {
#define ARTN_N3_GO(K2, NTV)                                                                               \
  {                                                                                                       \
    auto kern = artn_k_bits<KB1, K2, false, 0, false, NTV, true, false, 2>;                               \
    if (hipError_t e = ensure_lds<artn_k_bits<KB1, K2, false, 0, false, NTV, true, false, 2>>(lds); e != hipSuccess) return e; \
    hipLaunchKernelGGL(kern, grid, block, lds, st, A, B1, B2, C, p.bits);                                 \
    return hipGetLastError();                                                                             \
  }
      if (k2 == 5) { if (p.bits.nt_loads) ARTN_N3_GO(5, true) else ARTN_N3_GO(5, false) }
      else { if (p.bits.nt_loads) ARTN_N3_GO(6, true) else ARTN_N3_GO(6, false) }
#undef ARTN_N3_GO
    }
  }
#endif
#if ARTN_K2_LO
  if (p.bits.gather_dim >= 0) { // fused row gather: single stage, fp32 chains
    if (k2 != 0) return hipErrorInvalidValue;
    if (KB1 == 6 && p.bits.st[0].k > 6) {
      auto kern = artn_k_bits<(KB1 == 6 ? 6 : 1), 0, true, 0, true>;
      if (hipError_t e = ensure_lds<artn_k_bits<(KB1 == 6 ? 6 : 1), 0, true, 0, true>>(lds); e != hipSuccess) return e;
      hipLaunchKernelGGL(kern, grid, block, lds, st, A, B1, B2, C, p.bits);
    } else {
      auto kern = artn_k_bits<KB1, 0, false, 0, true>;
      if (hipError_t e = ensure_lds<artn_k_bits<KB1, 0, false, 0, true>>(lds); e != hipSuccess) return e;
      hipLaunchKernelGGL(kern, grid, block, lds, st, A, B1, B2, C, p.bits);
    }
    return hipGetLastError();
  }
  if (KB1 == 6 && k2 == 0 && p.bits.st[0].k > 6 && split == 1) { // bf16 operands
    auto kern = artn_k_bits<(KB1 == 6 ? 6 : 1), 0, true, 1>;
    if (hipError_t e = ensure_lds<artn_k_bits<(KB1 == 6 ? 6 : 1), 0, true, 1>>(lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, grid, block, lds, st, A, B1, B2, C, p.bits);
    return hipGetLastError();
  }
  if (KB1 == 6 && k2 == 0 && p.bits.st[0].k > 6) {
    auto kern = artn_k_bits<(KB1 == 6 ? 6 : 1), 0, true>;
    if (hipError_t e = ensure_lds<artn_k_bits<(KB1 == 6 ? 6 : 1), 0, true>>(lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, grid, block, lds, st, A, B1, B2, C, p.bits);
    return hipGetLastError();
  }
#endif
  switch (k2) {
#if ARTN_K2_LO
    ARTN_LAUNCH(0)
    ARTN_LAUNCH(1)
    ARTN_LAUNCH(2)
    ARTN_LAUNCH(3)
#endif
#if ARTN_K2_HI
    ARTN_LAUNCH(4)
    ARTN_LAUNCH(5)
    ARTN_LAUNCH(6)
#endif
    default: return hipErrorInvalidValue;
  }
#undef ARTN_LAUNCH
#undef ARTN_LAUNCH_NP
  return hipGetLastError();
}

// complex128 plans of make_bits: artn_k_bits128<KB1, KB2, ACC> (ACC: ArtnBitsPlan::accumulate; its own translation unit)
#if defined(ARTN_TU_B128) || (!defined(ARTN_TU_PART) && !defined(ARTN_TU_MAIN))
template <bool ACC>
static hipError_t launch_bits128_t(const ArtnPlan &p, const void *A, const void *B1, const void *B2, void *C, hipStream_t st) {
  dim3 grid(p.info.grid), block(ARTN_WG_THREADS);
  const size_t lds = (size_t)p.info.lds_bytes;
  const int k1 = p.bits.st[0].k, k2 = p.bits.n_stages == 2 ? p.bits.st[1].k : 0;
  const double2 *a = (const double2 *)A, *b1 = (const double2 *)B1, *b2 = (const double2 *)B2;
  double2 *c = (double2 *)C;
#define ARTN_B128_GO(K1, K2)                                                                        \
  {                                                                                                 \
    auto kern = artn_k_bits128<K1, K2, ACC>;                                                        \
    if (hipError_t e = ensure_lds<artn_k_bits128<K1, K2, ACC>>(lds); e != hipSuccess) return e;     \
    hipLaunchKernelGGL(kern, grid, block, lds, st, a, b1, b2, c, p.bits);                           \
    return hipGetLastError();                                                                       \
  }
#define ARTN_B128_K2(K1)                                                                            \
  case K1:                                                                                          \
    switch (k2) {                                                                                   \
      case 0: ARTN_B128_GO(K1, 0)                                                                   \
      case 1: ARTN_B128_GO(K1, 1)                                                                   \
      case 2: ARTN_B128_GO(K1, 2)                                                                   \
      case 3: ARTN_B128_GO(K1, 3)                                                                   \
      case 4: ARTN_B128_GO(K1, 4)                                                                   \
      case 5: ARTN_B128_GO(K1, 5)                                                                   \
      case 6: ARTN_B128_GO(K1, 6)                                                                   \
      default: return hipErrorInvalidValue;                                                         \
    }
  switch (k1) {
    ARTN_B128_K2(1)
    ARTN_B128_K2(2)
    ARTN_B128_K2(3)
    ARTN_B128_K2(4)
    ARTN_B128_K2(5)
    ARTN_B128_K2(6)
    default: return hipErrorInvalidValue;
  }
#undef ARTN_B128_K2
#undef ARTN_B128_GO
}
#if !defined(ARTN_TU_B128A)
hipError_t artn_launch_bits128(const ArtnPlan &p, const void *A, const void *B1, const void *B2, void *C, hipStream_t st) {
  return launch_bits128_t<false>(p, A, B1, B2, C, st);
}
#endif
#if defined(ARTN_TU_B128A) || !defined(ARTN_TU_PART)
hipError_t artn_launch_bits128_acc(const ArtnPlan &p, const void *A, const void *B1, const void *B2, void *C, hipStream_t st) {
  return launch_bits128_t<true>(p, A, B1, B2, C, st);
}
#endif
#elif defined(ARTN_TU_MAIN)
hipError_t artn_launch_bits128(const ArtnPlan &p, const void *A, const void *B1, const void *B2, void *C, hipStream_t st);
hipError_t artn_launch_bits128_acc(const ArtnPlan &p, const void *A, const void *B1, const void *B2, void *C, hipStream_t st);
#endif

// fused pairs of 2^12-element tiles (ArtnBitsPlan::wide8): artn_k_wide<KB1, KB2>, 3..6 contracted bits per stage
#if defined(ARTN_TU_WIDE) || (!defined(ARTN_TU_PART) && !defined(ARTN_TU_MAIN))
hipError_t artn_launch_wide(const ArtnPlan &p, const void *A, const void *B1, const void *B2, void *C, hipStream_t st) {
  dim3 grid(p.info.grid), block(ARTN_WIDE_THREADS);
  const size_t lds = (size_t)p.info.lds_bytes;
  const int k1 = p.bits.st[0].k, k2 = p.bits.st[1].k;
  const float2 *a = (const float2 *)A, *b1 = (const float2 *)B1, *b2 = (const float2 *)B2;
  float2 *c = (float2 *)C;
#define ARTN_WIDE_GO(K1, K2)                                                                        \
  if (k1 == K1 && k2 == K2) {                                                                       \
    auto kern = artn_k_wide<K1, K2>;                                                                \
    if (hipError_t e = ensure_lds<artn_k_wide<K1, K2>>(lds); e != hipSuccess) return e;             \
    hipLaunchKernelGGL(kern, grid, block, lds, st, a, b1, b2, c, p.bits);                           \
    return hipGetLastError();                                                                       \
  }
#define ARTN_WIDE_ROW(K1) ARTN_WIDE_GO(K1, 3) ARTN_WIDE_GO(K1, 4) ARTN_WIDE_GO(K1, 5) ARTN_WIDE_GO(K1, 6)
  ARTN_WIDE_ROW(3) ARTN_WIDE_ROW(4) ARTN_WIDE_ROW(5) ARTN_WIDE_ROW(6)
#undef ARTN_WIDE_ROW
#undef ARTN_WIDE_GO
  return hipErrorInvalidValue;
}
#elif defined(ARTN_TU_MAIN)
hipError_t artn_launch_wide(const ArtnPlan &p, const void *A, const void *B1, const void *B2, void *C, hipStream_t st);
#endif

#define ARTN_CAT2(a, b) a##b
#define ARTN_CAT(a, b) ARTN_CAT2(a, b)
// fused triples (make_bits3): artn_k_bits3<KB1, KB2, KB3, M3>, 3..5 contracted bits per stage, fragments of at most 80 registers;
// M3 exactly when a stage contracts 5 bits
#if defined(ARTN_TU_BITS3) || (defined(ARTN_DEV_BITS3) && !defined(ARTN_TU_PART) && !defined(ARTN_TU_MAIN))
template <int KB1>
static hipError_t launch_bits3_k(const ArtnPlan &p, const float2 *A, const float2 *B1, const float2 *B2, const float2 *B3, float2 *C,
                                 hipStream_t st) {
  dim3 grid(p.info.grid), block(ARTN_WG_THREADS);
  const size_t lds = (size_t)p.info.lds_bytes;
  const int k2 = p.bits.st[1].k, k3 = p.bits.st[2].k;
#define ARTN_B3_GO(K2, K3)                                                                                \
  if (k2 == K2 && k3 == K3) {                                                                             \
    if constexpr ((1 << KB1) + (1 << K2) + (1 << K3) <= 80) {                                             \
      constexpr bool M3V = KB1 == 5 || K2 == 5 || K3 == 5;                                                \
      if ((p.bits.m3 != 0) != M3V) return hipErrorInvalidValue;                                           \
      if (p.bits.nt_loads) {                                                                              \
        auto kern = artn_k_bits3<KB1, K2, K3, M3V, true>;                                                 \
        if (hipError_t e = ensure_lds<artn_k_bits3<KB1, K2, K3, M3V, true>>(lds); e != hipSuccess) return e; \
        hipLaunchKernelGGL(kern, grid, block, lds, st, A, B1, B2, B3, C, p.bits);                         \
      } else {                                                                                            \
        auto kern = artn_k_bits3<KB1, K2, K3, M3V, false>;                                                \
        if (hipError_t e = ensure_lds<artn_k_bits3<KB1, K2, K3, M3V, false>>(lds); e != hipSuccess) return e; \
        hipLaunchKernelGGL(kern, grid, block, lds, st, A, B1, B2, B3, C, p.bits);                         \
      }                                                                                                   \
      return hipGetLastError();                                                                           \
    }                                                                                                     \
  }
  ARTN_B3_GO(3, 3) ARTN_B3_GO(3, 4) ARTN_B3_GO(3, 5)
  ARTN_B3_GO(4, 3) ARTN_B3_GO(4, 4) ARTN_B3_GO(4, 5)
  ARTN_B3_GO(5, 3) ARTN_B3_GO(5, 4) ARTN_B3_GO(5, 5)
#undef ARTN_B3_GO
  return hipErrorInvalidValue;
}
#endif
#if defined(ARTN_TU_BITS3)
hipError_t ARTN_CAT(artn_launch_bits3_k, ARTN_TU_BITS3)(const ArtnPlan &p, const float2 *A, const float2 *B1, const float2 *B2,
                                                       const float2 *B3, float2 *C, hipStream_t st) {
  return launch_bits3_k<ARTN_TU_BITS3>(p, A, B1, B2, B3, C, st);
}
#endif
#if defined(ARTN_TU_B128) || defined(ARTN_TU_BITS3) || defined(ARTN_TU_WIDE)
// (nothing else in this translation unit)
#elif defined(ARTN_TU_BITS) && ARTN_TU_HALF >= 0
hipError_t ARTN_CAT(ARTN_CAT(ARTN_CAT(artn_launch_bits_k, ARTN_TU_BITS), h), ARTN_TU_HALF)(const ArtnPlan &p, const float2 *A, const float2 *B1,
                                                                                        const float2 *B2, float2 *C, hipStream_t st) {
  return launch_bits_k2<ARTN_TU_BITS>(p, A, B1, B2, C, st);
}
#elif defined(ARTN_TU_BITS)
hipError_t ARTN_CAT(artn_launch_bits_k, ARTN_TU_BITS)(const ArtnPlan &p, const float2 *A, const float2 *B1, const float2 *B2, float2 *C,
                                                     hipStream_t st) {
  return launch_bits_k2<ARTN_TU_BITS>(p, A, B1, B2, C, st);
}
#else
#ifdef ARTN_TU_MAIN
hipError_t artn_launch_bits_k1(const ArtnPlan &, const float2 *, const float2 *, const float2 *, float2 *, hipStream_t);
hipError_t artn_launch_bits_k2(const ArtnPlan &, const float2 *, const float2 *, const float2 *, float2 *, hipStream_t);
hipError_t artn_launch_bits_k3(const ArtnPlan &, const float2 *, const float2 *, const float2 *, float2 *, hipStream_t);
hipError_t artn_launch_bits_k4(const ArtnPlan &, const float2 *, const float2 *, const float2 *, float2 *, hipStream_t);
hipError_t artn_launch_bits_k5h0(const ArtnPlan &, const float2 *, const float2 *, const float2 *, float2 *, hipStream_t);
hipError_t artn_launch_bits_k5h1(const ArtnPlan &, const float2 *, const float2 *, const float2 *, float2 *, hipStream_t);
hipError_t artn_launch_bits_k6h0(const ArtnPlan &, const float2 *, const float2 *, const float2 *, float2 *, hipStream_t);
hipError_t artn_launch_bits_k6h1(const ArtnPlan &, const float2 *, const float2 *, const float2 *, float2 *, hipStream_t);
#endif
// Three-step fusion (artn_k_bits3 / artn_contract3, round 4) was built, is parity-green and shortens no committed workload
// (DESIGN.md 4.1c): it is compiled only into development builds (make dev: -DARTN_DEV_BITS3).
#if defined(ARTN_DEV_BITS3)
#ifdef ARTN_TU_MAIN
hipError_t artn_launch_bits3_k3(const ArtnPlan &, const float2 *, const float2 *, const float2 *, const float2 *, float2 *, hipStream_t);
hipError_t artn_launch_bits3_k4(const ArtnPlan &, const float2 *, const float2 *, const float2 *, const float2 *, float2 *, hipStream_t);
hipError_t artn_launch_bits3_k5(const ArtnPlan &, const float2 *, const float2 *, const float2 *, const float2 *, float2 *, hipStream_t);
#endif
static hipError_t launch_bits3(const ArtnPlan &p, const void *A, const void *B1, const void *B2, const void *B3, void *C, hipStream_t st) {
  const float2 *a = (const float2 *)A, *b1 = (const float2 *)B1, *b2 = (const float2 *)B2, *b3 = (const float2 *)B3;
  float2 *c = (float2 *)C;
  switch (p.bits.st[0].k) {
#ifdef ARTN_TU_MAIN
    case 3: return artn_launch_bits3_k3(p, a, b1, b2, b3, c, st);
    case 4: return artn_launch_bits3_k4(p, a, b1, b2, b3, c, st);
    case 5: return artn_launch_bits3_k5(p, a, b1, b2, b3, c, st);
#else
    case 3: return launch_bits3_k<3>(p, a, b1, b2, b3, c, st);
    case 4: return launch_bits3_k<4>(p, a, b1, b2, b3, c, st);
    case 5: return launch_bits3_k<5>(p, a, b1, b2, b3, c, st);
#endif
  }
  return hipErrorInvalidValue;
}
#endif // ARTN_DEV_BITS3

static hipError_t launch_bits(const ArtnPlan &p, const void *A, const void *B1, const void *B2, void *C,
                              hipStream_t st) {
  if (p.bits.c128) return p.bits.accumulate ? artn_launch_bits128_acc(p, A, B1, B2, C, st) : artn_launch_bits128(p, A, B1, B2, C, st);
  if (p.bits.wide8) return artn_launch_wide(p, A, B1, B2, C, st);
  const float2 *a = (const float2 *)A, *b1 = (const float2 *)B1, *b2 = (const float2 *)B2;
  float2 *c = (float2 *)C;
#ifdef ARTN_DEV_FEW // development builds only: one family of artn_k_bits instantiations (compiles in under a minute)
  if (p.bits.st[0].k == ARTN_DEV_FEW) return launch_bits_k2<ARTN_DEV_FEW>(p, a, b1, b2, c, st);
  return hipErrorInvalidValue;
#else
  switch (std::min(p.bits.st[0].k, 6)) {
#ifdef ARTN_TU_MAIN
    case 1: return artn_launch_bits_k1(p, a, b1, b2, c, st);
    case 2: return artn_launch_bits_k2(p, a, b1, b2, c, st);
    case 3: return artn_launch_bits_k3(p, a, b1, b2, c, st);
    case 4: return artn_launch_bits_k4(p, a, b1, b2, c, st);
    case 5: return ((p.bits.n_stages == 2 && p.bits.st[1].k >= 4) ? artn_launch_bits_k5h1 : artn_launch_bits_k5h0)(p, a, b1, b2, c, st);
    case 6: return ((p.bits.n_stages == 2 && p.bits.st[1].k >= 4) ? artn_launch_bits_k6h1 : artn_launch_bits_k6h0)(p, a, b1, b2, c, st);
#else
    case 1: return launch_bits_k2<1>(p, a, b1, b2, c, st);
    case 2: return launch_bits_k2<2>(p, a, b1, b2, c, st);
    case 3: return launch_bits_k2<3>(p, a, b1, b2, c, st);
    case 4: return launch_bits_k2<4>(p, a, b1, b2, c, st);
    case 5: return launch_bits_k2<5>(p, a, b1, b2, c, st);
    case 6: return launch_bits_k2<6>(p, a, b1, b2, c, st);
#endif
  }
  return hipErrorInvalidValue;
#endif
}

static hipError_t launch_pgemm(const ArtnPlan &p, const void *A, const void *B, void *C, void *ws, hipStream_t st) {
  const ArtnPackPlan &g = p.pack;
  const float2 *a = (const float2 *)(g.swapped ? B : A), *b = (const float2 *)(g.swapped ? A : B);
  // 16-byte units of the packed copies: bf16 -- 4 chunk values of one row; fp32 -- one chunk value of a row pair
  const int unit_bits = g.arith == 0 ? g.kc_bits - 2 : g.kc_bits - 1;
  const long a_units = 1L << (g.n_mo + g.n_ko + unit_bits + ARTN_PG_MT), b_units = 1L << (g.n_no + g.n_ko + unit_bits + ARTN_PG_NT);
  unsigned char *Ap = (unsigned char *)ws, *Bp = Ap + a_units * 16;
  auto blocks = [&](long units) { return dim3((unsigned)std::min<long>((units + ARTN_WG_THREADS - 1) / ARTN_WG_THREADS, (long)p.n_cu * 16)); };
  const size_t lds = (size_t)p.info.lds_bytes;
  if (g.arith == 0) {
    hipLaunchKernelGGL(artn_k_pack_bf16, blocks(a_units), dim3(ARTN_WG_THREADS), 0, st, a, (u32x4_t *)Ap, g.a, g.n_ko, a_units);
    hipLaunchKernelGGL(artn_k_pack_bf16, blocks(b_units), dim3(ARTN_WG_THREADS), 0, st, b, (u32x4_t *)Bp, g.b, g.n_ko, b_units);
    if (artn::tuning().pgemm16 >= 2) {
      if (hipError_t e = ensure_lds<artn_k_pgemm<2>>(lds); e != hipSuccess) return e;
      hipLaunchKernelGGL(artn_k_pgemm<2>, dim3(p.info.grid), dim3(ARTN_PG_THREADS), lds, st, Ap, Bp, (float2 *)C, g);
    } else if (artn::tuning().pgemm16 == 1) {
      if (hipError_t e = ensure_lds<artn_k_pgemm<1>>(lds); e != hipSuccess) return e;
      hipLaunchKernelGGL(artn_k_pgemm<1>, dim3(p.info.grid), dim3(ARTN_PG_THREADS), lds, st, Ap, Bp, (float2 *)C, g);
    } else {
      if (hipError_t e = ensure_lds<artn_k_pgemm<0>>(lds); e != hipSuccess) return e;
      hipLaunchKernelGGL(artn_k_pgemm<0>, dim3(p.info.grid), dim3(ARTN_PG_THREADS), lds, st, Ap, Bp, (float2 *)C, g);
    }
  } else {
    hipLaunchKernelGGL(artn_k_pack_f32, blocks(a_units), dim3(ARTN_WG_THREADS), 0, st, a, (f32x4 *)Ap, g.a, g.n_ko, a_units);
    hipLaunchKernelGGL(artn_k_pack_f32, blocks(b_units), dim3(ARTN_WG_THREADS), 0, st, b, (f32x4 *)Bp, g.b, g.n_ko, b_units);
    if (hipError_t e = ensure_lds<artn_k_pgemm3m>(lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(artn_k_pgemm3m, dim3(p.info.grid), dim3(ARTN_PG_THREADS), lds, st, Ap, Bp, (float2 *)C, g);
  }
  return hipGetLastError();
}

static hipError_t launch_xgemm128(const ArtnPlan &p, const void *A, const void *B, void *C, hipStream_t st) {
  const ArtnXGemmPlan &g = p.xg;
  const double2 *a = (const double2 *)(g.swapped ? B : A), *b = (const double2 *)(g.swapped ? A : B);
  dim3 grid(p.info.grid), block(ARTN_WG_THREADS);
  const size_t lds = (size_t)p.info.lds_bytes;
  if (g.kc != ARTN_XG128_KC || g.pc) return hipErrorInvalidValue;
  if (g.nb == 1) {
    if (hipError_t e = ensure_lds<artn_k_xgemm128<1>>(lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(artn_k_xgemm128<1>, grid, block, lds, st, a, b, (double2 *)C, g);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
// chunks of 16 contracted values: one of the six (blocks per tile, operand roles) instantiations
static hipError_t launch_xgemm16(const ArtnXGemmPlan &g, int n_wg, size_t lds, const float2 *a, const float2 *b, float2 *c, hipStream_t st) {
  dim3 grid(n_wg), block(ARTN_WG_THREADS);
#define ARTN_XGEMM_LAUNCH(NBV, TRV)                                                                  \
  {                                                                                                  \
    auto kern = artn_k_xgemm<NBV, TRV>;                                                              \
    if (hipError_t e = ensure_lds<artn_k_xgemm<NBV, TRV>>(lds); e != hipSuccess) return e;           \
    hipLaunchKernelGGL(kern, grid, block, lds, st, a, b, c, g);                                      \
    return hipGetLastError();                                                                        \
  }
  switch (g.nb * 2 + (g.trans ? 1 : 0)) {
    case 2: ARTN_XGEMM_LAUNCH(1, false)
    case 3: ARTN_XGEMM_LAUNCH(1, true)
    case 4: ARTN_XGEMM_LAUNCH(2, false)
    case 5: ARTN_XGEMM_LAUNCH(2, true)
    case 6: ARTN_XGEMM_LAUNCH(3, false)
    case 7: ARTN_XGEMM_LAUNCH(3, true)
  }
#undef ARTN_XGEMM_LAUNCH
  return hipErrorInvalidValue;
}
static hipError_t launch_xgemm(const ArtnPlan &p, const void *A, const void *B, void *C, hipStream_t st) {
  const ArtnXGemmPlan &g = p.xg;
  if (g.c128) return launch_xgemm128(p, A, B, C, st);
  const float2 *a = (const float2 *)(g.swapped ? B : A), *b = (const float2 *)(g.swapped ? A : B);
  float2 *c = (float2 *)C;
  dim3 grid(p.info.grid), block(ARTN_WG_THREADS);
  const size_t lds = (size_t)p.info.lds_bytes;
  if (g.rowmode == 2) { // the row-streaming form with a lane per row (64-row superblocks)
#define ARTN_XROW64_LAUNCH(SV)                                                                       \
  case SV:                                                                                           \
    if (artn_xrow_nbk(g.n.total) == 1) hipLaunchKernelGGL((artn_k_xrow64<SV, 1>), grid, block, lds, st, a, b, c, g); \
    else hipLaunchKernelGGL((artn_k_xrow64<SV, 2>), grid, block, lds, st, a, b, c, g);                 \
    break;
    if (artn_xrow_nbk(g.n.total) > 2) return hipErrorInvalidValue;
    switch (artn_xrow_steps(g.k.total)) {
      ARTN_XROW64_LAUNCH(1) ARTN_XROW64_LAUNCH(2) ARTN_XROW64_LAUNCH(3) ARTN_XROW64_LAUNCH(4)
      ARTN_XROW64_LAUNCH(5) ARTN_XROW64_LAUNCH(6) ARTN_XROW64_LAUNCH(7) ARTN_XROW64_LAUNCH(8)
      default: return hipErrorInvalidValue;
    }
#undef ARTN_XROW64_LAUNCH
    return hipGetLastError();
  }
  if (g.rowmode) { // the row-streaming form: the small operand in registers (1..8 MFMA steps of four contracted values, 1 or 2 column blocks)
#define ARTN_XROW_LAUNCH(SV)                                                                         \
  case SV:                                                                                           \
    switch (artn_xrow_nbk(g.n.total)) {                                                              \
      case 1: hipLaunchKernelGGL((artn_k_xrow<SV, 1>), grid, block, lds, st, a, b, c, g); break;     \
      case 2: hipLaunchKernelGGL((artn_k_xrow<SV, 2>), grid, block, lds, st, a, b, c, g); break;     \
      case 3: hipLaunchKernelGGL((artn_k_xrow<SV, 3>), grid, block, lds, st, a, b, c, g); break;     \
      default: return hipErrorInvalidValue;                                                          \
    }                                                                                                \
    break;
    switch (artn_xrow_steps(g.k.total)) {
      ARTN_XROW_LAUNCH(1) ARTN_XROW_LAUNCH(2) ARTN_XROW_LAUNCH(3) ARTN_XROW_LAUNCH(4)
      ARTN_XROW_LAUNCH(5) ARTN_XROW_LAUNCH(6) ARTN_XROW_LAUNCH(7) ARTN_XROW_LAUNCH(8)
      ARTN_XROW_LAUNCH(9) ARTN_XROW_LAUNCH(10) ARTN_XROW_LAUNCH(11) ARTN_XROW_LAUNCH(12)
      default: return hipErrorInvalidValue;
    }
#undef ARTN_XROW_LAUNCH
    return hipGetLastError();
  }
#ifdef ARTN_DEV_XGPC
  if (g.pc) { // one 8-wave workgroup per CU: four consumer waves (MFMAs, epilogue), four producer waves (tables, loads, LDS fills)
    dim3 pblock(ARTN_XGPC_THREADS);
#define ARTN_XGPC_LAUNCH(NBV, TRV)                                                                   \
  {                                                                                                  \
    auto kern = artn_k_xgemm_pc<NBV, TRV>;                                                           \
    if (hipError_t e = ensure_lds<artn_k_xgemm_pc<NBV, TRV>>(lds); e != hipSuccess) return e;        \
    hipLaunchKernelGGL(kern, grid, pblock, lds, st, a, b, c, g);                                     \
    return hipGetLastError();                                                                        \
  }
    switch (g.nb * 2 + (g.trans ? 1 : 0)) {
      case 2: ARTN_XGPC_LAUNCH(1, false)
      case 3: ARTN_XGPC_LAUNCH(1, true)
      case 4: ARTN_XGPC_LAUNCH(2, false)
      case 5: ARTN_XGPC_LAUNCH(2, true)
      case 6: ARTN_XGPC_LAUNCH(3, false)
      case 7: ARTN_XGPC_LAUNCH(3, true)
    }
#undef ARTN_XGPC_LAUNCH
    return hipErrorInvalidValue;
  }
#else
  if (g.pc) return hipErrorInvalidValue; // (development builds only)
#endif
  if (g.kc == 8) { // few contracted values, one block of columns: chunks of 8, four workgroups per CU
    if (g.nb != 1) return hipErrorInvalidValue;
    if (g.trans) {
      auto kern = artn_k_xgemm<1, true, 8>;
      if (hipError_t e = ensure_lds<artn_k_xgemm<1, true, 8>>(lds); e != hipSuccess) return e;
      hipLaunchKernelGGL(kern, grid, block, lds, st, a, b, c, g);
    } else {
      auto kern = artn_k_xgemm<1, false, 8>;
      if (hipError_t e = ensure_lds<artn_k_xgemm<1, false, 8>>(lds); e != hipSuccess) return e;
      hipLaunchKernelGGL(kern, grid, block, lds, st, a, b, c, g);
    }
    return hipGetLastError();
  }
  if (hipError_t e = launch_xgemm16(g, p.info.grid, lds, a, b, c, st); e != hipSuccess) return e;
  if (g.tail_nb) { // the columns behind the full column tiles: a second launch of the instantiation they need (artn_xg_tail_plan)
    const ArtnXGemmPlan t = artn_xg_tail_plan(g);
    return launch_xgemm16(t, g.tail_grid, (size_t)g.tail_lds, a, b, c, st);
  }
  return hipSuccess;
}

static hipError_t launch_gemm(const ArtnPlan &p, const void *A, const void *B, void *C, hipStream_t st) {
  const ArtnGemmPlan &g = p.gemm;
  const float2 *a = (const float2 *)(g.swapped ? B : A), *b = (const float2 *)(g.swapped ? A : B);
  float2 *c = (float2 *)C;
  dim3 grid(p.info.grid), block(ARTN_WG_THREADS);
  const size_t lds = (size_t)p.info.lds_bytes;
#define ARTN_GEMM_LAUNCH(MBV, NBV)                                                                   \
  {                                                                                                  \
    auto kern = artn_k_gemm<MBV, NBV>;                                                               \
    if (hipError_t e = ensure_lds<artn_k_gemm<MBV, NBV>>(lds); e != hipSuccess) return e;            \
    hipLaunchKernelGGL(kern, grid, block, lds, st, a, b, c, g);                                      \
    return hipGetLastError();                                                                        \
  }
#define ARTN_GEMM_LAUNCH_BF(MBV, NBV)                                                                \
  {                                                                                                  \
    auto kern = artn_k_gemm<MBV, NBV, true>;                                                         \
    if (hipError_t e = ensure_lds<artn_k_gemm<MBV, NBV, true>>(lds); e != hipSuccess) return e;      \
    hipLaunchKernelGGL(kern, grid, block, lds, st, a, b, c, g);                                      \
    return hipGetLastError();                                                                        \
  }
#define ARTN_GEMM_LAUNCH_M3(MBV, NBV)                                                                \
  {                                                                                                  \
    auto kern = artn_k_gemm<MBV, NBV, false, true>;                                                  \
    if (hipError_t e = ensure_lds<artn_k_gemm<MBV, NBV, false, true>>(lds); e != hipSuccess) return e; \
    hipLaunchKernelGGL(kern, grid, block, lds, st, a, b, c, g);                                      \
    return hipGetLastError();                                                                        \
  }
  if (g.split == 2) { // complex128 on v_mfma_f64_16x16x4_f64
    const double2 *a2 = (const double2 *)(g.swapped ? B : A), *b2 = (const double2 *)(g.swapped ? A : B);
    double2 *c2 = (double2 *)C;
#define ARTN_GEMM128_LAUNCH(NBV)                                                                     \
  {                                                                                                  \
    auto kern = artn_k_gemm128<NBV>;                                                                 \
    if (hipError_t e = ensure_lds<artn_k_gemm128<NBV>>(lds); e != hipSuccess) return e;              \
    hipLaunchKernelGGL(kern, grid, block, lds, st, a2, b2, c2, g);                                   \
    return hipGetLastError();                                                                        \
  }
#define ARTN_GEMM128_LAUNCH_G(NBV)                                                                   \
  {                                                                                                  \
    auto kern = artn_k_gemm128<NBV, true>;                                                           \
    if (hipError_t e = ensure_lds<artn_k_gemm128<NBV, true>>(lds); e != hipSuccess) return e;        \
    hipLaunchKernelGGL(kern, grid, block, lds, st, a2, b2, c2, g);                                   \
    return hipGetLastError();                                                                        \
  }
    if (g.gather_dim >= 0) { // row gather (artn_contract_gather) in complex128
      switch (g.nb_log2) {
        case 0: ARTN_GEMM128_LAUNCH_G(1)
        case 1: ARTN_GEMM128_LAUNCH_G(2)
        case 2: ARTN_GEMM128_LAUNCH_G(4)
      }
      return hipErrorInvalidValue;
    }
    switch (g.nb_log2) {
      case 0: ARTN_GEMM128_LAUNCH(1)
      case 1: ARTN_GEMM128_LAUNCH(2)
      case 2: ARTN_GEMM128_LAUNCH(4)
    }
#undef ARTN_GEMM128_LAUNCH_G
#undef ARTN_GEMM128_LAUNCH
    return hipErrorInvalidValue;
  }
  const int key = g.mb_log2 * 4 + g.nb_log2;
  if (g.gather_dim >= 0) { // row gather (artn_contract_gather): fp32, chunks of 2^4
    if (g.split || g.kc != ARTN_GEMM_KC || g.pitch_log2 != ARTN_GEMM_PITCH_LOG2) return hipErrorInvalidValue;
#define ARTN_GEMM_LAUNCH_G(MBV, NBV, M3V)                                                            \
  {                                                                                                  \
    auto kern = artn_k_gemm<MBV, NBV, false, M3V, false, true>;                                      \
    if (hipError_t e = ensure_lds<artn_k_gemm<MBV, NBV, false, M3V, false, true>>(lds); e != hipSuccess) return e; \
    hipLaunchKernelGGL(kern, grid, block, lds, st, a, b, c, g);                                      \
    return hipGetLastError();                                                                        \
  }
#define ARTN_GEMM_LAUNCH_DEEP_G(MBV, NBIV)                                                           \
  {                                                                                                  \
    auto kern = artn_k_gemm_deep<MBV, 1, 4, NBIV, true>;                                             \
    if (hipError_t e = ensure_lds<artn_k_gemm_deep<MBV, 1, 4, NBIV, true>>(lds); e != hipSuccess) return e; \
    hipLaunchKernelGGL(kern, grid, block, lds, st, a, b, c, g);                                      \
    return hipGetLastError();                                                                        \
  }
    if (g.m3) {
      if (g.nb_log2 == 0 && g.wk_log2 == 0 && g.n_ko >= 1 && g.n_ko <= 8 && g.ta_bits == 11 && artn::tuning().gemm_deep) {
        if (g.mb_log2 == 0 && g.tb_bits == 9) ARTN_GEMM_LAUNCH_DEEP_G(1, 1)
        if (g.mb_log2 == 1 && g.tb_bits == 10 && artn::tuning().gemm_deep >= 2) ARTN_GEMM_LAUNCH_DEEP_G(2, 2)
      }
      switch (key) {
        case 0: ARTN_GEMM_LAUNCH_G(1, 1, true)
        case 1: ARTN_GEMM_LAUNCH_G(1, 2, true)
        case 4: ARTN_GEMM_LAUNCH_G(2, 1, true)
      }
      return hipErrorInvalidValue;
    }
    if (key == 0 && g.wk_log2 == 0 && g.n_ko >= 1 && g.n_ko <= 8 && g.ta_bits == 11 && g.tb_bits <= 9 && artn::tuning().gemm_deep) {
      // 4M, 16 columns or fewer: the chunk steps of the sparse executor
      auto kern = artn_k_gemm_deep<1, 1, 4, 1, true, false>;
      if (hipError_t e = ensure_lds<artn_k_gemm_deep<1, 1, 4, 1, true, false>>(lds); e != hipSuccess) return e;
      hipLaunchKernelGGL(kern, grid, block, lds, st, a, b, c, g);
      return hipGetLastError();
    }
    switch (key) {
      case 0: ARTN_GEMM_LAUNCH_G(1, 1, false)
      case 1: ARTN_GEMM_LAUNCH_G(1, 2, false)
      case 2: ARTN_GEMM_LAUNCH_G(1, 4, false)
      case 5: ARTN_GEMM_LAUNCH_G(2, 2, false)
      case 6: ARTN_GEMM_LAUNCH_G(2, 4, false)
    }
#undef ARTN_GEMM_LAUNCH_G
#undef ARTN_GEMM_LAUNCH_DEEP_G
    return hipErrorInvalidValue;
  }
  if (g.kc == ARTN_GEMM_KC_TALL && !g.split) { // 32 x 32 tiles, chunks of 2^6 contracted values
    if (key != 0) return hipErrorInvalidValue;
    if (g.m3) {
      auto kern = artn_k_gemm<1, 1, false, true, true>;
      if (hipError_t e = ensure_lds<artn_k_gemm<1, 1, false, true, true>>(lds); e != hipSuccess) return e;
      hipLaunchKernelGGL(kern, grid, block, lds, st, a, b, c, g);
    } else {
      auto kern = artn_k_gemm<1, 1, false, false, true>;
      if (hipError_t e = ensure_lds<artn_k_gemm<1, 1, false, false, true>>(lds); e != hipSuccess) return e;
      hipLaunchKernelGGL(kern, grid, block, lds, st, a, b, c, g);
    }
    return hipGetLastError();
  }
  // memory-bound 3M steps (few free bits in the second operand: one block column per wave): operand loads two chunks ahead
  if (g.m3 && g.nb_log2 == 0 && g.wk_log2 == 0 && g.n_ko >= 1 && g.n_ko <= 8 && g.ta_bits == 11 &&
      g.pitch_log2 == ARTN_GEMM_PITCH_LOG2 && g.kc == ARTN_GEMM_KC && artn::tuning().gemm_deep) {
#define ARTN_GEMM_LAUNCH_DEEP(MBV, NBIV)                                                             \
  {                                                                                                  \
    auto kern = artn_k_gemm_deep<MBV, 1, 4, NBIV>;                                                   \
    if (hipError_t e = ensure_lds<artn_k_gemm_deep<MBV, 1, 4, NBIV>>(lds); e != hipSuccess) return e; \
    hipLaunchKernelGGL(kern, grid, block, lds, st, a, b, c, g);                                      \
    return hipGetLastError();                                                                        \
  }
    if (g.mb_log2 == 0 && g.tb_bits == 9) ARTN_GEMM_LAUNCH_DEEP(1, 1)
    if (g.mb_log2 == 1 && g.tb_bits == 10 && artn::tuning().gemm_deep >= 2) ARTN_GEMM_LAUNCH_DEEP(2, 2)
#undef ARTN_GEMM_LAUNCH_DEEP
  }
  if (g.m3) {
    switch (key) {
      case 0: ARTN_GEMM_LAUNCH_M3(1, 1)
      case 1: ARTN_GEMM_LAUNCH_M3(1, 2)
      case 4: ARTN_GEMM_LAUNCH_M3(2, 1)
    }
    return hipErrorInvalidValue;
  }
  if (g.split) {
    switch (key) {
      case 0: ARTN_GEMM_LAUNCH_BF(1, 1)
      case 1: ARTN_GEMM_LAUNCH_BF(1, 2)
      case 5: ARTN_GEMM_LAUNCH_BF(2, 2)
    }
    return hipErrorInvalidValue;
  }
  if (key == 0 && g.wk_log2 == 0 && g.n_ko >= 1 && g.n_ko <= 8 && g.ta_bits == 11 && g.tb_bits <= 9 &&
      g.pitch_log2 == ARTN_GEMM_PITCH_LOG2 && g.kc == ARTN_GEMM_KC && artn::tuning().gemm_deep) {
    // 4M, memory-bound: operand loads two chunks ahead
    auto kern = artn_k_gemm_deep<1, 1, 4, 1, false, false>;
    if (hipError_t e = ensure_lds<artn_k_gemm_deep<1, 1, 4, 1, false, false>>(lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, grid, block, lds, st, a, b, c, g);
    return hipGetLastError();
  }
  switch (key) {
    case 0: ARTN_GEMM_LAUNCH(1, 1)
    case 1: ARTN_GEMM_LAUNCH(1, 2)
    case 2: ARTN_GEMM_LAUNCH(1, 4)
    case 5: ARTN_GEMM_LAUNCH(2, 2)
    case 6: ARTN_GEMM_LAUNCH(2, 4)
  }
#undef ARTN_GEMM_LAUNCH
#undef ARTN_GEMM_LAUNCH_BF
#undef ARTN_GEMM_LAUNCH_M3
  return hipErrorInvalidValue;
}

// out[g][c] = sum_r in[g][r][c] over float4 columns (two complex64 each): the sum-out that closes a
// split-K contraction.  256 threads = 64 columns x 4 row lanes; the lanes' partial sums are added in
// a fixed order, so results do not depend on the launch shape.
template <typename V>
__global__ __launch_bounds__(ARTN_WG_THREADS) void artn_k_sum_axis(const V *__restrict__ in, V *__restrict__ out,
                                                                  long n_rows, long n_cols4, long col_tiles) {
  __shared__ V part[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const long g = blockIdx.x / col_tiles, ct = blockIdx.x % col_tiles;
  const long c = ct * 64 + tx;
  V acc = V(0);
  if (c < n_cols4) {
    const V *p = in + (g * n_rows) * n_cols4 + c;
    long r = ty;
    for (; r + 12 < n_rows; r += 16) { // four independent loads in flight per thread
      const V v0 = p[r * n_cols4], v1 = p[(r + 4) * n_cols4], v2 = p[(r + 8) * n_cols4], v3 = p[(r + 12) * n_cols4];
      acc += (v0 + v1) + (v2 + v3);
    }
    for (; r < n_rows; r += 4) acc += p[r * n_cols4];
  }
  part[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && c < n_cols4) {
    V t = part[0][tx];
#pragma unroll
    for (int q = 1; q < 4; ++q) t += part[q][tx];
    out[g * n_cols4 + c] = t;
  }
}
typedef double f64x2 __attribute__((ext_vector_type(2)));
template <typename V>
__global__ __launch_bounds__(ARTN_WG_THREADS) void artn_k_axpy_v(V *__restrict__ acc, const V *__restrict__ x, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) acc[i] += x[i];
}

// ----------------------------------------------------------------------------------------
// Matrix-pipe rate probe (measurement aid of bench.py: the peak a complex128 roofline is priced against is the
// f64 MFMA rate measured on the device it runs on -- the guide tabulates no f64 figure).  Every wave issues
// independent accumulator chains back to back; nothing touches memory inside the loop.
// ----------------------------------------------------------------------------------------
typedef double f64x4_t __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ __launch_bounds__(ARTN_WG_THREADS) void artn_k_mfma_probe(float *sink, int iters) {
  const int lane = threadIdx.x & 63;
  if constexpr (KIND == 2) { // v_mfma_f64_16x16x4_f64: 2048 FLOP
    f64x4_t acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = f64x4_t{0.0, 0.0, 0.0, 0.0};
    const double a = 1.0 + lane * 1e-3, b = 1.0 - lane * 1e-3;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
    double t = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) t += acc[q][0] + acc[q][3];
    if (t == 12345.678) sink[0] = (float)t;
  } else if constexpr (KIND == 0) { // v_mfma_f32_32x32x2_f32: 4096 FLOP
    f32x16 acc[2];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;
    const float a = 1.f + lane * 1e-3f, b = 1.f - lane * 1e-3f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int q = 0; q < 2; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
    float t = acc[0][0] + acc[1][5];
    if (t == 12345.678f) sink[0] = t;
  } else { // v_mfma_f32_32x32x16_bf16: 32768 FLOP
    f32x16 acc[2];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;
    u32x4_t a = {0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = {0x3f803f80u, 0x3f803f80u - lane, 0x3f803f80u, 0x3f803f80u};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int q = 0; q < 2; ++q)
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc[q], 0, 0, 0);
    float t = acc[0][0] + acc[1][5];
    if (t == 12345.678f) sink[0] = t;
  }
}

extern "C" {

int artn_abi_version(void) { return ARTN_ABI_VERSION; }
const char *artn_last_error(void) { return g_err.c_str(); }

int artn_device_count(void) {
  std::call_once(g_once, probe_devices);
  return g_ndev;
}

int artn_contract_query(const ArtnStepDesc *d, ArtnStepInfo *info) {
  if (!info) return fail(ARTN_E_INVALID, "null info");
  ArtnPlan p;
  std::string err;
  const bool no_bits = env_flag("ARTN_FORCE_GENERIC");
  const int64_t min_tiles = env_flag("ARTN_FORCE_BITS") ? 1 : 32;
  int rc = artn::make_plan(d, p, err, g_ncu, !no_bits, min_tiles, -1, true, true);
  if (rc) return fail(rc, err);
  *info = p.info;
  g_note = p.kernel == ARTN_KERNEL_GENERIC ? p.why_generic : std::string();
  return ARTN_OK;
}

const char *artn_last_plan_note(void) { return g_note.c_str(); }

int artn_contract(const ArtnStepDesc *d, const void *A, const void *B, void *C, void *stream) {
  if (artn_device_count() < 1) return fail(ARTN_E_NODEVICE, "no gfx950 device visible");
  if (!A || !B || !C) return fail(ARTN_E_INVALID, "null operand pointer");
  ArtnPlan p;
  std::string err;
  const bool aligned = (((uintptr_t)A | (uintptr_t)C) & 15) == 0;
  const bool no_bits = env_flag("ARTN_FORCE_GENERIC") || !aligned;
  const int64_t min_tiles = env_flag("ARTN_FORCE_BITS") ? 1 : 32;
  int rc = artn::make_plan(d, p, err, g_ncu, !no_bits, min_tiles);
  if (rc) return fail(rc, err);
  if (p.kernel == ARTN_KERNEL_GEMM_MFMA && (((uintptr_t)B) & 15) != 0) { // the GEMM kernel moves both operands in 16-byte lanes
    rc = artn::make_plan(d, p, err, g_ncu, !no_bits, min_tiles, -1, false);
    if (rc) return fail(rc, err);
  }
  hipStream_t st = (hipStream_t)stream;
  if (p.kernel == ARTN_KERNEL_BITS_MFMA) {
    HIP_TRY(launch_bits(p, A, B, nullptr, C, st));
    return ARTN_OK;
  }
  if (p.kernel == ARTN_KERNEL_GEMM_MFMA) {
    HIP_TRY(launch_gemm(p, A, B, C, st));
    return ARTN_OK;
  }
  if (p.kernel == ARTN_KERNEL_XGEMM) {
    HIP_TRY(launch_xgemm(p, A, B, C, st));
    return ARTN_OK;
  }
  dim3 grid(p.info.grid), block(ARTN_WG_THREADS);
  if (p.gen.out_numel == 0) return ARTN_OK;
  if (p.gen.out_numel <= 4096 && p.gen.red_numel >= 64) { // few results of sums: a workgroup per result
    dim3 rgrid((unsigned)p.gen.out_numel);
    if (d->dtype != ARTN_C128)
      hipLaunchKernelGGL((artn_k_generic_red<float2, float>), rgrid, block, 0, st, (const float2 *)A, (const float2 *)B, (float2 *)C, p.gen);
    else
      hipLaunchKernelGGL((artn_k_generic_red<double2, double>), rgrid, block, 0, st, (const double2 *)A, (const double2 *)B, (double2 *)C, p.gen);
    HIP_TRY(hipGetLastError());
    return ARTN_OK;
  }
  if (d->dtype != ARTN_C128) // (small steps of the reduced-precision mode run in fp32: they are launch-bound)
    hipLaunchKernelGGL((artn_k_generic<float2, float>), grid, block, 0, st, (const float2 *)A,
                       (const float2 *)B, (float2 *)C, p.gen);
  else
    hipLaunchKernelGGL((artn_k_generic<double2, double>), grid, block, 0, st, (const double2 *)A,
                       (const double2 *)B, (double2 *)C, p.gen);
  HIP_TRY(hipGetLastError());
  return ARTN_OK;
}

int artn_contract_ws(const ArtnStepDesc *d, const void *A, const void *B, void *C, void *ws, int64_t ws_bytes, void *stream) {
  if (artn_device_count() < 1) return fail(ARTN_E_NODEVICE, "no gfx950 device visible");
  if (!A || !B || !C) return fail(ARTN_E_INVALID, "null operand pointer");
  if (ws && ws_bytes > 0 && (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)ws) & 15) == 0 && !env_flag("ARTN_FORCE_GENERIC")) {
    ArtnPlan p;
    std::string err;
    const int64_t min_tiles = env_flag("ARTN_FORCE_BITS") ? 1 : 32; // (the same switches as artn_contract / artn_contract_query)
    int rc = artn::make_plan(d, p, err, g_ncu, true, min_tiles, -1, true, true);
    if (rc) return fail(rc, err);
    if (p.kernel == ARTN_KERNEL_PGEMM && p.info.workspace_bytes <= ws_bytes) {
      HIP_TRY(launch_pgemm(p, A, B, C, ws, (hipStream_t)stream));
      return ARTN_OK;
    }
  }
  return artn_contract(d, A, B, C, stream);
}

int artn_contract_gather(const ArtnStepDesc *d, const void *A, const void *B, void *C, int label,
                         const int64_t *rows_a, int64_t src_rows_a, const int64_t *rows_b, int64_t src_rows_b,
                         int32_t *err_flag, void *stream) {
  if (artn_device_count() < 1) return fail(ARTN_E_NODEVICE, "no gfx950 device visible");
  if (!d || !A || !B || !C) return fail(ARTN_E_INVALID, "null pointer");
  if (label < 0 || label >= d->n_labels) return fail(ARTN_E_INVALID, "gather label out of range");
  if (d->stride_c[label] < 0) return fail(ARTN_E_INVALID, "the gathered label must be an output label");
  if ((rows_a && (d->stride_a[label] < 0 || src_rows_a < 1)) || (rows_b && (d->stride_b[label] < 0 || src_rows_b < 1)))
    return fail(ARTN_E_INVALID, "row indices for an operand that does not carry the label");
  // the tiled kernel moves 16-byte lanes (artn_contract falls back to the strided kernel instead;
  // there is no strided gather, so the caller gathers explicitly)
  if ((((uintptr_t)A | (uintptr_t)C) & 15) != 0) return fail(ARTN_E_UNSUPPORTED, "row gather needs 16-byte aligned operands");
  ArtnPlan p;
  std::string err;
  int rc = artn::make_plan(d, p, err, g_ncu, true, 1, label);
  if (rc) return fail(rc, err);
  if (p.kernel == ARTN_KERNEL_GEMM_MFMA) { // (the kernel's first operand is the caller's B when the plan swapped them)
    if ((((uintptr_t)B) & 15) != 0) return fail(ARTN_E_UNSUPPORTED, "row gather needs 16-byte aligned operands");
    const bool sw = p.gemm.swapped != 0;
    p.gemm.rows_a = sw ? rows_b : rows_a;
    p.gemm.rows_b = sw ? rows_a : rows_b;
    p.gemm.src_rows_a = sw ? src_rows_b : src_rows_a;
    p.gemm.src_rows_b = sw ? src_rows_a : src_rows_b;
    p.gemm.gather_err = err_flag;
    if (hipError_t e = launch_gemm(p, A, B, C, (hipStream_t)stream); e != hipSuccess) {
      const ArtnGemmPlan &g = p.gemm;
      return fail(ARTN_E_LAUNCH, std::string("row gather on the GEMM kernel (tile 2^") + std::to_string(g.mt) + " x 2^" + std::to_string(g.nt) +
                                  ", chunk 2^" + std::to_string(g.kc) + ", blocks " + std::to_string(1 << g.mb_log2) + " x " +
                                  std::to_string(1 << g.nb_log2) + (g.m3 ? ", 3M" : "") + ", lds " + std::to_string(p.info.lds_bytes) +
                                  ", grid " + std::to_string(p.info.grid) + "): " + hipGetErrorString(e));
    }
    return ARTN_OK;
  }
  p.bits.rows_a = rows_a;
  p.bits.rows_b = rows_b;
  p.bits.src_rows_a = src_rows_a;
  p.bits.src_rows_b = src_rows_b;
  p.bits.gather_err = err_flag;
  HIP_TRY(launch_bits(p, A, B, nullptr, C, (hipStream_t)stream));
  return ARTN_OK;
}

int artn_contract2_query(const ArtnStepDesc *d1, const ArtnStepDesc *d2, ArtnStepInfo *info) {
  if (!info) return fail(ARTN_E_INVALID, "null info");
  ArtnPlan p;
  std::string err;
  if (env_flag("ARTN_NO_FUSE")) return fail(ARTN_E_UNSUPPORTED, "not fusable: ARTN_NO_FUSE is set");
  const int64_t min_tiles = env_flag("ARTN_FORCE_BITS") ? 1 : 32;
  int rc = artn::make_plan_fused(d1, d2, p, err, g_ncu, min_tiles);
  if (rc) return fail(rc, err);
  *info = p.info;
  return ARTN_OK;
}

int artn_contract2(const ArtnStepDesc *d1, const ArtnStepDesc *d2, const void *A, const void *B1, const void *B2,
                   void *C, void *stream) {
  if (artn_device_count() < 1) return fail(ARTN_E_NODEVICE, "no gfx950 device visible");
  if (!A || !B1 || !B2 || !C) return fail(ARTN_E_INVALID, "null operand pointer");
  if ((((uintptr_t)A | (uintptr_t)C) & 15) != 0) return fail(ARTN_E_UNSUPPORTED, "not fusable: operands not 16-byte aligned");
  if (env_flag("ARTN_NO_FUSE")) return fail(ARTN_E_UNSUPPORTED, "not fusable: ARTN_NO_FUSE is set");
  ArtnPlan p;
  std::string err;
  const int64_t min_tiles = env_flag("ARTN_FORCE_BITS") ? 1 : 32;
  int rc = artn::make_plan_fused(d1, d2, p, err, g_ncu, min_tiles);
  if (rc) return fail(rc, err);
  HIP_TRY(launch_bits(p, A, B1, B2, C, (hipStream_t)stream));
  return ARTN_OK;
}

int artn_contract2_acc(const ArtnStepDesc *d1, const ArtnStepDesc *d2, const void *A, const void *B1, const void *B2,
                       void *C, void *stream) {
  if (artn_device_count() < 1) return fail(ARTN_E_NODEVICE, "no gfx950 device visible");
  if (!A || !B1 || !B2 || !C) return fail(ARTN_E_INVALID, "null operand pointer");
  if ((((uintptr_t)A | (uintptr_t)C) & 15) != 0) return fail(ARTN_E_UNSUPPORTED, "not fusable: operands not 16-byte aligned");
  if (env_flag("ARTN_NO_FUSE") || env_flag("ARTN_NO_ACC")) return fail(ARTN_E_UNSUPPORTED, "not fusable: ARTN_NO_FUSE / ARTN_NO_ACC is set");
  ArtnPlan p;
  std::string err;
  const int64_t min_tiles = env_flag("ARTN_FORCE_BITS") ? 1 : 32;
  int rc = artn::make_plan_fused(d1, d2, p, err, g_ncu, min_tiles);
  if (rc) return fail(rc, err);
  if (p.bits.wide8) return fail(ARTN_E_UNSUPPORTED, "accumulate: not in artn_k_wide");
  if (!artn::bits_can_accumulate(p)) return fail(ARTN_E_UNSUPPORTED, "accumulate: this pair's store phase cannot add");
  p.bits.accumulate = 1;
  HIP_TRY(launch_bits(p, A, B1, B2, C, (hipStream_t)stream));
  return ARTN_OK;
}

int artn_contract_acc(const ArtnStepDesc *d, const void *A, const void *B, void *C, void *stream) {
  if (artn_device_count() < 1) return fail(ARTN_E_NODEVICE, "no gfx950 device visible");
  if (!A || !B || !C) return fail(ARTN_E_INVALID, "null operand pointer");
  if (env_flag("ARTN_NO_ACC")) return fail(ARTN_E_UNSUPPORTED, "accumulate: ARTN_NO_ACC is set");
  ArtnPlan p;
  std::string err;
  const bool aligned = (((uintptr_t)A | (uintptr_t)C) & 15) == 0;
  const int64_t min_tiles = env_flag("ARTN_FORCE_BITS") ? 1 : 32;
  int rc = artn::make_plan(d, p, err, g_ncu, aligned && !env_flag("ARTN_FORCE_GENERIC"), min_tiles);
  if (rc) return fail(rc, err);
  if (!artn::bits_can_accumulate(p)) return fail(ARTN_E_UNSUPPORTED, "accumulate: this step's kernel cannot add in its store phase");
  p.bits.accumulate = 1;
  HIP_TRY(launch_bits(p, A, B, nullptr, C, (hipStream_t)stream));
  return ARTN_OK;
}

#if defined(ARTN_DEV_BITS3)
int artn_contract3_query(const ArtnStepDesc *d1, const ArtnStepDesc *d2, const ArtnStepDesc *d3, ArtnStepInfo *info) {
  if (!info || !d1 || !d2 || !d3) return fail(ARTN_E_INVALID, "null argument");
  if (env_flag("ARTN_NO_FUSE")) return fail(ARTN_E_UNSUPPORTED, "not fusable: ARTN_NO_FUSE is set");
  ArtnPlan p;
  std::string err;
  int rc = artn::make_plan_fused3(d1, d2, d3, p, err, g_ncu);
  if (rc) return fail(rc, err);
  *info = p.info;
  return ARTN_OK;
}

int artn_contract3(const ArtnStepDesc *d1, const ArtnStepDesc *d2, const ArtnStepDesc *d3, const void *A, const void *B1,
                   const void *B2, const void *B3, void *C, void *stream) {
  if (artn_device_count() < 1) return fail(ARTN_E_NODEVICE, "no gfx950 device visible");
  if (!d1 || !d2 || !d3 || !A || !B1 || !B2 || !B3 || !C) return fail(ARTN_E_INVALID, "null pointer");
  if ((((uintptr_t)A | (uintptr_t)C) & 15) != 0) return fail(ARTN_E_UNSUPPORTED, "not fusable: operands not 16-byte aligned");
  if (env_flag("ARTN_NO_FUSE")) return fail(ARTN_E_UNSUPPORTED, "not fusable: ARTN_NO_FUSE is set");
  ArtnPlan p;
  std::string err;
  int rc = artn::make_plan_fused3(d1, d2, d3, p, err, g_ncu);
  if (rc) return fail(rc, err);
  HIP_TRY(launch_bits3(p, A, B1, B2, B3, C, (hipStream_t)stream));
  return ARTN_OK;
}
#endif // ARTN_DEV_BITS3

#if defined(ARTN_STAMPS) || defined(ARTN_PHASES)
// diagnostic builds only
int artn_debug_read_phases(unsigned long long *host) {
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(host, HIP_SYMBOL(artn_phase_buf), sizeof(unsigned long long) * 1024 * 20));
  return ARTN_OK;
}
#endif
#ifdef ARTN_STAMPS
int artn_debug_read_stamps(unsigned long long *host, int n_waves) {
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(host, HIP_SYMBOL(artn_stamp_buf), sizeof(unsigned long long) * ARTN_N_STAMPS * n_waves));
  return ARTN_OK;
}
#endif

int64_t artn_program_record_bytes(void) { return (int64_t)sizeof(ArtnProgStep); }

static int64_t prog_image_layout(int32_t n_steps, int32_t n_groups, int64_t n_levels, int64_t n_wtasks, int64_t n_terms, ArtnProgHeader *h) {
  int64_t off = sizeof(ArtnProgHeader);
  auto take = [&](int64_t bytes) { const int64_t o = off; off += (bytes + 15) / 16 * 16; return o; };
  const int64_t g = take((int64_t)n_groups * sizeof(ArtnProgGroup)), l = take(n_levels * sizeof(ArtnProgLevel));
  const int64_t w = take(n_wtasks * sizeof(ArtnProgWTask)), r = take((int64_t)n_steps * sizeof(ArtnProgStep));
  const int64_t t = take(n_terms * 8 + 64); // (+64: the 8-entry table loads of the last step may run past its end)
  if (h) { h->off_groups = g; h->off_levels = l; h->off_wtasks = w; h->off_records = r; h->off_tables = t; }
  return off;
}

int64_t artn_program_image_bytes(int32_t n_steps, const ArtnStepDesc *const *descs, int32_t n_groups) {
  if (n_steps < 0 || n_groups < 0 || (n_steps && !descs)) return -1;
  int64_t wt = 0, terms = 0;
  for (int s = 0; s < n_steps; ++s) {
    double f, na, nb, nc;
    artn::step_cost(descs[s], f, na, nb, nc);
    // wave tasks per step as artn_program_build cuts them: ceil(out / 128) for a general step, 2^(m bits - 5) *
    // max(1, 2^(n bits - 4)) blocks for a matrix-core step -- out / 32 when the second operand is fully contracted
    wt += (int64_t)((nc + 31.0) / 32.0) + 1;
    ArtnPlan p;
    std::string err;
    if (artn::validate(descs[s], err) || !artn::make_generic(descs[s], p, err)) return -1;
    terms += p.gen.red_numel;
  }
  return prog_image_layout(n_steps, n_groups, n_steps, wt, terms, nullptr); // (at most one level per step)
}

int artn_program_build(int32_t n_steps, const ArtnStepDesc *const *descs, const int64_t *loc_a, const int64_t *loc_b,
                       const int64_t *loc_c, const uint8_t *keep, int32_t n_groups, const int32_t *group_start,
                       void *host_image, int64_t image_bytes) {
  if (n_steps < 0 || n_groups < 0 || !descs || !loc_a || !loc_b || !loc_c || !group_start || !host_image)
    return fail(ARTN_E_INVALID, "null argument");
  if (group_start[0] != 0 || group_start[n_groups] != n_steps) return fail(ARTN_E_INVALID, "group_start must cover the steps");
  std::vector<ArtnProgStep> rec(n_steps);
  const bool c128 = n_steps > 0 && descs[0]->dtype == ARTN_C128;
  const int64_t esz = c128 ? 16 : 8; // bytes per element in the LDS arena (workspace offsets are the caller's)
  for (int s = 0; s < n_steps; ++s) {
    const ArtnStepDesc *d = descs[s];
    std::string err;
    int rc = artn::validate(d, err);
    if (rc) return fail(rc, err);
    if ((d->dtype == ARTN_C128) != (descs[0]->dtype == ARTN_C128)) return fail(ARTN_E_INVALID, "the steps of a program share one element type");
    ArtnPlan p;
    if (!artn::make_generic(d, p, err)) return fail(ARTN_E_UNSUPPORTED, err);
    const ArtnGenericPlan &g = p.gen;
    if (g.n_out > ARTN_PROG_MAX_OUT || g.n_red > ARTN_PROG_MAX_RED || g.red_numel > ARTN_PROG_MAX_REDN ||
        g.out_numel >= (1LL << 30) || loc_c[s] < 0)
      return fail(ARTN_E_UNSUPPORTED, "step does not fit a small-step record");
    ArtnProgStep &r = rec[s];
    memset(&r, 0, sizeof(r));
    r.n_out = g.n_out; r.n_red = g.n_red; r.out_numel = (int32_t)g.out_numel; r.red_numel = (int32_t)g.red_numel;
    r.loc_a = loc_a[s]; r.loc_b = loc_b[s]; r.loc_c = loc_c[s];
    r.lds_a = r.lds_b = r.lds_c = -1;
    {
      double f, na, nb, nc;
      artn::step_cost(d, f, na, nb, nc);
      if (na >= (double)(1 << 30) || nb >= (double)(1 << 30)) return fail(ARTN_E_UNSUPPORTED, "operand too large for a small-step record");
      r.a_numel = (int32_t)na; r.b_numel = (int32_t)nb;
      // operands are addressed as dense arrays of that many elements
      for (int which = 0; which < 2; ++which) {
        std::vector<std::pair<int64_t, int64_t>> v;
        for (int l = 0; l < d->n_labels; ++l) {
          const int64_t st = which ? d->stride_b[l] : d->stride_a[l];
          if (st >= 0 && d->extent[l] > 1) v.push_back({st, d->extent[l]});
        }
        std::sort(v.begin(), v.end());
        int64_t expect = 1;
        for (auto &pr : v) {
          if (pr.first != expect) return fail(ARTN_E_UNSUPPORTED, "small-step programs take dense operands");
          expect *= pr.second;
        }
      }
    }
    {
      // the generic plan lists the output axes fastest-in-C first: C strides are the running product; then the axes
      // are put in first-operand order (see ArtnProgStep)
      std::vector<int> ax(g.n_out);
      std::vector<int64_t> sc(g.n_out);
      int64_t run = 1;
      for (int i = 0; i < g.n_out; ++i) {
        if (g.out_sA[i] >= (1LL << 30) || g.out_sB[i] >= (1LL << 30)) return fail(ARTN_E_UNSUPPORTED, "stride too large for a small-step record");
        ax[i] = i; sc[i] = run; run *= g.out_ext[i];
      }
      std::stable_sort(ax.begin(), ax.end(), [&](int x, int y) {
        const bool nx = g.out_sA[x] == 0, ny = g.out_sA[y] == 0;
        return nx != ny ? ny : (!nx && g.out_sA[x] < g.out_sA[y]);
      });
      for (int q = 0; q < g.n_out; ++q) {
        const int i = ax[q];
        r.out_ext[q] = (int32_t)g.out_ext[i]; r.out_sA[q] = (int32_t)g.out_sA[i]; r.out_sB[q] = (int32_t)g.out_sB[i]; r.out_sC[q] = (int32_t)sc[i];
        r.out_lg[q] = artn::ilog2_exact(g.out_ext[i]);
      }
    }
    for (int i = 0; i < g.n_red; ++i) {
      if (g.red_sA[i] >= (1LL << 30) || g.red_sB[i] >= (1LL << 30)) return fail(ARTN_E_UNSUPPORTED, "stride too large for a small-step record");
      r.red_ext[i] = (int32_t)g.red_ext[i]; r.red_sA[i] = (int32_t)g.red_sA[i]; r.red_sB[i] = (int32_t)g.red_sB[i];
      r.red_lg[i] = artn::ilog2_exact(g.red_ext[i]);
    }
  }
  // ---- dependencies (a workspace offset names one result), levels, order by level inside each group
  std::map<int64_t, int> producer; // workspace offset -> step
  std::vector<int> level(n_steps, 1), last_use(n_steps, 0), group_of(n_steps, 0);
  for (int g = 0; g < n_groups; ++g)
    for (int s = group_start[g]; s < group_start[g + 1]; ++s) group_of[s] = g;
  for (int s = 0; s < n_steps; ++s) {
    for (int64_t loc : {loc_a[s], loc_b[s]}) {
      if (loc < 0) continue;
      auto it = producer.find(loc);
      if (it == producer.end()) return fail(ARTN_E_INVALID, "a step reads a workspace offset no earlier step wrote");
      if (group_of[it->second] != group_of[s]) return fail(ARTN_E_INVALID, "steps of different groups must be independent");
      level[s] = std::max(level[s], level[it->second] + 1);
    }
    if (producer.count(loc_c[s])) return fail(ARTN_E_INVALID, "two steps write the same workspace offset");
    producer[loc_c[s]] = s;
  }
  // fast steps (prog_mfma_task): every extent a power of two, no output axis in both operands, 5+ output bits in the
  // first operand, 2+ contracted values; a stride per output bit
  for (int s = 0; s < n_steps; ++s) {
    ArtnProgStep &r = rec[s];
    bool ok = r.red_numel >= 2 && (r.red_numel & (r.red_numel - 1)) == 0;
    for (int d = 0; d < r.n_red && ok; ++d) ok = r.red_lg[d] >= 0;
    int mb = 0, nb = 0;
    for (int d = 0; d < r.n_out && ok; ++d) {
      const int lg = r.out_lg[d];
      if (lg < 0 || (r.out_sA[d] != 0 && r.out_sB[d] != 0) || (r.out_sA[d] == 0 && r.out_sB[d] == 0)) { ok = false; break; }
      for (int b = 0; b < lg && ok; ++b) {
        if (r.out_sB[d] == 0) {
          if (mb >= 14) { ok = false; break; }
          r.mbit_sA[mb] = r.out_sA[d] << b; r.mbit_sC[mb] = r.out_sC[d] << b; ++mb;
        } else {
          if (nb >= 10) { ok = false; break; }
          r.nbit_sB[nb] = r.out_sB[d] << b; r.nbit_sC[nb] = r.out_sC[d] << b; ++nb;
        }
      }
    }
    r.fast = (ok && mb >= 5 && !c128) ? 1 : 0; // (complex128: the general path only)
    r.n_mbits = r.fast ? mb : 0; r.n_nbits = r.fast ? nb : 0;
  }
  for (int s = 0; s < n_steps; ++s)
    for (int which = 0; which < 2; ++which) {
      const int64_t loc = which ? loc_b[s] : loc_a[s];
      if (loc >= 0) last_use[producer[loc]] = std::max(last_use[producer[loc]], level[s]);
    }
  std::vector<int> order(n_steps), where(n_steps);
  for (int s = 0; s < n_steps; ++s) order[s] = s;
  std::stable_sort(order.begin(), order.end(), [&](int x, int y) {
    return group_of[x] != group_of[y] ? group_of[x] < group_of[y] : level[x] < level[y];
  });
  for (int q = 0; q < n_steps; ++q) where[order[q]] = q;
  // ---- per group: reduction tables, the LDS arena (first fit; a block is freed after its last consumer's level)
  struct Block { int off, size; };
  std::vector<ArtnProgGroup> groups(n_groups);
  std::vector<ArtnProgLevel> levels;
  std::vector<ArtnProgWTask> wtasks;
  for (int g = 0; g < n_groups; ++g) {
    const int b = group_start[g], e = group_start[g + 1];
    std::vector<int> steps(order.begin() + b, order.begin() + e); // by level
    int red = 0;
    int n_fast = 0;
    for (int s : steps) {
      rec[s].red_base = red; red += rec[s].red_numel; rec[s].level = level[s];
      if (rec[s].fast) {
        if (n_fast < ARTN_PROG_FAST_MAX) rec[s].fast_index = n_fast++;
        else rec[s].fast = 0; // (the general path takes what the bit-stride area cannot hold)
      }
    }
    if (red > ARTN_PROG_RED_ENTRIES) return fail(ARTN_E_UNSUPPORTED, "reduction tables of a group exceed their LDS share");
    std::vector<Block> free_list = {{0, ARTN_PROG_ARENA_BYTES}};
    // (two-ended: blocks of 16 KiB and more from the top of the arena, the many small ones from the bottom -- with
    //  one first-fit list the long-lived leaves of n12 left no room for the second 32 KiB buffer of its stem)
    auto alloc = [&](int bytes) {
      bytes = (bytes + 15) / 16 * 16;
      if (bytes >= 16384) {
        for (size_t i = free_list.size(); i-- > 0;)
          if (free_list[i].size >= bytes) {
            free_list[i].size -= bytes;
            const int off = free_list[i].off + free_list[i].size;
            if (free_list[i].size == 0) free_list.erase(free_list.begin() + i);
            return off;
          }
        return -1;
      }
      for (size_t i = 0; i < free_list.size(); ++i)
        if (free_list[i].size >= bytes) {
          const int off = free_list[i].off;
          free_list[i].off += bytes; free_list[i].size -= bytes;
          if (free_list[i].size == 0) free_list.erase(free_list.begin() + i);
          return off;
        }
      return -1;
    };
    auto release = [&](int off, int bytes) {
      bytes = (bytes + 15) / 16 * 16;
      size_t i = 0;
      while (i < free_list.size() && free_list[i].off < off) ++i;
      free_list.insert(free_list.begin() + i, {off, bytes});
      for (size_t k = 0; k + 1 < free_list.size();)
        if (free_list[k].off + free_list[k].size == free_list[k + 1].off) { free_list[k].size += free_list[k + 1].size; free_list.erase(free_list.begin() + k + 1); }
        else ++k;
    };
    const int max_level = steps.empty() ? 0 : level[steps.back()];
    // external operands (live from the start to their last reader)
    struct Ext { int lds, numel, last; };
    std::map<int64_t, Ext> exts;
    for (int s : steps)
      for (int which = 0; which < 2; ++which) {
        const int64_t loc = which ? loc_b[s] : loc_a[s];
        if (loc >= 0) continue;
        const int numel = which ? rec[s].b_numel : rec[s].a_numel;
        auto it = exts.find(loc);
        if (it == exts.end()) {
          Ext x = {-1, numel, level[s]};
          if (numel <= ARTN_PROG_PRELOAD_MAX) x.lds = alloc(numel * (int)esz);
          if (x.lds >= 0) (which ? rec[s].pre_b : rec[s].pre_a) = 1;
          it = exts.insert({loc, x}).first;
        } else if (it->second.numel != numel) {
          return fail(ARTN_E_INVALID, "an external operand is used with two sizes");
        }
        it->second.last = std::max(it->second.last, level[s]);
        (which ? rec[s].lds_b : rec[s].lds_a) = it->second.lds;
      }
    groups[g].step_begin = b; groups[g].step_end = e;
    groups[g].level_begin = (int)levels.size();
    size_t q = 0;
    for (int L = 1; L <= max_level; ++L) {
      ArtnProgLevel lv = {(int)wtasks.size(), 0, 0, 0};
      const size_t q0 = q;
      for (; q < steps.size() && level[steps[q]] == L; ++q) {
        const int s = steps[q];
        ArtnProgStep &r = rec[s];
        const bool read_inside = last_use[s] > 0; // (through the arena)
        r.to_ws = (!keep || keep[s] || !read_inside) ? 1 : 0;
        if (read_inside) r.lds_c = alloc(r.out_numel * (int)esz);
        if (r.lds_c < 0) r.to_ws = 1;
        if (r.fast) { // 32 x 16 blocks: first-operand sub-tile fastest
          const int n_tasks = (1 << (r.n_mbits - 5)) * (r.n_nbits > 4 ? 1 << (r.n_nbits - 4) : 1);
          for (int t = 0; t < n_tasks; ++t) wtasks.push_back({where[s], t});
        } else {
          for (int first = 0; first < r.out_numel; first += ARTN_PROG_TASK_ELEMS) wtasks.push_back({where[s], first});
        }
      }
      lv.wt_count = (int)wtasks.size() - lv.wt_begin;
      lv.first_step = lv.wt_count ? wtasks[lv.wt_begin].step : 0;
      levels.push_back(lv);
      // operands whose last reader ran at this level
      for (size_t k = q0; k < q; ++k) {
        const int s = steps[k];
        for (int which = 0; which < 2; ++which) {
          const int64_t loc = which ? loc_b[s] : loc_a[s];
          if (loc < 0) {
            Ext &x = exts[loc];
            if (x.lds >= 0 && x.last == L) { release(x.lds, x.numel * (int)esz); x.last = -1; }
          } else {
            const int p = producer[loc];
            (which ? rec[s].lds_b : rec[s].lds_a) = rec[p].lds_c;
            if (rec[p].lds_c >= 0 && last_use[p] == L) { release(rec[p].lds_c, rec[p].out_numel * (int)esz); last_use[p] = -1; }
          }
        }
      }
    }
    groups[g].level_end = (int)levels.size();
  }
  ArtnProgHeader h;
  memset(&h, 0, sizeof(h));
  h.magic = ARTN_PROG_MAGIC; h.n_groups = n_groups; h.n_steps = n_steps; h.n_levels = (int32_t)levels.size(); h.n_wtasks = (int32_t)wtasks.size();
  int64_t n_terms = 0;
  for (int s = 0; s < n_steps; ++s) n_terms += rec[s].red_numel;
  const int64_t need = prog_image_layout(n_steps, n_groups, (int64_t)levels.size(), (int64_t)wtasks.size(), n_terms, &h);
  if (need > image_bytes) return fail(ARTN_E_INVALID, "image buffer too small (artn_program_image_bytes)");
  char *img = (char *)host_image;
  memset(img, 0, (size_t)need);
  memcpy(img, &h, sizeof(h));
  if (n_groups) memcpy(img + h.off_groups, groups.data(), groups.size() * sizeof(ArtnProgGroup));
  if (!levels.empty()) memcpy(img + h.off_levels, levels.data(), levels.size() * sizeof(ArtnProgLevel));
  if (!wtasks.empty()) memcpy(img + h.off_wtasks, wtasks.data(), wtasks.size() * sizeof(ArtnProgWTask));
  {
    int64_t toff = h.off_tables;
    for (int s = 0; s < n_steps; ++s) {
      ArtnProgStep &r = rec[s];
      r.tab_off = toff;
      int32_t *tab = (int32_t *)(img + toff);
      for (int q = 0; q < r.red_numel; ++q) {
        int rr = q, ka = 0, kb = 0;
        for (int d = 0; d < r.n_red; ++d) {
          const int x = rr % r.red_ext[d];
          rr /= r.red_ext[d];
          ka += x * r.red_sA[d];
          kb += x * r.red_sB[d];
        }
        tab[2 * q] = ka; tab[2 * q + 1] = kb;
      }
      toff += (int64_t)r.red_numel * 8;
    }
  }
  for (int q = 0; q < n_steps; ++q) memcpy(img + h.off_records + (int64_t)q * sizeof(ArtnProgStep), &rec[order[q]], sizeof(ArtnProgStep));
  return ARTN_OK;
}

int artn_program_run(const void *dev_image, int32_t n_groups, const void *const *ext, int32_t n_ext, void *workspace, int32_t dtype,
                     void *stream) {
  if (artn_device_count() < 1) return fail(ARTN_E_NODEVICE, "no gfx950 device visible");
  if (n_groups < 0 || n_ext < 0 || n_ext > ARTN_PROGRAM_MAX_EXT) return fail(ARTN_E_INVALID, "bad group or pointer count");
  if (n_groups == 0) return ARTN_OK;
  if (!dev_image || !workspace || (n_ext && !ext)) return fail(ARTN_E_INVALID, "null pointer");
  ArtnExtPtrs e;
  memset(&e, 0, sizeof(e));
  for (int i = 0; i < n_ext; ++i) e.p[i] = ext[i];
  if (dtype == ARTN_C128) {
    HIP_TRY(ensure_lds<artn_k_program<double>>(ARTN_PROG_LDS_BYTES));
    hipLaunchKernelGGL(artn_k_program<double>, dim3(n_groups), dim3(1024), ARTN_PROG_LDS_BYTES, (hipStream_t)stream, (const char *)dev_image, e,
                       (char *)workspace);
  } else if (dtype == ARTN_C64 || dtype == ARTN_C64_BF16) {
    HIP_TRY(ensure_lds<artn_k_program<float>>(ARTN_PROG_LDS_BYTES));
    hipLaunchKernelGGL(artn_k_program<float>, dim3(n_groups), dim3(1024), ARTN_PROG_LDS_BYTES, (hipStream_t)stream, (const char *)dev_image, e,
                       (char *)workspace);
  } else {
    return fail(ARTN_E_INVALID, "dtype must be ARTN_C64, ARTN_C64_BF16 or ARTN_C128");
  }
  HIP_TRY(hipGetLastError());
  return ARTN_OK;
}

int artn_gather_rows(const void *src, const int64_t *idx, void *dst, int64_t nrows, int64_t row_bytes,
                     int64_t src_rows, int32_t *err_flag, void *stream) {
  if (artn_device_count() < 1) return fail(ARTN_E_NODEVICE, "no gfx950 device visible");
  if (nrows < 0 || row_bytes <= 0 || (row_bytes & 7)) return fail(ARTN_E_INVALID, "row_bytes must be a positive multiple of 8");
  if (nrows == 0) return ARTN_OK;
  if (!src || !idx || !dst) return fail(ARTN_E_INVALID, "null pointer");
  hipStream_t st = (hipStream_t)stream;
  const bool v16 = (row_bytes % 16 == 0) && ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0);
  const long vecs = v16 ? row_bytes / 16 : row_bytes / 8;
  const long total = nrows * vecs;
  const int grid = (int)std::min<long>((total + ARTN_WG_THREADS - 1) / ARTN_WG_THREADS, 256L * 8);
  if (v16)
    hipLaunchKernelGGL((artn_k_gather_rows<float4>), dim3(grid), dim3(ARTN_WG_THREADS), 0, st,
                       (const float4 *)src, idx, (float4 *)dst, (long)nrows, vecs, (long)src_rows, err_flag);
  else
    hipLaunchKernelGGL((artn_k_gather_rows<float2>), dim3(grid), dim3(ARTN_WG_THREADS), 0, st,
                       (const float2 *)src, idx, (float2 *)dst, (long)nrows, vecs, (long)src_rows, err_flag);
  HIP_TRY(hipGetLastError());
  return ARTN_OK;
}

int artn_axpy_c64(void *acc, const void *x, int64_t n, void *stream) {
  if (artn_device_count() < 1) return fail(ARTN_E_NODEVICE, "no gfx950 device visible");
  if (n < 0) return fail(ARTN_E_INVALID, "negative length");
  if (n == 0) return ARTN_OK;
  if (!acc || !x) return fail(ARTN_E_INVALID, "null pointer");
  hipStream_t st = (hipStream_t)stream;
  const bool v16 = (n % 2 == 0) && ((((uintptr_t)acc | (uintptr_t)x) & 15) == 0);
  if (v16) {
    const long n4 = n / 2;
    const int grid = (int)std::min<long>((n4 + ARTN_WG_THREADS - 1) / ARTN_WG_THREADS, 256L * 8);
    hipLaunchKernelGGL(artn_k_axpy4, dim3(grid), dim3(ARTN_WG_THREADS), 0, st, (float4 *)acc, (const float4 *)x, n4);
  } else {
    const long n1 = n * 2;
    const int grid = (int)std::min<long>((n1 + ARTN_WG_THREADS - 1) / ARTN_WG_THREADS, 256L * 8);
    hipLaunchKernelGGL(artn_k_axpy1, dim3(grid), dim3(ARTN_WG_THREADS), 0, st, (float *)acc, (const float *)x, n1);
  }
  HIP_TRY(hipGetLastError());
  return ARTN_OK;
}

int artn_sum_axis_c64(const void *in, void *out, int64_t n_groups, int64_t n_rows, int64_t n_cols, void *stream) {
  if (artn_device_count() < 1) return fail(ARTN_E_NODEVICE, "no gfx950 device visible");
  if (n_groups < 0 || n_rows < 1 || n_cols < 0) return fail(ARTN_E_INVALID, "bad extent");
  if (n_groups == 0 || n_cols == 0) return ARTN_OK;
  if (!in || !out) return fail(ARTN_E_INVALID, "null pointer");
  if ((((uintptr_t)in | (uintptr_t)out) & 7) != 0) return fail(ARTN_E_UNSUPPORTED, "artn_sum_axis_c64 needs 8-byte aligned buffers");
  if ((n_cols & 1) || ((((uintptr_t)in | (uintptr_t)out) & 15) != 0)) {
    // an odd column count (3^12 amplitudes of a bond-dimension-3 network) or buffers that are only 8-byte aligned: one element per lane
    const long col_tiles = (n_cols + 63) / 64;
    if (n_groups * col_tiles > (1L << 30)) return fail(ARTN_E_UNSUPPORTED, "too many workgroups");
    hipLaunchKernelGGL(artn_k_sum_axis<v2f_t>, dim3((unsigned)(n_groups * col_tiles)), dim3(ARTN_WG_THREADS), 0, (hipStream_t)stream,
                       (const v2f_t *)in, (v2f_t *)out, (long)n_rows, (long)n_cols, col_tiles);
    HIP_TRY(hipGetLastError());
    return ARTN_OK;
  }
  const long n4 = n_cols / 2, col_tiles = (n4 + 63) / 64;
  if (n_groups * col_tiles > (1L << 30)) return fail(ARTN_E_UNSUPPORTED, "too many workgroups");
  hipLaunchKernelGGL(artn_k_sum_axis<f32x4>, dim3((unsigned)(n_groups * col_tiles)), dim3(ARTN_WG_THREADS), 0, (hipStream_t)stream,
                     (const f32x4 *)in, (f32x4 *)out, (long)n_rows, n4, col_tiles);
  HIP_TRY(hipGetLastError());
  return ARTN_OK;
}

int artn_sum_axis_c128(const void *in, void *out, int64_t n_groups, int64_t n_rows, int64_t n_cols, void *stream) {
  if (artn_device_count() < 1) return fail(ARTN_E_NODEVICE, "no gfx950 device visible");
  if (n_groups < 0 || n_rows < 1 || n_cols < 0) return fail(ARTN_E_INVALID, "bad extent");
  if (n_groups == 0 || n_cols == 0) return ARTN_OK;
  if (!in || !out) return fail(ARTN_E_INVALID, "null pointer");
  if ((((uintptr_t)in | (uintptr_t)out) & 15) != 0) return fail(ARTN_E_UNSUPPORTED, "artn_sum_axis_c128 needs 16-byte aligned buffers");
  const long col_tiles = (n_cols + 63) / 64;
  if (n_groups * col_tiles > (1L << 30)) return fail(ARTN_E_UNSUPPORTED, "too many workgroups");
  hipLaunchKernelGGL(artn_k_sum_axis<f64x2>, dim3((unsigned)(n_groups * col_tiles)), dim3(ARTN_WG_THREADS), 0, (hipStream_t)stream,
                     (const f64x2 *)in, (f64x2 *)out, (long)n_rows, (long)n_cols, col_tiles);
  HIP_TRY(hipGetLastError());
  return ARTN_OK;
}

int artn_axpy_c128(void *acc, const void *x, int64_t n, void *stream) {
  if (artn_device_count() < 1) return fail(ARTN_E_NODEVICE, "no gfx950 device visible");
  if (n < 0) return fail(ARTN_E_INVALID, "negative length");
  if (n == 0) return ARTN_OK;
  if (!acc || !x) return fail(ARTN_E_INVALID, "null pointer");
  if ((((uintptr_t)acc | (uintptr_t)x) & 15) != 0) return fail(ARTN_E_UNSUPPORTED, "artn_axpy_c128 needs 16-byte aligned buffers");
  const int grid = (int)std::min<long>((n + ARTN_WG_THREADS - 1) / ARTN_WG_THREADS, 256L * 8);
  hipLaunchKernelGGL(artn_k_axpy_v<f64x2>, dim3(grid), dim3(ARTN_WG_THREADS), 0, (hipStream_t)stream, (f64x2 *)acc, (const f64x2 *)x, (long)n);
  HIP_TRY(hipGetLastError());
  return ARTN_OK;
}

int artn_probe_mfma_rate(int kind, void *scratch4, double *tflops) {
  if (artn_device_count() < 1) return fail(ARTN_E_NODEVICE, "no gfx950 device visible");
  if (kind < 0 || kind > 2 || !scratch4 || !tflops) return fail(ARTN_E_INVALID, "bad argument");
  const int iters = 20000, waves_per_cu = 8;
  const double flop_per_mfma[3] = {4096.0, 32768.0, 2048.0}, per_iter[3] = {2.0, 2.0, 4.0};
  dim3 grid((unsigned)(g_ncu * waves_per_cu / 4)), block(ARTN_WG_THREADS);
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  float ms = 0.f;
  for (int rep = 0; rep < 2; ++rep) { // (the first launch warms the clocks up)
    HIP_TRY(hipEventRecord(e0, nullptr));
    if (kind == 0) hipLaunchKernelGGL(artn_k_mfma_probe<0>, grid, block, 0, nullptr, (float *)scratch4, iters);
    else if (kind == 1) hipLaunchKernelGGL(artn_k_mfma_probe<1>, grid, block, 0, nullptr, (float *)scratch4, iters);
    else hipLaunchKernelGGL(artn_k_mfma_probe<2>, grid, block, 0, nullptr, (float *)scratch4, iters);
    HIP_TRY(hipEventRecord(e1, nullptr));
    HIP_TRY(hipEventSynchronize(e1));
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  const double mfmas = (double)g_ncu * waves_per_cu * iters * per_iter[kind];
  *tflops = mfmas * flop_per_mfma[kind] / (ms * 1e-3) / 1e12;
  return ARTN_OK;
}

int artn_absmax_normalize_c64(void *x, int64_t n, float *out_absmax, void *stream) {
  if (artn_device_count() < 1) return fail(ARTN_E_NODEVICE, "no gfx950 device visible");
  if (n <= 0 || !x || !out_absmax) return fail(ARTN_E_INVALID, "bad argument");
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(hipMemsetAsync(out_absmax, 0, sizeof(float), st));
  const int grid = (int)std::min<long>((n + ARTN_WG_THREADS - 1) / ARTN_WG_THREADS, 256L * 8);
  hipLaunchKernelGGL(artn_k_absmax, dim3(grid), dim3(ARTN_WG_THREADS), 0, st, (const float2 *)x, (long)n,
                     (unsigned int *)out_absmax);
  hipLaunchKernelGGL(artn_k_divide, dim3(grid), dim3(ARTN_WG_THREADS), 0, st, (float2 *)x, (long)n,
                     (const float *)out_absmax);
  HIP_TRY(hipGetLastError());
  return ARTN_OK;
}

int artn_absmax_normalize_c128(void *x, int64_t n, double *out_absmax, void *stream) {
  if (artn_device_count() < 1) return fail(ARTN_E_NODEVICE, "no gfx950 device visible");
  if (n <= 0 || !x || !out_absmax) return fail(ARTN_E_INVALID, "bad argument");
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(hipMemsetAsync(out_absmax, 0, sizeof(double), st));
  const int grid = (int)std::min<long>((n + ARTN_WG_THREADS - 1) / ARTN_WG_THREADS, 256L * 8);
  hipLaunchKernelGGL(artn_k_absmax128, dim3(grid), dim3(ARTN_WG_THREADS), 0, st, (const double2 *)x, (long)n,
                     (unsigned long long *)out_absmax);
  hipLaunchKernelGGL(artn_k_divide128, dim3(grid), dim3(ARTN_WG_THREADS), 0, st, (double2 *)x, (long)n,
                     (const double *)out_absmax);
  HIP_TRY(hipGetLastError());
  return ARTN_OK;
}

} // extern "C"
#endif // !ARTN_TU_PART
