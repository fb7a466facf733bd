// Probe (round 5): would THREE workgroups per CU pay for the state-streaming pair kernel?
//
// artn_k_bits runs two workgroups per CU (two 32 KiB LDS regions each, 190-256 VGPRs: the small operand's fragments live in
// registers) and the two fall into lockstep -- both in their MFMA stages, then both in their copy phases.
// The structure proposed instead: ONE region per workgroup, stage results held in 32 registers per lane to a barrier and
// written over the input (in place), the small operand's fragments read from LDS next to the tile operand -> ~52 KiB of LDS and
// < 168 VGPRs per workgroup: three per CU.  This kernel has that structure with synthetic addresses (a tile = 32 KiB contiguous
// in, 32 KiB contiguous out; every LDS access conflict-free) and the instruction mix of a 3M pair: per triple of MFMAs one
// ds_read_b64 of the tile operand, one of the fragment (or none: fragments in registers), two v_add.
//
//   tri_probe L1 L2 [RUN_IN RUN_OUT STRIDE XCD]   (runs of 2^RUN bytes at stride 2^STRIDE; RUN 0: contiguous 32 KiB tiles)   L1 / L2 = MFMA triples per wave and tile in stage 1 / 2 (6-bit stage 32, 5-bit 16, 4-bit 8, 3-bit 4)
// prints ms per pass over 262 144 tiles (8 GiB in + 8 GiB out: one n30 pair) for: fragments in LDS x {3, 2 workgroups per CU},
// fragments in registers x 2 per CU (today's structure on one region), and the copy-only / MFMA-only ablations of the first.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) v2f lds_v2f;
typedef __attribute__((address_space(3))) f32x4 lds_f4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
#define REGION 32768u
#define W1_OFF REGION
#define W2_OFF (REGION + 16384u)
#define LDS_MIN (REGION + 16384u + 4096u)

__device__ __forceinline__ v2f ldr8(unsigned a) { return *(lds_v2f *)(unsigned long)a; }
__device__ __forceinline__ void ldw8(unsigned a, v2f v) { *(lds_v2f *)(unsigned long)a = v; }
__device__ __forceinline__ f32x4 ldr16(unsigned a) { return *(lds_f4 *)(unsigned long)a; }
__device__ __forceinline__ void ldw16(unsigned a, f32x4 v) { *(lds_f4 *)(unsigned long)a = v; }

// one stage: L triples; operands of 4 triples in flight (ping-pong of 4); results -> hold[32]
template <int L, bool WLDS, bool MFMA>
__device__ __forceinline__ void stage(unsigned a_base, unsigned w_base, const v2f *wreg, float (&hold)[32]) {
  f32x16 t1, t2, t3;
#pragma unroll
  for (int e = 0; e < 16; ++e) t1[e] = t2[e] = t3[e] = 0.f;
  constexpr int U = 4;
  v2f a[2][U], w[2][U];
  auto fetch = [&](int s0, int buf) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      a[buf][u] = ldr8((a_base & ~(REGION - 1)) | ((a_base + (unsigned)((s0 + u) * 2 * 128 * 8)) & (REGION - 1)));   // k = 2 s (+ lane half): 128 columns per k
      if (WLDS) w[buf][u] = ldr8(w_base + (unsigned)((s0 + u) * 512));
      else w[buf][u] = wreg[s0 + u];
    }
  };
  fetch(0, 0);
#pragma unroll
  for (int s0 = 0; s0 < L; s0 += U) {
    const int buf = (s0 / U) & 1;
    if (s0 + U < L) fetch(s0 + U, buf ^ 1);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const v2f x = a[buf][u], y = w[buf][u];
      if (MFMA) {
        t1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y.x, x.x, t1, 0, 0, 0);
        t2 = __builtin_amdgcn_mfma_f32_32x32x2f32(y.y, x.y, t2, 0, 0, 0);
        t3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y.x + y.y, x.x + x.y, t3, 0, 0, 0);
      } else {
        t1[u] += y.x * x.x; t2[u] += y.y * x.y; t3[u] += (y.x + y.y) * (x.x + x.y);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) { hold[2 * e] = t1[e] - t2[e]; hold[2 * e + 1] = t3[e] - t1[e] - t2[e]; }
}

// byte offset of 16-byte chunk c (0 .. 2047) of tile t: contiguous tiles (rs == 0), or runs of 2^rb bytes (rb = rs >> 8: 6 .. 9) at
// stride 2^(rs & 255) -- the shape of a real tile (its lowest run bits + scattered ones); the tiles of a block interleave
__device__ long g_skew = 0; // bytes added to the row stride (0: a power of two, as in a dense bit-addressed tensor)
__device__ __forceinline__ long gaddr(long t, int c, int rs) {
  if (rs == 0) return t * 32768L + c * 16;
  const int rb = rs >> 8, st = rs & 255;
  // (shifts and masks only: a 64-bit division per chunk here once looked like a 1 ms penalty of scattered tiles)
  const long rows = 32768L >> rb;
  return (t >> (st - rb)) * (rows << st) + ((t & ((1L << (st - rb)) - 1)) << rb) + (long)(c >> (rb - 4)) * ((1L << st) + g_skew) + (c & ((1 << (rb - 4)) - 1)) * 16;
}
__device__ unsigned long long g_marks[1024 * 10];
// (round 6: random operands.  On all-zero data the chip holds 2.38 GHz where the real kernel, on real data, holds 1.84 --
//  MI355X_MICROARCH.md "DVFS give-back" -- and the probe looked 25 % faster than it is.)
__global__ void fill_random(float *p, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    unsigned s = (unsigned)i * 2654435761u + 12345u;
    s = s * 1664525u + 1013904223u;
    p[i] = ((int)(s >> 9) - (1 << 22)) * (1.0f / (1 << 22));
  }
}
#define MARK(i) do { if (marks && iter == 20 && tid == 0) g_marks[blockIdx.x * 10 + (i)] = wall_clock64(); } while (0)
// CONFL (round 6): the operand reads of both stages are 2^CONFL-way bank conflicted (lanes j, j + 32 / 2^CONFL ... of a
// half wave share a bank), the scatter writes 2^CONFL-way too -- what a real tile's bit layout does to artn_k_bits
template <int L1, int L2, int WPC, bool WLDS, bool MFMA, bool COPY, bool TWOREG = false, int PRIO = 0, int CONFL = 0>
__global__ __launch_bounds__(256, WPC) void tri(const char *__restrict__ A, char *__restrict__ C, long n_tiles, int rs_in, int rs_out, int marks) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // fragments (synthetic): in LDS, or in registers for the whole kernel
  v2f wreg1[WLDS ? 1 : L1], wreg2[WLDS ? 1 : L2];
  if (WLDS) {
    for (unsigned o = tid * 8; o < 16384u + 4096u; o += 256 * 8) ldw8(W1_OFF + o, v2f{1.0f / 64, 0.5f / 64});
  } else {
#pragma unroll
    for (int s = 0; s < L1; ++s) { unsigned q = (unsigned)(lane * 131 + s * 7919 + 1) * 2654435761u; wreg1[s] = v2f{((int)(q >> 9) - (1 << 22)) * (1.0f / (1 << 26)), ((int)((q * 1664525u) >> 9) - (1 << 22)) * (1.0f / (1 << 26))}; }
#pragma unroll
    for (int s = 0; s < L2; ++s) { unsigned q = (unsigned)(lane * 137 + s * 7907 + 5) * 2654435761u; wreg2[s] = v2f{((int)(q >> 9) - (1 << 22)) * (1.0f / (1 << 26)), ((int)((q * 1664525u) >> 9) - (1 << 22)) * (1.0f / (1 << 26))}; }
  }
  // column = wave * 32 + lane & 31, k parity = lane >> 5; CONFL: the top CONFL bits of the column select 2^CONFL slots that are
  // 2 KiB apart (same banks) instead of neighbours
  const unsigned jc = (unsigned)(lane & 31);
  const unsigned col_off = CONFL == 0 ? jc * 8u : ((jc & ((32u >> CONFL) - 1u)) * 8u + (jc >> (5 - CONFL)) * 2048u);
  const unsigned a_lane = (unsigned)(((lane >> 5) * 128 + wave * 32) * 8) + col_off;
  const unsigned w_lane = (unsigned)lane * 8u;
  const unsigned sc_lane = (unsigned)((wave * 32) * 8 + (lane >> 5) * 16 * 1024) + col_off; // scatter: 16 rows of 1 KiB per lane half
  const long G = gridDim.x;
  long t = blockIdx.x;
  if (rs_in >> 16) t = (long)(blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3); // XCD-aware: each XCD takes a contiguous eighth of a period
  rs_in &= 0xffff;
  f32x4 v[8];
  if (COPY && t < n_tiles) {
#pragma unroll
    for (int i = 0; i < 8; ++i) ldw16((unsigned)(tid + 256 * i) * 16u, *(const f32x4 *)(A + gaddr(t, tid + 256 * i, rs_in)));
    if (t + G < n_tiles)
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = *(const f32x4 *)(A + gaddr(t + G, tid + 256 * i, rs_in));
  }
  __syncthreads();
  int iter = 0;
  for (; t < n_tiles; t += G, ++iter) {
    float hold[32];
    MARK(0);
    if constexpr (TWOREG) {
      // artn_k_bits' structure (round 6): two regions, a stage scatters its result into the OTHER region right after its
      // chain (no barrier between chain and scatter), four barriers per tile
      if (PRIO == 2 && (__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 1)) __builtin_amdgcn_s_setprio(2);
      stage<L1, WLDS, MFMA>(a_lane, W1_OFF + w_lane, wreg1, hold);
#pragma unroll
      for (int e = 0; e < 16; ++e) ldw8(32768u + ((sc_lane + (unsigned)e * 1024u) & (REGION - 1)), v2f{hold[2 * e], hold[2 * e + 1]});
      MARK(1);
      __syncthreads();
      MARK(2);
      stage<L2, WLDS, MFMA>(32768u + a_lane, W2_OFF + w_lane, wreg2, hold);
#pragma unroll
      for (int e = 0; e < 16; ++e) ldw8((sc_lane + (unsigned)e * 1024u) & (REGION - 1), v2f{hold[2 * e], hold[2 * e + 1]});
      if (PRIO == 2) __builtin_amdgcn_s_setprio(0);
      MARK(3);
      __syncthreads();
      MARK(4);
      if (PRIO >= 1) __builtin_amdgcn_s_setprio(3);
    } else {
    stage<L1, WLDS, MFMA>(a_lane, W1_OFF + w_lane, wreg1, hold);
    MARK(1);
    __syncthreads();                       // every wave has read the tile: results go over it
#pragma unroll
    for (int e = 0; e < 16; ++e) ldw8((sc_lane + (unsigned)e * 1024u) & (REGION - 1), v2f{hold[2 * e], hold[2 * e + 1]});
    __syncthreads();
    MARK(2);
    stage<L2, WLDS, MFMA>(a_lane, W2_OFF + w_lane, wreg2, hold);
    MARK(3);
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 16; ++e) ldw8((sc_lane + (unsigned)e * 1024u) & (REGION - 1), v2f{hold[2 * e], hold[2 * e + 1]});
    __syncthreads();
    MARK(4);
    }
    if (COPY) {
      f32x4 x[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = ldr16((unsigned)(tid + 256 * i) * 16u);
      __syncthreads();
      MARK(5);
      if (t + G < n_tiles)
#pragma unroll
        for (int i = 0; i < 8; ++i) ldw16((unsigned)(tid + 256 * i) * 16u, v[i]);
#pragma unroll
      for (int i = 0; i < 8; ++i) __builtin_nontemporal_store(x[i], (f32x4 *)(C + gaddr(t, tid + 256 * i, rs_out)));
      if (t + 2 * G < n_tiles)
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = __builtin_nontemporal_load((const f32x4 *)(A + gaddr(t + 2 * G, tid + 256 * i, rs_in)));
      MARK(6);
    } else if (hold[0] == 12345.f) {
      C[tid] = 1;
    }
    if (TWOREG && PRIO >= 1) __builtin_amdgcn_s_setprio(0);
    __syncthreads();
    MARK(7);
    if (marks && iter == 21 && tid == 0) g_marks[blockIdx.x * 10 + 8] = wall_clock64();
  }
}

static int g_want_marks = 0;
template <int L1, int L2>
static void run_all(const char *a, char *c, long n_tiles, hipEvent_t e0, hipEvent_t e1, int rs_in, int rs_out) {
  auto timeit = [&](const char *name, auto kern, int wpc, unsigned lds) {
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 256, lds));
    float best = 1e9;
    for (int r = 0; r < 4; ++r) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(kern, dim3(256 * wpc), dim3(256), lds, 0, a, c, n_tiles, rs_in, rs_out, g_want_marks);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (r > 0 && ms < best) best = ms;
    }
    if (g_want_marks) {
      static unsigned long long m[1024 * 10];
      CK(hipMemcpyFromSymbol(m, HIP_SYMBOL(g_marks), sizeof(m)));
      const int nb = 256 * wpc;
      double seg[8] = {0};
      for (int b = 0; b < nb; ++b) {
        for (int i = 0; i < 7; ++i) seg[i] += (double)(m[b * 10 + i + 1] - m[b * 10 + i]);
        seg[7] += (double)(m[b * 10 + 8] - m[b * 10 + 0]);   // (mark 8 is taken one iteration later, after the end barrier: a period + the tail)
      }
      const char *nm[7] = {"stage 1", "barrier scatter barrier", "stage 2", "barrier scatter barrier", "x reads + barrier", "refill stores loads", "end barrier"};
      printf("      phases (ns, mean over workgroups, iteration 20):");
      for (int i = 0; i < 7; ++i) printf(" %s %.0f |", nm[i], seg[i] / nb * 10);
      printf("\n      raw marks of workgroup 0:");
      for (int i = 0; i < 9; ++i) printf(" %llu", m[i] - m[0]);
      printf("\n");
    }
    const double flop = 8.0 / 6.0 * 3.0 * (L1 + L2) * 4.0 * 32 * 32 * 2 * 2 * (double)n_tiles; // nominal (8 per complex MAC) of the 3M triples
    printf("  %-58s occ %d  %7.3f ms  %6.2f TB/s  %6.1f TFLOP/s nominal  %5.2f us per tile and workgroup\n", name, occ, best,
           2.0 * 32768.0 * n_tiles / best / 1e9, flop / best / 1e9, best * 1e3 / ((double)n_tiles / (256.0 * wpc)));
  };
  printf("L1 %d L2 %d  input runs 2^%d B, output runs 2^%d B (0: contiguous tiles), stride 2^%d B, %s tile order\n", L1, L2, (rs_in >> 8) & 255, rs_out >> 8, rs_out & 255, (rs_in >> 16) ? "XCD-aware" : "round-robin");
  timeit("fragments in LDS, 3 workgroups per CU", tri<L1, L2, 3, true, true, true>, 3, LDS_MIN);
  timeit("fragments in LDS, 2 workgroups per CU (LDS padded)", tri<L1, L2, 3, true, true, true>, 2, 70 * 1024);
  timeit("fragments in registers, 2 per CU", tri<L1, L2, 2, false, true, true>, 2, LDS_MIN);
  timeit("  ... two regions, scatter after the chain, 4 barriers", tri<L1, L2, 2, false, true, true, true, 0>, 2, 65536 + 4096);
  timeit("  ... 2-way bank conflicts on operand reads and scatters", tri<L1, L2, 2, false, true, true, true, 0, 1>, 2, 65536 + 4096);
  timeit("  ... 4-way bank conflicts on operand reads and scatters", tri<L1, L2, 2, false, true, true, true, 0, 2>, 2, 65536 + 4096);
  timeit("  ... and s_setprio 3 in the copy phases", tri<L1, L2, 2, false, true, true, true, 1>, 2, 65536 + 4096);
  timeit("  ... and s_setprio 2 in the odd workgroup's stages", tri<L1, L2, 2, false, true, true, true, 2>, 2, 65536 + 4096);
  timeit("fragments in registers, ONE per CU (one wave per SIMD)", tri<L1, L2, 2, false, true, true>, 1, LDS_MIN);
  timeit("fragments in registers, ONE per CU, no global traffic", tri<L1, L2, 2, false, true, false>, 1, LDS_MIN);
  timeit("fragments in registers, 2 per CU, no global traffic", tri<L1, L2, 2, false, true, false>, 2, LDS_MIN);
  timeit("fragments in LDS, 3 per CU, no global traffic", tri<L1, L2, 3, true, true, false>, 3, LDS_MIN);
  timeit("fragments in LDS, 3 per CU, no MFMA", tri<L1, L2, 3, true, false, true>, 3, LDS_MIN);
}

int main(int argc, char **argv) {
  const int l1 = argc > 1 ? atoi(argv[1]) : 32, l2 = argc > 2 ? atoi(argv[2]) : 8;
  // RUN_IN RUN_OUT: log2 bytes per run (0: contiguous tiles), STRIDE: log2 bytes between runs, XCD: 1 = XCD-aware tile order
  const int rb_in = argc > 3 ? atoi(argv[3]) : 0, rb_out = argc > 4 ? atoi(argv[4]) : 0, st = argc > 5 ? atoi(argv[5]) : 20, xcd = argc > 6 ? atoi(argv[6]) : 0;
  const int rs_in = (rb_in ? (rb_in << 8) | st : 0) | (xcd << 16), rs_out = rb_out ? (rb_out << 8) | st : 0;
  const long n_tiles = 262144;
  const long skew = argc > 7 ? atol(argv[7]) : 0;
  g_want_marks = argc > 8 ? atoi(argv[8]) : 0;
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_skew), &skew, sizeof(long)));
  printf("row stride skew %ld bytes\n", skew);
  char *a, *c;
  CK(hipMalloc(&a, n_tiles * 32768L + 512L * skew + (1L << 30)));
  CK(hipMalloc(&c, n_tiles * 32768L + 512L * skew + (1L << 30)));
  if (getenv("TRI_ZERO")) CK(hipMemset(a, 0, n_tiles * 32768L));
  else hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, (float *)a, n_tiles * 8192L);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  if (l1 == 32 && l2 == 8) run_all<32, 8>(a, c, n_tiles, e0, e1, rs_in, rs_out);        // 6 + 4
  else if (l1 == 16 && l2 == 16) run_all<16, 16>(a, c, n_tiles, e0, e1, rs_in, rs_out); // 5 + 5
  else if (l1 == 16 && l2 == 8) run_all<16, 8>(a, c, n_tiles, e0, e1, rs_in, rs_out);   // 5 + 4
  else if (l1 == 8 && l2 == 4) run_all<8, 4>(a, c, n_tiles, e0, e1, rs_in, rs_out);     // 4 + 3
  else if (l1 == 32 && l2 == 32) run_all<32, 32>(a, c, n_tiles, e0, e1, rs_in, rs_out); // 6 + 6
  else printf("instantiated: 32 8 | 16 16 | 16 8 | 8 4 | 32 32\n");
  return 0;
}
