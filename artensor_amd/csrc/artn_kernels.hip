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
#include <stdio.h>
#include <stdlib.h>
#include <mutex>
#include <string>

#include "artn_plan.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ----------------------------------------------------------------------------------------
// error plumbing
// ----------------------------------------------------------------------------------------
static thread_local std::string g_err;
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
// bit-permuted complex GEMM
// ----------------------------------------------------------------------------------------
// Lane roles inside one v_mfma_f32_32x32x2_f32 (D[i][j] += sum_kk Aop[i][kk] * Bop[kk][j]):
//   i = real output column n' = 2*n_local + (0: re, 1: im)   -> Aop lane l: [i = l&31][kk = l>>5]
//   j = tile column m (32 elements of the A tile)              -> Bop lane l: [kk = l>>5][j = l&31]
//   kk = h = l>>5 selects complex K index kc = 2*s + h; the re and im parts of that A
//   element are fed by two consecutive MFMAs (phase p = 0, 1), so one 8-byte LDS read
//   serves two MFMAs.
// Accumulator (guide section 3): lane l holds column j = l&31 and rows
//   i = (r&3) + 8*(r>>2) + 4*(l>>5), r = 0..15  =>  n_local = (r&3)/2 + 2*h + 4*(r>>2),
//   (acc[4q+2b], acc[4q+2b+1]) = (re, im) of n_local = b + 2h + 4q.
// Register discipline: everything that is the same for all lanes lives in SGPRs and is
// recomputed from the kernel argument per tile; per-lane state is a handful of 32-bit
// offsets.  OPAQUE() stops the compiler from hoisting per-chunk address arithmetic out of
// the tile loop (that hoisting, not the algorithm, is what used to cost >100 VGPRs).
#define OPAQUE_V(x) asm volatile("" : "+v"(x))

// Tile index -> element offsets of the tile in A, B, C (all wave-uniform: SALU only).
__device__ __forceinline__ void tile_offsets(const ArtnBitsPlan &P, long tile, long &offA, long &offB, long &offC) {
  long r = tile;
  offA = offB = offC = 0;
  for (int d = 0; d < P.n_outer; ++d) {
    const long ext = P.outer[d].ext;
    long x;
    if (P.outer[d].log2ext >= 0) {
      x = r & (ext - 1);
      r >>= P.outer[d].log2ext;
    } else {
      x = r % ext;
      r /= ext;
    }
    offA += x * P.outer[d].sA;
    offB += x * P.outer[d].sB;
    offC += x * P.outer[d].sC;
  }
}

// Copy-in, split in two so the loads of tile t+1 can be in flight while tile t is computed:
// issue_loads puts 8 x 16 B per thread in flight (uniform 64-bit base in SGPRs + one 32-bit
// per-lane byte offset); store_lds writes them to LDS linearly.  With fewer than 8 chunks
// per thread (small tiles) the surplus slots re-load an earlier chunk and are not stored.
__device__ __forceinline__ void issue_loads(f32x4 (&v)[8], const char *__restrict__ Abase, const ArtnBitsPlan &P,
                                            unsigned lane_off, int i0, int n_iters) {
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int i = (i0 + u) & (n_iters - 1);
    long off = 0;
#pragma unroll
    for (int b = 9; b < ARTN_TILE_BITS_MAX; ++b)
      if (b < P.T_in && ((i >> (b - 9)) & 1)) off += P.in_stride[b];
    v[u] = *reinterpret_cast<const f32x4 *>(Abase + off * 8 + lane_off);
  }
}
__device__ __forceinline__ void store_lds(const f32x4 (&v)[8], char *ldsb, unsigned tid16, int i0, int n_iters) {
#pragma unroll
  for (int u = 0; u < 8; ++u)
    if (i0 + u < n_iters) *reinterpret_cast<f32x4 *>(ldsb + tid16 + (i0 + u) * (ARTN_WG_THREADS * 16)) = v[u];
}

template <int KB, int PM>
__global__ __launch_bounds__(ARTN_WG_THREADS, (KB <= 5 ? 4 : 3)) void artn_k_bits(const float2 *__restrict__ A,
                                                               const float2 *__restrict__ B,
                                                               float2 *__restrict__ C,
                                                               const ArtnBitsPlan P) {
  constexpr int S = 1 << (KB - 1); // complex K pairs
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  char *ldsb = reinterpret_cast<char *>(smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int wn = wave & ((1 << P.wn_log2) - 1);
  const int wm = wave >> P.wn_log2;
  const int nt_eff = P.nt < 4 ? P.nt : 4;

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
  const unsigned tid16 = tid * 16;
  const int n_in_iters = 1 << (P.T_in - 9), n_out_iters = 1 << (P.T_out - 9);

  // ---- MFMA phase: per-lane tile-local byte offsets
  unsigned lane_in = (unsigned)h << (P.k_in_pos[0] + 3), lane_out = 0;
#pragma unroll
  for (int b = 0; b < 5; ++b) {
    if ((j >> b) & 1) {
      lane_in += 8u << P.lane_in_pos[b];
      lane_out += 8u << P.lane_out_pos[b];
    }
  }
  if (P.nt > 1) lane_out += (unsigned)h << (P.n_out_pos[1] + 3);

  // ---- small-operand fragments: lane (i = lane&31, h) needs W[n'=i][(kc = 2s+h, p)]
  const int ro = j & 1, nloc = j >> 1;
  const bool w_valid = (nloc >> nt_eff) == 0;
  unsigned lane_b = (unsigned)h * (unsigned)P.k_b_stride[0] * 8u;
#pragma unroll
  for (int b = 0; b < 4; ++b)
    if (b < nt_eff && ((nloc >> b) & 1)) lane_b += (unsigned)P.n_b_stride[b] * 8u;
  float W0[S], W1[S];
  long prev_offB = -1;

  // software pipeline: the loads of the next tile are issued before this tile's MFMA phase
  f32x4 v[8];
  const bool prefetch = n_in_iters <= 8;
  if (prefetch && (long)blockIdx.x < P.n_tiles) {
    long oa, ob, oc;
    tile_offsets(P, blockIdx.x, oa, ob, oc);
    issue_loads(v, reinterpret_cast<const char *>(A + oa), P, in_lane, 0, n_in_iters);
  }

  for (long tile = blockIdx.x; tile < P.n_tiles; tile += gridDim.x) {
    long offA, offB, offC;
    tile_offsets(P, tile, offA, offB, offC);

    if (offB != prev_offB) {
      prev_offB = offB;
      long wn_b = 0;
#pragma unroll
      for (int b = 0; b < 2; ++b)
        if (b < P.wn_log2 && ((wn >> b) & 1)) wn_b += P.n_b_stride[4 + b];
      const char *Bbase = reinterpret_cast<const char *>(B + offB + wn_b);
#pragma unroll
      for (int s = 0; s < S; ++s) {
        long ko = 0;
#pragma unroll
        for (int b = 1; b < KB; ++b)
          if ((s >> (b - 1)) & 1) ko += P.k_b_stride[b];
        float2 bv = make_float2(0.f, 0.f);
        if (w_valid) bv = *reinterpret_cast<const float2 *>(Bbase + ko * 8 + lane_b);
        W0[s] = ro ? bv.y : bv.x;
        W1[s] = ro ? bv.x : -bv.y;
      }
    }

    // ---- copy-in: global (16 B per lane, runs of 2^run_in elements) -> LDS (linear)
    {
      unsigned lo = in_lane, t16 = tid16;
      OPAQUE_V(lo);
      OPAQUE_V(t16);
      if (prefetch) {
        __syncthreads(); // previous tile's copy-out has finished reading LDS
        store_lds(v, ldsb, t16, 0, n_in_iters);
        __syncthreads();
        const long next = tile + gridDim.x;
        if (next < P.n_tiles) {
          long oa, ob, oc;
          tile_offsets(P, next, oa, ob, oc);
          issue_loads(v, reinterpret_cast<const char *>(A + oa), P, lo, 0, n_in_iters);
        }
      } else {
        const char *Abase = reinterpret_cast<const char *>(A + offA);
        for (int i0 = 0; i0 < n_in_iters; i0 += 8) {
          issue_loads(v, Abase, P, lo, i0, n_in_iters);
          if (i0 == 0) __syncthreads();
          store_lds(v, ldsb, t16, i0, n_in_iters);
        }
        __syncthreads();
      }
    }

    // ---- MFMA
    f32x16 acc[PM];
    unsigned msub_out[PM];
    {
      unsigned li = lane_in;
      OPAQUE_V(li);
      unsigned msub_in[PM];
#pragma unroll
      for (int pm = 0; pm < PM; ++pm) {
        const int msub = wm * PM + pm;
        unsigned oi = 0, oo = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          if (b < P.mt - 5 && ((msub >> b) & 1)) {
            oi += 8u << P.msub_in_pos[b];
            oo += 8u << P.msub_out_pos[b];
          }
        }
        msub_in[pm] = oi;
        msub_out[pm] = oo;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[pm][e] = 0.f;
      }
#pragma unroll
      for (int s = 0; s < S; ++s) {
        unsigned ko = 0;
#pragma unroll
        for (int b = 1; b < KB; ++b)
          if ((s >> (b - 1)) & 1) ko += 8u << P.k_in_pos[b];
#pragma unroll
        for (int pm = 0; pm < PM; ++pm) {
          const float2 a = *reinterpret_cast<const float2 *>(ldsb + li + msub_in[pm] + ko);
          acc[pm] = __builtin_amdgcn_mfma_f32_32x32x2f32(W0[s], a.x, acc[pm], 0, 0, 0);
          acc[pm] = __builtin_amdgcn_mfma_f32_32x32x2f32(W1[s], a.y, acc[pm], 0, 0, 0);
        }
      }
    }
    __syncthreads(); // every wave is done reading the input tile

    // ---- accumulators -> LDS in output-tile order (in place over the input tile)
    {
      unsigned lo = lane_out;
      OPAQUE_V(lo);
#pragma unroll
      for (int b = 0; b < 2; ++b)
        if (b < P.wn_log2 && ((wn >> b) & 1)) lo += 8u << P.n_out_pos[4 + b];
      const unsigned o0 = P.nt > 0 ? 8u << P.n_out_pos[0] : 0;
      const unsigned o2 = P.nt > 2 ? 8u << P.n_out_pos[2] : 0;
      const unsigned o3 = P.nt > 3 ? 8u << P.n_out_pos[3] : 0;
#pragma unroll
      for (int pm = 0; pm < PM; ++pm) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
          for (int b0 = 0; b0 < 2; ++b0) {
            const int nl = b0 + 2 * h + 4 * (q & 1) + 8 * (q >> 1);
            if ((nl >> nt_eff) == 0) {
              const unsigned o = lo + msub_out[pm] + b0 * o0 + (q & 1) * o2 + (q >> 1) * o3;
              *reinterpret_cast<float2 *>(ldsb + o) = make_float2(acc[pm][4 * q + 2 * b0], acc[pm][4 * q + 2 * b0 + 1]);
            }
          }
        }
      }
    }
    __syncthreads();

    // ---- copy-out: LDS (linear) -> global (16 B per lane, runs of 2^run_out elements)
    {
      char *Cbase = reinterpret_cast<char *>(C + offC);
      unsigned lo = out_lane, t16 = tid16;
      OPAQUE_V(lo);
      OPAQUE_V(t16);
      for (int i = 0; i < n_out_iters; ++i) {
        long off = 0;
#pragma unroll
        for (int b = 9; b < ARTN_TILE_BITS_MAX; ++b)
          if (b < P.T_out && ((i >> (b - 9)) & 1)) off += P.out_stride[b];
        const f32x4 v = *reinterpret_cast<const f32x4 *>(ldsb + t16 + i * (ARTN_WG_THREADS * 16));
        *reinterpret_cast<f32x4 *>(Cbase + off * 8 + lo) = v;
      }
    }
  }
}

// ----------------------------------------------------------------------------------------
// strided fallback: one thread per C element
// ----------------------------------------------------------------------------------------
template <typename T2, typename T>
__global__ __launch_bounds__(ARTN_WG_THREADS) void artn_k_generic(const T2 *__restrict__ A,
                                                                  const T2 *__restrict__ B,
                                                                  T2 *__restrict__ C,
                                                                  const ArtnGenericPlan G) {
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < G.out_numel;
       idx += (long)gridDim.x * blockDim.x) {
    long r = idx, oa = 0, ob = 0;
    for (int d = 0; d < G.n_out; ++d) {
      const long e = G.out_ext[d];
      const long x = r % e;
      r /= e;
      oa += x * G.out_sA[d];
      ob += x * G.out_sB[d];
    }
    T re = 0, im = 0;
    for (long q = 0; q < G.red_numel; ++q) {
      long rr = q, ka = 0, kbo = 0;
      for (int d = 0; d < G.n_red; ++d) {
        const long e = G.red_ext[d];
        const long x = rr % e;
        rr /= e;
        ka += x * G.red_sA[d];
        kbo += x * G.red_sB[d];
      }
      const T2 a = A[oa + ka], b = B[ob + kbo];
      re += a.x * b.x - a.y * b.y;
      im += a.x * b.y + a.y * b.x;
    }
    T2 o;
    o.x = re;
    o.y = im;
    C[idx] = o;
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

template <int KB>
static hipError_t launch_bits_pm(const ArtnPlan &p, const float2 *A, const float2 *B, float2 *C,
                                 hipStream_t st) {
  dim3 grid(p.info.grid), block(ARTN_WG_THREADS);
  const size_t lds = (size_t)p.info.lds_bytes;
  switch (p.bits.pm) {
    case 1: hipLaunchKernelGGL((artn_k_bits<KB, 1>), grid, block, lds, st, A, B, C, p.bits); break;
    case 2: hipLaunchKernelGGL((artn_k_bits<KB, 2>), grid, block, lds, st, A, B, C, p.bits); break;
    case 4: hipLaunchKernelGGL((artn_k_bits<KB, 4>), grid, block, lds, st, A, B, C, p.bits); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

static hipError_t launch_bits(const ArtnPlan &p, const void *A, const void *B, void *C, hipStream_t st) {
  const float2 *a = (const float2 *)A, *b = (const float2 *)B;
  float2 *c = (float2 *)C;
  switch (p.bits.k) {
    case 1: return launch_bits_pm<1>(p, a, b, c, st);
    case 2: return launch_bits_pm<2>(p, a, b, c, st);
    case 3: return launch_bits_pm<3>(p, a, b, c, st);
    case 4: return launch_bits_pm<4>(p, a, b, c, st);
    case 5: return launch_bits_pm<5>(p, a, b, c, st);
    case 6: return launch_bits_pm<6>(p, a, b, c, st);
  }
  return hipErrorInvalidValue;
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
  int rc = artn::make_plan(d, p, err, g_ncu, !no_bits, min_tiles);
  if (rc) return fail(rc, err);
  *info = p.info;
  return ARTN_OK;
}

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
  hipStream_t st = (hipStream_t)stream;
  if (p.kernel == ARTN_KERNEL_BITS_MFMA) {
    HIP_TRY(launch_bits(p, A, B, C, st));
    return ARTN_OK;
  }
  dim3 grid(p.info.grid), block(ARTN_WG_THREADS);
  if (p.gen.out_numel == 0) return ARTN_OK;
  if (d->dtype == ARTN_C64)
    hipLaunchKernelGGL((artn_k_generic<float2, float>), grid, block, 0, st, (const float2 *)A,
                       (const float2 *)B, (float2 *)C, p.gen);
  else
    hipLaunchKernelGGL((artn_k_generic<double2, double>), grid, block, 0, st, (const double2 *)A,
                       (const double2 *)B, (double2 *)C, p.gen);
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

} // extern "C"
