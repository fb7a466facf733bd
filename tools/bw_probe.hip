// Development probe (not product): what HBM rate do copy kernels shaped like artn_k_bits reach?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// V0: classic grid-stride float4 copy
__global__ __launch_bounds__(256) void v0(const f32x4* __restrict__ a, f32x4* __restrict__ c, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) c[i] = a[i];
}
// tile copy: tile = 2048 chunks of 16 B; chunk c of tile t lives at byte offset f(t, c).
// mode 0: contiguous 32 KiB tiles. mode 1: 8 streams of 4 KiB (strides like n30 step 78).
// mode 2: 128-B runs, 256 of them, scattered with stride 2^(7+sh).
template <int MODE, bool LDS, bool PREF>
__global__ __launch_bounds__(256) void vt(const char* __restrict__ a, char* __restrict__ c, long ntiles, int sh) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const unsigned tid = threadIdx.x;
  auto off = [&](long t, int i) -> long {
    // chunk index within tile: ch = tid + 256*i  (0..2047), 16 B each
    unsigned ch = tid + 256u * i;
    if (MODE == 0) return t * 32768 + (long)ch * 16;
    if (MODE == 1) { // 4 KiB blocks: 256 chunks; i selects stream (3 bits) -> bits 27,28,30 of byte address
      long s = ((long)(i & 1) << 27) | ((long)((i >> 1) & 1) << 28) | ((long)((i >> 2) & 1) << 30);
      // tile index fills the remaining bits: low 15 bits -> byte bits 12..26, next 1 -> bit 29, next -> 31,32
      long lo = (t & 0x7fff) << 12, r = t >> 15;
      long hi = ((r & 1) << 29) | (((r >> 1) & 3) << 31);
      return s | lo | hi | ((long)tid * 16);
    }
    // MODE 2: run = 8 lanes x 16 B = 128 B; run id = ch >> 3 (0..255); tile bits: 5 low run bits contiguous, 3 far
    unsigned run = ch >> 3, in = ch & 7;
    long rlo = run & 31, rhi = run >> 5;
    long base = (t & ((1L << sh) - 1)) << 12 | (rhi << (12 + sh)) | ((t >> sh) << (15 + sh));
    return base + rlo * 128 + in * 16;
  };
  f32x4 v[8];
  long t = blockIdx.x;
  if (PREF && t < ntiles) for (int i = 0; i < 8; ++i) v[i] = *(const f32x4*)(a + off(t, i));
  for (; t < ntiles; t += gridDim.x) {
    if (!PREF) for (int i = 0; i < 8; ++i) v[i] = *(const f32x4*)(a + off(t, i));
    if (LDS) {
      __syncthreads();
      for (int i = 0; i < 8; ++i) *(f32x4*)(lds + tid * 16 + i * 4096) = v[i];
      __syncthreads();
    }
    f32x4 w[8];
    if (LDS) { for (int i = 0; i < 8; ++i) w[i] = *(f32x4*)(lds + ((tid * 16 + i * 4096) ^ 0x10)); }
    else for (int i = 0; i < 8; ++i) w[i] = v[i];
    long tn = t + gridDim.x;
    if (PREF && tn < ntiles) for (int i = 0; i < 8; ++i) v[i] = *(const f32x4*)(a + off(tn, i));
    for (int i = 0; i < 8; ++i) *(f32x4*)(c + off(t, i)) = w[i];
  }
}
// blocked assignment: WG b copies tiles [b*tpw, (b+1)*tpw) (contiguous), one tile (32 KiB) at a time
template <bool LDS, bool PREF>
__global__ __launch_bounds__(256) void vb(const char* __restrict__ a, char* __restrict__ c, int tpw) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const unsigned tid = threadIdx.x;
  f32x4 v[8];
  long t0 = (long)blockIdx.x * tpw;
  auto off = [&](long t, int i) -> long { return t * 32768 + (long)(tid + 256u * i) * 16; };
  if (PREF) for (int i = 0; i < 8; ++i) v[i] = *(const f32x4*)(a + off(t0, i));
  for (int k = 0; k < tpw; ++k) {
    long t = t0 + k;
    if (!PREF) for (int i = 0; i < 8; ++i) v[i] = *(const f32x4*)(a + off(t, i));
    f32x4 w[8];
    if (LDS) {
      __syncthreads();
      for (int i = 0; i < 8; ++i) *(f32x4*)(lds + tid * 16 + i * 4096) = v[i];
      __syncthreads();
      for (int i = 0; i < 8; ++i) w[i] = *(f32x4*)(lds + ((tid * 16 + i * 4096) ^ 0x10));
    } else for (int i = 0; i < 8; ++i) w[i] = v[i];
    if (PREF && k + 1 < tpw) for (int i = 0; i < 8; ++i) v[i] = *(const f32x4*)(a + off(t + 1, i));
    for (int i = 0; i < 8; ++i) *(f32x4*)(c + off(t, i)) = w[i];
  }
}
int main(int argc, char** argv) {
  const long bytes = 8L << 30;
  char *a, *c;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&c, bytes));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(c, 0, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto launch) {
    launch(); CK(hipDeviceSynchronize());
    float best = 1e9;
    for (int r = 0; r < 5; ++r) { CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
    printf("%-40s %7.3f ms  %7.1f GB/s (r+w)\n", name, best, 2.0 * bytes / best / 1e6);
  };
  const long ntiles = bytes / 32768;
  for (int g : {1024, 2048, 4096}) {
    char nm[64]; snprintf(nm, 64, "v0 float4 grid-stride grid=%d", g);
    timeit(nm, [&] { hipLaunchKernelGGL(v0, dim3(g), dim3(256), 0, 0, (const f32x4*)a, (f32x4*)c, bytes / 16); });
  }
  timeit("v0 float4 one-chunk-per-thread", [&] { hipLaunchKernelGGL(v0, dim3(bytes / 16 / 256), dim3(256), 0, 0, (const f32x4*)a, (f32x4*)c, bytes / 16); });
  int g = 1024;
  timeit("tile contiguous, regs", [&] { hipLaunchKernelGGL((vt<0, false, false>), dim3(g), dim3(256), 0, 0, a, c, ntiles, 0); });
  timeit("tile contiguous, regs, prefetch", [&] { hipLaunchKernelGGL((vt<0, false, true>), dim3(g), dim3(256), 0, 0, a, c, ntiles, 0); });
  timeit("tile contiguous, LDS", [&] { hipLaunchKernelGGL((vt<0, true, false>), dim3(g), dim3(256), 32768, 0, a, c, ntiles, 0); });
  timeit("tile contiguous, LDS, prefetch", [&] { hipLaunchKernelGGL((vt<0, true, true>), dim3(g), dim3(256), 32768, 0, a, c, ntiles, 0); });
  timeit("tile 8x4KiB streams, regs", [&] { hipLaunchKernelGGL((vt<1, false, false>), dim3(g), dim3(256), 0, 0, a, c, ntiles, 0); });
  timeit("tile 8x4KiB streams, LDS, prefetch", [&] { hipLaunchKernelGGL((vt<1, true, true>), dim3(g), dim3(256), 32768, 0, a, c, ntiles, 0); });
  for (int sh : {0, 3, 8, 12}) {
    char nm[64]; snprintf(nm, 64, "tile 128B runs sh=%d, LDS, prefetch", sh);
    timeit(nm, [&] { hipLaunchKernelGGL((vt<2, true, true>), dim3(g), dim3(256), 32768, 0, a, c, ntiles, sh); });
  }
  for (int gg : {256, 512, 2048, 4096}) {
    char nm[64]; snprintf(nm, 64, "tile contiguous, LDS, prefetch grid=%d", gg);
    timeit(nm, [&] { hipLaunchKernelGGL((vt<0, true, true>), dim3(gg), dim3(256), 32768, 0, a, c, ntiles, 0); });
  }
  for (int tpw : {1, 2, 4, 8, 16, 64, 256}) {
    char nm[64];
    snprintf(nm, 64, "blocked tiles/WG=%d regs", tpw);
    timeit(nm, [&] { hipLaunchKernelGGL((vb<false, false>), dim3(ntiles / tpw), dim3(256), 0, 0, a, c, tpw); });
    snprintf(nm, 64, "blocked tiles/WG=%d LDS", tpw);
    timeit(nm, [&] { hipLaunchKernelGGL((vb<true, false>), dim3(ntiles / tpw), dim3(256), 32768, 0, a, c, tpw); });
    snprintf(nm, 64, "blocked tiles/WG=%d LDS pref", tpw);
    timeit(nm, [&] { hipLaunchKernelGGL((vb<true, true>), dim3(ntiles / tpw), dim3(256), 32768, 0, a, c, tpw); });
  }
  // strided assignment with many WGs (grid = ntiles / tpw, tile = b + k*grid)
  for (int tpw : {1, 2, 4, 8}) {
    char nm[64];
    snprintf(nm, 64, "strided tiles/WG=%d LDS pref", tpw);
    timeit(nm, [&] { hipLaunchKernelGGL((vt<0, true, true>), dim3(ntiles / tpw), dim3(256), 32768, 0, a, c, ntiles, 0); });
  }
  return 0;
}
