// Development probe (round 6): does the fp32 MFMA shape change the clock the chip holds under load?  Bare loops of
// v_mfma_f32_32x32x2_f32 and v_mfma_f32_16x16x4_f32 (the same 64 FLOP per clock and SIMD), operands in registers,
// random or all-zero data, three accumulation chains per wave (the 3M stages), 1 or 2 waves per SIMD, ~1 s per row.
// MI355X_MICROARCH.md (DVFS give-back 7) measured 1.12-1.15 x for the bf16 pair of shapes on random data.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__device__ __forceinline__ float rnd(unsigned &s) { s = s * 1664525u + 1013904223u; return ((int)(s >> 9) - (1 << 22)) * (1.0f / (1 << 22)); }
template <int SHAPE, int NOPS>
__global__ __launch_bounds__(256) void probe(float *sink, int iters, int zero, unsigned long long *clk) {
  unsigned seed = threadIdx.x * 2654435761u + blockIdx.x * 97u + 1u;
  float a[NOPS], b[NOPS];
#pragma unroll
  for (int i = 0; i < NOPS; ++i) { a[i] = zero ? 0.f : rnd(seed); b[i] = zero ? 0.f : rnd(seed); }
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float t = 0;
  if constexpr (SHAPE == 32) {
    f32x16 acc[3];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NOPS; ++i)
#pragma unroll
        for (int q = 0; q < 3; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[(i + q) % NOPS], acc[q], 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) t += acc[q][0] + acc[q][7];
  } else {
    f32x4 acc[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NOPS; ++i)
#pragma unroll
        for (int q = 0; q < 6; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[(i + q) % NOPS], acc[q], 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) t += acc[q][0] + acc[q][3];
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x < 512) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
  if (t == 12345.678f) sink[0] = t;
}
int main() {
  float *sink; CK(hipMalloc(&sink, 16));
  unsigned long long *clk; CK(hipMalloc(&clk, 1024 * 8));
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  const int ncu = pr.multiProcessorCount;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char *name, auto kern, double flop_per_iter_wave, int wg_per_cu, int zero) {
    const int iters = 60000;
    float best = 1e9; double ghz = 0;
    for (int r = 0; r < 4; ++r) {
      CK(hipEventRecord(e0)); hipLaunchKernelGGL(kern, dim3(ncu * wg_per_cu), dim3(256), 0, 0, sink, iters, zero, clk); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r > 0 && ms < best) {
        best = ms;
        unsigned long long h[1024]; CK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
        double s = 0; int n = 0;
        for (int b = 0; b < 512 && b < ncu * wg_per_cu; ++b) { s += (double)h[2 * b] / ((double)h[2 * b + 1] * 10.0); ++n; }   // shader cycles per ns
        ghz = s / n;
      }
    }
    const double flop = (double)ncu * wg_per_cu * 4 * iters * flop_per_iter_wave;
    printf("%-22s %s data  waves/SIMD %d : %7.2f TFLOP/s  in-kernel clock %.3f GHz  (%.1f ms)\n", name, zero ? "zero  " : "random", wg_per_cu, flop / (best * 1e-3) / 1e12, ghz, best);
  };
  for (int zero : {0, 1})
    for (int w : {1, 2}) {
      run("32x32x2, 3 chains", probe<32, 4>, 4 * 3 * 4096.0, w, zero);
      run("16x16x4, 6 chains", probe<16, 4>, 4 * 6 * 2048.0, w, zero);
    }
  return 0;
}
