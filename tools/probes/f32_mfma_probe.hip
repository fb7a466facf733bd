// Development probe (not product): v_mfma_f32_32x32x2_f32 rate by dependent chains per wave and waves per SIMD
// (is a single accumulation chain -- every MFMA waiting for the one before it -- slower than two interleaved ones?)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
template <int CH>
__global__ __launch_bounds__(256) void probe(float *sink, int iters) {
  const int lane = threadIdx.x & 63;
  f32x16 acc[CH];
#pragma unroll
  for (int q = 0; q < CH; ++q)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;
  float a = 1.f + lane * 1e-3f, b = 1.f - lane * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < CH; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
  }
  float t = 0;
#pragma unroll
  for (int q = 0; q < CH; ++q) t += acc[q][0] + acc[q][7];
  if (t == 12345.678f) sink[0] = t;
}
int main() {
  float *sink; CK(hipMalloc(&sink, 16));
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  const int ncu = pr.multiProcessorCount;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](auto kern, int ch, int wg_per_cu) {
    const int iters = 20000;
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
      CK(hipEventRecord(e0)); hipLaunchKernelGGL(kern, dim3(ncu * wg_per_cu), dim3(256), 0, 0, sink, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const double mf = (double)ncu * wg_per_cu * 4 * iters * ch;
    printf("chains/wave %d  waves/SIMD %d : %7.2f TFLOP/s\n", ch, wg_per_cu, mf * 4096 / (best * 1e-3) / 1e12);
  };
  for (int w : {1, 2}) { run(probe<1>, 1, w); run(probe<2>, 2, w); run(probe<3>, 3, w); run(probe<4>, 4, w); }
  return 0;
}
