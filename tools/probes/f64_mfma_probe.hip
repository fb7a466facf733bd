// Development probe (not product): what rate do back-to-back v_mfma_f64_16x16x4_f64 reach, by independent chains per wave
// and waves per SIMD?   hipcc -O3 --offload-arch=gfx950 tools/probes/f64_mfma_probe.hip -o tools/probes/f64_mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double f64x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
template <int CH>
__global__ __launch_bounds__(256) void probe(float *sink, int iters) {
  const int lane = threadIdx.x & 63;
  f64x4 acc[CH];
#pragma unroll
  for (int q = 0; q < CH; ++q) acc[q] = f64x4{0.0, 0.0, 0.0, 0.0};
  double a = 1.0 + lane * 1e-3, b = 1.0 - lane * 1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < CH; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
  }
  double t = 0;
#pragma unroll
  for (int q = 0; q < CH; ++q) t += acc[q][0] + acc[q][3];
  if (t == 12345.678) sink[0] = (float)t;
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
    printf("chains/wave %d  waves/SIMD %d : %7.2f TFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", ch, wg_per_cu, mf * 2048 / (best * 1e-3) / 1e12,
           best * 1e-3 * 2.4e9 / ((double)wg_per_cu * iters * ch));
  };
  for (int w : {1, 2, 4}) { run(probe<1>, 1, w); run(probe<2>, 2, w); run(probe<4>, 4, w); run(probe<8>, 8, w); }
  return 0;
}
