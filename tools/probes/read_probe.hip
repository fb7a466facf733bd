// Development probe (not product): HBM READ rate of this chip (8 GiB, every 16-byte chunk read once, results folded into a
// few bytes), by chunks per thread and workgroups per CU -- the ceiling of reduction-like steps (a big first operand, a small result).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
template <int U>
__global__ __launch_bounds__(256) void rd(const char* __restrict__ a, float* __restrict__ sink, long n16) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (long base = (long)blockIdx.x * 256 * U; base < n16; base += (long)gridDim.x * 256 * U) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load((const f32x4*)(a + (base + u * 256 + threadIdx.x) * 16));
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}
int main() {
  const long bytes = 8L << 30, n16 = bytes / 16;
  char* a; float* sink;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&sink, 16)); CK(hipMemset(a, 1, bytes));
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* nm, auto kern, int grid) {
    float best = 1e9;
    for (int r = 0; r < 4; ++r) {
      CK(hipEventRecord(e0)); hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, a, sink, n16); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%-40s grid %6d : %7.3f ms  %7.1f GB/s\n", nm, grid, best, bytes / best / 1e6);
  };
  const int ncu = pr.multiProcessorCount;
  for (int w : {2, 4, 8, 16}) {
    run("1 chunk per thread and trip", rd<1>, ncu * w);
    run("4 chunks per thread and trip", rd<4>, ncu * w);
    run("8 chunks per thread and trip", rd<8>, ncu * w);
  }
  run("one trip: 4 chunks per thread", rd<4>, (int)(n16 / (256 * 4)));
  run("one trip: 1 chunk per thread", rd<1>, (int)(n16 / 256));
  return 0;
}
