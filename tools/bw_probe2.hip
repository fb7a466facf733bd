// probe 2: effect of chunks-per-thread and occupancy on copy bandwidth (non-persistent)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
template <int U, bool INTERLEAVE>
__global__ __launch_bounds__(256) void vu(const char* __restrict__ a, char* __restrict__ c) {
  extern __shared__ char lds[];
  const long base = (long)blockIdx.x * (4096L * U);
  f32x4 v[U];
  if (INTERLEAVE) {
    for (int i = 0; i < U; ++i) { v[i] = *(const f32x4*)(a + base + i * 4096 + threadIdx.x * 16); *(f32x4*)(c + base + i * 4096 + threadIdx.x * 16) = v[i]; }
  } else {
    for (int i = 0; i < U; ++i) v[i] = *(const f32x4*)(a + base + i * 4096 + threadIdx.x * 16);
    for (int i = 0; i < U; ++i) *(f32x4*)(c + base + i * 4096 + threadIdx.x * 16) = v[i];
  }
  if (lds[0] == 77 && threadIdx.x == 999) c[0] = 1;
}
int main() {
  const long bytes = 8L << 30;
  char *a, *c;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&c, bytes));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(c, 0, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto launch) {
    launch(); CK(hipDeviceSynchronize());
    float best = 1e9;
    for (int r = 0; r < 5; ++r) { CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
    printf("%-44s %7.3f ms  %7.1f GB/s\n", name, best, 2.0 * bytes / best / 1e6);
  };
#define RUN(U, IL, LDSB) { char nm[80]; snprintf(nm, 80, "U=%d %s lds=%dK", U, IL ? "ld-st interleaved" : "loads then stores", LDSB / 1024); \
    timeit(nm, [&] { hipLaunchKernelGGL((vu<U, IL>), dim3(bytes / (4096L * U)), dim3(256), LDSB, 0, a, c); }); }
  RUN(1, false, 0) RUN(2, false, 0) RUN(4, false, 0) RUN(8, false, 0) RUN(16, false, 0)
  RUN(2, true, 0) RUN(4, true, 0) RUN(8, true, 0)
  RUN(1, false, 16384) RUN(1, false, 32768) RUN(1, false, 65536)
  RUN(8, false, 16384) RUN(8, false, 32768) RUN(8, false, 65536)
  RUN(4, false, 32768) RUN(2, false, 32768)
  return 0;
}
