// Diagnostic: which SIMD does each wave of a 512-thread workgroup land on?  (HW_ID: wave_id [3:0], simd_id [5:4], cu_id [11:8])
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(512, 1) void probe(unsigned *out) {
  extern __shared__ char smem[];
  const unsigned id = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4); // HW_ID, 32 bits
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
  if (threadIdx.x == 9999) smem[0] = 1;
}
int main() {
  unsigned *d, h[256 * 8];
  hipMalloc(&d, sizeof(h));
  hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
  hipLaunchKernelGGL(probe, dim3(256), dim3(512), 140 * 1024, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int pattern[8][4] = {{0}};
  for (int b = 0; b < 256; ++b)
    for (int w = 0; w < 8; ++w) pattern[w][(h[b * 8 + w] >> 4) & 3]++;
  for (int w = 0; w < 8; ++w) printf("wave %d: simd0 %d simd1 %d simd2 %d simd3 %d\n", w, pattern[w][0], pattern[w][1], pattern[w][2], pattern[w][3]);
  for (int b = 0; b < 4; ++b) {
    printf("block %d:", b);
    for (int w = 0; w < 8; ++w) printf(" w%d(simd %u slot %u cu %u)", w, (h[b * 8 + w] >> 4) & 3, h[b * 8 + w] & 15, (h[b * 8 + w] >> 8) & 15);
    printf("\n");
  }
  return 0;
}
