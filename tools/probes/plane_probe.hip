// plane_probe.hip -- which access SHAPE does a row-streaming kernel need?  (round 6, artn_k_xrow)
// C[n][m] = A[n][m] for NP planes of M rows (8-byte elements, M odd: no plane is aligned to a cache line), every wave takes
// blocks of R rows round-robin; a wave-instruction moves R rows x (64 / R) planes: R = 16 -> four 128-byte segments,
// R = 32 -> two 256-byte segments, R = 64 -> one 512-byte segment.  D blocks of loads in flight per wave, buffer or global
// instructions, stores with or without the nontemporal hint.   hipcc -O3 --offload-arch=gfx950 plane_probe.hip -o plane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef int v2i __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// One superblock = 64 rows x NP planes.  Loads in the shape RL rows x (64 / RL) planes per wave-instruction, stores in the shape
// RS x (64 / RS) (the stored VALUES are only right when RL == RS: this measures the memory system, not a transpose);
// superblocks round-robin over the waves (or a contiguous range per wave), D superblocks of loads in flight.
template <int RL, int RS, int NPT, int D, bool NT, bool CONTIG>
__global__ __launch_bounds__(256) void k_plane(const float2 *__restrict__ A, float2 *__restrict__ C, unsigned M, unsigned bytes) {
  constexpr int GL = 64 / RL, SL = (NPT + GL - 1) / GL, NL = (64 / RL) * SL;
  constexpr int GS = 64 / RS, SS = (NPT + GS - 1) / GS, NS = (64 / RS) * SS;
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(A), 0, (int)bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(C, 0, (int)bytes, 0x00020000);
  const unsigned n_blocks = (M + 63) / 64, n_waves = 4 * gridDim.x;
  const unsigned wg = gridDim.x % 8u == 0u ? (blockIdx.x % 8u) * (gridDim.x / 8u) + blockIdx.x / 8u : blockIdx.x;
  const unsigned wid = 4 * wg + wave;
  unsigned lo[NL], so[NS]; // byte offset of (row within the superblock, plane) of every instruction; 0xffffffff: no such plane
#pragma unroll
  for (int i = 0; i < NL; ++i) { const unsigned sub = i / SL, s = i % SL, p = s * GL + lane / RL; lo[i] = p < NPT ? (p * M + sub * RL + lane % RL) * 8u : 0xffffffffu; }
#pragma unroll
  for (int i = 0; i < NS; ++i) { const unsigned sub = i / SS, s = i % SS, p = s * GS + lane / RS; so[i] = p < NPT ? (p * M + sub * RS + lane % RS) * 8u : 0xffffffffu; }
  const unsigned per = (n_blocks + n_waves - 1) / n_waves;
  const unsigned n_it = (per + D) / (D + 1) * (D + 1);
  unsigned b = CONTIG ? wid * per : wid;
  const unsigned bstep = CONTIG ? 1 : n_waves, bend = CONTIG ? (wid * per + per < n_blocks ? wid * per + per : n_blocks) : n_blocks;
  v2f x[D + 1][NL];
  unsigned ro[D + 1];
  auto issue = [&](int slot) {
    const unsigned r8 = b < bend ? b * 512u : 0xffffffffu;
    ro[slot] = r8;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const unsigned sub = i / SL;
      const bool ok = r8 != 0xffffffffu && lo[i] != 0xffffffffu && b * 64u + sub * RL + lane % RL < M;
      x[slot][i] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rA, (int)(ok ? r8 + lo[i] : 0xffffffffu), 0, 0));
    }
    b += bstep;
  };
#pragma unroll
  for (int d = 0; d < D; ++d) issue(d);
  for (unsigned it = 0; it < n_it; it += D + 1) {
#pragma unroll
    for (int u = 0; u <= D; ++u) {
      const unsigned bcur = b - (unsigned)D * bstep;
      issue((u + D) % (D + 1));
      const unsigned r8 = ro[u];
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const unsigned sub = i / SS;
        const bool ok = r8 != 0xffffffffu && so[i] != 0xffffffffu && bcur * 64u + sub * RS + lane % RS < M;
        v2f v = x[u][i % NL];
        v.x += 1.0f;
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, v), rC, (int)(ok ? r8 + so[i] : 0xffffffffu), 0, NT ? 2 : 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

template <int RL, int RS, int NPT, int D, bool NT, bool CONTIG>
static void run(const char *name, const float2 *A, float2 *C, unsigned M, int wg_per_cu) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const unsigned bytes = M * NPT * 8u;
  const int grid = 256 * wg_per_cu;
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k_plane<RL, RS, NPT, D, NT, CONTIG>), dim3(grid), dim3(256), 0, 0, A, C, M, bytes);
  CK(hipEventRecord(e0));
  const int reps = 5;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_plane<RL, RS, NPT, D, NT, CONTIG>), dim3(grid), dim3(256), 0, 0, A, C, M, bytes);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  printf("planes %2d rows %8u  load %2d-row  store %2d-row  D=%d %s%s wg/cu %d  %7.3f ms  %6.2f TB/s   %s\n", NPT, M, RL, RS, D, NT ? "nt " : "   ", CONTIG ? "contig " : "       ", wg_per_cu, ms, 2.0 * bytes / ms * 1e-9, name);
  fflush(stdout);
}

int main(int argc, char **argv) {
  const unsigned M = 14348907u /* 3^15 */, NP = 27;
  const size_t bytes = (size_t)M * NP * 8;
  float2 *A, *C;
  CK(hipMalloc(&A, bytes)); CK(hipMalloc(&C, bytes));
  std::vector<float2> h((size_t)1 << 20);
  for (auto &v : h) v = float2{(float)rand() / RAND_MAX, (float)rand() / RAND_MAX};
  for (size_t o = 0; o < bytes; o += h.size() * 8) CK(hipMemcpy((char *)A + o, h.data(), std::min(h.size() * 8, bytes - o), hipMemcpyHostToDevice));
  // check one variant's result
  hipLaunchKernelGGL((k_plane<16, 16, 27, 1, true, false>), dim3(1024), dim3(256), 0, 0, A, C, M, (unsigned)bytes);
  CK(hipDeviceSynchronize());
  { std::vector<float2> a(1000), c(1000); size_t o = ((size_t)M * 13 + 777777) * 8;
    CK(hipMemcpy(a.data(), (char *)A + o, 8000, hipMemcpyDeviceToHost)); CK(hipMemcpy(c.data(), (char *)C + o, 8000, hipMemcpyDeviceToHost));
    for (int i = 0; i < 1000; ++i) if (c[i].x != a[i].x + 1.0f || c[i].y != a[i].y) { printf("MISMATCH %d\n", i); return 1; } }
  const int w = 4;
  run<16, 16, 27, 1, true, false>("", A, C, M, w);
  run<16, 16, 27, 1, false, false>("", A, C, M, w);
  run<32, 32, 27, 1, false, false>("", A, C, M, w);
  run<64, 64, 27, 1, false, false>("", A, C, M, w);
  run<64, 64, 27, 1, false, false>("", A, C, M, 2);
  run<64, 64, 27, 1, false, false>("", A, C, M, 3);
  run<16, 64, 27, 1, false, false>("which side matters", A, C, M, w);
  run<64, 16, 27, 1, false, false>("", A, C, M, w);
  run<32, 64, 27, 1, false, false>("", A, C, M, w);
  run<64, 32, 27, 1, false, false>("", A, C, M, w);
  run<64, 64, 27, 2, false, false>("deeper", A, C, M, 2);
  // 9 planes of 3^16 rows (the 9 x 9 step of the bond-dimension-3 network)
  const unsigned M9 = 3 * M;
  run<16, 16, 9, 1, false, false>("", A, C, M9, w);
  run<16, 16, 9, 3, false, false>("", A, C, M9, w);
  run<16, 16, 9, 3, false, false>("", A, C, M9, 8);
  run<32, 32, 9, 1, false, false>("", A, C, M9, w);
  run<32, 32, 9, 3, false, false>("", A, C, M9, 8);
  run<64, 64, 9, 1, false, false>("", A, C, M9, w);
  run<64, 64, 9, 3, false, false>("", A, C, M9, w);
  run<64, 64, 9, 3, false, false>("", A, C, M9, 8);
  run<64, 64, 9, 3, true, false>("", A, C, M9, 8);
  run<16, 64, 9, 3, false, false>("", A, C, M9, 8);
  run<64, 16, 9, 3, false, false>("", A, C, M9, 8);
  return 0;
}
