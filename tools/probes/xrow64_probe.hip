// xrow64_probe.hip -- would a 64-ROW shape pay for artn_k_xrow?  (round 6)
// C[n][m] = sum_k W[n][k] A[k][m] for K planes of M rows into N planes (8-byte complex elements, M odd), 3M arithmetic on
// v_mfma_f32_16x16x4_f32 as in the kernel -- but lane l of a wave is ROW 64 b + l: every load and store instruction moves 512
// contiguous bytes, and v_permlane16_swap / v_permlane32_swap butterflies turn four loads (contracted values 4 s .. 4 s + 3 of
// 64 rows) into the four 16-row MFMA operands (rows 16 q .. 16 q + 15, lane group g = contracted value 4 s + g) and the four
// blocks' accumulators back into 64-row columns.  Loads of the next superblock are issued group by group as the MFMAs of this
// one free their registers.     hipcc -O3 -std=c++17 --offload-arch=gfx950 xrow64_probe.hip -o xrow64_probe
#include <hip/hip_runtime.h>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// registers i = 0..3, lane groups q = 0..3:  out[i] group q = in[q] group i  (a 4 x 4 transpose of 16-lane groups).
// v_permlane16_swap: odd 16-lane groups of the first register <-> even groups of the second; v_permlane32_swap: upper 32 lanes of
// the first <-> lower 32 of the second.  Inline assembly: this hipcc's __builtin_amdgcn_permlane16_swap drops the second result
// (only `extractvalue 0` of the intrinsic's pair reaches the IR); the s_nop cover the VALU -> permlane -> VALU / MFMA wait states
// the hazard recognizer cannot see inside an asm statement.
__device__ __forceinline__ void butterfly(float &a0, float &a1, float &a2, float &a3) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\ts_nop 1\n\t"
               "v_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\ts_nop 1"
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
}

template <int S, int NBK, int WAVES>
__global__ __launch_bounds__(256, WAVES) void k_xrow64(const float2 *__restrict__ A, const float2 *__restrict__ W, float2 *__restrict__ C,
                                                     unsigned M, unsigned K, unsigned N, unsigned bytes_a, unsigned bytes_c) {
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned j = lane & 15, g = lane >> 4;
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(A), 0, (int)bytes_a, 0x00020000);
  const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(C, 0, (int)bytes_c, 0x00020000);
  float wr[NBK][S], wi[NBK][S];
#pragma unroll
  for (int s = 0; s < S; ++s)
#pragma unroll
    for (int blk = 0; blk < NBK; ++blk) {
      const unsigned n = 16u * blk + j, k = 4u * s + g;
      const float2 w = (n < N && k < K) ? W[n * K + k] : float2{0.f, 0.f};
      wr[blk][s] = w.x; wi[blk][s] = w.y;
    }
  const unsigned n_sb = (M + 63) / 64, per_it = 4 * gridDim.x;
  const unsigned wg = gridDim.x % 8u == 0u ? (blockIdx.x % 8u) * (gridDim.x / 8u) + blockIdx.x / 8u : blockIdx.x;
  const unsigned n_it = (n_sb + per_it - 1) / per_it;
  const unsigned M8 = M * 8u;
  unsigned m = 64u * (4u * wg + wave) + lane;
  float xr[S][4], xi[S][4];
  auto issue = [&](int s, unsigned row8) { // the four loads of contracted values 4 s .. 4 s + 3 (64 consecutive rows each)
#pragma unroll
    for (int gg = 0; gg < 4; ++gg) {
      const unsigned k = 4u * s + gg;
      const v2f v = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rA, (int)(k < K ? row8 : 0xffffffffu), (int)(k * M8), 0));
      xr[s][gg] = v.x; xi[s][gg] = v.y;
    }
  };
  unsigned row8 = m < M ? m * 8u : 0xffffffffu;
#pragma unroll
  for (int s = 0; s < S; ++s) issue(s, row8);
  for (unsigned it = 0; it < n_it; ++it) {
    const unsigned cur8 = row8;
    m += 64u * per_it;
    row8 = m < M ? m * 8u : 0xffffffffu;
    f32x4 t1[4][NBK], t2[4][NBK], t3[4][NBK];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int blk = 0; blk < NBK; ++blk) t1[q][blk] = t2[q][blk] = t3[q][blk] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < S; ++s) {
      butterfly(xr[s][0], xr[s][1], xr[s][2], xr[s][3]); // x[s][q]: rows 16 q .. 16 q + 15, lane group g = contracted value 4 s + g
      butterfly(xi[s][0], xi[s][1], xi[s][2], xi[s][3]);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float xs = xr[s][q] + xi[s][q];
#pragma unroll
        for (int blk = 0; blk < NBK; ++blk) {
          const float ws = wr[blk][s] + wi[blk][s];
          t1[q][blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[blk][s], xr[s][q], t1[q][blk], 0, 0, 0);
          t2[q][blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(wi[blk][s], xi[s][q], t2[q][blk], 0, 0, 0);
          t3[q][blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(ws, xs, t3[q][blk], 0, 0, 0);
        }
      }
      issue(s, row8); // the next superblock's loads of this group: its registers are free
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int blk = 0; blk < NBK; ++blk)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float re[4], im[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { re[q] = t1[q][blk][r] - t2[q][blk][r]; im[q] = t3[q][blk][r] - t1[q][blk][r] - t2[q][blk][r]; }
        butterfly(re[0], re[1], re[2], re[3]); // re[gg]: column 16 blk + 4 gg + r of the 64 rows
        butterfly(im[0], im[1], im[2], im[3]);
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
          const unsigned n = 16u * blk + 4u * gg + (unsigned)r;
          const v2f val = {re[gg], im[gg]};
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, val), rC, (int)(n < N ? cur8 : 0xffffffffu), (int)(n * M8), 0);
        }
      }
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int S, int NBK, int WAVES>
static void run(const float2 *A, const float2 *W, float2 *C, unsigned M, unsigned K, unsigned N, const std::vector<float2> &hA, const std::vector<float2> &hW) {
  const unsigned ba = M * K * 8u, bc = M * N * 8u;
  const int grid = 256 * WAVES;
  hipLaunchKernelGGL((k_xrow64<S, NBK, WAVES>), dim3(grid), dim3(256), 0, 0, A, W, C, M, K, N, ba, bc);
  CK(hipDeviceSynchronize());
  // check rows near the start, the middle and the end
  double worst = 0;
  for (unsigned m : {0u, 1u, 17u, 63u, 64u, 1000003u, M - 65, M - 2, M - 1}) {
    std::vector<float2> a(K), c(N);
    for (unsigned k = 0; k < K; ++k) CK(hipMemcpy(&a[k], (const char *)A + ((size_t)k * M + m) * 8, 8, hipMemcpyDeviceToHost));
    for (unsigned n = 0; n < N; ++n) CK(hipMemcpy(&c[n], (const char *)C + ((size_t)n * M + m) * 8, 8, hipMemcpyDeviceToHost));
    for (unsigned n = 0; n < N; ++n) {
      std::complex<double> s = 0;
      for (unsigned k = 0; k < K; ++k) s += std::complex<double>(hW[n * K + k].x, hW[n * K + k].y) * std::complex<double>(a[k].x, a[k].y);
      worst = std::max(worst, std::abs(s - std::complex<double>(c[n].x, c[n].y)));
    }
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  const int reps = 5;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_xrow64<S, NBK, WAVES>), dim3(grid), dim3(256), 0, 0, A, W, C, M, K, N, ba, bc);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  printf("K %2u -> N %2u on %8u rows  S=%d NBK=%d waves/SIMD %d  %7.3f ms  %6.2f TB/s  worst error %.2e\n", K, N, M, S, NBK, WAVES, ms, (double)(ba + bc) / ms * 1e-9, worst);
  fflush(stdout);
}

int main() {
  const unsigned M27 = 14348907u /* 3^15 */, M9 = 43046721u /* 3^16 */;
  const size_t bytes = (size_t)M27 * 27 * 8;
  float2 *A, *C, *W;
  CK(hipMalloc(&A, bytes)); CK(hipMalloc(&C, bytes)); CK(hipMalloc(&W, 32 * 32 * 8));
  std::vector<float2> h((size_t)1 << 20), hW(32 * 32);
  for (auto &v : h) v = float2{(float)(rand() % 2001 - 1000) * 1e-3f, (float)(rand() % 2001 - 1000) * 1e-3f};
  for (auto &v : hW) v = float2{(float)(rand() % 2001 - 1000) * 1e-3f, (float)(rand() % 2001 - 1000) * 1e-3f};
  for (size_t o = 0; o < bytes; o += h.size() * 8) CK(hipMemcpy((char *)A + o, h.data(), std::min(h.size() * 8, bytes - o), hipMemcpyHostToDevice));
  CK(hipMemcpy(W, hW.data(), hW.size() * 8, hipMemcpyHostToDevice));
  run<3, 1, 4>(A, W, C, M9, 9, 9, h, hW);
  run<3, 1, 5>(A, W, C, M9, 9, 9, h, hW);
  run<3, 1, 3>(A, W, C, M9, 9, 9, h, hW);
  run<7, 2, 2>(A, W, C, M27, 27, 27, h, hW);
  run<7, 1, 2>(A, W, C, M27, 27, 16, h, hW);
  return 0;
}
