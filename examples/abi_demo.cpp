// Using libartn_hip.so from C/C++ with no Python and no torch: one pairwise contraction
//     C[a b d e x] = sum_{c f} A[a b c d e f ...] * B[f c x]
// through artn_contract, checked against a host loop.
//
//   hipcc --offload-arch=gfx950 -Iinclude examples/abi_demo.cpp -Lartensor_amd -lartn_hip \
//         -Wl,-rpath,$PWD/artensor_amd -o /tmp/abi_demo && /tmp/abi_demo
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "artn.h"

#define CHECK_HIP(x)                                                        \
  do {                                                                      \
    hipError_t e_ = (x);                                                    \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } \
  } while (0)

int main() {
  if (artn_abi_version() != ARTN_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 2; }
  if (artn_device_count() < 1) { fprintf(stderr, "no gfx950 device\n"); return 3; }
  // A: 22 binary labels (2^22 complex64 = 32 MiB); B: labels {21, 3, 22}; label 3 and 21 contracted
  const int RA = 22;
  const int64_t NA = int64_t(1) << RA;
  ArtnStepDesc d;
  memset(&d, 0, sizeof(d));
  d.dtype = ARTN_C64;
  d.n_labels = RA + 1;
  const int K0 = 3, K1 = 21, X = 22; // label ids
  int64_t sc = 1;
  // C labels: all of A's except K0 and K1, then X fastest
  std::vector<int> c_labels;
  for (int l = 0; l < RA; ++l) if (l != K0 && l != K1) c_labels.push_back(l);
  c_labels.push_back(X);
  for (int l = 0; l <= RA; ++l) { d.extent[l] = 2; d.stride_a[l] = d.stride_b[l] = d.stride_c[l] = -1; }
  for (int l = 0; l < RA; ++l) d.stride_a[l] = int64_t(1) << (RA - 1 - l); // row-major, label 0 slowest
  d.stride_b[K1] = 4; d.stride_b[K0] = 2; d.stride_b[X] = 1;                 // B[K1][K0][X]
  for (int i = (int)c_labels.size() - 1; i >= 0; --i) { d.stride_c[c_labels[i]] = sc; sc *= 2; }
  const int64_t NC = sc;

  ArtnStepInfo info;
  if (artn_contract_query(&d, &info) != ARTN_OK) { fprintf(stderr, "query: %s\n", artn_last_error()); return 2; }
  printf("planner: kernel=%d k_bits=%d tile_in=2^%d tiles=%lld lds=%d B\n", info.kernel, info.k_bits, info.tile_in_bits,
         (long long)info.n_tiles, info.lds_bytes);

  std::vector<float> hA(2 * NA), hB(2 * 8), hC(2 * NC);
  srand(1);
  for (auto &v : hA) v = (float)(rand() % 2001) / 2000.f - 0.5f;
  for (auto &v : hB) v = (float)(rand() % 2001) / 2000.f - 0.5f;
  float *dA, *dB, *dC;
  CHECK_HIP(hipMalloc(&dA, hA.size() * 4));
  CHECK_HIP(hipMalloc(&dB, hB.size() * 4));
  CHECK_HIP(hipMalloc(&dC, hC.size() * 4));
  CHECK_HIP(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
  hipStream_t st;
  CHECK_HIP(hipStreamCreate(&st));
  if (artn_contract(&d, dA, dB, dC, st) != ARTN_OK) { fprintf(stderr, "contract: %s\n", artn_last_error()); return 2; }
  CHECK_HIP(hipStreamSynchronize(st));
  CHECK_HIP(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));

  // host check on a sample of output elements
  double worst = 0, scale = 0;
  for (int64_t s = 0; s < 4096; ++s) {
    const int64_t ci = (s * 2654435761u) % NC;
    int64_t a_base = 0, r = ci;
    const int x = (int)(r & 1);
    r >>= 1;
    for (int i = (int)c_labels.size() - 2; i >= 0; --i) { if (r & 1) a_base += d.stride_a[c_labels[i]]; r >>= 1; }
    double re = 0, im = 0;
    for (int k1 = 0; k1 < 2; ++k1)
      for (int k0 = 0; k0 < 2; ++k0) {
        const int64_t ai = a_base + k1 * d.stride_a[K1] + k0 * d.stride_a[K0];
        const int bi = k1 * 4 + k0 * 2 + x;
        const double ar = hA[2 * ai], aim = hA[2 * ai + 1], br = hB[2 * bi], bim = hB[2 * bi + 1];
        re += ar * br - aim * bim;
        im += ar * bim + aim * br;
      }
    worst = fmax(worst, fmax(fabs(re - hC[2 * ci]), fabs(im - hC[2 * ci + 1])));
    scale = fmax(scale, fmax(fabs(re), fabs(im)));
  }
  printf("max |diff| = %.3g (scale %.3g)\n", worst, scale);
  CHECK_HIP(hipFree(dA));
  CHECK_HIP(hipFree(dB));
  CHECK_HIP(hipFree(dC));
  return worst <= 1e-5 * scale ? 0 : 1;
}
