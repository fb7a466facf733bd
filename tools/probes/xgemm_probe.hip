// Stand-alone timing harness of artn_k_xgemm (diagnostics only; never part of the product): builds the plan of one step whose
// label layout is given low -> high stride per operand (letters K, M, N, H), launches the kernel a few times and prints ms.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DARTN_DEV_SWITCHES -DARTN_DEV_XGPC -Iinclude -Iartensor_amd/csrc [-DXG_ABLATE_MFMA] [-DXG_ABLATE_MEM] [-DXG_ABLATE_STORE]
//         tools/probes/xgemm_probe.hip -o tools/probes/xgemm_probe
//   tools/probes/xgemm_probe KMMMMMMMMKMMMKMMK KNNKNKKNN 3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string>
#include <vector>
#include "artn_plan.h"
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define OPAQUE_V(x) asm volatile("" : "+v"(x))
typedef float v2f_t __attribute__((ext_vector_type(2)));
typedef v2f_t __attribute__((address_space(3))) lds_v2f_t;
typedef __attribute__((address_space(3))) unsigned char lds_byte_t;
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v2f_t lds_read8(unsigned a) { return *(lds_v2f_t *)(unsigned long)a; }
__device__ __forceinline__ void lds_write8(unsigned a, v2f_t v) { *(lds_v2f_t *)(unsigned long)a = v; }
__device__ __forceinline__ void lds_write4(unsigned a, unsigned v) { *(__attribute__((address_space(3))) unsigned *)(unsigned long)a = v; }
#include "artn_xgemm_kernel.h"
#include "artn_xgemm_pc_kernel.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int NB, bool TR>
static int run_pc(const ArtnPlan &p, const float2 *a, const float2 *b, float2 *c, int reps) {
  auto kern = artn_k_xgemm_pc<NB, TR>;
  CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, p.info.lds_bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(p.info.grid), dim3(512), p.info.lds_bytes, 0, a, b, c, p.xg);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(p.info.grid), dim3(512), p.info.lds_bytes, 0, a, b, c, p.xg);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const ArtnXGemmPlan &x = p.xg;
  const double flops = 8.0 * (double)x.m.total * (double)x.n.total * (double)x.k.total;
  printf("PC  M %ld N %ld K %ld nb %d tiles %ld grid %d: %.3f ms  %.1f TFLOP/s\n", (long)x.m.total, (long)x.n.total, (long)x.k.total, x.nb,
         (long)x.n_tiles, p.info.grid, ms, flops / ms / 1e9);
  return 0;
}

template <int NB, bool TR, int KC = 16>
static int run(const ArtnPlan &p, const float2 *a, const float2 *b, float2 *c, int reps) {
  auto kern = artn_k_xgemm<NB, TR, KC>;
  CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, p.info.lds_bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(p.info.grid), dim3(256), p.info.lds_bytes, 0, a, b, c, p.xg);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(p.info.grid), dim3(256), p.info.lds_bytes, 0, a, b, c, p.xg);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const ArtnXGemmPlan &x = p.xg;
  const double flops = 8.0 * (double)x.m.total * (double)x.n.total * (double)x.k.total;
  const double bytes = 8.0 * ((double)x.m.total * x.k.total + (double)x.n.total * x.k.total + (double)x.m.total * x.n.total);
#ifdef XG_STAMPS
  unsigned long long st[64];
  CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(xg_stamp_buf), sizeof(st)));
  for (int t = 0; t < 2; ++t) {
    printf("  tile %d marks (cycles since mark 0):", 8 + t);
    for (int k = 1; k <= 8; ++k) printf(" %d:%lld", k, (long long)(st[t * 16 + k] - st[t * 16]));
    if (t == 1) printf("  | tile period %lld", (long long)(st[16] - st[0]));
    printf("\n");
  }
#endif
  printf("M %ld N %ld K %ld kc %d nb %d amode %d bmode %d trans %d tiles %ld grid %d: %.3f ms  %.1f TFLOP/s  %.2f TB/s\n", (long)x.m.total, (long)x.n.total,
         (long)x.k.total, x.kc, x.nb, x.amode, x.bmode, x.trans, (long)x.n_tiles, p.info.grid, ms, flops / ms / 1e9, bytes / ms / 1e9);
  return 0;
}

int main(int argc, char **argv) {
  if (argc < 4) { printf("usage: xgemm_probe <A layout> <B layout> <D> [reps]\n"); return 1; }
  const std::string al = argv[1], bl = argv[2];
  const int D = atoi(argv[3]), reps = argc > 4 ? atoi(argv[4]) : 5;
  // labels: every letter of A is a label; B's K / H letters pair with A's in order of appearance; C = M labels (A order) lowest, then N (B order), then H
  ArtnStepDesc d;
  memset(&d, 0, sizeof(d));
  d.dtype = ARTN_C64;
  std::vector<int> ak, ah;
  int64_t sa = 1, sb = 1, sc = 1;
  int n = 0;
  for (char ch : al) {
    d.extent[n] = D; d.stride_a[n] = sa; d.stride_b[n] = -1; d.stride_c[n] = -1;
    sa *= D;
    if (ch == 'K') ak.push_back(n);
    if (ch == 'H') ah.push_back(n);
    ++n;
  }
  size_t ik = 0, ih = 0;
  std::vector<int> bn;
  for (char ch : bl) {
    if (ch == 'K') { d.stride_b[ak.at(ik++)] = sb; }
    else if (ch == 'H') { d.stride_b[ah.at(ih++)] = sb; }
    else { d.extent[n] = D; d.stride_a[n] = -1; d.stride_b[n] = sb; d.stride_c[n] = -1; bn.push_back(n); ++n; }
    sb *= D;
  }
  d.n_labels = n;
  for (int l = 0; l < (int)al.size(); ++l) if (al[l] == 'M') { d.stride_c[l] = sc; sc *= D; }
  for (int l : bn) { d.stride_c[l] = sc; sc *= D; }
  for (int l : ah) { d.stride_c[l] = sc; sc *= D; }
  ArtnPlan p;
  memset(&p.info, 0, sizeof(p.info));
  if (!artn::make_xgemm(&d, p, 256, 1)) { printf("declined: %s\n", p.why_generic.c_str()); return 1; }
  float2 *a, *b, *c;
  CK(hipMalloc(&a, sa * 8));
  CK(hipMalloc(&b, sb * 8));
  CK(hipMalloc(&c, sc * 8));
  CK(hipMemset(a, 0x3c, sa * 8));
  CK(hipMemset(b, 0x3c, sb * 8));
  const ArtnXGemmPlan &x = p.xg;
  const float2 *pa = x.swapped ? b : a, *pb = x.swapped ? a : b;
  if (x.pc) {
    switch (x.nb * 2 + (x.trans ? 1 : 0)) {
      case 2: return run_pc<1, false>(p, pa, pb, c, reps);
      case 3: return run_pc<1, true>(p, pa, pb, c, reps);
      case 4: return run_pc<2, false>(p, pa, pb, c, reps);
      case 5: return run_pc<2, true>(p, pa, pb, c, reps);
      case 6: return run_pc<3, false>(p, pa, pb, c, reps);
      case 7: return run_pc<3, true>(p, pa, pb, c, reps);
    }
  }
  if (x.kc == 8) return x.trans ? run<1, true, 8>(p, pa, pb, c, reps) : run<1, false, 8>(p, pa, pb, c, reps);
  switch (x.nb * 2 + (x.trans ? 1 : 0)) {
    case 2: return run<1, false>(p, pa, pb, c, reps);
    case 3: return run<1, true>(p, pa, pb, c, reps);
    case 4: return run<2, false>(p, pa, pb, c, reps);
    case 5: return run<2, true>(p, pa, pb, c, reps);
    case 6: return run<3, false>(p, pa, pb, c, reps);
    case 7: return run<3, true>(p, pa, pb, c, reps);
  }
  return 1;
}
