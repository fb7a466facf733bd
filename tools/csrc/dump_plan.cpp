// Diagnostic (host only): print the tile layout the planner chooses for a step or a fused pair.
#include "artn_plan.h"
#include <cstdio>
extern "C" int artn_dump_plan(const ArtnStepDesc *d1, const ArtnStepDesc *d2) {
  ArtnPlan p;
  std::string err;
  int rc = d2 ? artn::make_plan_fused(d1, d2, p, err, 256, 1) : artn::make_plan(d1, p, err, 256, true, 1);
  if (rc) { printf("planner: rc %d %s\n", rc, err.c_str()); return rc; }
  if (p.kernel != ARTN_KERNEL_BITS_MFMA) { printf("kernel %d\n", p.kernel); return 0; }
  const ArtnBitsPlan &b = p.bits;
  printf("T_in %d T_mid %d T_out %d tiles %ld m3 %d\n", b.T_in, b.T_mid, b.T_out, (long)b.n_tiles, b.m3);
  printf(" in strides (log2):");
  for (int i = 0; i < b.T_in; ++i) printf(" %d", 63 - __builtin_clzl((unsigned long)b.in_stride[i]));
  printf("\n out strides (log2):");
  for (int i = 0; i < b.T_out; ++i) printf(" %d", 63 - __builtin_clzl((unsigned long)b.out_stride[i]));
  printf("\n");
  for (int s = 0; s < b.n_stages; ++s) printf(" stage %d: k %d mt %d nt %d m3 %d\n", s, b.st[s].k, b.st[s].m_bits, b.st[s].nt, b.st[s].m3);
  printf(" run_in %d run_out %d blocked %d nt_loads %d\n outer axes, fastest first (ext: log2 sA, log2 sC, sB1, sB2):", b.run_in, b.run_out, b.blocked, b.nt_loads);
  auto l2 = [](long x) { return x > 0 ? 63 - __builtin_clzl((unsigned long)x) : -1; };
  for (int i = 0; i < b.n_outer; ++i) printf(" [%ld: %d %d %ld %ld]", (long)b.outer[i].ext, l2(b.outer[i].sA), l2(b.outer[i].sC), (long)b.outer[i].sB1, (long)b.outer[i].sB2);
  printf("\n");
  return 0;
}
