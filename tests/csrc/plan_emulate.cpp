// CPU emulation of artn_k_bits (artensor_amd/csrc/artn_kernels.hip), TEST-ONLY.
// It replays the kernel thread by thread -- copy-in chunks, per-lane fragment offsets,
// the 32x32x2 MFMA lane maps of the gfx950 guide, the scatter into the second LDS region,
// the optional fused second stage and the copy-out -- from the same ArtnBitsPlan the GPU
// receives, so the planner's index algebra can be checked against the oracle on a box
// without a GPU.  Never linked into the product.
#include <complex>
#include <vector>
#include "artn_plan.h"

typedef std::complex<float> cf;

// element-offset form of the kernel's byte-offset swizzle
static int swz(int off, const ArtnStage *z) {
  if (z)
    for (int i = 0; i < z->swz_n; ++i)
      if ((off >> z->swz_src[i]) & 1) off ^= 1 << z->swz_dst[i];
  return off;
}

static void run_stage(const ArtnStage &st, const ArtnStage *zin, const cf *in, cf *out, const cf *B, int64_t offB) {
  const int KB = st.k, S = 1 << (KB - 1);
  const int nt_eff = st.nt < 4 ? st.nt : 4;
  const int wm_count = 4 >> st.wn_log2, msubs = 1 << (st.m_bits - 5);
  const int o0 = st.nt > 0 ? 1 << st.n_out_pos[0] : 0, o2 = st.nt > 2 ? 1 << st.n_out_pos[2] : 0,
            o3 = st.nt > 3 ? 1 << st.n_out_pos[3] : 0;
  for (int wave = 0; wave < 4; ++wave) {
    const int wn = wave & ((1 << st.wn_log2) - 1), wm = wave >> st.wn_log2;
    for (int msub = wm; msub < msubs; msub += wm_count) {
      int oi = 0, oo = 0;
      for (int b = 0; b < st.m_bits - 5; ++b)
        if ((msub >> b) & 1) { oi += 1 << st.msub_in_pos[b]; oo += 1 << st.msub_out_pos[b]; }
      float acc[64][16];
      for (auto &r : acc) for (float &x : r) x = 0.f;
      int lane_out[64];
      for (int s = 0; s < S; ++s) {
        int ko = 0; int64_t kbo = 0;
        for (int b = 1; b < KB; ++b) if ((s >> (b - 1)) & 1) { ko += 1 << st.k_in_pos[b]; kbo += st.k_b_stride[b]; }
        float W0[64], W1[64], ax[64], ay[64];
        for (int lane = 0; lane < 64; ++lane) {
          const int j = lane & 31, h = lane >> 5, ro = j & 1, nloc = j >> 1;
          int li = h << st.k_in_pos[0], lo = 0;
          for (int b = 0; b < 5; ++b) if ((j >> b) & 1) { li += 1 << st.lane_in_pos[b]; lo += 1 << st.lane_out_pos[b]; }
          if (st.nt > 1) lo += h << st.n_out_pos[1];
          int64_t lb = (int64_t)h * st.k_b_stride[0];
          for (int b = 0; b < nt_eff; ++b) if ((nloc >> b) & 1) lb += st.n_b_stride[b];
          for (int b = 0; b < st.wn_log2; ++b) if ((wn >> b) & 1) { lo += 1 << st.n_out_pos[4 + b]; lb += st.n_b_stride[4 + b]; }
          lane_out[lane] = lo;
          cf bv(0.f, 0.f);
          if ((nloc >> nt_eff) == 0) bv = B[offB + lb + kbo];
          W0[lane] = ro ? bv.imag() : bv.real();
          W1[lane] = ro ? bv.real() : -bv.imag();
          const cf a = in[swz(li + oi + ko, zin)];
          ax[lane] = a.real(); ay[lane] = a.imag();
        }
        // two v_mfma_f32_32x32x2_f32: D[i][j] += sum_kk Aop[i][kk] * Bop[kk][j]
        for (int phase = 0; phase < 2; ++phase) {
          const float *Wp = phase ? W1 : W0, *ap = phase ? ay : ax;
          for (int lane = 0; lane < 64; ++lane)
            for (int rr = 0; rr < 16; ++rr) {
              const int i = (rr & 3) + 8 * (rr >> 2) + 4 * (lane >> 5), jj = lane & 31;
              for (int kk = 0; kk < 2; ++kk) acc[lane][rr] += Wp[i + 32 * kk] * ap[jj + 32 * kk];
            }
        }
      }
      for (int lane = 0; lane < 64; ++lane) {
        const int h = lane >> 5;
        for (int q = 0; q < 4; ++q)
          for (int b0 = 0; b0 < 2; ++b0) {
            const int nl = b0 + 2 * h + 4 * (q & 1) + 8 * (q >> 1);
            if ((nl >> nt_eff) == 0)
              out[swz(lane_out[lane] + oo + b0 * o0 + (q & 1) * o2 + (q >> 1) * o3, &st)] =
                  cf(acc[lane][4 * q + 2 * b0], acc[lane][4 * q + 2 * b0 + 1]);
          }
      }
    }
  }
}

static void run_bits(const ArtnBitsPlan &P, const cf *A, const cf *B1, const cf *B2, cf *C) {
  std::vector<cf> R0((size_t)1 << P.r0_bits), R1((size_t)1 << P.T_mid);
  const int n_in_iters = 1 << (P.T_in - 9), n_out_iters = P.T_out >= 9 ? 1 << (P.T_out - 9) : 1;
  for (int64_t tile = 0; tile < P.n_tiles; ++tile) {
    int64_t r = tile, offA = 0, offB1 = 0, offB2 = 0, offC = 0;
    for (int d = 0; d < P.n_outer; ++d) {
      int64_t ext = P.outer[d].ext, x;
      if (P.outer[d].log2ext >= 0) { x = r & (ext - 1); r >>= P.outer[d].log2ext; }
      else { x = r % ext; r /= ext; }
      int64_t xa = x, xb = x;
      if (d == P.gather_dim) { // fused row gather (host pointers here)
        if (P.rows_a) { xa = P.rows_a[x]; if (xa < 0 || xa >= P.src_rows_a) { xa = 0; if (P.gather_err) *P.gather_err = 1; } }
        if (P.rows_b) { xb = P.rows_b[x]; if (xb < 0 || xb >= P.src_rows_b) { xb = 0; if (P.gather_err) *P.gather_err = 1; } }
      }
      offA += xa * P.outer[d].sA; offB1 += xb * P.outer[d].sB1; offB2 += x * P.outer[d].sB2; offC += x * P.outer[d].sC;
    }
    for (int tid = 0; tid < 256; ++tid) {
      int64_t in_lane = 0;
      for (int b = 1; b <= 8; ++b) if ((tid >> (b - 1)) & 1) in_lane += P.in_stride[b];
      for (int i = 0; i < n_in_iters; ++i) {
        int64_t off = 0;
        for (int b = 9; b < P.T_in; ++b) if ((i >> (b - 9)) & 1) off += P.in_stride[b];
        const cf *src = A + offA + in_lane + off;
        R0[2 * (tid + 256 * i)] = src[0];
        R0[2 * (tid + 256 * i) + 1] = src[1];
      }
    }
    run_stage(P.st[0], nullptr, R0.data(), R1.data(), B1, offB1);
    const cf *outr = R1.data();
    const ArtnStage *zout = &P.st[0];
    if (P.n_stages == 2) {
      run_stage(P.st[1], &P.st[0], R1.data(), R0.data(), B2, offB2);
      outr = R0.data();
      zout = &P.st[1];
    }
    for (int tid = 0; tid < 256; ++tid) {
      if (P.T_out < 9 && tid >= (1 << (P.T_out - 1))) continue;
      int64_t out_lane = 0;
      for (int b = 1; b <= 8; ++b) if ((tid >> (b - 1)) & 1) out_lane += P.out_stride[b];
      for (int i = 0; i < n_out_iters; ++i) {
        int64_t off = 0;
        for (int b = 9; b < P.T_out; ++b) if ((i >> (b - 9)) & 1) off += P.out_stride[b];
        cf *dst = C + offC + out_lane + off;
        dst[0] = outr[swz(2 * (tid + 256 * i), zout)];
        dst[1] = outr[swz(2 * (tid + 256 * i), zout) + 1];
      }
    }
  }
}

static void run_generic(const ArtnGenericPlan &G, const cf *A, const cf *B, cf *C) {
  for (int64_t idx = 0; idx < G.out_numel; ++idx) {
    int64_t r = idx, oa = 0, ob = 0;
    for (int d = 0; d < G.n_out; ++d) { int64_t x = r % G.out_ext[d]; r /= G.out_ext[d]; oa += x * G.out_sA[d]; ob += x * G.out_sB[d]; }
    cf sum(0.f, 0.f);
    for (int64_t q = 0; q < G.red_numel; ++q) {
      int64_t rr = q, ka = 0, kb = 0;
      for (int d = 0; d < G.n_red; ++d) { int64_t x = rr % G.red_ext[d]; rr /= G.red_ext[d]; ka += x * G.red_sA[d]; kb += x * G.red_sB[d]; }
      sum += A[oa + ka] * B[ob + kb];
    }
    C[idx] = sum;
  }
}

extern "C" int artn_emulate(const ArtnStepDesc *d, const void *A, const void *B, void *C, int force_generic,
                            int *kernel_used) {
  ArtnPlan p;
  std::string err;
  int rc = artn::make_plan(d, p, err, 256, !force_generic, 1);
  if (rc) return rc;
  if (kernel_used) *kernel_used = p.kernel;
  if (d->dtype != ARTN_C64) return ARTN_E_UNSUPPORTED;
  if (p.kernel == ARTN_KERNEL_BITS_MFMA) run_bits(p.bits, (const cf *)A, (const cf *)B, nullptr, (cf *)C);
  else run_generic(p.gen, (const cf *)A, (const cf *)B, (cf *)C);
  return 0;
}

// artn_contract_gather: row indices are host arrays here.
extern "C" int artn_emulate_gather(const ArtnStepDesc *d, const void *A, const void *B, void *C, int label,
                                   const int64_t *rows_a, int64_t src_rows_a, const int64_t *rows_b,
                                   int64_t src_rows_b, int32_t *err_flag) {
  ArtnPlan p;
  std::string err;
  int rc = artn::make_plan(d, p, err, 256, true, 1, label);
  if (rc) return rc;
  p.bits.rows_a = rows_a; p.bits.rows_b = rows_b;
  p.bits.src_rows_a = src_rows_a; p.bits.src_rows_b = src_rows_b;
  p.bits.gather_err = err_flag;
  run_bits(p.bits, (const cf *)A, (const cf *)B, nullptr, (cf *)C);
  return 0;
}

// Fused pair.  Returns ARTN_E_UNSUPPORTED when the planner declines to fuse.
extern "C" int artn_emulate2(const ArtnStepDesc *d1, const ArtnStepDesc *d2, const void *A, const void *B1,
                             const void *B2, void *C, ArtnStepInfo *info) {
  ArtnPlan p;
  std::string err;
  int rc = artn::make_plan_fused(d1, d2, p, err, 256, 1);
  if (rc) return rc;
  if (info) *info = p.info;
  run_bits(p.bits, (const cf *)A, (const cf *)B1, (const cf *)B2, (cf *)C);
  return 0;
}

// Diagnostic: LDS cycles per ds_read_b64 of the stage-1 operand reads (1 = conflict free): the 32
// lanes of a half wave read 8 bytes each, bank = (byte address / 4) mod 64.
extern "C" int artn_read_conflicts(const ArtnStepDesc *d1, const ArtnStepDesc *d2, int *stage1, int *stage2) {
  ArtnPlan p;
  std::string err;
  int rc = d2 ? artn::make_plan_fused(d1, d2, p, err, 256, 1) : artn::make_plan(d1, p, err, 256, true, 1);
  if (rc) return rc;
  if (p.kernel != ARTN_KERNEL_BITS_MFMA) return ARTN_E_UNSUPPORTED;
  for (int s = 0; s < p.bits.n_stages; ++s) {
    const ArtnStage &st = p.bits.st[s];
    const ArtnStage *zin = s == 0 ? nullptr : &p.bits.st[0];
    int count[32] = {0};
    bool seen[32][64];
    memset(seen, 0, sizeof(seen));
    int worst = 1;
    for (int j = 0; j < 32; ++j) {
      int off = 0;
      for (int b = 0; b < 5; ++b) if ((j >> b) & 1) off += 1 << st.lane_in_pos[b];
      off = swz(off, zin); // element index (8-byte units)
      const int slot = off & 31, hi = (off >> 5) & 63;
      if (!seen[slot][hi]) { seen[slot][hi] = true; count[slot]++; }
    }
    for (int b = 0; b < 32; ++b) worst = std::max(worst, count[b]);
    *(s == 0 ? stage1 : stage2) = worst;
  }
  return 0;
}
