// CPU emulation of artn_k_bits (artensor_amd/csrc/artn_kernels.hip), TEST-ONLY.
// It replays the kernel thread by thread -- copy-in chunks, per-lane fragment offsets,
// the 32x32x2 MFMA lane maps of the gfx950 guide, the in-place LDS transpose and the
// copy-out -- from the same ArtnBitsPlan the GPU receives, so the planner's index algebra
// can be checked against the oracle on a box without a GPU.  Never linked into the product.
#include <complex>
#include <vector>
#include "artn_plan.h"

typedef std::complex<float> cf;

static void run_bits(const ArtnBitsPlan &P, const cf *A, const cf *B, cf *C) {
  const int KB = P.k, PM = P.pm, S = 1 << (KB - 1);
  const int nt_eff = P.nt < 4 ? P.nt : 4;
  std::vector<cf> lds((size_t)1 << std::max(P.T_in, P.T_out));
  const int n_in_iters = 1 << (P.T_in - 9), n_out_iters = 1 << (P.T_out - 9);
  for (int64_t tile = 0; tile < P.n_tiles; ++tile) {
    int64_t r = tile, offA = 0, offB = 0, offC = 0;
    for (int d = 0; d < P.n_outer; ++d) {
      int64_t ext = P.outer[d].ext, x;
      if (P.outer[d].log2ext >= 0) { x = r & (ext - 1); r >>= P.outer[d].log2ext; }
      else { x = r % ext; r /= ext; }
      offA += x * P.outer[d].sA; offB += x * P.outer[d].sB; offC += x * P.outer[d].sC;
    }
    // copy-in
    for (int tid = 0; tid < 256; ++tid) {
      int64_t in_lane = 0;
      for (int b = 1; b <= 8; ++b) if ((tid >> (b - 1)) & 1) in_lane += P.in_stride[b];
      for (int i = 0; i < n_in_iters; ++i) {
        int64_t off = 0;
        for (int b = 9; b < P.T_in; ++b) if ((i >> (b - 9)) & 1) off += P.in_stride[b];
        const cf *src = A + offA + in_lane + off;
        lds[2 * (tid + 256 * i)] = src[0];
        lds[2 * (tid + 256 * i) + 1] = src[1];
      }
    }
    // MFMA phase, wave by wave; results kept per wave/lane until every wave has read LDS
    std::vector<float> acc((size_t)4 * 64 * PM * 16, 0.f);
    for (int wave = 0; wave < 4; ++wave) {
      const int wn = wave & ((1 << P.wn_log2) - 1), wm = wave >> P.wn_log2;
      int64_t wn_b = 0;
      for (int b = 0; b < P.wn_log2; ++b) if ((wn >> b) & 1) wn_b += P.n_b_stride[4 + b];
      for (int pm = 0; pm < PM; ++pm) {
        const int msub = wm * PM + pm;
        int msub_in = 0;
        for (int b = 0; b < P.mt - 5; ++b) if ((msub >> b) & 1) msub_in += 1 << P.msub_in_pos[b];
        for (int s = 0; s < S; ++s) {
          int ko = 0; int64_t kbo = 0;
          for (int b = 1; b < KB; ++b) if ((s >> (b - 1)) & 1) { ko += 1 << P.k_in_pos[b]; kbo += P.k_b_stride[b]; }
          float W0[64], W1[64], ax[64], ay[64];
          for (int lane = 0; lane < 64; ++lane) {
            const int j = lane & 31, h = lane >> 5;
            int lane_in = h << P.k_in_pos[0];
            for (int b = 0; b < 5; ++b) if ((j >> b) & 1) lane_in += 1 << P.lane_in_pos[b];
            const int ro = j & 1, nloc = j >> 1;
            int64_t lane_b = (int64_t)h * P.k_b_stride[0] + wn_b;
            for (int b = 0; b < nt_eff; ++b) if ((nloc >> b) & 1) lane_b += P.n_b_stride[b];
            cf bv(0.f, 0.f);
            if ((nloc >> nt_eff) == 0) bv = B[offB + lane_b + kbo];
            W0[lane] = ro ? bv.imag() : bv.real();
            W1[lane] = ro ? bv.real() : -bv.imag();
            cf a = lds[lane_in + msub_in + ko];
            ax[lane] = a.real(); ay[lane] = a.imag();
          }
          // two v_mfma_f32_32x32x2_f32: D[i][j] += sum_kk Aop[i][kk] * Bop[kk][j]
          for (int phase = 0; phase < 2; ++phase) {
            const float *Wp = phase ? W1 : W0, *ap = phase ? ay : ax;
            for (int lane = 0; lane < 64; ++lane)
              for (int rr = 0; rr < 16; ++rr) {
                const int i = (rr & 3) + 8 * (rr >> 2) + 4 * (lane >> 5), jj = lane & 31;
                float sum = acc[((size_t)(wave * 64 + lane) * PM + pm) * 16 + rr];
                for (int kk = 0; kk < 2; ++kk) sum += Wp[i + 32 * kk] * ap[jj + 32 * kk];
                acc[((size_t)(wave * 64 + lane) * PM + pm) * 16 + rr] = sum;
              }
          }
        }
      }
    }
    // accumulators -> LDS (output-tile order)
    const int o0 = P.nt > 0 ? 1 << P.n_out_pos[0] : 0, o2 = P.nt > 2 ? 1 << P.n_out_pos[2] : 0,
              o3 = P.nt > 3 ? 1 << P.n_out_pos[3] : 0;
    for (int wave = 0; wave < 4; ++wave) {
      const int wn = wave & ((1 << P.wn_log2) - 1), wm = wave >> P.wn_log2;
      int wn_out = 0;
      for (int b = 0; b < P.wn_log2; ++b) if ((wn >> b) & 1) wn_out += 1 << P.n_out_pos[4 + b];
      for (int lane = 0; lane < 64; ++lane) {
        const int j = lane & 31, h = lane >> 5;
        int lane_out = wn_out;
        for (int b = 0; b < 5; ++b) if ((j >> b) & 1) lane_out += 1 << P.lane_out_pos[b];
        if (P.nt > 1) lane_out += h << P.n_out_pos[1];
        for (int pm = 0; pm < PM; ++pm) {
          const int msub = wm * PM + pm;
          int msub_out = 0;
          for (int b = 0; b < P.mt - 5; ++b) if ((msub >> b) & 1) msub_out += 1 << P.msub_out_pos[b];
          const float *a = &acc[((size_t)(wave * 64 + lane) * PM + pm) * 16];
          for (int q = 0; q < 4; ++q)
            for (int b0 = 0; b0 < 2; ++b0) {
              const int nl = b0 + 2 * h + 4 * (q & 1) + 8 * (q >> 1);
              if ((nl >> nt_eff) == 0)
                lds[lane_out + msub_out + b0 * o0 + (q & 1) * o2 + (q >> 1) * o3] = cf(a[4 * q + 2 * b0], a[4 * q + 2 * b0 + 1]);
            }
        }
      }
    }
    // copy-out
    for (int tid = 0; tid < 256; ++tid) {
      int64_t out_lane = 0;
      for (int b = 1; b <= 8; ++b) if ((tid >> (b - 1)) & 1) out_lane += P.out_stride[b];
      for (int i = 0; i < n_out_iters; ++i) {
        int64_t off = 0;
        for (int b = 9; b < P.T_out; ++b) if ((i >> (b - 9)) & 1) off += P.out_stride[b];
        cf *dst = C + offC + out_lane + off;
        dst[0] = lds[2 * (tid + 256 * i)];
        dst[1] = lds[2 * (tid + 256 * i) + 1];
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
  if (p.kernel == ARTN_KERNEL_BITS_MFMA) run_bits(p.bits, (const cf *)A, (const cf *)B, (cf *)C);
  else run_generic(p.gen, (const cf *)A, (const cf *)B, (cf *)C);
  return 0;
}
