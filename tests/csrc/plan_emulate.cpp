// CPU emulation of artn_k_bits (artensor_amd/csrc/artn_kernels.hip), TEST-ONLY.
// It replays the kernel thread by thread -- copy-in chunks, per-lane fragment offsets,
// the 32x32x2 MFMA lane maps of the gfx950 guide, the scatter into the second LDS region,
// the optional fused second stage and the copy-out -- from the same ArtnBitsPlan the GPU
// receives, so the planner's index algebra can be checked against the oracle on a box
// without a GPU.  Never linked into the product.
#include <algorithm>
#include <cmath>
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
      if (st.m3 == 2) { // 4-bit stage of a 3M launch: v_mfma_f32_16x16x4_f32 blocks (A[l & 15][l >> 4], B[l >> 4][l & 15]; D: column
                        // l & 15, rows 4 (l >> 4) + r), two halves of 16 columns per 32-column sub-tile, three products
        for (int half = 0; half < 2; ++half) {
          float t1[64][4] = {}, t2[64][4] = {}, t3[64][4] = {};
          int lane_out4[64];
          for (int s = 0; s < (1 << (KB - 2)); ++s) { // chain steps of 4 contracted values (KB = 2..4; 5..6: ArtnBitsPlan::narrow3)
            float wre[64], wim[64], ax[64], ay[64];
            for (int lane = 0; lane < 64; ++lane) {
              const int jj = lane & 15, g = lane >> 4;
              int li = ((g & 1) << st.k_in_pos[0]) + ((g >> 1) << st.k_in_pos[1]);
              for (int b = 2; b < KB; ++b) if ((s >> (b - 2)) & 1) li += 1 << st.k_in_pos[b];
              int lo = 0;
              for (int b = 0; b < 4; ++b) if ((jj >> b) & 1) { li += 1 << st.lane_in_pos[b]; lo += 1 << st.lane_out_pos[b]; }
              if (half) { li += 1 << st.lane_in_pos[4]; lo += 1 << st.lane_out_pos[4]; }
              if (st.nt > 2) lo += (g & 1) << st.n_out_pos[2];
              if (st.nt > 3) lo += (g >> 1) << st.n_out_pos[3];
              int64_t lb = (int64_t)(g & 1) * st.k_b_stride[0] + (int64_t)(g >> 1) * st.k_b_stride[1];
              for (int b = 2; b < KB; ++b) if ((s >> (b - 2)) & 1) lb += st.k_b_stride[b];
              for (int b = 0; b < nt_eff; ++b) if ((jj >> b) & 1) lb += st.n_b_stride[b];
              for (int b = 0; b < st.wn_log2; ++b) if ((wn >> b) & 1) { lo += 1 << st.n_out_pos[4 + b]; lb += st.n_b_stride[4 + b]; }
              lane_out4[lane] = lo;
              cf bv(0.f, 0.f);
              if ((jj >> nt_eff) == 0) bv = B[offB + lb];
              wre[lane] = bv.real(); wim[lane] = bv.imag();
              const cf a = in[swz(li + oi, zin)];
              ax[lane] = a.real(); ay[lane] = a.imag();
            }
            for (int lane = 0; lane < 64; ++lane)
              for (int r = 0; r < 4; ++r) {
                const int i = 4 * (lane >> 4) + r, jc = lane & 15;
                for (int kk = 0; kk < 4; ++kk) {
                  t1[lane][r] += wre[i + 16 * kk] * ax[jc + 16 * kk];
                  t2[lane][r] += wim[i + 16 * kk] * ay[jc + 16 * kk];
                  t3[lane][r] += (wre[i + 16 * kk] + wim[i + 16 * kk]) * (ax[jc + 16 * kk] + ay[jc + 16 * kk]);
                }
              }
          }
          for (int lane = 0; lane < 64; ++lane)
            for (int r = 0; r < 4; ++r) {
              const int n = 4 * (lane >> 4) + r;
              if ((n >> nt_eff) != 0) continue;
              const int o = lane_out4[lane] + oo + ((r & 1) && st.nt > 0 ? 1 << st.n_out_pos[0] : 0) + ((r & 2) && st.nt > 1 ? 1 << st.n_out_pos[1] : 0);
              const float a1 = t1[lane][r], a2 = t2[lane][r], a3 = t3[lane][r];
              out[swz(o, &st)] = cf(a1 - a2, a3 - a1 - a2);
            }
        }
        continue;
      }
      if (st.m3) { // three real products per complex product: rows = 32 columns n of the small operand
        std::vector<float> t1(64 * 16, 0.f), t2(64 * 16, 0.f), t3(64 * 16, 0.f);
        int lane_out3[64];
        for (int s = 0; s < S; ++s) {
          int ko = 0; int64_t kbo = 0;
          for (int b = 1; b < KB; ++b) if ((s >> (b - 1)) & 1) { ko += 1 << st.k_in_pos[b]; kbo += st.k_b_stride[b]; }
          float wre[64], wim[64], ax[64], ay[64];
          for (int lane = 0; lane < 64; ++lane) {
            const int j = lane & 31, h = lane >> 5;
            int li = h << st.k_in_pos[0], lo = h << st.n_out_pos[2];
            for (int b = 0; b < 5; ++b) if ((j >> b) & 1) { li += 1 << st.lane_in_pos[b]; lo += 1 << st.lane_out_pos[b]; }
            int64_t lb = (int64_t)h * st.k_b_stride[0];
            for (int b = 0; b < 5; ++b) if ((j >> b) & 1) lb += st.n_b_stride[b];
            for (int b = 0; b < st.wn_log2; ++b) if ((wn >> b) & 1) { lo += 1 << st.n_out_pos[5 + b]; lb += st.n_b_stride[5 + b]; }
            lane_out3[lane] = lo;
            const cf bv = B[offB + lb + kbo];
            wre[lane] = bv.real(); wim[lane] = bv.imag();
            const cf a = in[swz(li + oi + ko, zin)];
            ax[lane] = a.real(); ay[lane] = a.imag();
          }
          for (int lane = 0; lane < 64; ++lane)
            for (int rr = 0; rr < 16; ++rr) {
              const int i = (rr & 3) + 8 * (rr >> 2) + 4 * (lane >> 5), jj = lane & 31;
              for (int kk = 0; kk < 2; ++kk) {
                t1[lane * 16 + rr] += wre[i + 32 * kk] * ax[jj + 32 * kk];
                t2[lane * 16 + rr] += wim[i + 32 * kk] * ay[jj + 32 * kk];
                t3[lane * 16 + rr] += (wre[i + 32 * kk] + wim[i + 32 * kk]) * (ax[jj + 32 * kk] + ay[jj + 32 * kk]);
              }
            }
        }
        for (int lane = 0; lane < 64; ++lane)
          for (int rr = 0; rr < 16; ++rr) {
            const int o = lane_out3[lane] + oo + ((rr & 1) << st.n_out_pos[0]) + (((rr >> 1) & 1) << st.n_out_pos[1]) +
                          (((rr >> 2) & 1) << st.n_out_pos[3]) + (((rr >> 3) & 1) << st.n_out_pos[4]);
            const float a1 = t1[lane * 16 + rr], a2 = t2[lane * 16 + rr], a3 = t3[lane * 16 + rr];
            out[swz(o, &st)] = cf(a1 - a2, a3 - a1 - a2);
          }
        continue;
      }
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

static void run_bits(const ArtnBitsPlan &P, const cf *A, const cf *B1, const cf *B2, cf *C, const cf *B3 = nullptr) {
  std::vector<cf> R0((size_t)1 << P.r0_bits), R1((size_t)1 << P.r1_bits);
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
    if (P.n_stages >= 2) {
      run_stage(P.st[1], &P.st[0], R1.data(), R0.data(), B2, offB2);
      outr = R0.data();
      zout = &P.st[1];
    }
    if (P.n_stages == 3) { // artn_k_bits3: the third stage reads region 0 and leaves the result in region 1
      run_stage(P.st[2], &P.st[1], R0.data(), R1.data(), B3, 0);
      outr = R1.data();
      zout = &P.st[2];
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


// ---- artn_k_bits128 (artn_bits128_kernel.h), replayed lane by lane from the same ArtnBitsPlan: 16-byte elements, sub-tiles
//      of 16 columns, v_mfma_f64_16x16x4_f64 (D[i][j] += sum_kk Aop[i][kk] Bop[kk][j]; lane = (j or i) + 16 kk; accumulator
//      register r of lane (j, g) = row g + 4 r)
typedef std::complex<double> cd;
static void run_stage128(const ArtnStage &st, const ArtnStage *zin, const cd *in, cd *out, const cd *B, int64_t offB) {
  const int KB = st.k, S = 1 << (KB - 1);
  const int nt3 = st.nt < 3 ? st.nt : 3, n_lim = 1 << nt3;
  const int wm_count = 4 >> st.wn_log2, msubs = 1 << (st.m_bits - 4);
  for (int wave = 0; wave < 4; ++wave) {
    const int wn = wave & ((1 << st.wn_log2) - 1), wm = wave >> st.wn_log2;
    for (int msub = wm; msub < msubs; msub += wm_count) {
      int oi = 0, oo = 0;
      for (int b = 0; b < st.m_bits - 4; ++b)
        if ((msub >> b) & 1) { oi += 1 << st.msub_in_pos[b]; oo += 1 << st.msub_out_pos[b]; }
      double acc[64][4];
      for (auto &r : acc) for (double &x : r) x = 0.0;
      int lane_out[64];
      for (int s = 0; s < S; ++s) {
        int ko = 0; int64_t kbo = 0;
        for (int b = 1; b < KB; ++b) if ((s >> (b - 1)) & 1) { ko += 1 << st.k_in_pos[b]; kbo += st.k_b_stride[b]; }
        double W[64], X[64];
        for (int lane = 0; lane < 64; ++lane) {
          const int j = lane & 15, g = lane >> 4, ro = j & 1, n_in = j >> 1, pp = g & 1, kcl = g >> 1;
          int li = kcl << st.k_in_pos[0], lo = 0;
          for (int b = 0; b < 4; ++b) if ((j >> b) & 1) { li += 1 << st.lane_in_pos[b]; lo += 1 << st.lane_out_pos[b]; }
          if (st.nt > 0) lo += kcl << st.n_out_pos[0];
          int64_t lb = (int64_t)kcl * st.k_b_stride[0];
          for (int b = 0; b < nt3; ++b) if ((n_in >> b) & 1) lb += st.n_b_stride[b];
          for (int b = 0; b < st.wn_log2; ++b) if ((wn >> b) & 1) { lo += 1 << st.n_out_pos[3 + b]; lb += st.n_b_stride[3 + b]; }
          lane_out[lane] = lo;
          cd bv(0.0, 0.0);
          if ((n_in >> nt3) == 0) bv = B[offB + lb + kbo];
          // (ro, p): (0,0) re  (0,1) -im  (1,0) im  (1,1) re
          W[lane] = (ro ^ pp) ? (ro == 0 ? -bv.imag() : bv.imag()) : bv.real();
          const cd a = in[swz(li + oi + ko, zin)];
          X[lane] = pp ? a.imag() : a.real();
        }
        for (int lane = 0; lane < 64; ++lane)
          for (int r = 0; r < 4; ++r) {
            const int i = (lane >> 4) + 4 * r, jj = lane & 15;
            for (int kk = 0; kk < 4; ++kk) acc[lane][r] += W[i + 16 * kk] * X[jj + 16 * kk];
          }
      }
      for (int lane = 0; lane < 64; ++lane) {
        const int g = lane >> 4;
        for (int r = 0; r < 4; ++r) {
          if ((g >> 1) + 2 * r >= n_lim) continue;
          const int o = lane_out[lane] + oo + ((r & 1) && st.nt > 1 ? 1 << st.n_out_pos[1] : 0) + ((r & 2) && st.nt > 2 ? 1 << st.n_out_pos[2] : 0);
          double *dst = reinterpret_cast<double *>(&out[swz(o, &st)]);
          dst[g & 1] = acc[lane][r];
        }
      }
    }
  }
}

static void run_bits128(const ArtnBitsPlan &P, const cd *A, const cd *B1, const cd *B2, cd *C) {
  std::vector<cd> R0((size_t)1 << P.r0_bits), R1((size_t)1 << P.T_mid);
  const int n_in_iters = 1 << (P.T_in - 8), n_out_iters = P.T_out >= 8 ? 1 << (P.T_out - 8) : 1;
  for (int64_t tile = 0; tile < P.n_tiles; ++tile) {
    int64_t r = tile, offA = 0, offB1 = 0, offB2 = 0, offC = 0;
    for (int d = 0; d < P.n_outer; ++d) {
      int64_t ext = P.outer[d].ext, x;
      if (P.outer[d].log2ext >= 0) { x = r & (ext - 1); r >>= P.outer[d].log2ext; }
      else { x = r % ext; r /= ext; }
      offA += x * P.outer[d].sA; offB1 += x * P.outer[d].sB1; offB2 += x * P.outer[d].sB2; offC += x * P.outer[d].sC;
    }
    for (auto &x : R0) x = cd(-777.0, -777.0);
    for (auto &x : R1) x = cd(-777.0, -777.0);
    for (int tid = 0; tid < 256; ++tid) {
      int64_t in_lane = 0;
      for (int b = 0; b < 8 && b < P.T_in; ++b) if ((tid >> b) & 1) in_lane += P.in_stride[b];
      for (int i = 0; i < n_in_iters; ++i) {
        int64_t off = 0;
        for (int b = 8; b < P.T_in; ++b) if ((i >> (b - 8)) & 1) off += P.in_stride[b];
        R0[tid + 256 * i] = A[offA + in_lane + off];
      }
    }
    run_stage128(P.st[0], nullptr, R0.data(), R1.data(), B1, offB1);
    const cd *outr = R1.data();
    const ArtnStage *zout = &P.st[0];
    if (P.n_stages == 2) {
      run_stage128(P.st[1], &P.st[0], R1.data(), R0.data(), B2, offB2);
      outr = R0.data();
      zout = &P.st[1];
    }
    for (int tid = 0; tid < 256; ++tid) {
      if (P.T_out < 8 && tid >= (1 << P.T_out)) continue;
      int64_t out_lane = 0;
      for (int b = 0; b < 8 && b < P.T_out; ++b) if ((tid >> b) & 1) out_lane += P.out_stride[b];
      for (int i = 0; i < n_out_iters; ++i) {
        int64_t off = 0;
        for (int b = 8; b < P.T_out; ++b) if ((i >> (b - 8)) & 1) off += P.out_stride[b];
        C[offC + out_lane + off] = outr[swz(tid + 256 * i, zout)];
      }
    }
  }
}

// ---- artn_k_gemm (artn_gemm_kernel.h), replayed thread by thread from the same ArtnGemmPlan ---------
static unsigned swzg(unsigned off, const ArtnGemmPlan &P) {
  for (int i = 0; i < P.swz_n; ++i)
    if ((off >> P.swz_src[i]) & 1) off ^= 1u << P.swz_dst[i];
  return off;
}

static float bf16r(float x) { // round to nearest even, as v_cvt_pk_bf16_f32 (finite inputs)
  uint32_t u;
  memcpy(&u, &x, 4);
  u += 0x7FFFu + ((u >> 16) & 1u);
  u &= 0xFFFF0000u;
  float y;
  memcpy(&y, &u, 4);
  return y;
}

static void run_gemm(const ArtnGemmPlan &P, const cf *A0, const cf *B0, cf *C) {
  const cf *A = P.swapped ? B0 : A0, *B = P.swapped ? A0 : B0;
  const int mt = P.mt, nt = P.nt, MB = 1 << P.mb_log2, NB = 1 << P.nb_log2;
  const int epi_bits = std::min(P.tc_bits, ARTN_GEMM_EPI_BITS);
  const int PL = P.split ? ARTN_GEMM_PITCH_LOG2 : P.pitch_log2; // (fp32: 7, or 5 for the tall chunks of 32 x 32 tiles)
  const int NS = P.split ? 8 : 1 << (P.kc - 1);                 // fp32 MFMA steps per chunk
  const int EB = P.split ? 4 : 8; // bytes per image element (bf16 pairs / fp32 pairs)
  std::vector<cf> imgA((size_t)(16 << ARTN_GEMM_PITCH_LOG2) * 8 / EB), imgB((size_t)(16 << ARTN_GEMM_PITCH_LOG2) * 8 / EB), res((size_t)1 << epi_bits);
  const int n_chunks = 1 << P.n_ko;
  for (int64_t tile = 0; tile < P.n_tiles; ++tile) {
    int64_t r = tile, offA = 0, offB = 0, offC = 0;
    for (int d = 0; d < P.n_outer; ++d) {
      int64_t ext = P.outer[d].ext, x;
      if (P.outer[d].log2ext >= 0) { x = r & (ext - 1); r >>= P.outer[d].log2ext; }
      else { x = r % ext; r /= ext; }
      int64_t xa = x, xb = x;
      if (d == P.gather_dim) { // row gather (host pointers here; rows_a belongs to the kernel's first operand)
        if (P.rows_a) { xa = P.rows_a[x]; if (xa < 0 || xa >= P.src_rows_a) { xa = 0; if (P.gather_err) *P.gather_err = 1; } }
        if (P.rows_b) { xb = P.rows_b[x]; if (xb < 0 || xb >= P.src_rows_b) { xb = 0; if (P.gather_err) *P.gather_err = 1; } }
      }
      offA += xa * P.outer[d].sA; offB += xb * P.outer[d].sB1; offC += x * P.outer[d].sC;
    }
    // accumulators of every (wave, block, lane, register)
    const int NACC = P.m3 ? 3 : 1, NBW = P.m3 ? 32 : 16;
    std::vector<float> acc((size_t)4 * MB * NB * NACC * 64 * 16, 0.f);
    auto ACC3 = [&](int wave, int a, int b, int t, int lane, int rr) -> float & {
      return acc[(((((size_t)wave * MB + a) * NB + b) * NACC + t) * 64 + lane) * 16 + rr];
    };
    auto ACC = [&](int wave, int a, int b, int lane, int rr) -> float & { return ACC3(wave, a, b, 0, lane, rr); };
    // epilogue: every 2^10 contracted values and at the end (the later partial sums are added to C)
    auto flush_tile = [&](bool accumulate) {
    auto m_off = [&](int m_local) { unsigned o = 0; for (int i = 0; i < mt; ++i) if ((m_local >> i) & 1) o |= 1u << P.m_pos[i]; return o; };
    auto n_off = [&](int n_local) { unsigned o = 0; for (int i = 0; i < nt; ++i) if ((n_local >> i) & 1) o |= 1u << P.n_pos[i]; return o; };
    const int n_lim = nt >= 4 ? 16 : 1 << nt;
    const int n_pass = 1 << (P.tc_bits - epi_bits);
    for (int pass = 0; pass < n_pass; ++pass) {
      for (auto &x : res) x = cf(-777.f, -777.f);
      for (int wave = 0; wave < 4; ++wave) { // (wave order = ascending wk: the first partial block is stored, the others added)
        const int wn = wave & ((1 << P.wn_log2) - 1), wm = (wave >> P.wn_log2) & ((1 << P.wm_log2) - 1);
        const bool add = (wave >> (P.wn_log2 + P.wm_log2)) > 0;
        if (P.m3) {
          for (int a = 0; a < MB; ++a) for (int b = 0; b < NB; ++b) for (int lane = 0; lane < 64; ++lane)
            for (int rr = 0; rr < 16; ++rr) {
              const int j = lane & 31, h = lane >> 5;
              const int n_loc = (rr & 3) + 8 * (rr >> 2) + 4 * h;
              const unsigned pos = swzg(m_off((wm * MB + a) * 32 + j) | n_off((wn * NB + b) * 32 + n_loc), P);
              if ((int)(pos >> ARTN_GEMM_EPI_BITS) != pass) continue;
              const float t1 = ACC3(wave, a, b, 0, lane, rr), t2 = ACC3(wave, a, b, 1, lane, rr), t3 = ACC3(wave, a, b, 2, lane, rr);
              cf &dst = res[pos & ((1u << ARTN_GEMM_EPI_BITS) - 1u)];
              dst = add ? dst + cf(t1 - t2, t3 - t1 - t2) : cf(t1 - t2, t3 - t1 - t2);
            }
          continue;
        }
        for (int a = 0; a < MB; ++a) for (int b = 0; b < NB; ++b) for (int lane = 0; lane < 64; ++lane)
          for (int q = 0; q < 4; ++q) for (int b0 = 0; b0 < 2; ++b0) {
            const int j = lane & 31, h = lane >> 5;
            const int n_loc = b0 + 2 * h + 4 * q;
            if (n_loc >= n_lim) continue;
            const unsigned pos = swzg(m_off((wm * MB + a) * 32 + j) | n_off((wn * NB + b) * 16 + n_loc), P);
            if ((int)(pos >> ARTN_GEMM_EPI_BITS) != pass) continue;
            cf &dst = res[pos & ((1u << ARTN_GEMM_EPI_BITS) - 1u)];
            const cf val(ACC(wave, a, b, lane, 4 * q + 2 * b0), ACC(wave, a, b, lane, 4 * q + 2 * b0 + 1));
            dst = add ? dst + val : val;
          }
      }
      const int cb = epi_bits - 1, iters = cb > 8 ? 1 << (cb - 8) : 1;
      const int64_t pass_off = P.tc_bits > epi_bits ? pass * P.out_stride[epi_bits] : 0;
      for (int tid = 0; tid < 256; ++tid) {
        if (!(cb >= 8 || tid < (1 << cb))) continue;
        for (int i = 0; i < iters; ++i) {
          const int chunk = tid + 256 * i;
          int64_t g = 0;
          for (int b = 1; b < epi_bits; ++b) if ((chunk >> (b - 1)) & 1) g += P.out_stride[b];
          const unsigned l = swzg(((unsigned)pass << ARTN_GEMM_EPI_BITS) | 2u * (unsigned)chunk, P) & ((1u << ARTN_GEMM_EPI_BITS) - 1u);
          if (accumulate) { C[offC + pass_off + g] += res[l]; C[offC + pass_off + g + 1] += res[l + 1]; }
          else { C[offC + pass_off + g] = res[l]; C[offC + pass_off + g + 1] = res[l + 1]; }
        }
      }
    }
    };
    const int flush_mask = P.split ? 0x7fffffff : (1 << (ARTN_GEMM_FLUSH_LOG2 - P.kc)) - 1;
    int64_t ka = 0, kb = 0;
    for (int c = 0; c < n_chunks; ++c) {
      if (c > 0) { // Gray code step from chunk c-1 to chunk c
        const int bit = __builtin_ctz((unsigned)c);
        const unsigned gn = (unsigned)c ^ ((unsigned)c >> 1);
        if ((gn >> bit) & 1) { ka += P.ko_sA[bit]; kb += P.ko_sB[bit]; } else { ka -= P.ko_sA[bit]; kb -= P.ko_sB[bit]; }
      }
      // global -> LDS images, one 16-byte chunk (two elements) per thread and iteration
      for (int which = 0; which < 2; ++which) {
        const int bits = which ? P.tb_bits : P.ta_bits;
        const int64_t *gs = which ? P.b_stride : P.a_stride;
        const int32_t *ls = which ? P.b_lds : P.a_lds;
        cf *img = which ? imgB.data() : imgA.data();
        const cf *src = which ? B + offB + kb : A + offA + ka;
        const int cb = bits - 1, iters = cb > 8 ? 1 << (cb - 8) : 1;
        for (int tid = 0; tid < 256; ++tid) {
          if (!(cb >= 8 || tid < (1 << cb))) continue;
          for (int u = 0; u < iters; ++u) {
            const int chunk = tid + 256 * u;
            int64_t g = 0; int l = 0;
            for (int b = 1; b < bits; ++b) if ((chunk >> (b - 1)) & 1) { g += gs[b]; l += ls[b]; }
            cf e0 = src[g], e1 = src[g + 1];
            if (P.split) { e0 = cf(bf16r(e0.real()), bf16r(e0.imag())); e1 = cf(bf16r(e1.real()), bf16r(e1.imag())); }
            img[l / EB] = e0;
            img[(l + ls[0]) / EB] = e1;
          }
        }
      }
      // MFMA pairs
      for (int wave = 0; wave < 4; ++wave) {
        const int wn = wave & ((1 << P.wn_log2) - 1), wm = (wave >> P.wn_log2) & ((1 << P.wm_log2) - 1);
        const int wk = wave >> (P.wn_log2 + P.wm_log2), WK = 1 << P.wk_log2; // waves sharing a block split the chunk
        if (P.m3) { // three real products: T1 = A_re B_re, T2 = A_im B_im, T3 = (A_re + A_im)(B_re + B_im); rows = 32 columns n
          for (int s = wk; s < NS; s += WK)
            for (int a = 0; a < MB; ++a)
              for (int b = 0; b < NB; ++b)
                for (int lane = 0; lane < 64; ++lane)
                  for (int rr = 0; rr < 16; ++rr) {
                    const int i = (rr & 3) + 8 * (rr >> 2) + 4 * (lane >> 5), jj = lane & 31;
                    for (int kk = 0; kk < 2; ++kk) {
                      const cf xv = imgA[((size_t)(2 * s + kk) << PL) + (wm * MB + a) * 32 + jj];
                      const cf bv = imgB[((size_t)(2 * s + kk) << PL) + (wn * NB + b) * 32 + i];
                      ACC3(wave, a, b, 0, lane, rr) += bv.real() * xv.real();
                      ACC3(wave, a, b, 1, lane, rr) += bv.imag() * xv.imag();
                      ACC3(wave, a, b, 2, lane, rr) += (bv.real() + bv.imag()) * (xv.real() + xv.imag());
                    }
                  }
          continue;
        }
        if (P.split) { // v_mfma_f32_32x32x16_bf16 groups: kc = 8t + 4h + u, image [kc >> 2][row][kc & 3]
          for (int t = wk; t < 4; t += WK)
            for (int a = 0; a < MB; ++a)
              for (int b = 0; b < NB; ++b)
                for (int lane = 0; lane < 64; ++lane)
                  for (int rr = 0; rr < 16; ++rr) {
                    const int i = (rr & 3) + 8 * (rr >> 2) + 4 * (lane >> 5), jj = lane & 31;
                    const int n_in = i >> 1, ro = i & 1;
                    if (!(nt >= 4 || n_in < (1 << nt))) continue;
                    float sum = 0.f;
                    for (int kk = 0; kk < 8; ++kk) {
                      const int kc = 8 * t + kk;
                      const cf xv = imgA[((size_t)((kc >> 2) << PL) + (wm * MB + a) * 32 + jj) * 4 + (kc & 3)];
                      const cf bv = imgB[((size_t)((kc >> 2) << PL) + (wn * NB + b) * 16 + n_in) * 4 + (kc & 3)];
                      const float w0 = ro ? bv.imag() : bv.real(), w1 = ro ? bv.real() : -bv.imag();
                      sum += w0 * xv.real() + w1 * xv.imag();
                    }
                    ACC(wave, a, b, lane, rr) += sum;
                  }
          continue;
        }
        for (int s = wk; s < NS; s += WK)
          for (int a = 0; a < MB; ++a)
            for (int b = 0; b < NB; ++b) {
              float W0[64], W1[64], ax[64], ay[64];
              for (int lane = 0; lane < 64; ++lane) {
                const int j = lane & 31, h = lane >> 5, ro = j & 1, n_in = j >> 1;
                const bool w_valid = nt >= 4 || n_in < (1 << nt);
                const cf xv = imgA[((size_t)(2 * s + h) << PL) + (wm * MB + a) * 32 + j];
                cf bv(0.f, 0.f);
                if (w_valid) bv = imgB[((size_t)(2 * s + h) << PL) + (wn * NB + b) * 16 + n_in];
                W0[lane] = ro ? bv.imag() : bv.real();
                W1[lane] = ro ? bv.real() : -bv.imag();
                ax[lane] = xv.real(); ay[lane] = xv.imag();
              }
              for (int phase = 0; phase < 2; ++phase) {
                const float *Wp = phase ? W1 : W0, *ap = phase ? ay : ax;
                for (int lane = 0; lane < 64; ++lane)
                  for (int rr = 0; rr < 16; ++rr) {
                    const int i = (rr & 3) + 8 * (rr >> 2) + 4 * (lane >> 5), jj = lane & 31;
                    for (int kk = 0; kk < 2; ++kk) ACC(wave, a, b, lane, rr) += Wp[i + 32 * kk] * ap[jj + 32 * kk];
                  }
              }
            }
      }
      if (c + 1 == n_chunks || ((c + 1) & flush_mask) == 0) {
        flush_tile(c > flush_mask);
        std::fill(acc.begin(), acc.end(), 0.f);
      }
    }
  }
}


// ---- artn_k_gemm128 (complex128 on v_mfma_f64_16x16x4_f64), replayed from the same plan --------------------

static void run_gemm128(const ArtnGemmPlan &P, const cd *A0, const cd *B0, cd *C) {
  const cd *A = P.swapped ? B0 : A0, *B = P.swapped ? A0 : B0;
  const int mt = P.mt, nt = P.nt, MB = 2, NB = 1 << P.nb_log2, PL = ARTN_GEMM_PITCH_LOG2;
  const int epi_bits = std::min(P.tc_bits, ARTN_GEMM128_EPI_BITS);
  std::vector<cd> imgA((size_t)8 << PL), imgB((size_t)8 << PL), res((size_t)1 << epi_bits);
  const int n_chunks = 1 << P.n_ko;
  for (int64_t tile = 0; tile < P.n_tiles; ++tile) {
    int64_t r = tile, offA = 0, offB = 0, offC = 0;
    for (int d = 0; d < P.n_outer; ++d) {
      int64_t ext = P.outer[d].ext, x;
      if (P.outer[d].log2ext >= 0) { x = r & (ext - 1); r >>= P.outer[d].log2ext; }
      else { x = r % ext; r /= ext; }
      int64_t xa = x, xb = x;
      if (d == P.gather_dim) { // fused row gather (tile_offsets<true> of the GATHER instantiation)
        if (P.rows_a) { xa = P.rows_a[x]; if (xa < 0 || xa >= P.src_rows_a) { xa = 0; if (P.gather_err) *P.gather_err = 1; } }
        if (P.rows_b) { xb = P.rows_b[x]; if (xb < 0 || xb >= P.src_rows_b) { xb = 0; if (P.gather_err) *P.gather_err = 1; } }
      }
      offA += xa * P.outer[d].sA; offB += xb * P.outer[d].sB1; offC += x * P.outer[d].sC;
    }
    std::vector<double> acc((size_t)4 * MB * NB * 64 * 4, 0.0);
    auto ACC = [&](int wave, int a, int b, int lane, int rr) -> double & { return acc[((((size_t)wave * MB + a) * NB + b) * 64 + lane) * 4 + rr]; };
    int64_t ka = 0, kb = 0;
    for (int c = 0; c < n_chunks; ++c) {
      if (c > 0) {
        const int bit = __builtin_ctz((unsigned)c);
        const unsigned gn = (unsigned)c ^ ((unsigned)c >> 1);
        if ((gn >> bit) & 1) { ka += P.ko_sA[bit]; kb += P.ko_sB[bit]; } else { ka -= P.ko_sA[bit]; kb -= P.ko_sB[bit]; }
      }
      for (int which = 0; which < 2; ++which) {
        const int bits = which ? P.tb_bits : P.ta_bits;
        const int64_t *gs = which ? P.b_stride : P.a_stride;
        const int32_t *ls = which ? P.b_lds : P.a_lds;
        cd *img = which ? imgB.data() : imgA.data();
        const cd *src = which ? B + offB + kb : A + offA + ka;
        const int iters = bits > 8 ? 1 << (bits - 8) : 1;
        for (int tid = 0; tid < 256; ++tid) {
          if (!(bits >= 8 || tid < (1 << bits))) continue;
          for (int u = 0; u < iters; ++u) {
            const int e = tid + 256 * u;
            int64_t g = 0; int l = 0;
            for (int b = 0; b < bits; ++b) if ((e >> b) & 1) { g += gs[b]; l += ls[b]; }
            img[l / 16] = src[g];
          }
        }
      }
      for (int wave = 0; wave < 4; ++wave) {
        const int wn = wave & ((1 << P.wn_log2) - 1), wm = wave >> P.wn_log2;
        if (wm >= (1 << P.wm_log2)) continue;
        for (int s = 0; s < 4; ++s)
          for (int a = 0; a < MB; ++a)
            for (int b = 0; b < NB; ++b) {
              double Wv[64], Xv[64];
              for (int lane = 0; lane < 64; ++lane) {
                const int j = lane & 15, g = lane >> 4, ro = j & 1, p = g & 1, n_in = j >> 1;
                const bool w_valid = nt >= 3 || n_in < (1 << nt);
                const cd xv = imgA[((size_t)(2 * s + (g >> 1)) << PL) + (wm * MB + a) * 16 + j];
                cd bv(0.0, 0.0);
                if (w_valid) bv = imgB[((size_t)(2 * s + (g >> 1)) << PL) + (wn * NB + b) * 8 + n_in];
                Xv[lane] = p ? xv.imag() : xv.real();
                const double comp = (ro ^ p) ? bv.imag() : bv.real();
                Wv[lane] = (ro == 0 && p == 1) ? -comp : comp;
              }
              for (int lane = 0; lane < 64; ++lane)
                for (int rr = 0; rr < 4; ++rr) {
                  const int i = (lane >> 4) + 4 * rr, jj = lane & 15; // guide: f64 C/D map
                  for (int kk = 0; kk < 4; ++kk) ACC(wave, a, b, lane, rr) += Wv[i + 16 * kk] * Xv[jj + 16 * kk];
                }
            }
      }
    }
    auto m_off = [&](int m_local) { unsigned o = 0; for (int i = 0; i < mt; ++i) if ((m_local >> i) & 1) o |= 1u << P.m_pos[i]; return o; };
    auto n_off = [&](int n_local) { unsigned o = 0; for (int i = 0; i < nt; ++i) if ((n_local >> i) & 1) o |= 1u << P.n_pos[i]; return o; };
    const int n_lim = nt >= 3 ? 8 : 1 << nt;
    const int n_pass = 1 << (P.tc_bits - epi_bits);
    for (int pass = 0; pass < n_pass; ++pass) {
      std::vector<double> img((size_t)2 << epi_bits, -777.0);
      for (int wave = 0; wave < 4; ++wave) {
        const int wn = wave & ((1 << P.wn_log2) - 1), wm = wave >> P.wn_log2;
        if (wm >= (1 << P.wm_log2)) continue;
        for (int a = 0; a < MB; ++a) for (int b = 0; b < NB; ++b) for (int lane = 0; lane < 64; ++lane)
          for (int rr = 0; rr < 4; ++rr) {
            const int j = lane & 15, g = lane >> 4;
            const int n_loc = (g >> 1) + 2 * rr;
            if (n_loc >= n_lim) continue;
            const unsigned pos = swzg(m_off((wm * MB + a) * 16 + j) | n_off((wn * NB + b) * 8 + n_loc), P);
            if ((int)(pos >> ARTN_GEMM128_EPI_BITS) != pass) continue;
            img[2 * (size_t)(pos & ((1u << ARTN_GEMM128_EPI_BITS) - 1u)) + (g & 1)] = ACC(wave, a, b, lane, rr);
          }
      }
      const int iters = epi_bits > 8 ? 1 << (epi_bits - 8) : 1;
      const int64_t pass_off = P.tc_bits > epi_bits ? pass * P.out_stride[epi_bits] : 0;
      for (int tid = 0; tid < 256; ++tid) {
        if (!(epi_bits >= 8 || tid < (1 << epi_bits))) continue;
        for (int i = 0; i < iters; ++i) {
          const int e = tid + 256 * i;
          int64_t g = 0;
          for (int b = 0; b < epi_bits; ++b) if ((e >> b) & 1) g += P.out_stride[b];
          const unsigned l = swzg(((unsigned)pass << ARTN_GEMM128_EPI_BITS) | (unsigned)e, P) & ((1u << ARTN_GEMM128_EPI_BITS) - 1u);
          C[offC + pass_off + g] = cd(img[2 * (size_t)l], img[2 * (size_t)l + 1]);
        }
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

// ---- packed-operand GEMM (artn_pgemm_kernel.h), replayed from the same ArtnPackPlan ---------------------------------
// The packing passes unit by unit (same index arithmetic as artn_k_pack_bf16 / artn_k_pack_f32), then artn_k_pgemm /
// artn_k_pgemm3m tile by tile: tile index -> (m-outer, n-outer), the chunk images addressed as the kernels address them
// (planes 2t + h of 4 values / contracted value 2s + h), the T1 / T2 / T3 arithmetic of the complex64 form with its
// partial-sum flush, the C-ordered swizzled result image pass by pass and the copy-out through out_stride.  (The MFMA
// lane and accumulator maps are those of artn_k_gemm, replayed lane by lane in run_gemm above; here an output element
// is computed directly from the image elements its lanes read.)
static unsigned swzp(unsigned off, const ArtnPackPlan &P) {
  for (int i = 0; i < P.swz_n; ++i)
    if ((off >> P.swz_src[i]) & 1) off ^= 1u << P.swz_dst[i];
  return off;
}
static void pack_side(const ArtnPackSide &S, int n_ko, int kc_bits, bool bf, const cf *X, std::vector<cf> &out, int n_to) {
  const int rb = S.n_row;
  const size_t n_elems = (size_t)1 << (n_to + n_ko + kc_bits + rb);
  out.assign(n_elems, cf(0, 0));
  if (bf) { // unit = [row][plane g][chunk][tile], 4 values kc = 4g + u per unit
    const size_t n_units = n_elems / 4;
    for (size_t unit = 0; unit < n_units; ++unit) {
      size_t r = unit;
      int64_t src = 0;
      for (int i = 0; i < rb; ++i) if ((r >> i) & 1) src += S.row[i];
      r >>= rb;
      for (int q = 2; q < kc_bits; ++q) if ((r >> (q - 2)) & 1) src += S.kc[q];
      r >>= kc_bits - 2;
      for (int q = 0; q < n_ko; ++q) if ((r >> q) & 1) src += S.ko[q];
      r >>= n_ko;
      for (int q = 0; q < S.n_to; ++q) if ((r >> q) & 1) src += S.to[q];
      for (int u = 0; u < 4; ++u) {
        const cf e = X[src + ((u & 1) ? S.kc[0] : 0) + ((u & 2) ? S.kc[1] : 0)];
        out[unit * 4 + u] = cf(bf16r(e.real()), bf16r(e.imag()));
      }
    }
  } else { // unit = [row pair][k][chunk][tile]
    const size_t n_units = n_elems / 2;
    for (size_t unit = 0; unit < n_units; ++unit) {
      size_t r = unit;
      int64_t src = 0;
      for (int i = 1; i < rb; ++i) if ((r >> (i - 1)) & 1) src += S.row[i];
      r >>= rb - 1;
      for (int q = 0; q < kc_bits; ++q) if ((r >> q) & 1) src += S.kc[q];
      r >>= kc_bits;
      for (int q = 0; q < n_ko; ++q) if ((r >> q) & 1) src += S.ko[q];
      r >>= n_ko;
      for (int q = 0; q < S.n_to; ++q) if ((r >> q) & 1) src += S.to[q];
      out[unit * 2] = X[src];
      out[unit * 2 + 1] = X[src + S.row[0]];
    }
  }
}
static void run_pgemm(const ArtnPackPlan &P, const cf *A0, const cf *B0, cf *C) {
  const cf *A = P.swapped ? B0 : A0, *B = P.swapped ? A0 : B0;
  const bool bf = P.arith == 0;
  const int KC = P.kc_bits, RA = 1 << ARTN_PG_MT, RB = 1 << ARTN_PG_NT, TC = ARTN_PG_MT + ARTN_PG_NT, EPI = ARTN_PG_EPI_BITS;
  std::vector<cf> Ap, Bp;
  pack_side(P.a, P.n_ko, KC, bf, A, Ap, P.n_mo);
  pack_side(P.b, P.n_ko, KC, bf, B, Bp, P.n_no);
  const int n_chunks = 1 << P.n_ko;
  const size_t a_chunk = (size_t)RA << KC, b_chunk = (size_t)RB << KC; // elements per (tile, chunk)
  const int seg_len = (!bf && P.flush_chunks > 0 && P.flush_chunks < n_chunks) ? P.flush_chunks : n_chunks;
  const int nl = std::min(P.n_no, 3), ml = std::min(P.n_mo, 2);
  auto m_off = [&](int m) { unsigned o = 0; for (int i = 0; i < ARTN_PG_MT; ++i) if ((m >> i) & 1) o |= 1u << P.m_pos[i]; return o; };
  auto n_off = [&](int n) { unsigned o = 0; for (int i = 0; i < ARTN_PG_NT; ++i) if ((n >> i) & 1) o |= 1u << P.n_pos[i]; return o; };
  std::vector<cf> img((size_t)1 << EPI);
  for (int64_t tile = 0; tile < P.n_tiles; ++tile) {
    int64_t r = tile;
    int64_t no = r & ((1LL << nl) - 1); r >>= nl;
    int64_t mo = r & ((1LL << ml) - 1); r >>= ml;
    no |= (r & ((1LL << (P.n_no - nl)) - 1)) << nl; r >>= P.n_no - nl;
    mo |= r << ml;
    int64_t c_off = 0;
    for (int q = 0; q < P.n_mo; ++q) if ((mo >> q) & 1) c_off += P.c_mo[q];
    for (int q = 0; q < P.n_no; ++q) if ((no >> q) & 1) c_off += P.c_no[q];
    const cf *At = Ap.data() + (size_t)mo * n_chunks * a_chunk, *Bt = Bp.data() + (size_t)no * n_chunks * b_chunk;
    for (int seg = 0; seg < n_chunks; seg += seg_len) {
      // acc[wave][lane][block a][block b][register]: complex value per (m_local, n_local), kept as T1/T2/T3 in the 3M form
      std::vector<float> t1((size_t)RA * RB * 2, 0.f), t2, t3;
      if (!bf) { t2.assign((size_t)RA * RB, 0.f); t3.assign((size_t)RA * RB, 0.f); t1.assign((size_t)RA * RB, 0.f); }
      for (int c = seg; c < seg + seg_len; ++c) {
        const cf *ia = At + (size_t)c * a_chunk, *ib = Bt + (size_t)c * b_chunk; // the chunk's LDS images, in element units
        // per output element (m_local, n_local) of the tile: the image elements its MFMAs read -- planes 2t + h, unit slot u
        // (bf16) / contracted value 2s + h (fp32) -- in the order the chains accumulate them
        for (int m = 0; m < RA; ++m)
          for (int n = 0; n < RB; ++n) {
            if (bf) {
              float re = t1[((size_t)m * RB + n) * 2], im = t1[((size_t)m * RB + n) * 2 + 1];
              for (int t = 0; t < 4; ++t)
                for (int h = 0; h < 2; ++h)
                  for (int u = 0; u < 4; ++u) {
                    const cf x = ia[((size_t)(2 * t + h) * RA + m) * 4 + u], w = ib[((size_t)(2 * t + h) * RB + n) * 4 + u];
                    re = fmaf(w.real(), x.real(), re); re = fmaf(-w.imag(), x.imag(), re);
                    im = fmaf(w.imag(), x.real(), im); im = fmaf(w.real(), x.imag(), im);
                  }
              t1[((size_t)m * RB + n) * 2] = re; t1[((size_t)m * RB + n) * 2 + 1] = im;
            } else {
              float a1 = t1[(size_t)m * RB + n], a2 = t2[(size_t)m * RB + n], a3 = t3[(size_t)m * RB + n];
              for (int sidx = 0; sidx < (1 << (KC - 1)); ++sidx)
                for (int h = 0; h < 2; ++h) {
                  const cf x = ia[(size_t)(2 * sidx + h) * RA + m], w = ib[(size_t)(2 * sidx + h) * RB + n];
                  a1 = fmaf(w.real(), x.real(), a1);
                  a2 = fmaf(w.imag(), x.imag(), a2);
                  a3 = fmaf(w.real() + w.imag(), x.real() + x.imag(), a3);
                }
              t1[(size_t)m * RB + n] = a1; t2[(size_t)m * RB + n] = a2; t3[(size_t)m * RB + n] = a3;
            }
          }
      }
      // epilogue: register (m_local, n_local) -> swizzled image position -> pass -> copy-out
      for (int pass = 0; pass < (1 << (TC - EPI)); ++pass) {
        std::fill(img.begin(), img.end(), cf(0, 0));
        for (int m = 0; m < RA; ++m)
          for (int n = 0; n < RB; ++n) {
            const unsigned pos = swzp(m_off(m) | n_off(n), P);
            if ((int)(pos >> EPI) != pass) continue;
            const cf v = bf ? cf(t1[((size_t)m * RB + n) * 2], t1[((size_t)m * RB + n) * 2 + 1])
                            : cf(t1[(size_t)m * RB + n] - t2[(size_t)m * RB + n], t3[(size_t)m * RB + n] - t1[(size_t)m * RB + n] - t2[(size_t)m * RB + n]);
            img[pos & ((1u << EPI) - 1u)] = v;
          }
        for (int tid = 0; tid < 512; ++tid)
          for (int i = 0; i < 8; ++i) {
            const unsigned e0 = (unsigned)tid * 2u + (unsigned)i * 1024u; // elements e0, e0 + 1 of the pass
            int64_t o = c_off;
            const unsigned full = e0 | ((unsigned)pass << EPI);
            for (int b = 1; b < TC; ++b) if ((full >> b) & 1) o += P.out_stride[b];
            const unsigned src = swzp(e0, P) ^ (swzp((unsigned)pass << EPI, P) & ((1u << EPI) - 1u));
            for (int q = 0; q < 2; ++q) {
              cf v = img[src + q];
              if (seg > 0) v += C[o + q * P.out_stride[0]];
              C[o + q * P.out_stride[0]] = v;
            }
          }
      }
    }
  }
}

// ---- artn_k_xgemm (artn_xgemm_kernel.h), replayed thread by thread from the same ArtnXGemmPlan -----------------------
// Level tables, the per-tile row / column tables (indices past the end clamped), the copy slots of both modes with the
// zero padding of a group's last chunk, the k walk (groups of k.L0 values x the outer contracted labels), the MFMA lane
// and accumulator maps of both operand roles (TRANS), the partial-sum flush and the predicated stores.
static void run_xgemm(const ArtnXGemmPlan &P, const cf *A0, const cf *B0, cf *C) {
  const cf *A = P.swapped ? B0 : A0, *B = P.swapped ? A0 : B0;
  const int NB = P.nb, TM = ARTN_XG_TM, TN = 32 * NB, KC = P.kc, KCL = KC == 16 ? 4 : 3, RSTEP = 256 / KC, PA = artn_xg_pitch_a(), PB = artn_xg_pitch_b(NB);
  // level tables
  std::vector<uint32_t> mA0(256), mC0(256), mA1(256), mC1(256), nB0(256), nC0(256), nB1(256), nC1(256), kA(256), kB(256);
  auto level = [&](const ArtnXSide &S, std::vector<uint32_t> &a0, std::vector<uint32_t> &c0, std::vector<uint32_t> *a1, std::vector<uint32_t> *c1) {
    for (int i = 0; i < S.L0; ++i) artn_xg_decode(S, 0, S.n0, (uint32_t)i, a0[i], c0[i]);
    if (a1) for (int i = 0; i < S.L1; ++i) artn_xg_decode(S, S.n0, S.n1, (uint32_t)i, (*a1)[i], (*c1)[i]);
  };
  level(P.m, mA0, mC0, &mA1, &mC1);
  level(P.n, nB0, nC0, &nB1, &nC1);
  level(P.k, kA, kB, nullptr, nullptr);
  const uint32_t K0 = (uint32_t)P.k.L0, Mtot = (uint32_t)P.m.total, Ntot = (uint32_t)P.n.total;
  const int64_t n_chunks = P.k_groups * P.cpg;
  std::vector<cf> imgA((size_t)KC * PA), imgB((size_t)KC * PB);
  std::vector<uint32_t> rowA(TM), rowC(TM), colB(TM), colC(TM);
  for (int64_t tile = 0; tile < P.n_tiles; ++tile) {
    const uint32_t tu = (uint32_t)tile, r = tu / (uint32_t)P.tiles_n, tn = tu - r * (uint32_t)P.tiles_n;
    uint32_t hh = r / (uint32_t)P.tiles_m;
    const uint32_t tm = r - hh * (uint32_t)P.tiles_m, m0 = tm * TM, n0 = (uint32_t)P.col0 + tn * TN;
    uint32_t hA = 0, hB = 0, hC = 0;
    for (int i = 0; i < P.n_h; ++i) {
      const uint32_t e = (uint32_t)P.h_ext[i], q = hh / e, d = hh - q * e;
      hA += d * (uint32_t)P.h_sA[i]; hB += d * (uint32_t)P.h_sB[i]; hC += d * (uint32_t)P.h_sC[i];
      hh = q;
    }
    auto side = [&](const ArtnXSide &S, uint32_t first, int count, const std::vector<uint32_t> &t0a, const std::vector<uint32_t> &t0c,
                    const std::vector<uint32_t> &t1a, const std::vector<uint32_t> &t1c, std::vector<uint32_t> &oa, std::vector<uint32_t> &oc) {
      for (int loc = 0; loc < count; ++loc) {
        uint32_t idx = first + (uint32_t)loc;
        if (idx >= (uint32_t)S.total) idx = (uint32_t)S.total - 1;
        const uint32_t q0 = idx / (uint32_t)S.L0, i0 = idx - q0 * (uint32_t)S.L0, q1 = q0 / (uint32_t)S.L1, i1 = q0 - q1 * (uint32_t)S.L1;
        uint32_t o0, o1;
        artn_xg_decode(S, S.n0 + S.n1, S.n_lab - S.n0 - S.n1, q1, o0, o1);
        oa[loc] = o0 + t0a[i0] + t1a[i1];
        oc[loc] = o1 + t0c[i0] + t1c[i1];
      }
    };
    side(P.m, m0, TM, mA0, mC0, mA1, mC1, rowA, rowC);
    side(P.n, n0, TN, nB0, nC0, nB1, nC1, colB, colC);
    // accumulators of every (wave, block, product, lane, register)
    std::vector<float> acc((size_t)4 * NB * 3 * 64 * 16, 0.f);
    auto ACC = [&](int wave, int b, int t, int lane, int rr) -> float & { return acc[((((size_t)wave * NB + b) * 3 + t) * 64 + lane) * 16 + rr]; };
    bool flushed_before = false;
    int since_flush = 0;
    for (int64_t c = 0; c < n_chunks; ++c) {
      const uint32_t ig = (uint32_t)(c / P.cpg), iq = (uint32_t)(c % P.cpg);
      uint32_t gA, gB;
      artn_xg_decode(P.k, P.k.n0, P.k.n_lab - P.k.n0, ig, gA, gB);
      const uint32_t kbase = iq * KC;
      const int kvalid = (int)std::min<uint32_t>(K0 - kbase, KC);
      for (auto &x : imgA) x = cf(-777.f, -777.f);
      for (auto &x : imgB) x = cf(-777.f, -777.f);
      for (int tid = 0; tid < 256; ++tid) {
        for (int u = 0; u < TM * KC / 256; ++u) {
          int row, kk;
          if (P.amode) { kk = tid & (KC - 1); row = (tid >> KCL) + RSTEP * u; } else { row = tid & (TM - 1); kk = (tid >> 7) + 2 * u; }
          uint32_t kc = kbase + (uint32_t)kk;
          if (kc >= K0) kc = K0 - 1;
          const cf v = A[(uint32_t)(hA + gA + rowA[row] + kA[kc])];
          imgA[(size_t)kk * PA + row] = kk >= kvalid ? cf(0.f, 0.f) : v;
        }
        for (int u = 0; u < TN * KC / 256; ++u) {
          int col, kk;
          if (P.bmode) { kk = tid & (KC - 1); col = (tid >> KCL) + RSTEP * u; } else { col = (tid & 31) + 32 * (u % NB); kk = (tid >> 5) + 8 * (u / NB); }
          uint32_t kc = kbase + (uint32_t)kk;
          if (kc >= K0) kc = K0 - 1;
          const cf v = B[(uint32_t)(hB + gB + colB[col] + kB[kc])];
          imgB[(size_t)kk * PB + col] = kk >= kvalid ? cf(0.f, 0.f) : v;
        }
      }
      const int trips = (kvalid + 3) >> 2;
      for (int wave = 0; wave < 4; ++wave)
        for (int s = 0; s < 2 * trips; ++s)
          for (int b = 0; b < NB; ++b)
            for (int lane = 0; lane < 64; ++lane)
              for (int rr = 0; rr < 16; ++rr) {
                const int i = (rr & 3) + 8 * (rr >> 2) + 4 * (lane >> 5), jj = lane & 31;
                // srcA lane l: [row i = l & 31][kk = l >> 5]; srcB lane l: [kk = l >> 5][column j = l & 31]
                const int xi = P.trans ? i : jj, wi = P.trans ? jj : i; // index into the A image (rows of m) / the B image (columns of n)
                for (int kk = 0; kk < 2; ++kk) {
                  const cf xv = imgA[(size_t)(2 * s + kk) * PA + 32 * wave + xi];
                  const cf wv = imgB[(size_t)(2 * s + kk) * PB + 32 * b + wi];
                  ACC(wave, b, 0, lane, rr) += wv.real() * xv.real();
                  ACC(wave, b, 1, lane, rr) += wv.imag() * xv.imag();
                  ACC(wave, b, 2, lane, rr) += (wv.real() + wv.imag()) * (xv.real() + xv.imag());
                }
              }
      ++since_flush;
      const bool last = c + 1 == n_chunks;
      if (last || (P.flush_chunks > 0 && since_flush == P.flush_chunks)) {
        for (int wave = 0; wave < 4; ++wave)
          for (int b = 0; b < NB; ++b)
            for (int lane = 0; lane < 64; ++lane)
              for (int rr = 0; rr < 16; ++rr) {
                const int i = (rr & 3) + 8 * (rr >> 2) + 4 * (lane >> 5), jj = lane & 31;
                const uint32_t m_loc = 32 * wave + (P.trans ? i : jj), n_loc = 32 * b + (P.trans ? jj : i);
                if (m0 + m_loc >= Mtot || n0 + n_loc >= Ntot) continue;
                const float t1 = ACC(wave, b, 0, lane, rr), t2 = ACC(wave, b, 1, lane, rr), t3 = ACC(wave, b, 2, lane, rr);
                cf &dst = C[(uint32_t)(hC + rowC[m_loc] + colC[n_loc])];
                const cf val(t1 - t2, t3 - t1 - t2);
                dst = flushed_before ? dst + val : val;
              }
        flushed_before = true;
        since_flush = 0;
        std::fill(acc.begin(), acc.end(), 0.f);
      }
    }
  }
}

// ---- artn_k_xrow (artn_xrow_kernel.h): the row-streaming form, replayed lane by lane -- the wave's contiguous range of 16-row
// blocks, the lane's row (clamped past the end: loads the last row, stores nothing), the small operand's fragments (zero
// outside its extents), the three products of every v_mfma_f32_16x16x4_f32 step summed over its four contracted values in
// order, accumulator register r of lane (j, g) <-> column 16 blk + 4 g + r of row 16 b + j, the predicated stores.
static void run_xrow(const ArtnXGemmPlan &P, const cf *A0, const cf *B0, cf *C, int grid) {
  // artn_k_xrow wave by wave: three levels of (A, C) BYTE offsets, the launch's round-robin deal of 16-row blocks with the XCD
  // swizzle, positions advanced by artn_xrow_advance (the kernel's own function), buffer range checks (offset 0xffffffff: dropped)
  const cf *A = P.swapped ? B0 : A0, *B = P.swapped ? A0 : B0;
  const uint32_t Mtot = (uint32_t)P.m.total, Ktot = (uint32_t)P.k.total, Ntot = (uint32_t)P.n.total;
  const int S = artn_xrow_steps(P.k.total), NBK = artn_xrow_nbk(P.n.total), D = artn_xrow_depth(S);
  if (S < 1 || S > 12 || NBK > 3 || (int)Ntot > 16 * NBK) abort();
  const uint32_t L0 = (uint32_t)P.m.L0, L1 = (uint32_t)P.m.L1, L2 = Mtot / (L0 * L1);
  if ((uint64_t)L0 * L1 * L2 != Mtot || L2 > ARTN_XROW_L2_MAX) abort();
  std::vector<uint32_t> t0(2 * 256), t1(2 * 256), t2(2 * (size_t)L2);
  for (uint32_t i = 0; i < L0; ++i) { uint32_t a, c; artn_xg_decode(P.m, 0, P.m.n0, i, a, c); t0[2 * i] = a << 3; t0[2 * i + 1] = c << 3; }
  for (uint32_t i = 0; i < L1; ++i) { uint32_t a, c; artn_xg_decode(P.m, P.m.n0, P.m.n1, i, a, c); t1[2 * i] = a << 3; t1[2 * i + 1] = c << 3; }
  for (uint32_t i = 0; i < L2; ++i) { uint32_t a, c; artn_xg_decode(P.m, P.m.n0 + P.m.n1, P.m.n_lab - P.m.n0 - P.m.n1, i, a, c); t2[2 * i] = a << 3; t2[2 * i + 1] = c << 3; }
  const uint32_t n_blocks = (Mtot + 15) >> 4, per_it = 4u * (uint32_t)grid;
  const uint32_t n_it = ((n_blocks + per_it - 1) / per_it + (uint32_t)D) / (uint32_t)(D + 1) * (uint32_t)(D + 1);
  auto loadA = [&](uint32_t off) -> cf { return off > P.row_bytes_a - 8u ? cf(0.f, 0.f) : A[off >> 3]; };
  for (uint32_t blockIdx = 0; blockIdx < (uint32_t)grid; ++blockIdx)
    for (uint32_t wave = 0; wave < 4; ++wave)
      for (int lane = 0; lane < 64; ++lane) {
        const uint32_t j = (uint32_t)lane & 15, g = (uint32_t)lane >> 4;
        const uint32_t wg = (uint32_t)grid % 8u == 0u ? (blockIdx % 8u) * ((uint32_t)grid / 8u) + blockIdx / 8u : blockIdx;
        uint32_t m = 16u * (4u * wg + wave) + j;
        ArtnXRowPos pos, step;
        artn_xrow_place(m < Mtot ? m : Mtot - 1u, L0, L1, pos);
        artn_xrow_place(16u * per_it, L0, L1, step);
        for (uint32_t it = 0; it < n_it + (uint32_t)D; ++it) { // (the kernel issues D blocks' loads past its last multiply)
          const uint32_t i2 = pos.i2 < L2 ? pos.i2 : L2 - 1u;
          if (pos.i0 >= L0 || pos.i1 >= L1) abort();
          const uint32_t ra = t0[2 * pos.i0] + t1[2 * pos.i1] + t2[2 * i2];
          uint32_t rc = t0[2 * pos.i0 + 1] + t1[2 * pos.i1 + 1] + t2[2 * i2 + 1];
          if (m >= Mtot) rc = 0xffffffffu;
          else { // the tables and the carries agree with the plain decode of the row
            uint32_t ea, ec;
            artn_xg_decode(P.m, 0, P.m.n_lab, m, ea, ec);
            if ((ea << 3) != ra || (ec << 3) != rc) abort();
          }
          if (it < n_it)
            for (int blk = 0; blk < NBK; ++blk)
              for (int r = 0; r < 4; ++r) {
                const uint32_t n = 16u * (uint32_t)blk + 4 * g + (uint32_t)r;
                uint32_t nB, nC;
                artn_xg_decode(P.n, 0, P.n.n_lab, n < Ntot ? n : 0, nB, nC);
                float s1 = 0.f, s2 = 0.f, s3 = 0.f;
                for (uint32_t k = 0; k < 4u * (uint32_t)S; ++k) {
                  uint32_t kA, kB;
                  artn_xg_decode(P.k, 0, P.k.n_lab, k < Ktot ? k : Ktot - 1, kA, kB);
                  const cf w = (k < Ktot && n < Ntot) ? B[nB + kB] : cf(0.f, 0.f);
                  const cf x = loadA(ra + (kA << 3));
                  s1 += w.real() * x.real();
                  s2 += w.imag() * x.imag();
                  s3 += (w.real() + w.imag()) * (x.real() + x.imag());
                }
                const uint32_t off = (rc != 0xffffffffu && n < Ntot) ? rc + (nC << 3) : 0xffffffffu;
                if (off <= P.row_bytes_c - 8u) C[off >> 3] = cf(s1 - s2, s3 - s1 - s2);
                else if (off != 0xffffffffu) abort(); // (a row and column that exist lie inside the result)
              }
          m += 16u * per_it;
          artn_xrow_advance(pos, step, L0, L1);
        }
      }
}

// artn_k_xrow64 lane by lane: a wave = 64 rows, the register butterflies (v_permlane16_swap: odd 16-lane groups of the first
// register <-> even groups of the second; v_permlane32_swap: upper half of the first <-> lower half of the second) and the lane
// map of v_mfma_f32_16x16x4_f32 (A operand: lane (i, kk) = row i, contracted kk; B operand: lane (n, kk); D: lane (n, g), register
// r = row 4 g + r), positions through the kernel's own artn_xrow_place / artn_xrow_advance, buffer range checks.
static void xrow_bfly(float *a0, float *a1, float *a2, float *a3) {
  auto swap16 = [](float *a, float *b) { for (int q = 0; q < 2; ++q) for (int l = 0; l < 16; ++l) std::swap(a[32 * q + 16 + l], b[32 * q + l]); };
  auto swap32 = [](float *a, float *b) { for (int l = 0; l < 32; ++l) std::swap(a[32 + l], b[l]); };
  swap16(a0, a1); swap16(a2, a3); swap32(a0, a2); swap32(a1, a3);
}
static void xrow_mfma(const float *a, const float *b, float (*d)[4]) { // d[lane][r] += sum_kk a[(4 g + r) + 16 kk] b[n + 16 kk]
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const int n = l & 15, i = 4 * (l >> 4) + r;
      float s = d[l][r];
      for (int kk = 0; kk < 4; ++kk) s += a[i + 16 * kk] * b[n + 16 * kk];
      d[l][r] = s;
    }
}
static void run_xrow64(const ArtnXGemmPlan &P, const cf *A0, const cf *B0, cf *C, int grid) {
  const cf *A = P.swapped ? B0 : A0, *B = P.swapped ? A0 : B0;
  const uint32_t Mtot = (uint32_t)P.m.total, Ktot = (uint32_t)P.k.total, Ntot = (uint32_t)P.n.total;
  const int S = artn_xrow_steps(P.k.total), NBK = artn_xrow_nbk(P.n.total);
  if (S < 1 || S > 8 || NBK > 2) abort();
  const uint32_t L0 = (uint32_t)P.m.L0, L1 = (uint32_t)P.m.L1, L2 = Mtot / (L0 * L1);
  if ((uint64_t)L0 * L1 * L2 != Mtot || L2 > ARTN_XROW_L2_MAX) abort();
  std::vector<uint32_t> t0(2 * 256), t1(2 * 256), t2(2 * (size_t)L2), tcol(16 * NBK), ka(4 * S);
  for (uint32_t i = 0; i < L0; ++i) { uint32_t a, c; artn_xg_decode(P.m, 0, P.m.n0, i, a, c); t0[2 * i] = a << 3; t0[2 * i + 1] = c << 3; }
  for (uint32_t i = 0; i < L1; ++i) { uint32_t a, c; artn_xg_decode(P.m, P.m.n0, P.m.n1, i, a, c); t1[2 * i] = a << 3; t1[2 * i + 1] = c << 3; }
  for (uint32_t i = 0; i < L2; ++i) { uint32_t a, c; artn_xg_decode(P.m, P.m.n0 + P.m.n1, P.m.n_lab - P.m.n0 - P.m.n1, i, a, c); t2[2 * i] = a << 3; t2[2 * i + 1] = c << 3; }
  for (uint32_t n = 0; n < 16u * NBK; ++n) { uint32_t b, c; artn_xg_decode(P.n, 0, P.n.n_lab, n < Ntot ? n : 0, b, c); tcol[n] = c << 3; }
  for (uint32_t k = 0; k < 4u * S; ++k) { uint32_t a, b; artn_xg_decode(P.k, 0, P.k.n_lab, k < Ktot ? k : Ktot - 1, a, b); ka[k] = a << 3; }
  // fragments of the small operand, per lane
  std::vector<float> wr((size_t)NBK * S * 64), wi((size_t)NBK * S * 64), ws((size_t)NBK * S * 64);
  for (int blk = 0; blk < NBK; ++blk)
    for (int s = 0; s < S; ++s)
      for (int l = 0; l < 64; ++l) {
        const uint32_t n = 16u * blk + (l & 15), k = 4u * s + (l >> 4);
        uint32_t kA, kB, nB, nC;
        artn_xg_decode(P.k, 0, P.k.n_lab, k < Ktot ? k : Ktot - 1, kA, kB);
        artn_xg_decode(P.n, 0, P.n.n_lab, n < Ntot ? n : 0, nB, nC);
        const cf w = (k < Ktot && n < Ntot) ? B[nB + kB] : cf(0.f, 0.f);
        const size_t o = ((size_t)blk * S + s) * 64 + l;
        wr[o] = w.real(); wi[o] = w.imag(); ws[o] = w.real() + w.imag();
      }
  const uint32_t n_sb = (Mtot + 63) >> 6, per_it = 4u * (uint32_t)grid, n_it = (n_sb + per_it - 1) / per_it;
  auto loadA = [&](uint32_t voff, uint32_t soff) -> cf { return voff > P.row_bytes_a - 8u ? cf(0.f, 0.f) : A[((uint64_t)voff + soff) >> 3]; };
  std::vector<float> xr((size_t)S * 4 * 64), xi((size_t)S * 4 * 64);
  for (uint32_t blockIdx = 0; blockIdx < (uint32_t)grid; ++blockIdx)
    for (uint32_t wave = 0; wave < 4; ++wave) {
      const uint32_t wg = (uint32_t)grid % 8u == 0u ? (blockIdx % 8u) * ((uint32_t)grid / 8u) + blockIdx / 8u : blockIdx;
      uint32_t m[64], ra[64], rc[64];
      ArtnXRowPos pos[64], step;
      artn_xrow_place(64u * per_it, L0, L1, step);
      auto offsets = [&]() {
        for (int l = 0; l < 64; ++l) {
          const uint32_t i2 = pos[l].i2 < L2 ? pos[l].i2 : L2 - 1u;
          if (pos[l].i0 >= L0 || pos[l].i1 >= L1) abort();
          const uint32_t a = t0[2 * pos[l].i0] + t1[2 * pos[l].i1] + t2[2 * i2], c = t0[2 * pos[l].i0 + 1] + t1[2 * pos[l].i1 + 1] + t2[2 * i2 + 1];
          if (m[l] < Mtot) { // the tables and the carries agree with the plain decode of the row
            uint32_t ea, ec;
            artn_xg_decode(P.m, 0, P.m.n_lab, m[l], ea, ec);
            if ((ea << 3) != a || (ec << 3) != c) abort();
          }
          ra[l] = m[l] < Mtot ? a : 0xffffffffu;
          rc[l] = m[l] < Mtot ? c : 0xffffffffu;
        }
      };
      auto issue = [&](int s) {
        for (int gg = 0; gg < 4; ++gg)
          for (int l = 0; l < 64; ++l) {
            const cf v = loadA(4u * s + gg < Ktot ? ra[l] : 0xffffffffu, ka[4 * s + gg]);
            xr[((size_t)s * 4 + gg) * 64 + l] = v.real();
            xi[((size_t)s * 4 + gg) * 64 + l] = v.imag();
          }
      };
      for (int l = 0; l < 64; ++l) { m[l] = 64u * (4u * wg + wave) + (uint32_t)l; artn_xrow_place(m[l] < Mtot ? m[l] : Mtot - 1u, L0, L1, pos[l]); }
      offsets();
      for (int s = 0; s < S; ++s) issue(s);
      for (uint32_t it = 0; it < n_it; ++it) {
        uint32_t rc_cur[64];
        for (int l = 0; l < 64; ++l) { rc_cur[l] = rc[l]; m[l] += 64u * per_it; artn_xrow_advance(pos[l], step, L0, L1); }
        offsets();
        std::vector<float> acc((size_t)3 * 4 * NBK * 64 * 4, 0.f); // product, block q, column block, lane, register
        auto T = [&](int t, int q, int blk) { return reinterpret_cast<float (*)[4]>(&acc[((((size_t)t * 4 + q) * NBK + blk) * 64) * 4]); };
        for (int s = 0; s < S; ++s) {
          float *X = &xr[(size_t)s * 4 * 64], *Y = &xi[(size_t)s * 4 * 64];
          xrow_bfly(X, X + 64, X + 128, X + 192);
          xrow_bfly(Y, Y + 64, Y + 128, Y + 192);
          for (int q = 0; q < 4; ++q) {
            float xs[64];
            for (int l = 0; l < 64; ++l) xs[l] = X[64 * q + l] + Y[64 * q + l];
            for (int blk = 0; blk < NBK; ++blk) {
              const size_t o = ((size_t)blk * S + s) * 64;
              xrow_mfma(&wr[o], X + 64 * q, T(0, q, blk));
              xrow_mfma(&wi[o], Y + 64 * q, T(1, q, blk));
              xrow_mfma(&ws[o], xs, T(2, q, blk));
            }
          }
          issue(s);
        }
        for (int blk = 0; blk < NBK; ++blk)
          for (int r = 0; r < 4; ++r) {
            float re[4][64], im[4][64];
            for (int q = 0; q < 4; ++q)
              for (int l = 0; l < 64; ++l) {
                const float a = T(0, q, blk)[l][r], b = T(1, q, blk)[l][r], c = T(2, q, blk)[l][r];
                re[q][l] = a - b;
                im[q][l] = c - a - b;
              }
            xrow_bfly(re[0], re[1], re[2], re[3]);
            xrow_bfly(im[0], im[1], im[2], im[3]);
            for (int gg = 0; gg < 4; ++gg) {
              const uint32_t n = 16u * blk + 4u * gg + (uint32_t)r;
              for (int l = 0; l < 64; ++l) {
                const uint32_t voff = n < Ntot ? rc_cur[l] : 0xffffffffu;
                if (voff > P.row_bytes_c - 8u) { if (voff != 0xffffffffu) abort(); continue; }
                C[((uint64_t)voff + tcol[n]) >> 3] = cf(re[gg][l], im[gg][l]);
              }
            }
          }
      }
    }
}

// artn_k_xgemm128: the same walk with 16-byte elements, chunks of 8, and the lane / accumulator map of v_mfma_f64_16x16x4_f64
// (wave w: rows 32 w + 16 a + j, a = 0, 1; blocks b of 8 complex columns; lane (j, g): W row 2 n_in + ro with j = 2 n_in + ro,
// contracted value 2 s + (g >> 1), component p = g & 1; accumulator register r: component g & 1 of column (g >> 1) + 2 r).
static void run_xgemm128(const ArtnXGemmPlan &P, const cd *A0, const cd *B0, cd *C) {
  const cd *A = P.swapped ? B0 : A0, *B = P.swapped ? A0 : B0;
  const int NB = P.nb, TM = ARTN_XG_TM, TN = 32 * NB, KC = P.kc, KCL = 3, RSTEP = 256 / KC, PA = artn_xg_pitch_a(), PB = artn_xg_pitch_b(NB);
  if (KC != 8 || NB != 1) abort();
  std::vector<uint32_t> mA0(256), mC0(256), mA1(256), mC1(256), nB0(256), nC0(256), nB1(256), nC1(256), kA(256), kB(256);
  auto level = [&](const ArtnXSide &S, std::vector<uint32_t> &a0, std::vector<uint32_t> &c0, std::vector<uint32_t> *a1, std::vector<uint32_t> *c1) {
    for (int i = 0; i < S.L0; ++i) artn_xg_decode(S, 0, S.n0, (uint32_t)i, a0[i], c0[i]);
    if (a1) for (int i = 0; i < S.L1; ++i) artn_xg_decode(S, S.n0, S.n1, (uint32_t)i, (*a1)[i], (*c1)[i]);
  };
  level(P.m, mA0, mC0, &mA1, &mC1);
  level(P.n, nB0, nC0, &nB1, &nC1);
  level(P.k, kA, kB, nullptr, nullptr);
  const uint32_t K0 = (uint32_t)P.k.L0, Mtot = (uint32_t)P.m.total, Ntot = (uint32_t)P.n.total;
  const int64_t n_chunks = P.k_groups * P.cpg;
  std::vector<cd> imgA((size_t)KC * PA), imgB((size_t)KC * PB);
  std::vector<uint32_t> rowA(TM), rowC(TM), colB(TM), colC(TM);
  for (int64_t tile = 0; tile < P.n_tiles; ++tile) {
    const uint32_t tu = (uint32_t)tile, r = tu / (uint32_t)P.tiles_n, tn = tu - r * (uint32_t)P.tiles_n;
    uint32_t hh = r / (uint32_t)P.tiles_m;
    const uint32_t tm = r - hh * (uint32_t)P.tiles_m, m0 = tm * TM, n0 = tn * TN;
    uint32_t hA = 0, hB = 0, hC = 0;
    for (int i = 0; i < P.n_h; ++i) {
      const uint32_t e = (uint32_t)P.h_ext[i], q = hh / e, d = hh - q * e;
      hA += d * (uint32_t)P.h_sA[i]; hB += d * (uint32_t)P.h_sB[i]; hC += d * (uint32_t)P.h_sC[i];
      hh = q;
    }
    auto side = [&](const ArtnXSide &S, uint32_t first, int count, const std::vector<uint32_t> &t0a, const std::vector<uint32_t> &t0c,
                    const std::vector<uint32_t> &t1a, const std::vector<uint32_t> &t1c, std::vector<uint32_t> &oa, std::vector<uint32_t> &oc) {
      for (int loc = 0; loc < count; ++loc) {
        uint32_t idx = first + (uint32_t)loc;
        if (idx >= (uint32_t)S.total) idx = (uint32_t)S.total - 1;
        const uint32_t q0 = idx / (uint32_t)S.L0, i0 = idx - q0 * (uint32_t)S.L0, q1 = q0 / (uint32_t)S.L1, i1 = q0 - q1 * (uint32_t)S.L1;
        uint32_t o0, o1;
        artn_xg_decode(S, S.n0 + S.n1, S.n_lab - S.n0 - S.n1, q1, o0, o1);
        oa[loc] = o0 + t0a[i0] + t1a[i1];
        oc[loc] = o1 + t0c[i0] + t1c[i1];
      }
    };
    side(P.m, m0, TM, mA0, mC0, mA1, mC1, rowA, rowC);
    side(P.n, n0, TN, nB0, nC0, nB1, nC1, colB, colC);
    std::vector<double> acc((size_t)4 * 2 * 4 * NB * 64 * 4, 0.0); // wave, a, b, lane, register
    auto ACC = [&](int wave, int a, int b, int lane, int rr) -> double & { return acc[((((size_t)wave * 2 + a) * 4 * NB + b) * 64 + lane) * 4 + rr]; };
    bool flushed_before = false;
    int since_flush = 0;
    for (int64_t c = 0; c < n_chunks; ++c) {
      const uint32_t ig = (uint32_t)(c / P.cpg), iq = (uint32_t)(c % P.cpg);
      uint32_t gA, gB;
      artn_xg_decode(P.k, P.k.n0, P.k.n_lab - P.k.n0, ig, gA, gB);
      const uint32_t kbase = iq * KC;
      const int kvalid = (int)std::min<uint32_t>(K0 - kbase, KC);
      for (auto &x : imgA) x = cd(-777., -777.);
      for (auto &x : imgB) x = cd(-777., -777.);
      for (int tid = 0; tid < 256; ++tid) {
        for (int u = 0; u < TM * KC / 256; ++u) {
          int row, kk;
          if (P.amode) { kk = tid & (KC - 1); row = (tid >> KCL) + RSTEP * u; } else { row = tid & (TM - 1); kk = (tid >> 7) + 2 * u; }
          uint32_t kc = kbase + (uint32_t)kk;
          if (kc >= K0) kc = K0 - 1;
          const cd v = A[(uint32_t)(hA + gA + rowA[row] + kA[kc])];
          imgA[(size_t)kk * PA + row] = kk >= kvalid ? cd(0., 0.) : v;
        }
        for (int u = 0; u < TN * KC / 256; ++u) {
          int col, kk;
          if (P.bmode) { kk = tid & (KC - 1); col = (tid >> KCL) + RSTEP * u; } else { col = (tid & 31) + 32 * u; kk = tid >> 5; }
          uint32_t kc = kbase + (uint32_t)kk;
          if (kc >= K0) kc = K0 - 1;
          const cd v = B[(uint32_t)(hB + gB + colB[col] + kB[kc])];
          imgB[(size_t)kk * PB + col] = kk >= kvalid ? cd(0., 0.) : v;
        }
      }
      const int pairs = (kvalid + 1) >> 1;
      for (int wave = 0; wave < 4; ++wave)
        for (int s = 0; s < pairs; ++s)
          for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 4 * NB; ++b)
              for (int lane = 0; lane < 64; ++lane)
                for (int rr = 0; rr < 4; ++rr) {
                  // D[i][j] += sum over the four lane groups gg of Aop[i][gg] * Bop[gg][j]; this lane holds D[i = g + 4 rr][j]
                  const int jj = lane & 15, g = lane >> 4, i = g + 4 * rr;
                  const int n_in = i >> 1, ro = i & 1;
                  for (int gg = 0; gg < 4; ++gg) {
                    const int kk = 2 * s + (gg >> 1), pp = gg & 1;
                    const cd xv = imgA[(size_t)kk * PA + 32 * wave + 16 * a + jj];
                    const cd wv = imgB[(size_t)kk * PB + 8 * b + n_in];
                    const double x = pp ? xv.imag() : xv.real();
                    const double w = (ro ^ pp) ? wv.imag() : wv.real();
                    ACC(wave, a, b, lane, rr) += ((ro == 0 && pp == 1) ? -w : w) * x;
                  }
                }
      ++since_flush;
      const bool last = c + 1 == n_chunks;
      if (last || (P.flush_chunks > 0 && since_flush == P.flush_chunks)) {
        for (int wave = 0; wave < 4; ++wave)
          for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 4 * NB; ++b)
              for (int lane = 0; lane < 64; ++lane)
                for (int rr = 0; rr < 4; ++rr) {
                  const int jj = lane & 15, g = lane >> 4;
                  const uint32_t m_loc = 32 * wave + 16 * a + jj, n_loc = 8 * b + (g >> 1) + 2 * rr;
                  if (m0 + m_loc >= Mtot || n0 + n_loc >= Ntot) continue;
                  double *dst = reinterpret_cast<double *>(&C[(uint32_t)(hC + rowC[m_loc] + colC[n_loc])]) + (g & 1);
                  const double val = ACC(wave, a, b, lane, rr);
                  *dst = flushed_before ? *dst + val : val;
                }
        flushed_before = true;
        since_flush = 0;
        std::fill(acc.begin(), acc.end(), 0.0);
      }
    }
  }
}

// Force the extent-based GEMM plan; ARTN_E_UNSUPPORTED if make_xgemm declines.
extern "C" int artn_emulate_xgemm(const ArtnStepDesc *d, const void *A, const void *B, void *C, ArtnStepInfo *info, int32_t *modes) {
  ArtnPlan p;
  std::string err;
  int rc = artn::validate(d, err);
  if (rc) return rc;
  memset(&p.info, 0, sizeof(p.info));
  if (!artn::make_xgemm(d, p, (modes && modes[8] > 0) ? modes[8] : 256, 1)) return ARTN_E_UNSUPPORTED; // (modes[8] on entry: the CU count to plan for)
  if (modes && modes[7] == 1 && p.xg.rowmode == 2) { // (modes[7] = 1 on entry: the 16-row shape where the planner takes the 64-row one)
    p.xg.rowmode = 1;
    artn::xrow_fill_info(p.xg, p.info, (modes[8] > 0) ? modes[8] : 256);
  }
  if (info) *info = p.info;
  if (modes) { modes[0] = p.xg.amode; modes[1] = p.xg.bmode; modes[2] = p.xg.trans; modes[3] = p.xg.swapped; modes[4] = p.xg.nb; modes[5] = p.xg.flush_chunks; modes[6] = p.xg.kc; }
  if (modes) modes[7] = p.xg.rowmode;
  if (p.xg.c128) run_xgemm128(p.xg, (const cd *)A, (const cd *)B, (cd *)C);
  else if (p.xg.rowmode == 2) run_xrow64(p.xg, (const cf *)A, (const cf *)B, (cf *)C, p.info.grid);
  else if (p.xg.rowmode) run_xrow(p.xg, (const cf *)A, (const cf *)B, (cf *)C, p.info.grid);
  else {
    run_xgemm(p.xg, (const cf *)A, (const cf *)B, (cf *)C);
    if (p.xg.tail_nb) run_xgemm(artn_xg_tail_plan(p.xg), (const cf *)A, (const cf *)B, (cf *)C); // (the launcher's second launch)
  }
  if (modes) modes[8] = p.xg.tail_nb;
  return 0;
}

extern "C" int artn_emulate_pgemm(const ArtnStepDesc *d, const void *A, const void *B, void *C, ArtnStepInfo *info) {
  ArtnPlan p;
  std::string err;
  int rc = artn::validate(d, err);
  if (rc) return rc;
  memset(&p.info, 0, sizeof(p.info));
  const char *e = getenv("ARTN_EMU_NCU");
  if (!artn::make_pgemm(d, p, e ? atoi(e) : 256, true)) return ARTN_E_UNSUPPORTED;
  if (info) *info = p.info;
  run_pgemm(p.pack, (const cf *)A, (const cf *)B, (cf *)C);
  return 0;
}

extern "C" int artn_emulate(const ArtnStepDesc *d, const void *A, const void *B, void *C, int force_generic,
                            int *kernel_used) {
  ArtnPlan p;
  std::string err;
  int rc = artn::make_plan(d, p, err, 256, !force_generic, 1);
  if (rc) return rc;
  if (kernel_used) *kernel_used = p.kernel;
  if (d->dtype == ARTN_C128) {
    if (p.kernel == ARTN_KERNEL_BITS_MFMA) { run_bits128(p.bits, (const cd *)A, (const cd *)B, nullptr, (cd *)C); return 0; }
    if (p.kernel == ARTN_KERNEL_XGEMM) { run_xgemm128(p.xg, (const cd *)A, (const cd *)B, (cd *)C); return 0; }
    if (p.kernel != ARTN_KERNEL_GEMM_MFMA) return ARTN_E_UNSUPPORTED;
    run_gemm128(p.gemm, (const cd *)A, (const cd *)B, (cd *)C);
    return 0;
  }
  if (d->dtype != ARTN_C64 && !(d->dtype == ARTN_C64_BF16 && p.kernel == ARTN_KERNEL_GEMM_MFMA)) return ARTN_E_UNSUPPORTED;
  if (p.kernel == ARTN_KERNEL_BITS_MFMA) run_bits(p.bits, (const cf *)A, (const cf *)B, nullptr, (cf *)C);
  else if (p.kernel == ARTN_KERNEL_GEMM_MFMA) run_gemm(p.gemm, (const cf *)A, (const cf *)B, (cf *)C);
  else if (p.kernel == ARTN_KERNEL_XGEMM) {
    if (p.xg.rowmode == 2) run_xrow64(p.xg, (const cf *)A, (const cf *)B, (cf *)C, p.info.grid);
    else if (p.xg.rowmode) run_xrow(p.xg, (const cf *)A, (const cf *)B, (cf *)C, p.info.grid);
    else {
      run_xgemm(p.xg, (const cf *)A, (const cf *)B, (cf *)C);
      if (p.xg.tail_nb) run_xgemm(artn_xg_tail_plan(p.xg), (const cf *)A, (const cf *)B, (cf *)C);
    }
  } else run_generic(p.gen, (const cf *)A, (const cf *)B, (cf *)C);
  return 0;
}

// Force the two-operand GEMM plan (whatever the planner would prefer); ARTN_E_UNSUPPORTED if it declines.
extern "C" int artn_emulate_gemm(const ArtnStepDesc *d, const void *A, const void *B, void *C, ArtnStepInfo *info, int use_3m) {
  ArtnPlan p;
  std::string err;
  int rc = artn::validate(d, err);
  if (rc) return rc;
  memset(&p.info, 0, sizeof(p.info));
  const char *e = getenv("ARTN_EMU_NCU"); // 1: never shrink tiles for want of workgroups (covers the big-tile layouts)
  if (!artn::make_gemm(d, p, e ? atoi(e) : 256, 1, false, use_3m)) return ARTN_E_UNSUPPORTED;
  if (info) *info = p.info;
  run_gemm(p.gemm, (const cf *)A, (const cf *)B, (cf *)C);
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
  if (p.kernel == ARTN_KERNEL_GEMM_MFMA) { // (as artn_contract_gather: the kernel's first operand is B when the plan swapped them)
    const bool sw = p.gemm.swapped != 0;
    p.gemm.rows_a = sw ? rows_b : rows_a; p.gemm.rows_b = sw ? rows_a : rows_b;
    p.gemm.src_rows_a = sw ? src_rows_b : src_rows_a; p.gemm.src_rows_b = sw ? src_rows_a : src_rows_b;
    p.gemm.gather_err = err_flag;
    if (d->dtype == ARTN_C128) run_gemm128(p.gemm, (const cd *)A, (const cd *)B, (cd *)C);
    else run_gemm(p.gemm, (const cf *)A, (const cf *)B, (cf *)C);
    return 2;
  }
  if (d->dtype == ARTN_C128) return ARTN_E_UNSUPPORTED; // (no gathering state-streaming kernel in complex128)
  p.bits.rows_a = rows_a; p.bits.rows_b = rows_b;
  p.bits.src_rows_a = src_rows_a; p.bits.src_rows_b = src_rows_b;
  p.bits.gather_err = err_flag;
  run_bits(p.bits, (const cf *)A, (const cf *)B, nullptr, (cf *)C);
  return 0;
}

// Force the state-streaming plan of one step (complex64 or complex128); ARTN_E_UNSUPPORTED if make_bits declines.
extern "C" int artn_emulate_bits(const ArtnStepDesc *d, const void *A, const void *B, void *C, ArtnStepInfo *info) {
  ArtnPlan p;
  std::string err;
  int rc = artn::validate(d, err);
  if (rc) return rc;
  memset(&p.info, 0, sizeof(p.info));
  if (!artn::make_bits(d, nullptr, p, 256, 1)) return ARTN_E_UNSUPPORTED;
  if (info) *info = p.info;
  if (p.bits.c128) run_bits128(p.bits, (const cd *)A, (const cd *)B, nullptr, (cd *)C);
  else run_bits(p.bits, (const cf *)A, (const cf *)B, nullptr, (cf *)C);
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
  if (p.bits.c128) run_bits128(p.bits, (const cd *)A, (const cd *)B1, (const cd *)B2, (cd *)C);
  else run_bits(p.bits, (const cf *)A, (const cf *)B1, (const cf *)B2, (cf *)C);
  return 0;
}

// Diagnostic: LDS cycles per ds_read_b64 of the stage-1 operand reads (1 = conflict free): the 32
// lanes of a half wave read 8 bytes each, bank = (byte address / 4) mod 64.
// Fused triple.  Returns ARTN_E_UNSUPPORTED when the planner declines.
extern "C" int artn_emulate3(const ArtnStepDesc *d1, const ArtnStepDesc *d2, const ArtnStepDesc *d3, const void *A, const void *B1,
                             const void *B2, const void *B3, void *C, ArtnStepInfo *info) {
  ArtnPlan p;
  std::string err;
  int rc = artn::make_plan_fused3(d1, d2, d3, p, err, 256, A ? 1 : (1 << 14)); // (emulated on tensors of 2^16+ elements)
  if (rc) return rc;
  if (info) *info = p.info;
  if (A) run_bits(p.bits, (const cf *)A, (const cf *)B1, (const cf *)B2, (cf *)C, (const cf *)B3);
  return 0;
}

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
