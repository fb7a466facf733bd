// artn_plan.h -- host-side lowering of one pairwise contraction step (ArtnStepDesc, the
// label-list form of the einsum strings built at /root/reference/artensor/contraction.py:13-20)
// into the launch plan consumed by the gfx950 kernels in artn_kernels.hip.
//
// Pure C++ (no HIP): it is compiled into libartn_hip.so and, separately, into the
// CPU-only plan emulator the tests use to check the index algebra without a GPU.
//
// Vocabulary.  Every label whose extent is a power of two is split into *bits* (extent-2
// axes); a contraction over all-dims-2 circuit tensors is then a permutation of address
// bits around a small complex GEMM:
//   K bits  -- carried by A and B, not C (contracted)
//   M bits  -- carried by A and C only   (free bits of the big "state" operand)
//   N bits  -- carried by B and C only   (free bits of the small operand)
//   H axes  -- carried by all three      (batch: the sparse path's shared row label)
// A *tile* is the set {all K bits} u {M_t: a subset of M bits} of A, staged in LDS by one
// workgroup; it yields the C tile {M_t} u {N_t}.  Everything else is an *outer* axis
// enumerated by the tile index.
#ifndef ARTN_PLAN_H
#define ARTN_PLAN_H

#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>

#include "artn.h"

#define ARTN_MAX_OUTER 40
#define ARTN_TILE_BITS_MAX 13 /* 2^13 complex64 = 64 KiB of LDS */
#define ARTN_TILE_BITS_TARGET 12
#define ARTN_WG_THREADS 256

struct ArtnOuterDim {
  int64_t ext;
  int64_t sA, sB, sC; /* element strides; 0 where the operand does not carry the axis */
  int32_t log2ext;    /* >= 0 for powers of two, -1 otherwise */
  int32_t pad_;
};

// Launch plan of the LDS-tiled bit-permuted complex GEMM (kernel argument, POD).
struct ArtnBitsPlan {
  int32_t k, mt, nt;        // K / M_t / N_t bit counts
  int32_t T_in, T_out;      // k + mt, mt + nt
  int32_t wn_log2;          // waves along n' tiles (16 complex n each): max(0, nt - 4)
  int32_t wm_log2;          // waves along m sub-tiles: 2 - wn_log2
  int32_t pm;               // m sub-tiles (32 m each) per wave: 2^(mt-5) / 2^wm_log2
  int32_t n_outer;
  int32_t run_in, run_out;  // tile-local bits [0,run) are global bits [0,run)
  int64_t n_tiles;
  int64_t in_stride[ARTN_TILE_BITS_MAX];  // tile-local input bit  -> A element stride
  int64_t out_stride[ARTN_TILE_BITS_MAX]; // tile-local output bit -> C element stride
  int32_t lane_in_pos[5], lane_out_pos[5];   // MFMA column bit j -> tile-local in/out bit
  int32_t msub_in_pos[4], msub_out_pos[4];   // m sub-tile bit    -> tile-local in/out bit
  int32_t k_in_pos[6];                       // K bit i (kc bit i) -> tile-local in bit
  int32_t n_out_pos[6];                      // N_t bit i          -> tile-local out bit
  int64_t k_b_stride[6];                     // K bit i   -> B element stride
  int64_t n_b_stride[6];                     // N_t bit i -> B element stride
  ArtnOuterDim outer[ARTN_MAX_OUTER];
};

// Plan of the strided fallback (one thread per C element).
#define ARTN_GEN_MAX_OUT 64
#define ARTN_GEN_MAX_RED 32
struct ArtnGenericPlan {
  int32_t n_out, n_red;
  int64_t out_numel, red_numel;
  // output axes, fastest (stride_c == 1) first; adjacent axes that are contiguous in A
  // and B are merged, so these hold far fewer entries than the label count
  int64_t out_ext[ARTN_GEN_MAX_OUT], out_sA[ARTN_GEN_MAX_OUT], out_sB[ARTN_GEN_MAX_OUT];
  // reduction axes, innermost first
  int64_t red_ext[ARTN_GEN_MAX_RED], red_sA[ARTN_GEN_MAX_RED], red_sB[ARTN_GEN_MAX_RED];
};

struct ArtnPlan {
  int kernel; // ARTN_KERNEL_*
  ArtnBitsPlan bits;
  ArtnGenericPlan gen;
  ArtnStepInfo info;
  std::string why_generic;
};

namespace artn {

// Development knobs (environment: ARTN_WG_PER_CU, ARTN_TILE_TARGET, ARTN_RUN_MAX); the
// defaults are what ships.
struct Tuning {
  int wg_per_cu = 4;    // persistent workgroups per CU (grid = CUs * this)
  int tile_target = ARTN_TILE_BITS_TARGET;
  int run_max = 4;      // longest contiguous run (log2 elements) the tile is forced to keep
};
static inline Tuning &tuning() {
  static Tuning t = [] {
    Tuning x;
    if (const char *e = getenv("ARTN_WG_PER_CU")) x.wg_per_cu = std::max(1, atoi(e));
    if (const char *e = getenv("ARTN_TILE_TARGET")) x.tile_target = std::min(ARTN_TILE_BITS_MAX, std::max(9, atoi(e)));
    if (const char *e = getenv("ARTN_RUN_MAX")) x.run_max = std::min(6, std::max(1, atoi(e)));
    return x;
  }();
  return t;
}

struct Axis {
  int64_t ext, sA, sB, sC; // stride -1 = absent
  bool bit;
  bool inA() const { return sA >= 0; }
  bool inB() const { return sB >= 0; }
  bool inC() const { return sC >= 0; }
};

static inline int ilog2_exact(int64_t v) {
  if (v <= 0 || (v & (v - 1))) return -1;
  int l = 0;
  while ((int64_t(1) << l) < v) ++l;
  return l;
}

// Returns 0 or a negative ARTN_E_* with `err` set.
static inline int validate(const ArtnStepDesc *d, std::string &err) {
  if (!d) { err = "null descriptor"; return ARTN_E_INVALID; }
  if (d->dtype != ARTN_C64 && d->dtype != ARTN_C128) { err = "unknown dtype"; return ARTN_E_UNSUPPORTED; }
  if (d->n_labels < 0 || d->n_labels > ARTN_MAX_LABELS) { err = "n_labels out of range"; return ARTN_E_INVALID; }
  for (int l = 0; l < d->n_labels; ++l) {
    if (d->extent[l] < 1) { err = "label extent < 1"; return ARTN_E_INVALID; }
    bool a = d->stride_a[l] >= 0, b = d->stride_b[l] >= 0, c = d->stride_c[l] >= 0;
    if (!a && !b) { err = "label carried by neither input operand"; return ARTN_E_INVALID; }
  }
  // C must be dense row-major over its labels: sorted by stride, stride_i = prod of faster extents
  std::vector<std::pair<int64_t, int64_t>> cs;
  for (int l = 0; l < d->n_labels; ++l)
    if (d->stride_c[l] >= 0 && d->extent[l] > 1) cs.push_back({d->stride_c[l], d->extent[l]});
  std::sort(cs.begin(), cs.end());
  int64_t expect = 1;
  for (auto &p : cs) {
    if (p.first != expect) { err = "C is not dense row-major over its labels"; return ARTN_E_INVALID; }
    expect *= p.second;
  }
  return ARTN_OK;
}

static inline bool make_generic(const ArtnStepDesc *d, ArtnPlan &p, std::string &err) {
  ArtnGenericPlan &g = p.gen;
  memset(&g, 0, sizeof(g));
  std::vector<int> outl, redl;
  for (int l = 0; l < d->n_labels; ++l) {
    if (d->extent[l] == 1) continue;
    (d->stride_c[l] >= 0 ? outl : redl).push_back(l);
  }
  std::sort(outl.begin(), outl.end(), [&](int x, int y) { return d->stride_c[x] < d->stride_c[y]; });
  g.out_numel = 1;
  for (int l : outl) {
    int64_t e = d->extent[l];
    int64_t sa = d->stride_a[l] >= 0 ? d->stride_a[l] : 0, sb = d->stride_b[l] >= 0 ? d->stride_b[l] : 0;
    g.out_numel *= e;
    if (g.n_out > 0) { // C is dense, so only A and B decide whether two axes fuse
      int q = g.n_out - 1;
      if (sa == g.out_sA[q] * g.out_ext[q] && sb == g.out_sB[q] * g.out_ext[q]) { g.out_ext[q] *= e; continue; }
    }
    if (g.n_out >= ARTN_GEN_MAX_OUT) { err = "more than 64 unfusable output axes"; return false; }
    g.out_ext[g.n_out] = e; g.out_sA[g.n_out] = sa; g.out_sB[g.n_out] = sb;
    ++g.n_out;
  }
  // innermost reduction axis = the one with the smallest A stride (best locality)
  std::sort(redl.begin(), redl.end(), [&](int x, int y) {
    int64_t sx = d->stride_a[x] >= 0 ? d->stride_a[x] : d->stride_b[x];
    int64_t sy = d->stride_a[y] >= 0 ? d->stride_a[y] : d->stride_b[y];
    return sx < sy;
  });
  g.red_numel = 1;
  for (int l : redl) {
    if (g.n_red >= ARTN_GEN_MAX_RED) { err = "more than 32 reduction axes"; return false; }
    g.red_ext[g.n_red] = d->extent[l];
    g.red_sA[g.n_red] = d->stride_a[l] >= 0 ? d->stride_a[l] : 0;
    g.red_sB[g.n_red] = d->stride_b[l] >= 0 ? d->stride_b[l] : 0;
    g.red_numel *= d->extent[l];
    ++g.n_red;
  }
  p.kernel = ARTN_KERNEL_GENERIC;
  p.info.kernel = ARTN_KERNEL_GENERIC;
  int64_t blocks = (g.out_numel + ARTN_WG_THREADS - 1) / ARTN_WG_THREADS;
  p.info.grid = (int32_t)std::min<int64_t>(blocks, 1 << 20);
  p.info.n_tiles = blocks;
  return true;
}

// The bit-GEMM planner.  Returns false (with p.why_generic set) when the step does not
// fit the MFMA kernel's envelope.
static inline bool make_bits(const ArtnStepDesc *d, ArtnPlan &p, int n_cu, int64_t min_tiles) {
  if (d->dtype != ARTN_C64) { p.why_generic = "dtype is not complex64"; return false; }
  std::vector<Axis> ax;
  for (int l = 0; l < d->n_labels; ++l) {
    int64_t e = d->extent[l];
    if (e == 1) continue;
    int lg = ilog2_exact(e);
    if (lg > 0) {
      for (int jb = 0; jb < lg; ++jb) {
        Axis a;
        a.ext = 2;
        a.bit = true;
        a.sA = d->stride_a[l] >= 0 ? d->stride_a[l] << jb : -1;
        a.sB = d->stride_b[l] >= 0 ? d->stride_b[l] << jb : -1;
        a.sC = d->stride_c[l] >= 0 ? d->stride_c[l] << jb : -1;
        ax.push_back(a);
      }
    } else {
      Axis a{e, d->stride_a[l], d->stride_b[l], d->stride_c[l], false};
      ax.push_back(a);
    }
  }
  std::vector<int> K, M, N, O; // O: outer-only axes (H batch axes, non power-of-two free axes)
  for (int i = 0; i < (int)ax.size(); ++i) {
    const Axis &a = ax[i];
    if (a.inA() && a.inB() && !a.inC()) {
      if (!a.bit) { p.why_generic = "contracted label with a non power-of-two extent"; return false; }
      K.push_back(i);
    } else if (a.inA() && !a.inB() && a.inC()) {
      (a.bit ? M : O).push_back(i);
    } else if (!a.inA() && a.inB() && a.inC()) {
      (a.bit ? N : O).push_back(i);
    } else if (a.inA() && a.inB() && a.inC()) {
      O.push_back(i);
    } else {
      p.why_generic = "label summed out of a single operand";
      return false;
    }
  }
  const int k = (int)K.size(), m = (int)M.size(), n = (int)N.size();
  if (k < 1 || k > 6) { p.why_generic = "contracted bit count outside 1..6"; return false; }
  auto byA = [&](int x, int y) { return ax[x].sA < ax[y].sA; };
  auto byC = [&](int x, int y) { return ax[x].sC < ax[y].sC; };
  std::sort(K.begin(), K.end(), byA);
  std::sort(M.begin(), M.end(), byA);
  std::sort(N.begin(), N.end(), byC);

  // contiguous run at the bottom of A (over K u M bits) and of C (over M u N bits)
  auto run_len = [&](bool in_side) {
    int r = 0;
    for (; r < tuning().run_max; ++r) {
      bool found = false;
      for (int i = 0; i < (int)ax.size() && !found; ++i) {
        const Axis &a = ax[i];
        if (!a.bit) continue;
        bool cls = in_side ? (a.inA() && !(a.inA() && a.inB() && a.inC()))
                           : (a.inC() && !(a.inA() && a.inB() && a.inC()));
        int64_t s = in_side ? a.sA : a.sC;
        if (cls && s == (int64_t(1) << r)) found = true;
      }
      if (!found) break;
    }
    return r;
  };
  int run_in = run_len(true), run_out = run_len(false);
  if (run_in < 1 || run_out < 1) { p.why_generic = "no contiguous 16-byte run at the bottom of A or C"; return false; }

  // N_t: forced by the output run, then lowest C positions
  int nt_target = std::min(n, std::min(6, std::max(k, 4)));
  std::vector<int> Mt, Nt;
  int mt_min = 0, mt_max = 0, wn_log2 = 0;
  for (;;) {
    Nt.clear();
    Mt.clear();
    for (int i : N)
      if (ax[i].sC < (int64_t(1) << run_out)) Nt.push_back(i);
    for (int i : N) {
      if ((int)Nt.size() >= nt_target) break;
      if (std::find(Nt.begin(), Nt.end(), i) == Nt.end()) Nt.push_back(i);
    }
    if ((int)Nt.size() > 6) { p.why_generic = "too many forced N bits"; return false; }
    int nt = (int)Nt.size();
    wn_log2 = std::max(0, nt - 4);
    mt_max = std::min(std::min(9 - wn_log2, ARTN_TILE_BITS_MAX - k), ARTN_TILE_BITS_MAX - nt);
    mt_min = std::max(5, 7 - wn_log2);
    for (int i : M)
      if (ax[i].sA < (int64_t(1) << run_in) || ax[i].sC < (int64_t(1) << run_out)) Mt.push_back(i);
    if ((int)Mt.size() <= mt_max) break;
    // too many forced bits: shorten the longer run and retry
    if (run_out >= run_in && run_out > 1) --run_out;
    else if (run_in > 1) --run_in;
    else { p.why_generic = "forced tile bits exceed the LDS tile"; return false; }
  }
  if (mt_min > mt_max) { p.why_generic = "tile envelope empty"; return false; }
  int mt_target = std::min(mt_max, std::max(mt_min, tuning().tile_target - k));
  for (int i : M) {
    if ((int)Mt.size() >= mt_target) break;
    if (std::find(Mt.begin(), Mt.end(), i) == Mt.end()) Mt.push_back(i);
  }
  if ((int)Mt.size() < mt_min) { p.why_generic = "too few free A bits for a tile"; return false; }
  const int mt = (int)Mt.size(), nt = (int)Nt.size();

  ArtnBitsPlan &b = p.bits;
  memset(&b, 0, sizeof(b));
  b.k = k; b.mt = mt; b.nt = nt;
  b.T_in = k + mt; b.T_out = mt + nt;
  b.wn_log2 = wn_log2; b.wm_log2 = 2 - wn_log2;
  b.pm = (1 << (mt - 5)) >> b.wm_log2;
  b.run_in = run_in; b.run_out = run_out;
  if (b.pm < 1 || b.pm > 4) { p.why_generic = "m sub-tiles per wave outside 1..4"; return false; }

  // tile-local orders
  std::vector<int> tin(K), tout(Mt);
  tin.insert(tin.end(), Mt.begin(), Mt.end());
  tout.insert(tout.end(), Nt.begin(), Nt.end());
  std::sort(tin.begin(), tin.end(), byA);
  std::sort(tout.begin(), tout.end(), byC);
  auto pos_in = [&](int axis) { return (int)(std::find(tin.begin(), tin.end(), axis) - tin.begin()); };
  auto pos_out = [&](int axis) { return (int)(std::find(tout.begin(), tout.end(), axis) - tout.begin()); };
  for (int i = 0; i < b.T_in; ++i) b.in_stride[i] = ax[tin[i]].sA;
  for (int i = 0; i < b.T_out; ++i) b.out_stride[i] = ax[tout[i]].sC;
  for (int i = 0; i < run_in; ++i)
    if (b.in_stride[i] != (int64_t(1) << i)) { p.why_generic = "internal: input run broken"; return false; }
  for (int i = 0; i < run_out; ++i)
    if (b.out_stride[i] != (int64_t(1) << i)) { p.why_generic = "internal: output run broken"; return false; }
  std::vector<int> Mts(Mt);
  std::sort(Mts.begin(), Mts.end(), [&](int x, int y) { return pos_in(x) < pos_in(y); });
  for (int i = 0; i < 5; ++i) { b.lane_in_pos[i] = pos_in(Mts[i]); b.lane_out_pos[i] = pos_out(Mts[i]); }
  for (int i = 5; i < mt; ++i) { b.msub_in_pos[i - 5] = pos_in(Mts[i]); b.msub_out_pos[i - 5] = pos_out(Mts[i]); }
  for (int i = 0; i < k; ++i) { b.k_in_pos[i] = pos_in(K[i]); b.k_b_stride[i] = ax[K[i]].sB; }
  for (int i = 0; i < nt; ++i) { b.n_out_pos[i] = pos_out(Nt[i]); b.n_b_stride[i] = ax[Nt[i]].sB; }

  // outer axes: N-outer fastest (tiles sharing an A tile run together), then M-outer by A stride,
  // then batch / generic axes by A (or B) stride
  std::vector<int> outer;
  for (int i : N) if (std::find(Nt.begin(), Nt.end(), i) == Nt.end()) outer.push_back(i);
  for (int i : M) if (std::find(Mt.begin(), Mt.end(), i) == Mt.end()) outer.push_back(i);
  std::sort(O.begin(), O.end(), [&](int x, int y) {
    int64_t sx = ax[x].inA() ? ax[x].sA : ax[x].sB, sy = ax[y].inA() ? ax[y].sA : ax[y].sB;
    return sx < sy;
  });
  outer.insert(outer.end(), O.begin(), O.end());
  b.n_tiles = 1;
  int64_t a_rereads = 1;
  for (int i : outer) {
    const Axis &a = ax[i];
    ArtnOuterDim od;
    od.ext = a.ext;
    od.sA = a.inA() ? a.sA : 0;
    od.sB = a.inB() ? a.sB : 0;
    od.sC = a.inC() ? a.sC : 0;
    od.log2ext = ilog2_exact(a.ext);
    od.pad_ = 0;
    b.n_tiles *= a.ext;
    if (!a.inA()) a_rereads *= a.ext;
    if (b.n_outer > 0) { // merge with the previous dim when both are powers of two and contiguous everywhere
      ArtnOuterDim &pr = b.outer[b.n_outer - 1];
      bool okA = (pr.sA == 0 && od.sA == 0) || (pr.sA != 0 && od.sA == pr.sA * pr.ext);
      bool okB = (pr.sB == 0 && od.sB == 0) || (pr.sB != 0 && od.sB == pr.sB * pr.ext);
      bool okC = (pr.sC == 0 && od.sC == 0) || (pr.sC != 0 && od.sC == pr.sC * pr.ext);
      if (pr.log2ext >= 0 && od.log2ext >= 0 && okA && okB && okC && pr.log2ext + od.log2ext < 31) {
        pr.ext *= od.ext;
        pr.log2ext += od.log2ext;
        continue;
      }
    }
    if (b.n_outer >= ARTN_MAX_OUTER) { p.why_generic = "too many outer axes"; return false; }
    b.outer[b.n_outer++] = od;
  }

  // the copy phases move 16 bytes (two elements) per lane and need every thread busy
  if (b.T_in < 9 || b.T_out < 9) { p.why_generic = "tile smaller than one copy pass"; return false; }
  for (int i = 1; i < b.T_in; ++i) if (b.in_stride[i] & 1) { p.why_generic = "odd A stride"; return false; }
  for (int i = 1; i < b.T_out; ++i) if (b.out_stride[i] & 1) { p.why_generic = "odd C stride"; return false; }
  for (int i = 0; i < b.n_outer; ++i)
    if ((b.outer[i].sA & 1) || (b.outer[i].sC & 1)) { p.why_generic = "odd outer stride"; return false; }
  {
    // per-lane byte offsets inside the kernel are 32-bit: copy chunks span tile bits 1..8,
    // the small operand is addressed by its N_t / K bits
    int64_t si = 0, so = 0, sb = 0;
    for (int i = 1; i <= 8; ++i) { si += b.in_stride[i]; so += b.out_stride[i]; }
    for (int i = 0; i < nt; ++i) sb += b.n_b_stride[i];
    for (int i = 0; i < k; ++i) sb += b.k_b_stride[i];
    const int64_t lim = (int64_t(1) << 28) - 1; // elements: * 8 B < 2^31
    if (si > lim || so > lim || sb > lim) { p.why_generic = "lane offsets exceed 32 bits"; return false; }
  }
  if (b.n_tiles < min_tiles) { p.why_generic = "too few tiles to fill the chip"; return false; }

  p.kernel = ARTN_KERNEL_BITS_MFMA;
  ArtnStepInfo &f = p.info;
  f.kernel = ARTN_KERNEL_BITS_MFMA;
  f.k_bits = k; f.m_tile_bits = mt; f.n_tile_bits = nt;
  f.tile_in_bits = b.T_in; f.tile_out_bits = b.T_out;
  f.run_in_bits = run_in; f.run_out_bits = run_out;
  f.lds_bytes = 8 << std::max(b.T_in, b.T_out);
  f.n_tiles = b.n_tiles;
  f.a_rereads = a_rereads;
  int wg_per_cu = std::max(1, std::min(tuning().wg_per_cu, (160 * 1024) / f.lds_bytes));
  f.grid = (int32_t)std::min<int64_t>(b.n_tiles, (int64_t)n_cu * wg_per_cu);
  (void)m;
  return true;
}

// min_tiles: below this many LDS tiles the strided kernel is used instead (a handful of
// workgroups cannot fill 256 CUs; such steps are launch-latency bound either way).
static inline int make_plan(const ArtnStepDesc *d, ArtnPlan &p, std::string &err, int n_cu = 256,
                            bool allow_bits = true, int64_t min_tiles = 32) {
  int rc = validate(d, err);
  if (rc) return rc;
  memset(&p.info, 0, sizeof(p.info));
  double prod = 1, na = 1, nb = 1, nc = 1;
  for (int l = 0; l < d->n_labels; ++l) {
    prod *= (double)d->extent[l];
    if (d->stride_a[l] >= 0) na *= (double)d->extent[l];
    if (d->stride_b[l] >= 0) nb *= (double)d->extent[l];
    if (d->stride_c[l] >= 0) nc *= (double)d->extent[l];
  }
  bool ok = allow_bits && make_bits(d, p, n_cu, min_tiles);
  if (!ok && !make_generic(d, p, err)) return ARTN_E_UNSUPPORTED;
  p.info.flops = 8.0 * prod;
  p.info.bytes = (d->dtype == ARTN_C64 ? 8.0 : 16.0) * (na + nb + nc);
  return ARTN_OK;
}

} // namespace artn
#endif
