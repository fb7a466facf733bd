"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the
same seeded inputs, against the golden fixtures the reference produced, and -- at the full
n30 size -- against committed statistics of the reference's own 2^30-amplitude output.

Tolerances (complex64, north_star: <= 1e-5 relative): per-step results are compared with
max|diff| <= 1e-5 * max|want|; whole-scheme amplitudes with
|got - want| <= 1e-5 * max(|want|, rms(want)) for EVERY amplitude, i.e. 1e-5 relative for
amplitudes at or above the typical magnitude and 1e-5 of the typical magnitude below it.
(A pure per-amplitude relative bound is not meaningful for the small amplitudes of a
chaotic circuit: the reference's own complex64 output is 2.9e-5 away from a complex128
run of the same scheme under that metric, and 4.1e-7 under this one -- measured on n12.)

The STRICT figure of SURVEY 8c -- max relative error per amplitude over |amp| >= 1e-3 rms -- is
computed too (`amp_strict`) and asserted against the reference's own spread: tests/golden/
c128_spread.npz holds complex128 runs of the reference's executors on the same leaves and schemes
(tests/golden/make_golden.py c128_spread) and how far the reference's complex64 results are from
them; the HIP results have to stay within STRICT_FACTOR x that distance of the complex128 truth."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import artensor_amd as A
from artensor_amd import _native as N
from artensor_amd.fixtures import load_case
from oracle import oracle
from helpers import GOLDEN, crandn, dense_scheme_shapes, shrink_step

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
STEP_TOL = 1e-5


def gpu(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def rel(got, want):
    want = np.asarray(want)
    return np.abs(np.asarray(got) - want).max() / max(np.abs(want).max(), 1e-30)


def amp_rel(got, want, rms=None):
    """max over amplitudes of |got - want| / max(|want|, rms).  (In float64: the open bond-dimension-6 network's
    amplitudes are ~1e-26, whose squares underflow float32 -- the rms came out 0 and the metric became the strict one.)"""
    got, want = np.asarray(got, dtype=np.complex128).reshape(-1), np.asarray(want, dtype=np.complex128).reshape(-1)
    if rms is None:
        rms = np.sqrt(np.mean(np.abs(want) ** 2))
    return (np.abs(got - want) / np.maximum(np.abs(want), rms)).max()


def amp_strict(got, want):
    """SURVEY 8c's contract: max over amplitudes with |want| >= 1e-3 rms of |got - want| / |want|."""
    got, want = np.asarray(got, dtype=np.complex128).reshape(-1), np.asarray(want, dtype=np.complex128).reshape(-1)
    rms = np.sqrt(np.mean(np.abs(want) ** 2))
    sel = np.abs(want) >= 1e-3 * rms
    return float((np.abs(got - want)[sel] / np.abs(want)[sel]).max())


STRICT_FACTOR = 2.0   # HIP-vs-complex128 strict error allowed, in units of the reference's own complex64-vs-complex128 one
# The factor is 2 everywhere except where a measured reason is written down here (profiles/r03_truth_report.md has every
# figure with 3M arithmetic on and off; the same table is regenerated per round by tools/truth_round.sh):
#   n53_m20_batch_slice0 -- 1 024 amplitudes of which 9 lie under 0.1 rms, the smallest at 0.02 rms: the strict figure IS the
#       absolute error on that one amplitude divided by it (0.9e-6 rms for the reference's complex64 run, 2.4e-6 rms here), a
#       maximum over a handful of samples that moved 6.3e-5 / 8.0e-5 / 1.2e-4 (1.5x / 1.9x / 2.9x the reference's 4.15e-5) over
#       three builds that only reordered fp32 additions, with 4M arithmetic 8.0e-5 as well.  Allowed: 4x; the statistically
#       stable figure -- rms error over all amplitudes within 2x the reference's -- is asserted beside it for every key.
# (n30_dense_block_sums, 2.2x with 3M / 1.7x with 4M, is not asserted through this function: the block sums are checked
#  in complex128 only.)
STRICT_EXCEPTIONS = {"n53_m20_batch_slice0": 4.0}
_spread = None


def c128_spread():
    global _spread
    if _spread is None:
        import json
        z = np.load(os.path.join(GOLDEN, "c128_spread.npz"))
        _spread = (json.loads(bytes(z["meta"]).decode()), {k: z[k] for k in z.files if k != "meta"})
    return _spread


_truth = None


_EXACT128 = {"rand_D3_nv112_slice0": "rand_D3_nv112.npz", "rand_D6_nv64_slice0": "rand_D6_nv64.npz",
             "rand_D3_open6_nv96_final": "rand_D3_open6_nv96.npz", "rand_D6_open4_nv60_final": "rand_D6_open4_nv60.npz"}


def gpu_truth(key):
    """complex128 value of a checked quantity of a big case, computed on an MI355X by this package's complex128
    path (tests/golden/make_c128_truth_gpu.py; 1e-12 against complex128 einsum where that fits) and committed.
    (The round-5 fixtures of _EXACT128 carry the REFERENCE executor's own complex128 value instead: `exact128`.)"""
    global _truth
    if key in _EXACT128:
        return load_case(os.path.join(GOLDEN, _EXACT128[key])).arrays["exact128"].reshape(-1)
    if _truth is None:
        _truth = np.load(os.path.join(GOLDEN, "c128_truth_gpu.npz"))
    return _truth[key].reshape(-1)


def assert_contract(got, ref_c64, key, rms=None):
    """The north_star contract (complex64, <= 1e-5 relative) against the complex128 TRUTH of the same leaves and
    scheme, with the reference's own complex64 value measured beside it:
      loose  |got - truth| <= 1e-5 max(|truth|, rms) for every amplitude;
      strict max relative error over |truth| >= 1e-3 rms within STRICT_FACTOR (2) x the reference's own (STRICT_EXCEPTIONS: one key, 4);
      and the HIP value is no farther from the reference's than 1e-5 + the reference's distance to the truth."""
    t = gpu_truth(key)
    got, ref_c64 = np.asarray(got).reshape(-1), np.asarray(ref_c64).reshape(-1)
    ref_loose, ref_strict = amp_rel(ref_c64, t, rms), amp_strict(ref_c64, t)
    assert amp_rel(got, t, rms) <= 1e-5, (key, amp_rel(got, t, rms))
    # (floor 5e-6: the single-amplitude slices, where the reference's own figure is as low as 8e-7 by luck of one rounding)
    assert amp_strict(got, t) <= STRICT_EXCEPTIONS.get(key, STRICT_FACTOR) * max(ref_strict, 5e-6), (key, amp_strict(got, t), ref_strict)
    typ = rms if rms is not None else np.sqrt(np.mean(np.abs(t) ** 2))
    rms_err = lambda x: np.sqrt(np.mean(np.abs(x - t) ** 2)) / typ
    if t.size >= 100:   # (a statistic: not for the single-amplitude slices, where it is the loose figure again)
        assert rms_err(got) <= 2.0 * max(rms_err(ref_c64), 1e-6), (key, rms_err(got), rms_err(ref_c64))
    assert amp_rel(got, ref_c64, rms) <= 1e-5 + ref_loose, (key, amp_rel(got, ref_c64, rms), ref_loose)


def hip_step(eq, a, b):
    return A.contract(eq, gpu(a), gpu(b)).cpu().numpy()


def test_library_loaded_and_device_visible():
    assert N.lib().artn_device_count() >= 1
    assert torch.cuda.is_available()


def test_no_cpu_fallback():
    a = torch.zeros(2, 2, dtype=torch.complex64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        A.contract("ab,bc->ac", a, a)
    with pytest.raises(RuntimeError):
        A.tensor_contraction({0: a, 1: a}, [((0, 1), "ab,bc->ac")])


@pytest.mark.parametrize("force", ["ARTN_FORCE_BITS", "ARTN_FORCE_GENERIC", None])
@pytest.mark.parametrize("k,n,ra", [(1, 1, 12), (2, 0, 13), (3, 3, 14), (4, 4, 15), (5, 5, 15), (6, 6, 16),
                                    (4, 7, 13), (6, 2, 16), (1, 6, 12), (5, 1, 14), (7, 3, 15), (8, 4, 15), (7, 6, 16), (8, 0, 14), (3, 4, 20)])
def test_random_bit_steps(monkeypatch, force, k, n, ra):
    if force:
        monkeypatch.setenv(force, "1")
    rng = np.random.default_rng(100 * k + n)
    for trial in range(2):
        la = [chr(65 + x) for x in range(ra)]
        kl = list(rng.choice(la, size=k, replace=False))
        nl = [chr(97 + x) for x in range(n)]
        lb = kl + nl
        rng.shuffle(lb)
        lo = [x for x in la if x not in kl] + nl
        rng.shuffle(lo)
        eq = "".join(la) + "," + "".join(lb) + "->" + "".join(lo)
        a, b = crandn(rng, (2,) * ra), crandn(rng, (2,) * len(lb))
        assert rel(hip_step(eq, a, b), oracle.einsum_pair(eq, a, b)) < STEP_TOL, eq


def test_mfma_layout_identity_asymmetric():
    """A = identity on the contracted index with an ASYMMETRIC small operand: a swapped
    accumulator row/column map or a wrong re/im lane split cannot hide (guide section 3)."""
    k = n = 4
    ra = 14
    la = [chr(65 + x) for x in range(ra)]
    kl, nl = la[2:2 + k], [chr(97 + x) for x in range(n)]
    lo = [x for x in la if x not in kl] + nl
    eq = "".join(la) + "," + "".join(kl + nl) + "->" + "".join(lo)
    b = (np.arange(2 ** (k + n)).reshape((2,) * (k + n)) + 1j * (1000 + 3 * np.arange(2 ** (k + n)).reshape((2,) * (k + n)))).astype(np.complex64)
    a = np.zeros((2,) * ra, dtype=np.complex64)
    # A[m, kc] = delta(kc, m mod 16) * (1 + 2j) so every output picks exactly one B row
    idx = np.indices((2,) * ra).reshape(ra, -1)
    kc = sum(idx[2 + t] << (k - 1 - t) for t in range(k))
    mm = sum(idx[t] << t for t in range(ra) if t < 2 or t >= 2 + k) & 15
    a.reshape(-1)[np.flatnonzero(kc == mm)] = 1 + 2j
    os.environ["ARTN_FORCE_BITS"] = "1"
    try:
        got = hip_step(eq, a, b)
    finally:
        del os.environ["ARTN_FORCE_BITS"]
    assert np.array_equal(got, oracle.einsum_pair(eq, a, b))  # small integers: exact


def test_n30_big_steps_surrogates():
    """The 28 big steps of the n30 m14 scheme, state operand truncated to 2^22 elements."""
    case = load_case(os.path.join(GOLDEN, "n30_dense.npz"))
    steps = dense_scheme_shapes(case)
    big = [(n, s) for n, s in enumerate(steps) if np.prod(s[1]) >= 2 ** 20]
    assert len(big) == 28
    for n, (eq, sa, sb) in big:
        eq2, sa2, sb2 = shrink_step(eq, sa, sb, max_log2=22)
        rng = np.random.default_rng(n)
        a, b = crandn(rng, sa2), crandn(rng, sb2)
        info = A.step_info(eq2, sa2, sb2)
        # (a growth step whose second operand is itself big runs, unfused, on the two-operand GEMM kernel)
        assert info["kernel"] in (N.KERNEL_BITS_MFMA, N.KERNEL_GEMM_MFMA)
        assert rel(hip_step(eq2, a, b), oracle.einsum_pair(eq2, a, b)) < STEP_TOL, (n, eq2)


def test_n30_fused_pairs_surrogates():
    """Two consecutive big steps in ONE pass (artn_contract2): all 13 fusable pairs of the
    n30 scheme, both steps truncated consistently to 2^22 elements, against the oracle run
    step by step."""
    from artensor_amd.contraction import fusion_schedule, contract2
    from helpers import shrink_pair
    case = load_case(os.path.join(GOLDEN, "n30_dense.npz"))
    steps = dense_scheme_shapes(case)
    pairs = [e for e in fusion_schedule(case.scheme) if e[0] == "pair" and np.prod(steps[e[1]][1]) >= 2 ** 22]
    assert len(pairs) == 13
    fused = 0
    for _, n, m in pairs:
        eq1, sa, sb1 = steps[n]
        eq2, _, sb2 = steps[m]
        e1, a_s, b1_s, e2, b2_s = shrink_pair(eq1, sa, sb1, eq2, sb2, max_log2=22)
        rng = np.random.default_rng(n)
        a, b1, b2 = crandn(rng, a_s), crandn(rng, b1_s), crandn(rng, b2_s)
        got = contract2(e1, gpu(a), gpu(b1), e2, gpu(b2))
        if got is None:
            continue
        fused += 1
        want = oracle.einsum_pair(e2, oracle.einsum_pair(e1, a, b1), b2)
        assert rel(got.cpu().numpy(), want) < STEP_TOL, (n, m)
    assert fused >= 10


def test_random_fused_pairs(monkeypatch):
    from artensor_amd.contraction import contract2
    monkeypatch.setenv("ARTN_FORCE_BITS", "1")
    rng = np.random.default_rng(3)
    done = 0
    for trial in range(40):
        ra = int(rng.integers(13, 19))
        k1, n1, k2, n2 = (int(x) for x in rng.integers(1, 7, size=4))
        la = [chr(65 + x) for x in range(ra)]
        kl1 = list(rng.choice(la, size=k1, replace=False))
        nl1 = [chr(97 + x) for x in range(n1)]
        lb1 = kl1 + nl1
        rng.shuffle(lb1)
        lo1 = [x for x in la if x not in kl1] + nl1
        rng.shuffle(lo1)
        if k2 > len(lo1) - 6:
            continue
        kl2 = list(rng.choice(lo1, size=k2, replace=False))
        nl2 = [chr(110 + x) for x in range(n2)]
        lb2 = kl2 + nl2
        rng.shuffle(lb2)
        lo2 = [x for x in lo1 if x not in kl2] + nl2
        rng.shuffle(lo2)
        eq1 = "".join(la) + "," + "".join(lb1) + "->" + "".join(lo1)
        eq2 = "".join(lo1) + "," + "".join(lb2) + "->" + "".join(lo2)
        a, b1, b2 = crandn(rng, (2,) * ra), crandn(rng, (2,) * len(lb1)), crandn(rng, (2,) * len(lb2))
        got = contract2(eq1, gpu(a), gpu(b1), eq2, gpu(b2))
        if got is None:
            continue
        want = oracle.einsum_pair(eq2, oracle.einsum_pair(eq1, a, b1), b2)
        assert rel(got.cpu().numpy(), want) < STEP_TOL, (eq1, eq2)
        done += 1
    assert done >= 15


def test_fused_pairs_with_batch_labels(monkeypatch):
    """Fused pairs whose steps carry a batch label (the shared rows of the sparse executor): batch
    in both steps, in the first only, in the second only; power-of-two and ragged extents."""
    from artensor_amd.contraction import contract2
    monkeypatch.setenv("ARTN_FORCE_BITS", "1")
    rng = np.random.default_rng(11)
    done = 0
    for trial in range(30):
        ra = int(rng.integers(13, 18))
        k1, n1, k2, n2 = (int(x) for x in rng.integers(1, 6, size=4))
        hext = int(rng.choice([2, 3, 4, 7]))
        mode = trial % 3
        la = [chr(65 + x) for x in range(ra)]
        kl1 = list(rng.choice(la, size=k1, replace=False))
        nl1 = [chr(97 + x) for x in range(n1)]
        lb1 = kl1 + nl1
        rng.shuffle(lb1)
        lo1 = [x for x in la if x not in kl1] + nl1
        rng.shuffle(lo1)
        if k2 > len(lo1) - 6:
            continue
        kl2 = list(rng.choice(lo1, size=k2, replace=False))
        nl2 = [chr(110 + x) for x in range(n2)]
        lb2 = kl2 + nl2
        rng.shuffle(lb2)
        lo2 = [x for x in lo1 if x not in kl2] + nl2
        rng.shuffle(lo2)
        la_, lo1_, lo2_ = ["z"] + la, ["z"] + lo1, ["z"] + lo2
        lb1_ = (["z"] if mode in (0, 1) else []) + lb1
        lb2_ = (["z"] if mode in (0, 2) else []) + lb2
        shape = lambda labs: tuple(hext if x == "z" else 2 for x in labs)
        eq1 = "".join(la_) + "," + "".join(lb1_) + "->" + "".join(lo1_)
        eq2 = "".join(lo1_) + "," + "".join(lb2_) + "->" + "".join(lo2_)
        a, b1, b2 = crandn(rng, shape(la_)), crandn(rng, shape(lb1_)), crandn(rng, shape(lb2_))
        got = contract2(eq1, gpu(a), gpu(b1), eq2, gpu(b2))
        if got is None:
            continue
        want = oracle.einsum_pair(eq2, oracle.einsum_pair(eq1, a, b1), b2)
        assert rel(got.cpu().numpy(), want) < STEP_TOL, (eq1, eq2, hext, mode)
        done += 1
    assert done >= 12


def test_fused_and_unfused_schemes_agree(monkeypatch):
    case = load_case(os.path.join(GOLDEN, "n30_dense_sliced3.npz"))
    sliced = A.apply_slice(case.fresh_tensors(device=DEV), case.slicing_indices, [0, 1, 1])
    fused = A.tensor_contraction(dict(sliced), case.scheme)
    monkeypatch.setenv("ARTN_NO_FUSE", "1")
    A.contraction._pair_cache.clear()
    plain = A.tensor_contraction(dict(sliced), case.scheme)
    A.contraction._pair_cache.clear()
    d = (fused - plain).abs().max().item()
    assert d <= 1e-5 * plain.abs().max().item()


def test_batch_generic_dims_and_edge_cases():
    rng = np.random.default_rng(5)
    cases = [
        ("zabcdefghijk,zkcxy->zabdefghijxy", (5,) + (2,) * 11, (5, 2, 2, 2, 2)),
        ("pabcdefghijkl,qlcx->pqabdefghijkx", (3,) + (2,) * 12, (3, 2, 2, 2)),
        ("abcdefg,gcx->abdefx", (4,) * 7, (4, 4, 4)),
        ("abcdef,fcx->abdex", (3,) * 6, (3, 3, 3)),
        ("abc,cd->a", (2, 3, 4), (4, 2)),
        ("ab,ab->", (4, 4), (4, 4)),
        ("ab,cd->acbd", (2, 2), (2, 2)),
        ("a,a->a", (7,), (7,)),
    ]
    for force in (None, "ARTN_FORCE_BITS"):
        if force:
            os.environ[force] = "1"
        try:
            for eq, sa, sb in cases:
                a, b = crandn(rng, sa), crandn(rng, sb)
                assert rel(hip_step(eq, a, b), oracle.einsum_pair(eq, a, b)) < STEP_TOL, eq
        finally:
            if force:
                del os.environ[force]
    # non-contiguous operand views (a permuted A), complex128, empty output
    a, b = crandn(rng, (2,) * 14), crandn(rng, (2,) * 6)
    ta = gpu(a).permute(*reversed(range(14)))
    la = "".join(chr(65 + x) for x in range(14))
    eq = la[::-1] + ",ABCxyz->" + la[3:][::-1] + "xyz"
    got = A.contract(eq, ta, gpu(b)).cpu().numpy()
    assert rel(got, oracle.einsum_pair(eq, a.transpose(*reversed(range(14))), b)) < STEP_TOL
    a128, b128 = a.astype(np.complex128), b.astype(np.complex128)
    eq = la + ",ABCxyz->" + la[3:] + "xyz"
    assert rel(hip_step(eq, a128, b128), oracle.einsum_pair(eq, a128, b128)) < 1e-12
    e = A.contract("ab,bc->ac", torch.zeros(0, 2, dtype=torch.complex64, device=DEV),
                   torch.zeros(2, 3, dtype=torch.complex64, device=DEV))
    assert e.shape == (0, 3)
    with pytest.raises(RuntimeError):
        A.contract("ab,bc->ac", gpu(a[0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]), gpu(crandn(rng, (3, 2))))


def test_contract_gathered():
    """artn_contract_gather against gather + einsum in numpy; ragged row counts; an out-of-range
    index is flagged; a step the tiled kernel cannot take reports None."""
    from artensor_amd.contraction import contract_gathered
    rng = np.random.default_rng(23)
    # (the last three take the GEMM kernel's row gather: 7-8 contracted bits, 5+ free bits in the second operand -- the chunk
    #  steps of the n30 x 100 scheme; (4, 4, 6, 4, 7, 7) with the operands swapped inside the plan)
    for (na, nb, n, free, kb, nn) in [(7, 5, 6, 12, 3, 2), (16, 16, 9, 13, 4, 3), (3, 9, 619, 10, 8, 3), (40, 1, 33, 11, 5, 5),
                                      (9, 11, 100, 8, 8, 6), (6, 3, 37, 9, 7, 5), (4, 4, 6, 4, 7, 7)]:
        la = ["z"] + [chr(65 + x) for x in range(free + kb)]
        kl = la[1:1 + kb]
        nl = [chr(97 + x) for x in range(nn)]
        lb = ["z"] + kl[::-1] + nl
        lo = ["z"] + [x for x in la[1:] if x not in kl] + nl
        eq = "".join(la) + "," + "".join(lb) + "->" + "".join(lo)
        a = crandn(rng, (na,) + (2,) * (len(la) - 1))
        b = crandn(rng, (nb,) + (2,) * (len(lb) - 1))
        ra, rb = torch.from_numpy(rng.integers(0, na, size=n)), torch.from_numpy(rng.integers(0, nb, size=n))
        got = contract_gathered(eq, gpu(a), ra, gpu(b), rb)
        assert got is not None, eq
        want = oracle.einsum_pair(eq, a[ra.numpy()], b[rb.numpy()])
        assert rel(got.cpu().numpy(), want) < STEP_TOL, eq
        # one operand gathered only
        got = contract_gathered(eq, gpu(a), ra, gpu(b[rb.numpy()]), None)
        assert got is not None and rel(got.cpu().numpy(), want) < STEP_TOL
    # an index outside the operand: refused on the host (the reference raises IndexError) ...
    bad = torch.tensor([0, 99, 1, 2, 3, 4])
    big, small = gpu(crandn(rng, (7,) + (2,) * 12)), gpu(crandn(rng, (6, 2, 2, 2)))
    with pytest.raises(RuntimeError, match="row index out of range"):
        contract_gathered("zABCDEFGHIJKL,zLKa->zABCDEFGHIJa", big, bad, small, None)
    # ... and, should it ever reach the kernel, flagged there and raised by the next flag check
    A.contraction.check_gather_flag()
    contract_gathered("zABCDEFGHIJKL,zLKa->zABCDEFGHIJa", big, bad, small, None, _validate=False)
    with pytest.raises(RuntimeError, match="outside its operand"):
        A.contraction.check_gather_flag("test")
    A.contraction.check_gather_flag()   # cleared
    # negative indices count from the end, as in the reference's tensors[i][idx]
    neg = torch.tensor([-1, 0, -7, 3])
    a_np, b_np = crandn(rng, (7,) + (2,) * 12), crandn(rng, (4, 2, 2, 2))
    got = contract_gathered("zABCDEFGHIJKL,zLKa->zABCDEFGHIJa", gpu(a_np), neg, gpu(b_np), None)
    want = oracle.einsum_pair("zABCDEFGHIJKL,zLKa->zABCDEFGHIJa", a_np[neg.numpy()], b_np)
    assert rel(got.cpu().numpy(), want) < STEP_TOL
    assert contract_gathered("zab,zbc->zac", gpu(crandn(rng, (4, 2, 2))), torch.tensor([0, 1]),
                             gpu(crandn(rng, (4, 2, 2))), torch.tensor([1, 1])) is None
    # the same edge cases on the GEMM kernel's gather (8 contracted bits, 5 free bits in the second operand)
    eq_g = "zABCDEFGHIJKLMNOP,zPONMLKJIabcde->zABCDEFGHabcde"
    a_np, b_np = crandn(rng, (7,) + (2,) * 16), crandn(rng, (5,) + (2,) * 13)
    neg, rb = torch.tensor([-1, 0, -7, 3, 2]), torch.tensor([4, -5, 0, 2, -1])
    got = contract_gathered(eq_g, gpu(a_np), neg, gpu(b_np), rb)
    want = oracle.einsum_pair(eq_g, a_np[neg.numpy()], b_np[rb.numpy()])
    assert rel(got.cpu().numpy(), want) < STEP_TOL
    contract_gathered(eq_g, gpu(a_np), torch.tensor([0, 99, 1]), gpu(b_np), torch.tensor([0, 1, 2]), _validate=False)
    with pytest.raises(RuntimeError, match="outside its operand"):
        A.contraction.check_gather_flag("test")
    A.contraction.check_gather_flag()


def test_gather_axpy_normalize():
    rng = np.random.default_rng(9)
    for shape in [(37, 2, 2, 2), (5, 3), (16, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2)]:
        t = crandn(rng, shape)
        idx = torch.from_numpy(rng.integers(0, shape[0], size=53))
        got = A.contraction.gather_rows(gpu(t), idx).cpu().numpy()
        assert np.array_equal(got, t[idx.numpy()])
    t = crandn(rng, (6, 2, 2))  # every row in order: the tensor itself comes back
    assert np.array_equal(A.contraction.gather_rows(gpu(t), torch.arange(6)).cpu().numpy(), t)
    assert np.array_equal(A.contraction.gather_rows(gpu(t), torch.arange(5)).cpu().numpy(), t[:5])
    empty = A.contraction.gather_rows(gpu(crandn(rng, (4, 2))), torch.zeros(0, dtype=torch.int64))
    assert empty.shape == (0, 2)
    # out-of-range rows: refused on the host; in the kernel zero-filled and flagged, never read
    with pytest.raises(RuntimeError, match="row index out of range"):
        A.contraction.gather_rows(gpu(crandn(rng, (4, 2))), torch.tensor([0, 9]))
    A.contraction.check_gather_flag()
    z = A.contraction.gather_rows(gpu(crandn(rng, (4, 2))), torch.tensor([0, 9]), _validate=False)
    assert np.all(z.cpu().numpy()[1] == 0)
    with pytest.raises(RuntimeError, match="outside its operand"):
        A.contraction.check_gather_flag("test")
    t4 = crandn(rng, (4, 2))
    assert np.array_equal(A.contraction.gather_rows(gpu(t4), torch.tensor([-1, 0, -4])).cpu().numpy(), t4[[-1, 0, -4]])
    for rows, cols in [(2, 2), (5, 6), (64, 1024), (4096, 130), (131072, 16), (37, 4098)]:
        t = crandn(rng, (rows, cols))
        got = A.contraction.sum_leading(gpu(t), rows).cpu().numpy()
        want = t.astype(np.complex128).sum(axis=0)
        assert np.abs(got - want).max() <= 2e-6 * np.abs(t).max() * np.sqrt(rows) + 1e-30, (rows, cols)
    for n in (1, 7, 4096, 100003):
        x, y = crandn(rng, (n,)), crandn(rng, (n,))
        acc = gpu(x)
        A.accumulate(acc, gpu(y))
        assert np.array_equal(acc.cpu().numpy(), x + y)
    v = crandn(rng, (1000, 3))
    t = gpu(v)
    amax = A.contraction._normalize_inplace(t)
    want = np.abs(v).max()
    assert abs(amax.item() - want) <= 1e-6 * want
    assert rel(t.cpu().numpy(), v / want) < 1e-6


def test_n12_dense_scheme():
    case = load_case(os.path.join(GOLDEN, "n12_dense.npz"))
    raw = A.tensor_contraction(case.fresh_tensors(device=DEV), case.scheme)
    assert raw.shape == (2,) * 12 and raw.is_cuda
    raw = raw.cpu().numpy()
    assert amp_rel(raw, case.arrays["raw"]) < 1e-5
    final = raw.transpose(case.meta["permute_dims"]).reshape(-1)
    # the reference's state_vec() is another contraction order in complex64: its own tensor-network result is this
    # far from it, and the HIP result may be 1e-5 farther
    ref_final = case.arrays["raw"].transpose(case.meta["permute_dims"]).reshape(-1)
    assert amp_rel(final, case.arrays["state_vec"]) <= 1e-5 + amp_rel(ref_final, case.arrays["state_vec"])
    ora = oracle.tensor_contraction({i: t.numpy().copy() for i, t in case.tensors.items()}, case.scheme)
    assert amp_rel(raw, ora) < 1e-5
    # strict per-amplitude figure against the reference's complex128 run of the same scheme, in units of
    # the reference's own complex64 distance from it (2.9e-5 on this case)
    meta, arrays = c128_spread()
    truth = arrays["n12_dense_c128"]
    assert amp_strict(raw.reshape(-1), truth) <= STRICT_FACTOR * meta["n12_dense"]["strict"]
    for bits, (re, im) in case.meta["table"].items():
        assert abs(final[int(bits, 2)] - complex(re, im)) <= 1e-4 * abs(complex(re, im))


@pytest.mark.parametrize("name", ["n12_sparse5", "n30_sparse100"])
def test_sparse_schemes(name):
    case = load_case(os.path.join(GOLDEN, name + ".npz"))
    out = A.tensor_contraction_sparse(case.fresh_tensors(device=DEV), case.scheme).cpu().numpy()
    assert out.shape == case.arrays["final"].shape
    assert amp_rel(out, case.arrays["final"]) < 1e-5
    meta, arrays = c128_spread()
    assert amp_strict(out, arrays[name + "_c128"]) <= STRICT_FACTOR * max(meta[name]["strict"], 2e-6)
    if "google" in case.arrays:
        g = case.arrays["google"]
        assert (np.abs(out - g) / np.abs(g)).max() < 1e-3


@pytest.mark.parametrize("name", ["n12_sparse5", "n30_sparse100"])
def test_sparse_schemes_in_complex128(name):
    """The sparse executor with complex128 leaves (reference simulation.py:90 takes any dtype): fused pairs on
    artn_k_bits128, single steps on artn_k_gemm128 / the strided kernel, gathers through artn_gather_rows (the fused
    row gather is complex64 only).  Against the reference's own complex128 run of the same scheme (c128_spread.npz)
    and, for n30 x 100 bitstrings, the truth computed in round 3 with un-fused complex128 GEMM passes."""
    case = load_case(os.path.join(GOLDEN, name + ".npz"))
    out = A.tensor_contraction_sparse(case.fresh_tensors(dtype=torch.complex128, device=DEV), case.scheme).cpu().numpy()
    assert out.dtype == np.complex128 and out.shape == case.arrays["final"].shape
    meta, arrays = c128_spread()
    want = arrays[name + "_c128"].reshape(out.shape)
    assert np.abs(out - want).max() <= 1e-11 * np.abs(want).max()
    if name == "n30_sparse100":
        t = gpu_truth("n30_sparse100_final")
        assert np.abs(out.reshape(-1) - t).max() <= 1e-11 * np.abs(t).max()


def test_small_step_program(monkeypatch):
    """The launch-latency tail of a dense scheme as ONE launch (artn_program_*): n12 is 68 tiny steps and
    nothing else; with the program switched off the same scheme goes step by step through artn_contract.
    Both must give the reference's amplitudes, and the program must actually be in use by default."""
    from artensor_amd import contraction as C
    case = load_case(os.path.join(GOLDEN, "n12_dense.npz"))
    shapes = {k: tuple(t.shape) for k, t in case.tensors.items()}
    prog, main = C._plan_small_program(case.scheme, shapes, torch.complex64)
    assert prog is not None and prog.n_steps == 68 and main == []
    a = A.tensor_contraction(case.fresh_tensors(device=DEV), case.scheme).cpu().numpy()
    monkeypatch.setenv("ARTN_NO_PROGRAM", "1")
    C._plan_cache.clear()
    b = A.tensor_contraction(case.fresh_tensors(device=DEV), case.scheme).cpu().numpy()
    monkeypatch.delenv("ARTN_NO_PROGRAM")
    C._plan_cache.clear()
    assert amp_rel(a, case.arrays["raw"]) < 1e-5 and amp_rel(b, case.arrays["raw"]) < 1e-5
    assert amp_rel(a, b) < 2e-6
    # a network with bond dimension 3 (non power-of-two extents are decoded with divisions)
    case = load_case(os.path.join(GOLDEN, "rand_D3_open.npz"))
    out = A.tensor_contraction(case.fresh_tensors(device=DEV), case.scheme).cpu().numpy()
    assert rel(out, case.arrays["final"]) < 1e-5
    # programs that stop at 256-element tensors: several results leave through the workspace and feed ordinary
    # launches (results too big for the LDS arena -- the 2^14-element ones of n30 -- are covered by the n30 tests)
    monkeypatch.setattr(C, "PROGRAM_MAX_NUMEL", 256)
    for name, key in (("n12_dense", "raw"), ("rand_D4_closed", "final")):
        C._plan_cache.clear()
        case = load_case(os.path.join(GOLDEN, name + ".npz"))
        out = A.tensor_contraction(case.fresh_tensors(device=DEV), case.scheme).cpu().numpy()
        assert amp_rel(out, case.arrays[key]) < 1e-5
    C._plan_cache.clear()


def test_small_step_program_in_complex128(monkeypatch):
    """artn_k_program<double>: the same one-launch program for complex128 tensors (16-byte elements in the LDS arena, vector
    ALU only).  n12 dense = 68 tiny steps in ONE launch; against the step-by-step executor (same sums: 1e-14) and against
    numpy's complex128 einsums of the same scheme (reference loop contraction.py:66-70 run in complex128)."""
    from artensor_amd import contraction as C
    case = load_case(os.path.join(GOLDEN, "n12_dense.npz"))
    shapes = {k: tuple(t.shape) for k, t in case.tensors.items()}
    prog, main = C._plan_small_program(case.scheme, shapes, torch.complex128)
    assert prog is not None and prog.n_steps == 68 and main == [] and prog.dtype == torch.complex128

    class Count:
        def __init__(self):
            self.kernels = []

        def record(self, info, e0, e1):
            self.kernels.append(info["kernel"])

    counter = Count()
    monkeypatch.setattr(C, "profiler", counter)
    a = A.tensor_contraction(case.fresh_tensors(dtype=torch.complex128, device=DEV), case.scheme).cpu().numpy()
    monkeypatch.setattr(C, "profiler", None)
    assert counter.kernels == [C.KERNEL_PROGRAM]
    monkeypatch.setenv("ARTN_NO_PROGRAM", "1")
    C._plan_cache.clear()
    b = A.tensor_contraction(case.fresh_tensors(dtype=torch.complex128, device=DEV), case.scheme).cpu().numpy()
    monkeypatch.delenv("ARTN_NO_PROGRAM")
    C._plan_cache.clear()
    assert a.dtype == np.complex128 and np.abs(a - b).max() <= 1e-13 * np.abs(b).max()
    want = oracle.tensor_contraction({k: v.numpy().astype(np.complex128) for k, v in case.tensors.items()}, case.scheme)
    assert np.abs(a - want).max() <= 1e-12 * np.abs(want).max()
    assert amp_rel(a.astype(np.complex64), case.arrays["raw"]) < 1e-5
    # bond dimension 3 (divisions in the decode), a sparse-state scheme (hoisted program) and the slice loop's batches
    case = load_case(os.path.join(GOLDEN, "rand_D3_open.npz"))
    out = A.tensor_contraction(case.fresh_tensors(dtype=torch.complex128, device=DEV), case.scheme).cpu().numpy()
    want = oracle.tensor_contraction({k: v.numpy().astype(np.complex128) for k, v in case.tensors.items()}, case.scheme)
    assert np.abs(out - want).max() <= 1e-12 * np.abs(want).max()
    case = load_case(os.path.join(GOLDEN, "n12_sparse_sliced.npz"))
    got = A.sliced_contraction(case.tensors, case.scheme, case.slicing_indices, case.arrays["final"].shape, sparse=True,
                               dtype=torch.complex128, device=DEV)
    assert got.dtype == torch.complex128 and amp_rel(got.cpu().numpy().astype(np.complex64), case.arrays["final"]) < 1e-5


def test_deferred_row_select(monkeypatch):
    """A branch-(C) row select of a big tensor (reference contraction.py:187) is deferred to its consumer: the chunk loop
    that follows it in the n30 x 10 000 scheme reads the un-selected tensor through composed indices (`base[idx][rows] ==
    base[idx[rows]]`), the 2 x 7.8 GB copy of the select never happens.  Same bits as the eager order, against the reference's
    amplitudes, and the big gather is gone."""
    from artensor_amd import contraction as C
    case = load_case(os.path.join(GOLDEN, "n30_sparse10000.npz"))
    seen = []
    orig = C.gather_rows

    def spy(t, idx, _validate=True):
        seen.append(int(t.numel()))
        return orig(t, idx, _validate)
    monkeypatch.setattr(C, "gather_rows", spy)
    lazy = A.tensor_contraction_sparse(case.fresh_tensors(device=DEV), case.scheme).cpu().numpy()
    big_lazy = [n for n in seen if n >= C.LAZY_SELECT_MIN_NUMEL]
    seen.clear()
    monkeypatch.setenv("ARTN_NO_LAZY_SELECT", "1")
    eager = A.tensor_contraction_sparse(case.fresh_tensors(device=DEV), case.scheme).cpu().numpy()
    big_eager = [n for n in seen if n >= C.LAZY_SELECT_MIN_NUMEL]
    # (the eager order gathers the 2^30-element result of step 145; deferred, only the 2^24.3-element gather of a later
    #  gathered step is left)
    assert max(big_eager) >= 2 ** 29 and max(big_lazy + [0]) < 2 ** 26 and len(big_lazy) == len(big_eager) - 1, (big_eager, big_lazy)
    assert np.array_equal(lazy, eager)
    assert amp_rel(lazy.reshape(-1), case.arrays["final"].reshape(-1)) <= 1e-5
    # composed indices: cached per (select, rows) pair, out-of-range rows refused like the reference's IndexError
    sel, rows = torch.tensor([5, 3, 9, 1]), torch.tensor([2, 0, -1])
    assert C._composed(sel, rows).tolist() == [9, 5, 1] and C._composed(sel, rows) is C._composed(sel, rows)
    with pytest.raises(RuntimeError, match="row index out of range"):
        C._composed(sel, torch.tensor([4]))
    # a deferred select that nobody indexes is materialised: (big select) then a plain step
    rng = np.random.default_rng(8)
    a = gpu(crandn(rng, (64,) + (2,) * 18))
    lz = C._RowsOf(a, torch.tensor([3, 1, 60]))
    assert lz.shape == (3,) + (2,) * 18 and torch.equal(C.rows_of(lz), a[[3, 1, 60]])


def test_sparse_scientific_notation():
    case = load_case(os.path.join(GOLDEN, "n12_sparse5_scinot.npz"))
    factor, out = A.tensor_contraction_sparse(case.fresh_tensors(device=DEV), case.scheme, scientific_notation=True)
    assert abs(factor.cpu().item().real - case.arrays["factor"].real) < 1e-4
    assert amp_rel(out.cpu().numpy(), case.arrays["final"]) < 1e-5


def test_sliced_sparse_loop():
    case = load_case(os.path.join(GOLDEN, "n12_sparse_sliced.npz"))
    sim = A.TensorNetworkSimulation.from_case(case)
    out = sim.contraction(device=DEV).cpu().numpy()
    assert amp_rel(out, case.arrays["final"]) < 1e-5
    assert amp_rel(out, case.arrays["state_vec_at"]) < 5e-5


@pytest.mark.parametrize("name", ["rand_D2_closed", "rand_D3_open", "rand_D4_closed",
                                  "rand_D2_open_sliced", "rand_D2_closed_sliced"])
def test_random_networks(name):
    case = load_case(os.path.join(GOLDEN, name + ".npz"))
    want = case.arrays["final"]
    out = A.sliced_contraction(case.tensors, case.scheme, case.slicing_indices or {}, want.shape,
                               device=DEV).cpu().numpy()
    assert rel(out, want) < 1e-5
    assert rel(out, case.arrays["exact128"]) < 1e-5
    if case.slicing_indices:
        # the three ways through the slice loop agree: reuse of small intermediates (default, Gray
        # order), plain loop, whole slices replayed from a captured HIP graph (dense executor)
        plain = A.sliced_contraction(case.tensors, case.scheme, case.slicing_indices, want.shape, device=DEV,
                                     reuse_small=False).cpu().numpy()
        replay = A.sliced_contraction(case.tensors, case.scheme, case.slicing_indices, want.shape, device=DEV,
                                      reuse_small=False, graph=True).cpu().numpy()
        assert rel(plain, want) < 1e-5 and np.array_equal(plain, replay)


def test_n30_sparse_10000():
    case = load_case(os.path.join(GOLDEN, "n30_sparse10000.npz"))
    out = A.tensor_contraction_sparse(case.fresh_tensors(device=DEV), case.scheme).cpu().numpy()
    assert out.shape == (10000,)
    assert amp_rel(out, case.arrays["final"]) < 1e-5
    assert_contract(out, case.arrays["final"], "n30_sparse10000_final")
    # against Google's Schroedinger-Feynman amplitudes the reference itself only reaches
    # 9.5e-4 (complex64 gate constants); allow the same order here
    g = case.arrays["google"]
    assert (np.abs(out - g) / np.abs(g)).max() < 2e-3


def test_n30_dense_full_size():
    """BASELINE config 2 at full size: all 2^30 amplitudes on one MI355X, checked against
    statistics of the reference's own output (tests/golden/make_golden.py::case_n30_run)
    and Google's Schroedinger-Feynman amplitudes."""
    case = load_case(os.path.join(GOLDEN, "n30_dense.npz"))
    raw = A.tensor_contraction(case.fresh_tensors(device=DEV), case.scheme)
    assert raw.numel() == 2 ** 30
    perm = case.meta["permute_dims"]
    flat = raw.reshape(-1)
    rms = 2.0 ** -15  # 2^30 amplitudes of a normalised state
    # final = raw.permute(perm) (reference simulation.py:115-116) is a view; torch cannot
    # materialise a 30-dim permutation on the GPU, so final positions are mapped to raw ones
    fpos = np.array([int(b, 2) for b in case.meta["google_bitstrings"]], dtype=np.int64)
    at = flat[torch.from_numpy(raw_index(fpos, perm)).to(DEV)].cpu().numpy()
    assert amp_rel(at, case.arrays["amps_at_google"], rms) < 1e-5
    assert_contract(at, case.arrays["amps_at_google"], "n30_dense_at_google", rms)
    g = case.arrays["google"]
    assert (np.abs(at - g) / np.abs(g)).max() < 1e-3
    spos = np.arange(len(case.arrays["strided"]), dtype=np.int64) * (2 ** 14 + 1)
    strided = flat[torch.from_numpy(raw_index(spos, perm)).to(DEV)].cpu().numpy()
    assert amp_rel(strided, case.arrays["strided"], rms) < 1e-5
    assert_contract(strided, case.arrays["strided"], "n30_dense_strided", rms)
    # the 1024 block sums of `final` fix its 10 leading qubits: in `raw` those are dims
    # perm[0..9]; move them to the front with a <=16-dim view and reduce the rest
    lead = perm[:10]
    rest = [d for d in range(30) if d not in lead]
    # group `rest` into contiguous runs so the view has few dims
    blocks = block_sums(raw, lead)
    want = case.arrays["block_sums"]
    assert np.abs(blocks - want).max() <= 1e-5 * np.abs(want).max()
    tb = gpu_truth("n30_dense_block_sums")   # sums of 2^20 amplitudes each, against the complex128 truth
    assert np.abs(blocks - tb).max() <= 1e-5 * np.abs(tb).max()
    norm2 = float((flat.real.double() ** 2 + flat.imag.double() ** 2).sum())
    assert abs(norm2 - case.meta["norm2"]) < 1e-5


def raw_index(final_pos, perm, n=30):
    """final = raw.permute(perm): final dim d is raw dim perm[d] (dim 0 = most significant bit)."""
    out = np.zeros_like(final_pos)
    for d in range(n):
        bit = (final_pos >> (n - 1 - d)) & 1
        out |= bit << (n - 1 - perm[d])
    return out


def block_sums(raw, lead):
    """sum of raw over every dim not in `lead`, result indexed by the lead dims in order."""
    n = raw.dim()
    acc = raw.to(torch.complex128) if raw.numel() <= 2 ** 26 else None
    flat = raw.reshape(-1)
    # reduce in chunks of the flat index: amplitude index -> block id via bit gather
    out = torch.zeros(2 ** len(lead), dtype=torch.complex128, device=raw.device)
    chunk = 2 ** 26
    for s in range(0, flat.numel(), chunk):
        idx = torch.arange(s, min(s + chunk, flat.numel()), device=raw.device, dtype=torch.int64)
        blk = torch.zeros_like(idx)
        for r, d in enumerate(lead):
            blk |= ((idx >> (n - 1 - d)) & 1) << (len(lead) - 1 - r)
        vals = flat[s:s + chunk].to(torch.complex128)
        out.index_add_(0, blk, vals)
    del acc
    return out.cpu().numpy()


def test_more_contracted_bits_than_a_tile_holds():
    """k = 11 and 15 contracted bits: split-K through a temporary batch label (sparse closing steps)."""
    rng = np.random.default_rng(21)
    for k, n, ra in ((11, 3, 21), (15, 2, 22)):
        la = [chr(65 + x) for x in range(ra)]
        kl = list(rng.choice(la, size=k, replace=False))
        nl = [chr(97 + x) for x in range(n)]
        lb = kl + nl
        rng.shuffle(lb)
        lo = [x for x in la if x not in kl] + nl
        rng.shuffle(lo)
        eq = "".join(la) + "," + "".join(lb) + "->" + "".join(lo)
        a, b = crandn(rng, (2,) * ra), crandn(rng, (2,) * len(lb))
        # (against the complex128 value: a complex64 matmul over 2^15 terms is itself 1e-5 off)
        assert rel(hip_step(eq, a, b), oracle.einsum_pair(eq, a.astype(np.complex128), b.astype(np.complex128))) < 1e-5, eq


def test_n53_slice_with_the_chain_cut_and_with_pairs_from_the_left(monkeypatch):
    """The sparse executor cuts every chain of steps on the state tensor into single steps and fused pairs by a priced
    dynamic programme (contraction._plan_chain); ARTN_CHAIN_PLAN=0 restores the pairs-from-the-left schedule of rounds
    1-4.  Same amplitudes either way (reference loop: contraction.py:140-191, one einsum per step), more fused pairs."""
    from artensor_amd import contraction as C
    case = load_case(os.path.join(GOLDEN, "n53_m14_sliced.npz"))
    n_b = len(case.slicing_indices)
    leaves = case.fresh_tensors(device=DEV)

    class Count:
        def __init__(self):
            self.n = 0

        def record(self, info, e0, e1):
            self.n += 1 if info.get("k2_bits", 0) > 0 else 0   # fused pairs

    def one(mode):
        monkeypatch.setenv("ARTN_CHAIN_PLAN", mode)
        C._schedule_cache.clear()
        sliced = A.apply_slice(leaves, case.slicing_indices, A.slice_assignments(n_b, 0))
        counter = Count()
        monkeypatch.setattr(C, "profiler", counter)
        out = A.tensor_contraction_sparse(sliced, case.scheme).reshape(-1).cpu().numpy()
        monkeypatch.setattr(C, "profiler", None)
        return out, counter.n

    left, n_left = one("0")
    cut, n_cut = one("1")
    C._schedule_cache.clear()
    assert np.abs(cut - left).max() <= 1e-5 * np.abs(left).max()
    assert n_cut > n_left   # more of the big steps run as fused pairs
    if "slice0" in case.arrays:
        assert_contract(cut, case.arrays["slice0"], "n53_m14_sliced_slice0")


@pytest.mark.parametrize("name", ["n30_sparse100", "n12_sparse5", "n12_sparse_sliced"])
def test_sparse_schemes_with_the_chain_cut_and_with_pairs_from_the_left(name, monkeypatch):
    """Every committed sparse-state scheme under both schedules (contraction._plan_chain / fusion_schedule): the same
    amplitudes to 1e-5 of the largest, and the reference's (fixture) either way."""
    from artensor_amd import contraction as C
    case = load_case(os.path.join(GOLDEN, name + ".npz"))
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("ARTN_CHAIN_PLAN", mode)
        C._schedule_cache.clear()
        if case.slicing_indices:
            out = A.sliced_contraction(case.tensors, case.scheme, case.slicing_indices, case.arrays["final"].shape, sparse=True,
                                       device=DEV).cpu().numpy()
        else:
            out = A.tensor_contraction_sparse(case.fresh_tensors(device=DEV), case.scheme).cpu().numpy()
        outs[mode] = out.reshape(-1)
        assert amp_rel(outs[mode], case.arrays["final"].reshape(-1)) < 1e-5
    C._schedule_cache.clear()
    assert np.abs(outs["0"] - outs["1"]).max() <= 1e-5 * np.abs(outs["0"]).max()


def test_n53_slices(monkeypatch):
    """BASELINE config 4 (Sycamore n53 m14, derived from the bundled m20 circuit; one bitstring,
    14 sliced bonds): slice 0 against the reference executor's CPU result, and the slice loop
    on a handful of slices against the sum of the single-slice results."""
    case = load_case(os.path.join(GOLDEN, "n53_m14_sliced.npz"))
    n_b = len(case.slicing_indices)
    assert n_b == 14 and len(case.scheme) == 326
    leaves = case.fresh_tensors(device=DEV)

    def one(s):
        sliced = A.apply_slice(leaves, case.slicing_indices, A.slice_assignments(n_b, s))
        return A.tensor_contraction_sparse(sliced, case.scheme).reshape(-1).cpu().numpy()

    got0 = one(0)
    if "slice0" in case.arrays:
        assert_contract(got0, case.arrays["slice0"], "n53_m14_sliced_slice0")
    slices = [0, 5, 777, 16383]
    singles = sum(one(s) for s in slices)
    loop = A.sliced_contraction(leaves, case.scheme, case.slicing_indices, (1,), sparse=True, device=DEV,
                                slices=slices).cpu().numpy()
    assert np.abs(loop - singles).max() <= 1e-5 * np.abs(singles).max()
    # the loop above kept the small intermediates across slices; without that reuse, and with whole
    # slices replayed from a captured HIP graph (same launches, same bits as the plain loop)
    plain_loop = A.sliced_contraction(leaves, case.scheme, case.slicing_indices, (1,), sparse=True, device=DEV,
                                      slices=slices, reuse_small=False).cpu().numpy()
    assert np.abs(loop - plain_loop).max() <= 1e-5 * np.abs(plain_loop).max()
    replay = A.sliced_contraction(leaves, case.scheme, case.slicing_indices, (1,), sparse=True, device=DEV,
                                  slices=slices, reuse_small=False, graph=True).cpu().numpy()
    assert np.array_equal(replay, plain_loop)
    # Gray-ordered shard: consecutive slices differ in one bond, so only a few small steps rerun
    runner = A.SliceRunner(leaves, case.scheme, case.slicing_indices, (1,), sparse=True, device=DEV)
    order = A.rank_slices(2 ** n_b, 3, 8, gray=True)[:6]
    runner.run(order[:1])
    first = runner.small_steps_run
    runner.run(order[1:])
    assert first > 250 and runner.small_steps_run - first < 5 * 40
    want = sum(one(s) for s in order)
    assert np.abs(runner.collect.cpu().numpy() - want).max() <= 1e-5 * np.abs(want).max()
    # fused and unfused execution agree on a slice
    monkeypatch.setenv("ARTN_NO_FUSE", "1")
    A.contraction._pair_cache.clear()
    plain = one(5)
    A.contraction._pair_cache.clear()
    monkeypatch.delenv("ARTN_NO_FUSE")
    assert np.abs(plain - one(5)).max() <= 1e-5 * np.abs(plain).max()


@pytest.mark.parametrize("name", ["rand_D2_nv260_sliced", "rand_D4_nv100"])
def test_random_network_bench_fixtures(name):
    """Benchmark-scale random 3-regular networks (north_star: "random tensor networks of stated
    bond dimension"): slice 0 / the whole contraction against the reference executor's value."""
    case = load_case(os.path.join(GOLDEN, name + ".npz"))
    n_b = len(case.slicing_indices)
    leaves = case.fresh_tensors(device=DEV)
    want = case.arrays["slice0"].reshape(-1)
    runner = A.SliceRunner(leaves, case.scheme, case.slicing_indices, (1,), device=DEV)
    got = runner.run([0]).reshape(-1).cpu().numpy().copy()
    assert_contract(got, want, name + "_slice0")   # (rand D = 2: 7.0e-6 from the truth, the reference 4.8e-6 on the other side)
    if n_b:
        # a few more slices: reuse of small intermediates on/off agree
        order = A.rank_slices(2 ** n_b, 1, 8, gray=True)[:4]
        a = A.sliced_contraction(leaves, case.scheme, case.slicing_indices, (1,), device=DEV, slices=order)
        b = A.sliced_contraction(leaves, case.scheme, case.slicing_indices, (1,), device=DEV, slices=order,
                                 reuse_small=False)
        assert (a - b).abs().max().item() <= 1e-5 * b.abs().max().item()


KERNEL_XGEMM = 5


def _random_extent_step(rng, exts, n_m, n_n, n_k, n_h=0):
    M, Nn, K, H = ([f"{c}{i}" for i in range(n)] for c, n in (("m", n_m), ("n", n_n), ("k", n_k), ("h", n_h)))
    ext = {x: int(rng.choice(exts)) for x in M + Nn + K + H}
    la, lb, lo = M + K + H, Nn + K + H, M + Nn + H
    for lst in (la, lb, lo):
        rng.shuffle(lst)
    return (tuple(la), tuple(lb), tuple(lo)), tuple(ext[x] for x in la), tuple(ext[x] for x in lb)


def _einsum128_labels(eq, a, b):
    sym = {}
    for x in eq[0] + eq[1] + eq[2]:
        sym.setdefault(x, chr(65 + len(sym)) if len(sym) < 26 else chr(97 + len(sym) - 26))
    return np.einsum("".join(sym[x] for x in eq[0]) + "," + "".join(sym[x] for x in eq[1]) + "->" + "".join(sym[x] for x in eq[2]),
                     a.astype(np.complex128), b.astype(np.complex128))


def test_extent_gemm_with_a_narrower_launch_for_the_last_columns():
    """Round 6 (ArtnXGemmPlan::tail_nb): on launches of many rounds the columns behind the full column tiles run as a second
    launch of a narrower instantiation -- 243 columns = 2 tiles of 96 + 1 of 64, 216 = 2 x 96 + 1 x 32 -- both operand roles;
    ARTN_XG_TAIL is read when the library loads, so the one-launch plan is the other tests' business (small steps)."""
    rng = np.random.default_rng(606)
    for cols, eq_out in (((3, 3, 3, 3, 3), ("n4", "n3", "n2", "n1", "n0", "m1", "m0")),      # rows fastest in the result
                         ((6, 6, 6), ("m1", "m0", "n2", "n1", "n0"))):                          # columns fastest (roles swapped)
        nl = tuple(f"n{i}" for i in reversed(range(len(cols))))
        eq = (("m1", "k0", "m0"), nl[:1] + ("k0",) + nl[1:], eq_out)
        sa, sb = (1100, 9, 250), cols[:1] + (9,) + cols[1:]
        info = A.step_info(eq, sa, sb)
        n_cols = int(np.prod(cols))
        full = (n_cols // 32) // 3
        assert info["kernel"] == KERNEL_XGEMM and info["a_rereads"] == full + 1, info                       # 2 full column tiles + the tail
        assert info["n_tiles"] == -(-1100 * 250 // 128) * (full + 1), info
        a, b = crandn(rng, sa), crandn(rng, sb)
        got = A.contract(eq, gpu(a), gpu(b)).cpu().numpy()
        want = _einsum128_labels(eq, a, b)
        assert np.abs(got - want).max() <= 3e-6 * np.abs(want).max(), cols


def test_row_streaming_form_of_the_extent_gemm():
    """artn_k_xrow / artn_k_xrow64 (round 6): a handful of contracted values into a handful of columns on 2^15+ rows -- every count of MFMA
    steps (1..12 x four contracted values), one to three column blocks, rows that end inside a block, the contracted labels
    inside / outside the row labels of the operand -- against complex128 einsums; and ARTN_XROW=0 plans no such launch."""
    rng = np.random.default_rng(2026)
    cases = 0
    for kk, nn in ((1, 3), (3, 3), (5, 16), (9, 9), (12, 5), (16, 27), (17, 2), (21, 32), (24, 7), (27, 27), (29, 17), (32, 32),
                   (36, 36), (33, 9), (40, 48), (45, 20), (48, 33), (7, 41)):
        for form in range(2):
            rows = max(int(rng.integers(1 << 15, 3 << 15)), (1 << 21) // (kk * nn) + 1000)   # (8 x rows x kk x nn >= 2^24: a tiled launch)
            m0 = int(rng.integers(180, 250))   # (the third level of the row-offset tables takes up to 4 096 values)
            m1 = -(-rows // m0)
            if form == 0:   # contracted label slowest in the operand: rows contiguous
                eq, sa, sb = (("k", "m1", "m0"), ("n", "k"), ("n", "m1", "m0")), (kk, m1, m0), (nn, kk)
            else:           # contracted label fastest: neighbouring rows kk elements apart
                eq, sa, sb = (("m1", "m0", "k"), ("k", "n"), ("n", "m1", "m0")), (m1, m0, kk), (kk, nn)
            info = A.step_info(eq, sa, sb)
            assert info["kernel"] == KERNEL_XGEMM and info["m_tile_bits"] == (6 if kk <= 32 and nn <= 32 else 4), (kk, nn, form, info)   # 64-row superblocks (a lane per row) up to 32 x 32, 16-row blocks beyond
            a, b = crandn(rng, sa), crandn(rng, sb)
            got = A.contract(eq, gpu(a), gpu(b)).cpu().numpy()
            want = _einsum128_labels(eq, a, b)
            assert np.abs(got - want).max() <= 3e-6 * np.abs(want).max(), (kk, nn, form)
            cases += 1
    assert cases == 36
    # bond dimension 3 with the contracted labels between the row labels, all three levels of the row-offset tables in use
    eq = (tuple("abcdKefgLhijk"), ("x", "K", "L", "y"), ("y", "x") + tuple("abcdefghijk"))
    a, b = crandn(rng, (3,) * 13), crandn(rng, (3,) * 4)
    assert A.step_info(eq, a.shape, b.shape)["m_tile_bits"] == 6
    got = A.contract(eq, gpu(a), gpu(b)).cpu().numpy()
    want = _einsum128_labels(eq, a, b)
    assert np.abs(got - want).max() <= 3e-6 * np.abs(want).max()


def test_non_power_of_two_extents_on_the_matrix_cores():
    """artn_k_xgemm (round 5): steps whose labels have extents that are not powers of two -- bond dimension 3, 5, 6, 7:
    the reference's einsum (contraction.py:70) takes any bond_dims (tensor_network.py:4-30) -- against complex128 einsums:
    every copy mode, both operand roles, batch labels, a contraction long enough to flush partial sums, strided views."""
    seen = set()
    for seed in range(36):
        rng = np.random.default_rng(seed)
        exts = [3] if seed % 2 == 0 else [2, 3, 5, 6, 7]
        eq, sa, sb = _random_extent_step(rng, exts, int(rng.integers(4, 9)), int(rng.integers(1, 5)), int(rng.integers(1, 5)),
                                         int(rng.integers(0, 2)) if seed % 2 else 0)
        ext = dict(zip(eq[0], sa))
        ext.update(zip(eq[1], sb))
        if np.prod([float(v) for v in ext.values()]) > 2e10:
            continue   # (one seed draws a 2.4e8-element operand: four minutes of numpy for the truth; benchmark-size networks below)
        a, b = crandn(rng, sa), crandn(rng, sb)
        info = A.step_info(eq, sa, sb)
        got = A.contract(eq, gpu(a), gpu(b)).cpu().numpy()
        want = _einsum128_labels(eq, a, b)
        assert np.abs(got - want).max() <= 3e-6 * np.abs(want).max(), (seed, eq, info["kernel"])
        seen.add(info["kernel"])
    assert KERNEL_XGEMM in seen
    rng = np.random.default_rng(100)
    # 5 103 contracted values: partial sums through C
    eq = (("m0", "k0", "k1", "k2", "m1"), ("k2", "n0", "k0", "k1"), ("m0", "n0", "m1"))
    a, b = crandn(rng, (20, 3, 81, 21, 14)), crandn(rng, (21, 12, 3, 81))   # (numpy's complex128 einsum of a bigger one takes minutes)
    assert A.step_info(eq, a.shape, b.shape)["kernel"] == KERNEL_XGEMM
    got = A.contract(eq, gpu(a), gpu(b)).cpu().numpy()
    want = _einsum128_labels(eq, a, b)
    assert np.abs(got - want).max() <= 5e-6 * np.abs(want).max()
    # few results of long sums (the closing steps of a closed network end in a dot product): one workgroup per result
    a, b = crandn(rng, (6, 6, 6, 6)), crandn(rng, (6, 6, 6, 6))
    got = A.contract((("a", "b", "c", "d"), ("c", "a", "d", "b"), ()), gpu(a), gpu(b)).cpu().numpy()
    want = np.einsum("abcd,cadb->", a.astype(np.complex128), b.astype(np.complex128))
    assert abs(got - want) <= 3e-6 * np.sqrt(1296) * 2
    a, b = crandn(rng, (3, 7, 5, 11, 9)), crandn(rng, (9, 2, 11, 7, 5))
    got = A.contract((("m", "i", "j", "k", "l"), ("l", "n", "k", "i", "j"), ("n", "m")), gpu(a), gpu(b)).cpu().numpy()
    want = np.einsum("mijkl,lnkij->nm", a.astype(np.complex128), b.astype(np.complex128))
    assert np.abs(got - want).max() <= 3e-6 * np.abs(want).max()
    # strided views of bigger tensors
    big_a, big_b = gpu(crandn(rng, (27, 9, 25, 6, 30))), gpu(crandn(rng, (6, 14, 9, 30)))
    va, vb = big_a[1:26, :, ::2, :, :], big_b[:, 1:12, :, :]
    eq = (("m", "k", "p", "q", "r"), ("q", "n", "k", "r"), ("p", "n", "m"))
    got = A.contract(eq, va, vb).cpu().numpy()
    want = _einsum128_labels(eq, va.cpu().numpy(), vb.cpu().numpy())
    assert np.abs(got - want).max() <= 3e-6 * np.abs(want).max()


def test_non_power_of_two_extents_in_complex128_on_the_f64_matrix_cores():
    """artn_k_xgemm128 (round 5): the same steps with complex128 operands run on v_mfma_f64_16x16x4_f64 instead of the strided
    kernel -- against numpy's complex128 einsum to 1e-12: copy modes, batch labels, partial tiles, a contraction long enough to
    flush partial sums, strided views; and a whole bond-dimension-3 network in complex128 against the oracle."""
    seen = set()
    c128 = lambda rng, shape: rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
    for seed in range(24):
        rng = np.random.default_rng(700 + seed)
        exts = [3] if seed % 2 == 0 else [2, 3, 5, 6, 7]
        eq, sa, sb = _random_extent_step(rng, exts, int(rng.integers(4, 9)), int(rng.integers(1, 5)), int(rng.integers(1, 5)),
                                         int(rng.integers(0, 2)) if seed % 2 else 0)
        a, b = c128(rng, sa), c128(rng, sb)
        info = A.step_info(eq, sa, sb, dtype=torch.complex128)
        got = A.contract(eq, torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV)).cpu().numpy()
        want = _einsum128_labels(eq, a, b)
        assert got.dtype == np.complex128 and np.abs(got - want).max() <= 1e-12 * np.abs(want).max(), (seed, eq, info["kernel"])
        seen.add(info["kernel"])
    assert KERNEL_XGEMM in seen
    rng = np.random.default_rng(800)
    eq = (("m0", "k0", "k1", "k2", "m1"), ("k2", "n0", "k0", "k1"), ("m0", "n0", "m1"))
    a, b = c128(rng, (20, 3, 81, 21, 14)), c128(rng, (21, 12, 3, 81))   # (5 103 contracted values: partial sums through C)
    assert A.step_info(eq, a.shape, b.shape, dtype=torch.complex128)["kernel"] == KERNEL_XGEMM
    got = A.contract(eq, torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV)).cpu().numpy()
    want = _einsum128_labels(eq, a, b)
    assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max()
    big_a, big_b = torch.from_numpy(c128(rng, (27, 9, 25, 6, 30))).to(DEV), torch.from_numpy(c128(rng, (6, 14, 9, 30))).to(DEV)
    va, vb = big_a[1:26, :, ::2, :, :], big_b[:, 1:12, :, :]
    eq = (("m", "k", "p", "q", "r"), ("q", "n", "k", "r"), ("p", "n", "m"))
    got = A.contract(eq, va, vb).cpu().numpy()
    want = _einsum128_labels(eq, va.cpu().numpy(), vb.cpu().numpy())
    assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max()
    # a whole network of bond dimension 3 (small-step program in complex128 + artn_k_xgemm128 for whatever is big enough)
    case = load_case(os.path.join(GOLDEN, "rand_D3_open.npz"))
    out = A.tensor_contraction(case.fresh_tensors(dtype=torch.complex128, device=DEV), case.scheme).cpu().numpy()
    want = oracle.tensor_contraction({k: v.numpy().astype(np.complex128) for k, v in case.tensors.items()}, case.scheme)
    assert np.abs(out - want).max() <= 1e-12 * np.abs(want).max()


@pytest.mark.parametrize("name", ["rand_D3_nv112", "rand_D6_nv64"])
def test_random_networks_whose_bond_dimension_is_not_a_power_of_two(name):
    """Benchmark-scale random 3-regular networks of bond dimension 3 and 6 = 2 x 3 (tests/golden/make_golden.py
    random_bench_nonpow2: planned and contracted by the reference; largest intermediate 3^18 / 6^11 elements): the whole
    contraction against the reference executor's complex64 value and its own complex128 run, every big step on the
    matrix cores (the strided kernel keeps only launch-bound leftovers)."""
    from artensor_amd import contraction as C
    case = load_case(os.path.join(GOLDEN, name + ".npz"))
    leaves = case.fresh_tensors(device=DEV)
    want = case.arrays["slice0"].reshape(-1)

    class Prof:
        def __init__(self):
            self.rows = []

        def record(self, info, e0, e1):
            self.rows.append((info, e0, e1))

    got = A.tensor_contraction(dict(leaves), case.scheme).reshape(-1).cpu().numpy()
    assert_contract(got, want, name + "_slice0")
    prof = Prof()
    C.profiler = prof
    try:
        again = A.tensor_contraction(dict(leaves), case.scheme).reshape(-1).cpu().numpy()
    finally:
        C.profiler = None
    assert np.array_equal(again, got)
    torch.cuda.synchronize()
    ms = {}
    for info, e0, e1 in prof.rows:
        ms[info["kernel"]] = ms.get(info["kernel"], 0.0) + e0.elapsed_time(e1)
    total = sum(ms.values())
    assert ms.get(KERNEL_XGEMM, 0.0) >= 0.85 * total, ms
    assert ms.get(0, 0.0) <= 0.10 * total, ms   # artn_k_generic: sum-outs and launch-bound leftovers only
    big = [info for info, _, _ in prof.rows if info["kernel"] == 0 and info["flops"] >= 8 * 2 ** 22]
    assert not big, big
    # the same scheme with the reference's own label order for every intermediate (no private layouts): same value
    os.environ["ARTN_OWN_LAYOUTS"] = "0"
    try:
        C._plan_cache.clear()
        plain = A.tensor_contraction(dict(leaves), list(case.scheme)).reshape(-1).cpu().numpy()
    finally:
        del os.environ["ARTN_OWN_LAYOUTS"]
        C._plan_cache.clear()
    assert amp_rel(plain, got) <= 1e-5


def test_extent_step_on_a_tensor_of_more_than_2_to_the_31_elements():
    """A bond-dimension-3 step whose operand AND result have 3^20 = 2^31.7 elements (27.9 GB of complex64 each: one
    simulated-annealing level above tests/golden/rand_D3_nv112.npz, whose largest tensor has 3^18): until round 6 the extent
    GEMM declined tensors of 2^31+ elements and such a step fell back to the strided kernel with a RuntimeWarning (VERDICT
    r05 missing #4; the reference's einsum at contraction.py:70 has no such limit).  Element offsets in artn_k_xgemm are
    unsigned 32-bit: tensors up to 2^32 elements run on the matrix cores.  Checked on 4 096 random result elements
    (including the last one) against a complex128 evaluation of the same sums from gathered operand elements."""
    free = torch.cuda.mem_get_info()[0]
    if free < 70 * 2 ** 30:
        pytest.skip("needs 70 GB of free device memory")
    import warnings
    n_lab, D = 20, 3
    la = tuple(range(n_lab))
    kpos = (3, 11)                                   # contracted labels of A
    lb = (3, 11, 20, 21)
    lo = tuple(20 if x == 3 else (21 if x == 11 else x) for x in la)
    info = A.step_info((la, lb, lo), (D,) * n_lab, (D,) * 4)
    assert info["kernel"] == KERNEL_XGEMM, info
    gen = torch.Generator(device=DEV).manual_seed(20)
    a = torch.view_as_complex(torch.randn((D,) * n_lab + (2,), device=DEV, generator=gen))
    b = torch.view_as_complex(torch.randn((D,) * 4 + (2,), device=DEV, generator=gen))
    assert a.numel() == 3 ** 20 > 2 ** 31
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)       # the strided fallback would warn
        c = A.contract((la, lb, lo), a, b)
    assert tuple(c.shape) == (D,) * n_lab
    n_s = 4096
    idx = torch.randint(0, D, (n_s, n_lab), device=DEV, generator=gen)
    idx[0] = D - 1                                          # the very last element of the result
    w = torch.tensor([D ** (n_lab - 1 - q) for q in range(n_lab)], device=DEV, dtype=torch.int64)
    got = c.reshape(-1)[(idx * w).sum(1)].to(torch.complex128)
    want = torch.zeros(n_s, dtype=torch.complex128, device=DEV)
    b128 = b.to(torch.complex128)
    for k0 in range(D):
        for k1 in range(D):
            ia = idx.clone()
            ia[:, kpos[0]], ia[:, kpos[1]] = k0, k1
            av = a.reshape(-1)[(ia * w).sum(1)].to(torch.complex128)
            want += av * b128[k0, k1][idx[:, kpos[0]], idx[:, kpos[1]]]
    err = (got - want).abs().max().item() / want.abs().max().item()
    assert err <= 2e-6, err
    del a, c
    torch.cuda.empty_cache()


@pytest.mark.parametrize("name,n_amp", [("rand_D3_open6_nv96", 3 ** 6), ("rand_D6_open4_nv60", 6 ** 4)])
def test_open_random_networks_whose_bond_dimension_is_not_a_power_of_two(name, n_amp):
    """The OPEN twins of the benchmark-scale fixtures above (round 6; tests/golden/make_golden.py
    random_bench_nonpow2_open): 3^6 = 729 and 6^4 = 1 296 output amplitudes instead of one closing dot product, so a
    layout slip in any extent step shows -- every amplitude through assert_contract against the reference executor's
    complex64 result and its complex128 run, in the scheme's own output order; then the same in complex128 (1e-11)."""
    case = load_case(os.path.join(GOLDEN, name + ".npz"))
    want = case.arrays["final"]
    assert want.size == n_amp and want.ndim == case.meta["n_open"]
    got = A.tensor_contraction(case.fresh_tensors(device=DEV), case.scheme)
    assert tuple(got.shape) == tuple(want.shape)
    assert_contract(got.cpu().numpy(), want, name + "_final")
    got128 = A.tensor_contraction(case.fresh_tensors(dtype=torch.complex128, device=DEV), case.scheme).cpu().numpy()
    t = case.arrays["exact128"]
    assert np.abs(got128 - t).max() <= 1e-11 * np.abs(t).max()
    # the reference's label orders for the intermediates (no private layouts): the same amplitudes in the same places
    from artensor_amd import contraction as C
    os.environ["ARTN_OWN_LAYOUTS"] = "0"
    try:
        C._plan_cache.clear()
        plain = A.tensor_contraction(case.fresh_tensors(device=DEV), list(case.scheme)).cpu().numpy()
    finally:
        del os.environ["ARTN_OWN_LAYOUTS"]
        C._plan_cache.clear()
    assert amp_rel(plain.reshape(-1), got.cpu().numpy().reshape(-1)) <= 1e-5


@pytest.mark.parametrize("n_slabs", [2, 4, 8])
def test_n30_slabs_of_the_replanned_reduced_network(n_slabs):
    """The unsliced n30 m14 contraction over N ranks without a collective (round 5): log2 N output qubits fixed at the leaves
    and the REDUCED network re-planned by the reference's order finder (tests/golden/make_golden.py::case_n30_dense_parts;
    0.85 / 0.88 / 0.97 x the unsliced plan's FLOP in total at N = 2 / 4 / 8, against 1.36 / 2.13 / 3.78 x for the one tree
    of round 4).  Slabs 0 and N - 1 against the matching amplitudes of the FULL result as the reference computed it
    (n30_dense.npz: Google's 10 000 bitstrings)."""
    full = load_case(os.path.join(GOLDEN, "n30_dense.npz"))
    part = load_case(os.path.join(GOLDEN, f"n30_dense_part{n_slabs}.npz"))
    k = int(np.log2(n_slabs))
    assert len(part.meta["fixed"]) == k and len(part.meta["out_qubits"]) == 30 - k
    assert part.meta["executed_flop_over_unsliced"] <= 1.0
    fpos = np.array([int(b, 2) for b in full.meta["google_bitstrings"]], dtype=np.int64)
    want_all = full.arrays["amps_at_google"]
    rms = float(np.sqrt(np.mean(np.abs(want_all) ** 2)))
    leaves = part.fresh_tensors(device=DEV)
    seen = 0
    for r in (0, n_slabs - 1):
        raw = A.slab_contraction(leaves, part.scheme, part.meta["fixed"], r, device=DEV)
        assert raw.numel() == 2 ** (30 - k)
        sel = np.ones(len(fpos), dtype=bool)
        for j, (_leaf, _dim, q) in enumerate(part.meta["fixed"]):
            sel &= ((fpos >> (29 - q)) & 1) == ((r >> j) & 1)
        oq = part.meta["out_qubits"]
        local = np.zeros(int(sel.sum()), dtype=np.int64)
        for x, q in enumerate(oq):
            local |= ((fpos[sel] >> (29 - q)) & 1) << (len(oq) - 1 - x)
        got = raw.reshape(-1)[torch.from_numpy(local).to(DEV)].cpu().numpy()
        assert sel.sum() > 10000 // n_slabs // 2
        assert amp_rel(got, want_all[sel], rms) <= 1e-5, (n_slabs, r, amp_rel(got, want_all[sel], rms))
        seen += int(sel.sum())
        del raw
    assert seen > 0


def test_state_vec_n12_and_n30():
    """circuit.py:155-175 on the device.  n12: the reference's own state vector.  n30: the state
    vector (1 270 gate applications on 2^30 amplitudes) against the tensor-network amplitudes the
    reference computed at Google's 10 000 bitstrings -- two unrelated contraction orders."""
    g12 = load_case(os.path.join(GOLDEN, "n12_gates.npz"))
    gates = [(g12.tensors[k], g12.meta["inds"][k]) for k in range(len(g12.meta["inds"]))]
    sv = A.state_vec(gates, g12.meta["n_qubits"], device=DEV).reshape(-1).cpu().numpy()
    want = load_case(os.path.join(GOLDEN, "n12_dense.npz")).arrays["state_vec"]
    assert amp_rel(sv, want) < 1e-5
    g30 = load_case(os.path.join(GOLDEN, "n30_gates.npz"))
    gates = [(g30.tensors[k], g30.meta["inds"][k]) for k in range(len(g30.meta["inds"]))]
    sv = A.state_vec(gates, 30, device=DEV)
    case = load_case(os.path.join(GOLDEN, "n30_dense.npz"))
    bits = case.meta["google_bitstrings"]
    flat = (sv._base if sv._base is not None else sv).reshape(-1)   # the permute is a view
    st = sv.stride()                                                 # (torch cannot index 30 dims at once)
    pos = torch.tensor([sum(int(c) * st[q] for q, c in enumerate(b)) for b in bits], device=DEV)
    at = flat[pos].cpu().numpy()
    want = case.arrays["amps_at_google"]
    rms = 2.0 ** -15
    # (1 270 gate applications, another contraction order: against the complex128 truth of the tensor-network amplitudes)
    assert amp_rel(at, gpu_truth("n30_dense_at_google"), rms) <= 1e-5
    assert amp_rel(at, want, rms) <= 1e-5 + amp_rel(want, gpu_truth("n30_dense_at_google"), rms)
    # the state is normalised
    norm2 = sum(float((torch.view_as_real(flat[k::4]) ** 2).sum().item()) for k in range(4))
    assert abs(norm2 - 1.0) < 1e-3


def test_n53_m20_slice0():
    """The bundled Sycamore n53 m20 circuit (BASELINE configs[4] runs it at reduced precision; here
    complex64): slice 0 of 2^29 against the reference executor's CPU value, plus one more slice
    with and without the reuse of small intermediates."""
    case = load_case(os.path.join(GOLDEN, "n53_m20_sliced.npz"))
    n_b = len(case.slicing_indices)
    assert n_b == 29 and len(case.scheme) == 454
    leaves = case.fresh_tensors(device=DEV)
    runner = A.SliceRunner(leaves, case.scheme, case.slicing_indices, (1,), sparse=True, device=DEV)
    got = runner.run([0]).reshape(-1).cpu().numpy().copy()
    assert_contract(got, case.arrays["slice0"], "n53_m20_sliced_slice0")
    a = A.sliced_contraction(leaves, case.scheme, case.slicing_indices, (1,), sparse=True, device=DEV, slices=[1, 3])
    b = A.sliced_contraction(leaves, case.scheme, case.slicing_indices, (1,), sparse=True, device=DEV, slices=[1, 3],
                             reuse_small=False)
    assert (a - b).abs().max().item() <= 1e-5 * b.abs().max().item()


def test_two_process_sliced_contraction(tmp_path):
    """The multi-GPU product path end to end: two fresh processes (world_size 2, gloo, both on cuda:0 --
    RCCL needs one GPU per rank, the code path does not) run artensor_amd.sliced_contraction with the
    real HIP executors and the real collective; their results are compared with the fixture and with
    the single-process sum.  (tests/test_distributed.py covers the same sharding logic on CPU boxes.)"""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
    s.close()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_gpu_worker.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", port, str(tmp_path)], env=env) for r in range(2)]
    codes = [p.wait(timeout=600) for p in procs]
    assert codes == [0, 0], codes
    case = load_case(os.path.join(GOLDEN, "n12_sparse_sliced.npz"))
    want = case.arrays["final"]
    for r in range(2):   # all_reduce: both ranks hold the full sum
        assert amp_rel(np.load(tmp_path / f"n12_{r}.npy"), want) < 1e-5
    case = load_case(os.path.join(GOLDEN, "n53_m14_sliced.npz"))
    single = A.sliced_contraction(case.fresh_tensors(device=DEV), case.scheme, case.slicing_indices, (1,), sparse=True,
                                  device=DEV, slices=list(range(8))).cpu().numpy()
    r0, r1 = np.load(tmp_path / "n53_0.npy"), np.load(tmp_path / "n53_1.npy")
    assert np.abs(r0 - single).max() <= 1e-5 * np.abs(single).max()      # reduce to root: rank 0 holds the sum
    assert np.abs(r1 - single).max() > 1e-3 * np.abs(single).max()       # rank 1 keeps its partial sum


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank: this box has fewer than 2")
def test_two_gpu_rccl_sliced_contraction(tmp_path):
    """The same two-rank run over backend "nccl" (= RCCL over xGMI), one GPU per rank: n12 sliced with its all_reduce and
    8 slices of n53 m14 reduced to rank 0 (reference simulation.py:107-114, the loop being sharded)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
    s.close()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_gpu_worker.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", port, str(tmp_path), "nccl"], env=env) for r in range(2)]
    codes = [p.wait(timeout=600) for p in procs]
    assert codes == [0, 0], codes
    case = load_case(os.path.join(GOLDEN, "n12_sparse_sliced.npz"))
    for r in range(2):
        assert amp_rel(np.load(tmp_path / f"n12_{r}.npy"), case.arrays["final"]) < 1e-5
    case = load_case(os.path.join(GOLDEN, "n53_m14_sliced.npz"))
    single = A.sliced_contraction(case.fresh_tensors(device=DEV), case.scheme, case.slicing_indices, (1,), sparse=True,
                                  device=DEV, slices=list(range(8))).cpu().numpy()
    assert np.abs(np.load(tmp_path / "n53_0.npy") - single).max() <= 1e-5 * np.abs(single).max()


def _bench_last_line(args, env_extra, timeout=900):
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")]
    assert p.returncode == 0 and lines, (p.returncode, p.stdout[-2000:], p.stderr[-2000:])
    return lines, json.loads(lines[-1])


def test_bench_gpus_2_as_a_plain_command():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment starts its own two ranks as a child process and relays
    rank 0's compact line (here: gloo, both ranks on cuda:0 through the self-test knobs -- a one-GPU box cannot run RCCL;
    on a multi-GPU node the same command runs backend nccl, test_bench_gpus_2_rccl).  The line must fit the driver's tail."""
    lines, line = _bench_last_line(["--gpus", "2", "--steps", "1", "--warmup", "1", "--slices", "2"],
                                   {"ARTN_BENCH_BACKEND": "gloo", "ARTN_BENCH_DEVICE": "0"})
    assert len(lines[-1]) < 2000
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["check"] == "ok"
    assert line["sliced"]["ranks_in_collective"] == 2 and line["sliced"]["backend"] == "gloo" and line["sliced"]["check"] == "ok"
    assert line["sliced"]["slices"] == 2 * 3 * 2
    assert line["sliced"]["rand_D2"][3] == "ok" and line["sliced"]["rand_D2"][0] > 10   # the random-network series, same sharding


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank: this box has fewer than 2")
def test_bench_gpus_2_rccl():
    lines, line = _bench_last_line(["--gpus", "2", "--steps", "2", "--warmup", "1"], {})
    assert len(lines[-1]) < 2000 and line["n_gpus"] == 2 and line["config"]["check"] == "ok"
    assert line["sliced"]["ranks_in_collective"] == 2 and line["sliced"]["backend"] == "nccl" and line["sliced"]["check"] == "ok"


def test_bench_default_line_is_compact():
    """N = 1: every leg on an earlier line, the LAST line under 2 000 characters with value, roofline and (here skipped)
    cpu_baseline; roofline fractions never above 1."""
    lines, line = _bench_last_line(["--steps", "2", "--warmup", "1", "--no-cpu-baseline"], {})
    assert len(lines[-1]) < 2000 and line["n_gpus"] == 1 and line["config"]["check"] == "ok"
    assert 0.3 < line["roofline"]["frac"] <= 1.0 and line["roofline"]["bound"] == "mfma"
    legs = [json.loads(ln) for ln in lines[:-1] if ln.startswith('{"leg"')]
    assert {leg["leg"] for leg in legs} == set(line["workloads"]) and len(legs) >= 9
    for leg in legs:
        assert leg["check"]["check"] == "ok", leg["leg"]
        assert leg["roofline"] is not None and 0.0 < leg["roofline"]["frac"] <= 1.0, (leg["leg"], leg["roofline"])
    assert line["sliced"]["ranks_in_collective"] == 1


def test_output_partitioned_contraction():
    """Build-side extension for the unsliced dense contraction at N > 1 (bench.py --gpus N): output labels
    fixed at the leaves give disjoint slabs of the result, no exchange.  All 8 slabs of n12 against the
    reference's full result; n30 planned on the host (scheme consistency, work per slab)."""
    case = load_case(os.path.join(GOLDEN, "n12_dense.npz"))
    raw = case.arrays["raw"]
    for part in range(8):
        slab, fixed, vals = A.partitioned_contraction(case.tensors, case.scheme, 3, part, device=DEV)
        idx = [slice(None)] * raw.ndim
        for d, v in zip(fixed, vals):
            idx[d] = v
        assert np.abs(slab.cpu().numpy() - raw[tuple(idx)]).max() <= 1e-5 * np.abs(raw).max(), part
    case = load_case(os.path.join(GOLDEN, "n30_dense.npz"))
    shapes = {k: tuple(t.shape) for k, t in case.tensors.items()}
    new_scheme, selects, fixed = A.partition_output(case.scheme, shapes, 3)
    assert len(new_scheme) == len(case.scheme) and len(set(fixed)) == 3
    assert len(new_scheme[-1][1].split("->")[1]) == 27


def _random_gemm_step(rng, m, n, k, batch=0):
    ml = [f"m{x}" for x in range(m)]
    kl = [f"k{x}" for x in range(k)]
    nl = [f"n{x}" for x in range(n)]
    la, lb, lo = ml + kl, kl + nl, ml + nl
    rng.shuffle(la), rng.shuffle(lb), rng.shuffle(lo)
    sa, sb = [2] * len(la), [2] * len(lb)
    if batch:
        la, lb, lo = ["z"] + la, ["z"] + lb, ["z"] + lo
        sa, sb = [batch] + sa, [batch] + sb
    return (tuple(la), tuple(lb), tuple(lo)), tuple(sa), tuple(sb)


def _einsum128(eq, a, b):
    import string
    la, lb, lo = eq
    labels = list(dict.fromkeys(list(la) + list(lb)))
    mp = {x: string.ascii_letters[i] for i, x in enumerate(labels)}
    eq = "".join(mp[x] for x in la) + "," + "".join(mp[x] for x in lb) + "->" + "".join(mp[x] for x in lo)
    # (torch's CPU einsum in complex128: the same sum as numpy's, threaded -- numpy takes minutes at these sizes)
    return torch.einsum(eq, torch.from_numpy(np.ascontiguousarray(a)).to(torch.complex128),
                        torch.from_numpy(np.ascontiguousarray(b)).to(torch.complex128)).numpy()


@pytest.mark.parametrize("m,n,k,batch", [(11, 11, 9, 0), (12, 11, 7, 0), (11, 12, 8, 0), (13, 6, 8, 0), (9, 9, 9, 5),
                                         (10, 10, 11, 0), (5, 5, 14, 0), (8, 8, 7, 3)])
def test_gemm_kernel_steps(m, n, k, batch):
    """Steps the planner gives to the two-operand GEMM kernel (7+ contracted bits with 6+ free bits on
    both sides, or more than 8 contracted bits): full 128 x 128 tiles with a two-pass epilogue, many
    looped contracted bits, a ragged batch axis; fp32 against a complex128 einsum, and under
    precision("bf16") against the same einsum of the bf16-rounded operands."""
    rng = np.random.default_rng(1000 * m + 10 * n + k)
    eq, sa, sb = _random_gemm_step(rng, m, n, k, batch)
    a, b = crandn(rng, sa), crandn(rng, sb)
    info = A.step_info(eq, sa, sb)
    # (the two-operand LDS GEMM, or -- 2^8+ contracted values, 8+ / 7+ free bits, no batch label -- its packed-operand form)
    assert info["kernel"] in (N.KERNEL_GEMM_MFMA, N.KERNEL_PGEMM), info
    got = A.contract(eq, gpu(a), gpu(b)).cpu().numpy()
    assert rel(got, _einsum128(eq, a, b)) < STEP_TOL, (eq, info)
    with A.precision("bf16"):
        got16 = A.contract(eq, gpu(a), gpu(b)).cpu().numpy()
    want16 = _einsum128(eq, _bf16_round(a), _bf16_round(b))
    assert rel(got16, want16) < 5e-6, (eq, info)
    assert rel(got16, got) > 1e-4   # really the reduced-precision arithmetic


def test_complex128_on_the_matrix_cores():
    """dtype=torch.complex128 (reference simulation.py:90 takes any dtype): the 28 big steps of the n30 scheme as
    surrogates (state operand truncated to 2^20) on v_mfma_f64_16x16x4_f64, parity 1e-12 against a complex128
    einsum; random GEMM-shaped steps with many contracted bits and a ragged batch axis; the whole n12 scheme."""
    case = load_case(os.path.join(GOLDEN, "n30_dense.npz"))
    steps = dense_scheme_shapes(case)
    big = [(n, s) for n, s in enumerate(steps) if np.prod(s[1]) >= 2 ** 20]
    assert len(big) == 28
    on_mfma = 0
    for n, (eq, sa, sb) in big:
        eq2, sa2, sb2 = shrink_step(eq, sa, sb, max_log2=20)
        rng = np.random.default_rng(n)
        a = rng.standard_normal(sa2) + 1j * rng.standard_normal(sa2)
        b = rng.standard_normal(sb2) + 1j * rng.standard_normal(sb2)
        info = A.step_info(eq2, sa2, sb2, dtype=torch.complex128)
        on_mfma += info["kernel"] == N.KERNEL_GEMM_MFMA
        got = A.contract(eq2, gpu(a), gpu(b)).cpu().numpy()
        want = torch.einsum(eq2, torch.from_numpy(a), torch.from_numpy(b)).numpy()
        assert got.dtype == np.complex128 and rel(got, want) < 1e-12, (n, eq2)
    assert on_mfma >= 26
    rng = np.random.default_rng(5)
    for (m, n_, k, batch) in [(11, 10, 9, 0), (12, 3, 5, 0), (9, 9, 12, 0), (8, 7, 4, 5), (4, 12, 6, 0)]:
        eq, sa, sb = _random_gemm_step(rng, m, n_, k, batch)
        a = rng.standard_normal(sa) + 1j * rng.standard_normal(sa)
        b = rng.standard_normal(sb) + 1j * rng.standard_normal(sb)
        assert A.step_info(eq, sa, sb, dtype=torch.complex128)["kernel"] == N.KERNEL_GEMM_MFMA
        assert rel(A.contract(eq, gpu(a), gpu(b)).cpu().numpy(), _einsum128(eq, a, b)) < 1e-12, eq
    case = load_case(os.path.join(GOLDEN, "n12_dense.npz"))
    raw = A.tensor_contraction(case.fresh_tensors(dtype=torch.complex128, device=DEV), case.scheme).cpu().numpy()
    meta, arrays = c128_spread()
    assert raw.dtype == np.complex128 and amp_rel(raw.reshape(-1), arrays["n12_dense_c128"]) < 1e-12


def test_complex128_row_gather_and_scientific_notation():
    """The sparse executor in complex128 (reference contraction.py:132-205 takes any dtype): (d) the chunk loop's row
    gather fused into the f64 GEMM kernel (artn_k_gemm128<NB, GATHER>, :149-156) against gather + einsum in numpy, and
    the n30 x 100 scheme (four chunk steps) against the REFERENCE's own complex128 run (c128_spread.npz); (c)
    scientific_notation renormalises in complex128 (:197-200) -- n12 against the oracle in complex128, and the
    factor / amplitudes recombine to the reference's complex128 amplitudes."""
    from artensor_amd.contraction import contract_gathered
    rng = np.random.default_rng(29)
    for (na, nb, n, free, kb, nn) in [(7, 5, 6, 12, 3, 2), (3, 9, 619, 10, 8, 3), (9, 11, 100, 8, 8, 6), (6, 3, 37, 9, 7, 5),
                                      (4, 4, 6, 4, 7, 7)]:
        la = ["z"] + [chr(65 + x) for x in range(free + kb)]
        kl = la[1:1 + kb]
        nl = [chr(97 + x) for x in range(nn)]
        lb = ["z"] + kl[::-1] + nl
        lo = ["z"] + [x for x in la[1:] if x not in kl] + nl
        eq = "".join(la) + "," + "".join(lb) + "->" + "".join(lo)
        a = rng.standard_normal((na,) + (2,) * (len(la) - 1)) + 1j * rng.standard_normal((na,) + (2,) * (len(la) - 1))
        b = rng.standard_normal((nb,) + (2,) * (len(lb) - 1)) + 1j * rng.standard_normal((nb,) + (2,) * (len(lb) - 1))
        ra, rb = torch.from_numpy(rng.integers(0, na, size=n)), torch.from_numpy(rng.integers(0, nb, size=n))
        got = contract_gathered(eq, gpu(a), ra, gpu(b), rb)
        assert got is not None and got.dtype == torch.complex128, eq
        want = torch.einsum(eq, torch.from_numpy(a[ra.numpy()]), torch.from_numpy(b[rb.numpy()])).numpy()
        assert rel(got.cpu().numpy(), want) < 1e-12, eq
        got = contract_gathered(eq, gpu(a[ra.numpy()]), None, gpu(b), rb)
        assert got is not None and rel(got.cpu().numpy(), want) < 1e-12
    A.contraction.check_gather_flag()
    meta, arrays = c128_spread()
    case = load_case(os.path.join(GOLDEN, "n30_sparse100.npz"))
    from artensor_amd import contraction as C
    calls = []
    orig = C.contract_gathered
    C.contract_gathered = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        out = A.tensor_contraction_sparse(case.fresh_tensors(dtype=torch.complex128, device=DEV), case.scheme)
    finally:
        C.contract_gathered = orig
    assert calls and out.dtype == torch.complex128
    assert amp_rel(out.cpu().numpy().reshape(-1), arrays["n30_sparse100_c128"]) < 1e-11
    # scientific notation in complex128
    case = load_case(os.path.join(GOLDEN, "n12_sparse5_scinot.npz"))
    factor, out = A.tensor_contraction_sparse(case.fresh_tensors(dtype=torch.complex128, device=DEV), case.scheme,
                                              scientific_notation=True)
    assert out.dtype == torch.complex128 and factor.dtype in (torch.float64, torch.complex128)   # (log10 of a float64 maximum)
    factor = factor.real if factor.is_complex() else factor
    leaves = {k: t.numpy().astype(np.complex128) for k, t in case.tensors.items()}
    of, oo = oracle.tensor_contraction_sparse(leaves, case.scheme, scientific_notation=True)
    assert abs(factor.cpu().item() - float(np.real(of))) < 1e-10
    assert amp_rel(out.cpu().numpy(), oo) < 1e-12
    assert abs(float(out.abs().max().item()) - 1.0) < 1e-12            # renormalised after the last step
    assert amp_rel(out.cpu().numpy().reshape(-1) * 10.0 ** factor.cpu().item(), arrays["n12_sparse5_c128"]) < 1e-11


def test_scheme_list_mutated_in_place_is_recompiled():
    """The reference re-reads the scheme list on every call (contraction.py:66); the compiled-plan caches are keyed on
    id(scheme) and must notice steps appended / removed / replaced in place."""
    case = load_case(os.path.join(GOLDEN, "n12_dense.npz"))
    leaves64 = {k: t.numpy() for k, t in case.tensors.items()}
    scheme = list(case.scheme)
    full = A.tensor_contraction(case.fresh_tensors(device=DEV), scheme).cpu().numpy()
    assert rel(full, oracle.tensor_contraction(dict(leaves64), list(case.scheme))) < 1e-5
    last = scheme.pop()                                    # same list object, one step fewer
    part = A.tensor_contraction(case.fresh_tensors(device=DEV), scheme).cpu().numpy()
    want = oracle.tensor_contraction(dict(leaves64), list(case.scheme[:-1]))
    assert part.shape == want.shape and rel(part, want) < 1e-5
    scheme.append(last)                                    # and back
    again = A.tensor_contraction(case.fresh_tensors(device=DEV), scheme).cpu().numpy()
    assert again.shape == full.shape and np.array_equal(again, full)
    # the sparse executor's schedule / program caches
    case = load_case(os.path.join(GOLDEN, "n12_sparse5.npz"))
    scheme = list(case.scheme)
    full = A.tensor_contraction_sparse(case.fresh_tensors(device=DEV), scheme).cpu().numpy()
    assert amp_rel(full, case.arrays["final"]) < 1e-5
    last = scheme.pop()
    part = A.tensor_contraction_sparse(case.fresh_tensors(device=DEV), scheme).cpu().numpy()
    want = oracle.tensor_contraction_sparse({k: t.numpy() for k, t in case.tensors.items()}, list(case.scheme[:-1]))
    assert part.shape == np.asarray(want).shape and rel(part, want) < 1e-5
    scheme.append(last)
    assert np.array_equal(A.tensor_contraction_sparse(case.fresh_tensors(device=DEV), scheme).cpu().numpy(), full)


TRUTH_CASES = {
    # key in c128_truth_gpu.npz -> (fixture, sparse, sliced): the cases torch's GPU einsum can run (tensors of up to 25
    # dims: its copy kernels refuse or crawl on the rank-30 operands of the other fixtures -- those are pinned by the torch-CPU
    # run of tests/golden/make_c128_truth_cpu.py instead, tests/test_oracle.py::test_gpu_truth_equals_the_independent_cpu_truth)
    "n30_sparse10000_final": ("n30_sparse10000.npz", True, False),
    "rand_D4_nv100_slice0": ("rand_D4_nv100.npz", False, True),
}


@pytest.mark.parametrize("key", sorted(TRUTH_CASES))
def test_c128_truth_against_torch_einsum_on_the_gpu(key):
    """The committed complex128 truth (tests/golden/c128_truth_gpu.npz, computed by THIS package's f64-MFMA path) against
    an INDEPENDENT complex128 computation of the same leaves and scheme, live: the reference's executor loop run by torch
    in complex128 on the GPU (oracle.tensor_contraction[_sparse]_torch: torch.einsum -> permute + bmm on the vendor BLAS;
    no planner, descriptor or kernel of this package).  Both must agree to 1e-11 of the rms amplitude."""
    fixture, sparse, sliced = TRUTH_CASES[key]
    case = load_case(os.path.join(GOLDEN, fixture))
    leaves = case.fresh_tensors(dtype=torch.complex128, device=DEV)
    if sliced and case.slicing_indices:
        leaves = A.apply_slice(leaves, case.slicing_indices, A.slice_assignments(len(case.slicing_indices), 0))
    fn = oracle.tensor_contraction_sparse_torch if sparse else oracle.tensor_contraction_torch
    out = fn(leaves, case.scheme)["result"]
    assert out.dtype == torch.complex128
    got = out.reshape(-1).cpu().numpy()
    del out, leaves
    torch.cuda.empty_cache()
    t = gpu_truth(key)
    rms = float(np.sqrt(np.mean(np.abs(t) ** 2)))
    assert np.abs(got - t).max() <= 1e-11 * rms, (key, float(np.abs(got - t).max() / rms))


@pytest.mark.skipif(not N.has("artn_contract3"), reason="three-step fusion is compiled into development builds only (make dev)")
def test_fused_triples_on_the_gpu(monkeypatch):
    """artn_contract3 (artn_k_bits3: three steps of reference contraction.py:66-70 on one tensor in ONE pass; region 0 -> 1
    -> 0 -> 1) against three oracle steps: the two triples of the n30 scheme that fit a 2^12 tile with 128-byte runs, on
    surrogates of 2^26 elements (the planner takes triples only for launches of 2^14+ tiles), four-product and 3M
    instantiations, against the same three steps run one by one (the lane-by-lane replay against the ORACLE is
    tests/test_plan_emulation.py::test_fused_triples_emulated / test_n30_triples_planned_and_emulated)."""
    from artensor_amd.contraction import contract3, triple_info
    from helpers import shrink_triple
    case = load_case(os.path.join(GOLDEN, "n30_dense.npz"))
    steps = dense_scheme_shapes(case)
    rng = np.random.default_rng(31)
    for (n1, n2, n3) in [(125, 128, 131), (149, 155, 159)]:
        (eq1, sa, sb1), (eq2, _, sb2), (eq3, _, sb3) = steps[n1], steps[n2], steps[n3]
        assert triple_info(eq1, sa, sb1, eq2, sb2, eq3, sb3) is not None
        e1, sa_, sb1_, e2, sb2_, e3, sb3_ = shrink_triple(eq1, sa, sb1, eq2, sb2, eq3, sb3, max_log2=26)
        info = triple_info(e1, sa_, sb1_, e2, sb2_, e3, sb3_)
        assert info is not None and info["n_tiles"] >= 2 ** 14, (n1, info)
        a = crandn(rng, sa_)
        b1, b2, b3 = crandn(rng, sb1_), crandn(rng, sb2_), crandn(rng, sb3_)
        got = contract3(e1, gpu(a), gpu(b1), e2, gpu(b2), e3, gpu(b3))
        assert got is not None, (n1, n2, n3)
        # the same three steps one by one on the GPU (single-step kernels: green against the oracle elsewhere)
        ref = A.contract(e3, A.contract(e2, A.contract(e1, gpu(a), gpu(b1)), gpu(b2)), gpu(b3))
        assert float((got - ref).abs().max().item()) <= 2e-6 * float(ref.abs().max().item()), (n1, n2, n3)
        del got, ref
        torch.cuda.empty_cache()


def test_accumulate_in_the_store_phase_in_complex128(monkeypatch):
    """The same for complex128 (artn_k_bits128<KB1, KB2, true>, round 5): the 13 fusable pairs of the n30 scheme on surrogates of
    2^22 elements against accumulator + the plain launch (one f64 add per element either way: equal to the last bit), through
    tensor_contraction(accumulate_into=...), and a sliced dense loop in complex128 with and without the fused add."""
    from artensor_amd.contraction import fusion_schedule, contract2, _pair_descriptors
    from helpers import shrink_pair
    import ctypes
    lib = N.lib()
    case = load_case(os.path.join(GOLDEN, "n30_dense.npz"))
    steps = dense_scheme_shapes(case)
    pairs = [e for e in fusion_schedule(case.scheme) if e[0] == "pair" and np.prod(steps[e[1]][1]) >= 2 ** 22]
    stream = N.current_stream_ptr(torch.device("cuda:0"))
    c128 = lambda rng, shape: torch.from_numpy(rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).to(DEV)
    fused = 0
    for _, n, m in pairs:
        eq1, sa, sb1 = steps[n]
        eq2, _, sb2 = steps[m]
        e1, a_s, b1_s, e2, b2_s = shrink_pair(eq1, sa, sb1, eq2, sb2, max_log2=22)
        rng = np.random.default_rng(300 + n)
        a, b1, b2 = c128(rng, a_s), c128(rng, b1_s), c128(rng, b2_s)
        plain = contract2(e1, a, b1, e2, b2)
        if plain is None:
            continue
        acc0 = c128(rng, tuple(plain.shape))
        acc = acc0.clone()
        d1, d2, _ = _pair_descriptors(e1, a, b1, e2, b2)
        rc = lib.artn_contract2_acc(ctypes.byref(d1), ctypes.byref(d2), a.data_ptr(), b1.data_ptr(), b2.data_ptr(), acc.data_ptr(), stream)
        if rc == -2:
            continue
        assert rc == 0, lib.artn_last_error()
        assert torch.equal(acc, acc0 + plain), (n, m)
        fused += 1
        tensors = {0: a.clone(), 1: b1, 2: b2}
        acc2 = acc0.clone()
        got = A.tensor_contraction(tensors, [((0, 1), e1), ((0, 2), e2)], accumulate_into=acc2)
        assert got is acc2 and torch.equal(acc2, acc0 + plain), (n, m)
    assert fused >= 10, fused   # (every tile shape of artn_k_bits128 can add)
    # the slice loop in complex128: `collect += slice` inside each slice's last launch against the separate artn_axpy_c128
    case = load_case(os.path.join(GOLDEN, "n30_dense_sliced3.npz"))
    shape = case.arrays["final"].shape if "final" in case.arrays else (2,) * 30
    fused_sum = A.sliced_contraction(case.tensors, case.scheme, case.slicing_indices, shape, dtype=torch.complex128, device=DEV,
                                     slices=[0, 1, 2])
    monkeypatch.setenv("ARTN_NO_ACC", "1")
    plain_sum = A.sliced_contraction(case.tensors, case.scheme, case.slicing_indices, shape, dtype=torch.complex128, device=DEV,
                                     slices=[0, 1, 2])
    monkeypatch.delenv("ARTN_NO_ACC")
    assert fused_sum.dtype == torch.complex128
    assert float((fused_sum - plain_sum).abs().max()) <= 1e-13 * float(plain_sum.abs().max())


def test_accumulate_in_the_store_phase():
    """`collect_tensor += tensor_contraction(...)` (reference simulation.py:114) with the add in the store phase of the last
    launch (artn_contract2_acc / artn_contract_acc): the 13 fusable pairs of the n30 scheme and its big single steps on
    surrogates of 2^22 elements, against accumulator + the same launch without the add (one fp32 add per element either
    way: equal to the last bit), and through tensor_contraction(accumulate_into=...) with a scheme whose last launch can and
    one whose last launch cannot add."""
    from artensor_amd.contraction import fusion_schedule, contract2, _pair_descriptors, _descriptor
    from helpers import shrink_pair
    import ctypes
    lib = N.lib()
    case = load_case(os.path.join(GOLDEN, "n30_dense.npz"))
    steps = dense_scheme_shapes(case)
    pairs = [e for e in fusion_schedule(case.scheme) if e[0] == "pair" and np.prod(steps[e[1]][1]) >= 2 ** 22]
    fused = 0
    stream = N.current_stream_ptr(torch.device("cuda:0"))
    for _, n, m in pairs:
        eq1, sa, sb1 = steps[n]
        eq2, _, sb2 = steps[m]
        e1, a_s, b1_s, e2, b2_s = shrink_pair(eq1, sa, sb1, eq2, sb2, max_log2=22)
        rng = np.random.default_rng(100 + n)
        a, b1, b2 = gpu(crandn(rng, a_s)), gpu(crandn(rng, b1_s)), gpu(crandn(rng, b2_s))
        plain = contract2(e1, a, b1, e2, b2)
        if plain is None:
            continue
        acc0 = gpu(crandn(rng, tuple(plain.shape)))
        acc = acc0.clone()
        d1, d2, _ = _pair_descriptors(e1, a, b1, e2, b2)
        rc = lib.artn_contract2_acc(ctypes.byref(d1), ctypes.byref(d2), a.data_ptr(), b1.data_ptr(), b2.data_ptr(), acc.data_ptr(), stream)
        if rc == -2:
            continue
        assert rc == 0, lib.artn_last_error()
        assert torch.equal(acc, acc0 + plain), (n, m)
        fused += 1
        # ... and through the executor: a two-step scheme whose last launch is this pair
        tensors = {0: a.clone(), 1: b1, 2: b2}
        scheme = [((0, 1), e1), ((0, 2), e2)]
        acc2 = acc0.clone()
        got = A.tensor_contraction(tensors, scheme, accumulate_into=acc2)
        assert got is acc2 and torch.equal(acc2, acc0 + plain), (n, m)
    assert fused >= 6, fused   # (the 3M pairs: launches this small have no non-temporal loads, the other way into a FULL instantiation)
    # single big steps
    singles = 0
    for n in (75, 93, 108, 139, 172):
        eq, sa, sb = steps[n]
        e, a_s, b_s = shrink_step(eq, sa, sb, max_log2=22)
        rng = np.random.default_rng(200 + n)
        a, b = gpu(crandn(rng, a_s)), gpu(crandn(rng, b_s))
        plain = A.contract(e, a, b)
        acc0 = gpu(crandn(rng, tuple(plain.shape)))
        acc = acc0.clone()
        la, rest = e.split(",")
        lb, lo = rest.split("->")
        d, _ = _descriptor(tuple(la), tuple(lb), tuple(lo), tuple(a.shape), tuple(a.stride()), tuple(b.shape), tuple(b.stride()), torch.complex64)
        rc = lib.artn_contract_acc(ctypes.byref(d), a.data_ptr(), b.data_ptr(), acc.data_ptr(), stream)
        if rc == -2:
            continue
        assert rc == 0, lib.artn_last_error()
        assert torch.equal(acc, acc0 + plain), n
        singles += 1
    assert singles >= 2, singles
    # random pairs of 2^24-element states (2^12 tiles: enough for the non-temporal instantiations too): wherever the entry point
    # accepts, the result equals the separate add; it must never accept a launch whose instantiation has no add (that traps)
    rng = np.random.default_rng(77)
    accepted = 0
    for trial in range(24):
        k1, k2 = int(rng.integers(2, 7)), int(rng.integers(2, 7))
        ra = 24
        la_ = [chr(65 + x) for x in range(ra)]
        kl1 = list(rng.choice(la_[:16], size=k1, replace=False))
        nl1 = [chr(97 + x) for x in range(k1)]
        lb1 = kl1 + nl1
        rng.shuffle(lb1)
        lo1 = [x for x in la_ if x not in kl1]
        for x in nl1:
            lo1.insert(int(rng.integers(0, len(lo1) + 1)), x)
        kl2 = list(rng.choice(lo1[:16], size=k2, replace=False))
        nl2 = [chr(110 + x) for x in range(k2)]
        lb2 = kl2 + nl2
        rng.shuffle(lb2)
        lo2 = [x for x in lo1 if x not in kl2]
        for x in nl2:
            lo2.insert(int(rng.integers(0, len(lo2) + 1)), x)
        e1 = "".join(la_) + "," + "".join(lb1) + "->" + "".join(lo1)
        e2 = "".join(lo1) + "," + "".join(lb2) + "->" + "".join(lo2)
        a, b1, b2 = gpu(crandn(rng, (2,) * ra)), gpu(crandn(rng, (2,) * (2 * k1))), gpu(crandn(rng, (2,) * (2 * k2)))
        plain = contract2(e1, a, b1, e2, b2)
        if plain is None:
            continue
        acc0 = gpu(crandn(rng, tuple(plain.shape)))
        acc = acc0.clone()
        d1, d2, _ = _pair_descriptors(e1, a, b1, e2, b2)
        rc = lib.artn_contract2_acc(ctypes.byref(d1), ctypes.byref(d2), a.data_ptr(), b1.data_ptr(), b2.data_ptr(), acc.data_ptr(), stream)
        if rc == -2:
            continue
        assert rc == 0, lib.artn_last_error()
        torch.cuda.synchronize()
        assert torch.equal(acc, acc0 + plain), (e1, e2)
        accepted += 1
    assert accepted >= 6, accepted
    # a scheme whose last launch cannot add (a tiny step: strided kernel) falls back to the separate add
    a, b = gpu(crandn(np.random.default_rng(5), (2, 2, 2))), gpu(crandn(np.random.default_rng(6), (2, 2)))
    acc0 = gpu(crandn(np.random.default_rng(7), (2, 2, 2)))
    acc = acc0.clone()
    got = A.tensor_contraction({0: a.clone(), 1: b}, [((0, 1), "abc,cd->abd")], accumulate_into=acc)
    assert got is acc and float((acc - (acc0 + torch.einsum("abc,cd->abd", a, b))).abs().max()) < 1e-5
    with pytest.raises(RuntimeError, match="accumulate_into"):
        A.tensor_contraction({0: a.clone(), 1: b}, [((0, 1), "abc,cd->abd")], accumulate_into=acc[:1])


@pytest.mark.parametrize("k,nt,ra,seed", [(5, 2, 22, 0), (5, 4, 22, 1), (6, 3, 22, 2), (6, 1, 21, 3), (5, 0, 21, 4), (6, 4, 23, 5), (5, 3, 20, 6)])
def test_shrinking_single_steps_on_narrow_three_product_blocks(k, nt, ra, seed):
    """ArtnBitsPlan::narrow3 on the GPU (artn_k_bits<KB1, 0, ..., N3>: the 16 x 16 x 4 three-product stage of artn_k_wide on four
    waves, for single steps with 5-6 contracted bits that keep at most 4 result bits in the tile -- n53's 2^30 -> 2^27 step):
    scattered bit positions, tiles of 2^12 and 2^13 elements, against the oracle."""
    rng = np.random.default_rng(950 + seed)
    la = [chr(65 + x) for x in range(ra)]
    kl = list(rng.choice(la, size=k, replace=False))
    nl = [chr(97 + x) for x in range(nt)]
    lb = kl + nl
    rng.shuffle(lb)
    lo = [x for x in la if x not in kl]
    for x in nl:
        lo.insert(int(rng.integers(0, len(lo) + 1)), x)
    eq = "".join(la) + "," + "".join(lb) + "->" + "".join(lo)
    info = A.step_info(eq, (2,) * ra, (2,) * len(lb))
    a, b = crandn(rng, (2,) * ra), crandn(rng, (2,) * len(lb))
    got = A.contract(eq, gpu(a), gpu(b)).cpu().numpy()
    want = oracle.einsum_pair(eq, a, b)
    assert rel(got, want) < STEP_TOL, (eq, info)


def test_wide_kernel_pairs():
    """artn_k_wide (ARTN_WIDE=1: one 8-wave workgroup per CU on one tile, LDS-DMA ring, every stage 3M on 16 x 16 x 4 blocks --
    the default only for pairs with 11+ contracted bits) on EVERY pair shape against the oracle, in a process of its own because the planner reads its
    tuning once: tests/wide_worker.py."""
    env = dict(os.environ, ARTN_WIDE="1", ARTN_WIDE_MIN_TILES="1")
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "wide_worker.py")], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "wide ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_complex128_fused_pairs():
    """complex128 pairs in ONE pass (artn_k_bits128: 16-byte elements, f64 MFMA stages, the intermediate in LDS): the 13
    fusable pairs of the n30 scheme truncated to 2^21 elements and random pairs, 1e-12 against complex128 einsums; single
    steps forced onto the same kernel (ARTN_FORCE_BITS) as well."""
    from artensor_amd.contraction import fusion_schedule, contract2
    from helpers import shrink_pair
    case = load_case(os.path.join(GOLDEN, "n30_dense.npz"))
    steps = dense_scheme_shapes(case)
    pairs = [e for e in fusion_schedule(case.scheme) if e[0] == "pair" and np.prod(steps[e[1]][1]) >= 2 ** 22]
    c128 = lambda rng, shape: rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
    e128 = lambda eq, x, y: torch.einsum(eq, torch.from_numpy(np.ascontiguousarray(x)), torch.from_numpy(np.ascontiguousarray(y))).numpy()
    fused = 0
    for _, n, m in pairs:
        eq1, sa, sb1 = steps[n]
        eq2, _, sb2 = steps[m]
        e1, a_s, b1_s, e2, b2_s = shrink_pair(eq1, sa, sb1, eq2, sb2, max_log2=21)
        rng = np.random.default_rng(n)
        a, b1, b2 = c128(rng, a_s), c128(rng, b1_s), c128(rng, b2_s)
        got = contract2(e1, gpu(a), gpu(b1), e2, gpu(b2))
        if got is None:
            continue
        fused += 1
        want = e128(e2, e128(e1, a, b1), b2)
        assert got.dtype == torch.complex128 and rel(got.cpu().numpy(), want) < 1e-12, (n, m)
    assert fused >= 10
    rng = np.random.default_rng(9)
    done = 0
    for trial in range(30):
        ra = int(rng.integers(13, 19))
        k1, n1, k2, n2 = (int(x) for x in rng.integers(1, 6, size=4))
        la = [chr(65 + x) for x in range(ra)]
        kl1 = list(rng.choice(la, size=k1, replace=False))
        nl1 = [chr(97 + x) for x in range(n1)]
        lb1 = kl1 + nl1
        rng.shuffle(lb1)
        lo1 = [x for x in la if x not in kl1] + nl1
        rng.shuffle(lo1)
        if k2 > len(lo1) - 6:
            continue
        kl2 = list(rng.choice(lo1, size=k2, replace=False))
        nl2 = [chr(110 + x) for x in range(n2)]
        lb2 = kl2 + nl2
        rng.shuffle(lb2)
        lo2 = [x for x in lo1 if x not in kl2] + nl2
        rng.shuffle(lo2)
        eq1 = "".join(la) + "," + "".join(lb1) + "->" + "".join(lo1)
        eq2 = "".join(lo1) + "," + "".join(lb2) + "->" + "".join(lo2)
        a, b1, b2 = c128(rng, (2,) * ra), c128(rng, (2,) * len(lb1)), c128(rng, (2,) * len(lb2))
        got = contract2(eq1, gpu(a), gpu(b1), eq2, gpu(b2))
        if got is None:
            continue
        want = e128(eq2, e128(eq1, a, b1), b2)
        assert rel(got.cpu().numpy(), want) < 1e-12, (eq1, eq2)
        done += 1
    assert done >= 12


def test_complex128_n30_dense_full_size_against_committed_truth():
    """The whole n30 contraction in complex128 (32 GiB of state buffers; fused pairs on artn_k_bits128, the rest on
    artn_k_gemm128) reproduces the committed complex128 truth (tests/golden/c128_truth_gpu.npz, computed in round 3 with
    un-fused GEMM passes only) to 1e-11: two different kernel families, the same amplitudes."""
    case = load_case(os.path.join(GOLDEN, "n30_dense.npz"))
    raw = A.tensor_contraction(case.fresh_tensors(dtype=torch.complex128, device=DEV), case.scheme)
    assert raw.dtype == torch.complex128 and raw.numel() == 2 ** 30
    perm = case.meta["permute_dims"]
    flat = raw.reshape(-1)
    fpos = np.array([int(b, 2) for b in case.meta["google_bitstrings"]], dtype=np.int64)
    at = flat[torch.from_numpy(raw_index(fpos, perm)).to(DEV)].cpu().numpy()
    want = gpu_truth("n30_dense_at_google")
    assert np.abs(at - want).max() <= 1e-11 * np.abs(want).max()
    blocks = block_sums(raw, perm[:10])
    tb = gpu_truth("n30_dense_block_sums")
    assert np.abs(blocks - tb).max() <= 1e-11 * np.abs(tb).max()
    del raw, flat
    torch.cuda.empty_cache()


@pytest.mark.parametrize("k1,k2", [(5, 2), (5, 3), (5, 4), (6, 2), (6, 3), (6, 4), (2, 5), (3, 5), (4, 5), (2, 6), (3, 6), (4, 6)])
def test_fused_pairs_3m_with_a_narrow_stage(k1, k2, monkeypatch):
    """3M pairs whose other stage contracts 2-4 bits (that stage on v_mfma_f32_16x16x4_f32 blocks, three products): full
    tiles (2^22-element state: the FULL / artn_k_alt instantiations) and smaller ones, size-preserving, growing and
    narrow-column stages, against a complex128 einsum on the host."""
    from artensor_amd.contraction import contract2, pair_info
    monkeypatch.setenv("ARTN_FORCE_BITS", "1")
    rng = np.random.default_rng(100 * k1 + k2)
    gen = torch.Generator(device=DEV).manual_seed(k1 * 10 + k2)
    done = 0
    for trial in range(5):
        ra = 22 if trial == 0 else int(rng.integers(15, 19))
        n1 = k1 + (1 if (trial == 2 and k1 <= 4) else 0) - (1 if (trial == 3 and k1 <= 4) else 0)
        n2 = k2 + (1 if (trial == 2 and k2 <= 4) else 0) - (1 if (trial == 3 and k2 <= 4) else 0)
        la = [chr(65 + x) for x in range(ra)]
        kl1 = list(rng.choice(la, size=k1, replace=False))
        nl1 = [chr(97 + x) for x in range(n1)]
        lb1 = kl1 + nl1
        rng.shuffle(lb1)
        lo1 = [x for x in la if x not in kl1] + nl1
        rng.shuffle(lo1)
        kl2 = list(rng.choice(lo1, size=k2, replace=False))
        nl2 = [chr(110 + x) for x in range(n2)]
        lb2 = kl2 + nl2
        rng.shuffle(lb2)
        lo2 = [x for x in lo1 if x not in kl2] + nl2
        rng.shuffle(lo2)
        eq1 = "".join(la) + "," + "".join(lb1) + "->" + "".join(lo1)
        eq2 = "".join(lo1) + "," + "".join(lb2) + "->" + "".join(lo2)
        mk = lambda n: torch.view_as_complex(torch.randn((2,) * n + (2,), device=DEV, generator=gen))
        a, b1, b2 = mk(ra), mk(len(lb1)), mk(len(lb2))
        got = contract2(eq1, a, b1, eq2, b2)
        if got is None:
            continue
        info = pair_info(eq1, (2,) * ra, (2,) * len(lb1), eq2, (2,) * len(lb2))
        want = torch.einsum(eq2, torch.einsum(eq1, a.cpu().to(torch.complex128), b1.cpu().to(torch.complex128)), b2.cpu().to(torch.complex128))
        err = float((got.cpu().to(torch.complex128) - want).abs().max() / want.abs().max())
        assert err < STEP_TOL, (eq1, eq2, err)
        done += info is not None and info["arith"] == 1
    assert done >= 2, done


def test_randomised_single_steps_pairs_and_gathers():
    """tools/stress_random.py: 120 random single steps, fused pairs and gathered steps (random label orders, 1-8 contracted
    bits, 0-7 new bits, ragged batch labels; complex64 and complex128) through contract / contract2 / contract_gathered
    against a complex128 einsum on the host: 2e-5 / 1e-11 (600 more cases were run when the round-3 planner rules went in)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("stress_random", os.path.join(root, "tools", "stress_random.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")   # (small random steps on the strided kernel warn by design)
        assert mod.main(120, seed=11) == 0


def test_randomised_steps_with_extents_that_are_not_powers_of_two():
    """tools/stress_extents.py: 120 random single steps with label extents 2..9 in random orders -- general shapes, row-streaming
    shapes (artn_k_xrow), many-tile shapes (artn_k_xgemm with its second launch for the last columns), the strided fallback for
    what is too small -- against torch.einsum in complex128 on the device: 3e-6 of the largest result (x sqrt(K / 256) for long
    sums).  1 400 more cases (seeds 0-3) ran when the round-6 planner rules went in: worst 1.56e-6 (gpurun_out/s_r7l, s_r7n)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("stress_extents", os.path.join(root, "tools", "stress_extents.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")   # (small random steps on the strided kernel warn by design)
        assert mod.main(120, seed=12) == 0


def test_randomised_row_streaming_steps_in_the_16_row_shape():
    """ARTN_XROW64=0 (read when the library loads: a process of its own): the steps the planner gives to artn_k_xrow64 run on the
    16-row instantiations of artn_k_xrow instead -- 90 random cases of tools/stress_extents.py, a third of them row-streaming shapes."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ARTN_XROW64="0")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_extents.py"), "90", "7"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "'xrow')" in r.stdout and "xrow64" not in r.stdout, r.stdout[-600:]


def test_gemm_kernel_strided_operands_and_split_k():
    """Operands that are views (the slice loop hands in selected leaves), and a closing step whose result
    is too small to fill the chip: contracted labels become a batch label (split-K) and are summed."""
    rng = np.random.default_rng(77)
    eq, sa, sb = _random_gemm_step(rng, 10, 9, 9)
    big_a, big_b = crandn(rng, (2,) + sa + (2,)), crandn(rng, sb + (2,))
    a, b = gpu(big_a)[1, ..., 0], gpu(big_b)[..., 1]
    want = _einsum128(eq, big_a[1, ..., 0], big_b[..., 1])
    assert rel(A.contract(eq, a, b).cpu().numpy(), want) < STEP_TOL
    eq, sa, sb = _random_gemm_step(rng, 3, 3, 18)
    a, b = crandn(rng, sa), crandn(rng, sb)
    assert rel(A.contract(eq, gpu(a), gpu(b)).cpu().numpy(), _einsum128(eq, a, b)) < STEP_TOL


def test_n53_m20_big_batch_slice0():
    """BASELINE configs[4]: the bundled n53 m20 circuit, big-batch sampling -- 1 024 correlated bitstrings
    (16 open qubits), sparse-state scheme compiled by the reference with chunked (A), gathered (B) and
    row-select (C) steps at n53 scale, 40 sliced bonds.  Slice 0 in complex64 against the reference's CPU
    executor (tests/golden/n53_m20_batch.npz, 321 s there), then under precision("bf16") -- the
    bf16-complex MFMA path -- by state fidelity against the complex64 result (parity unpinned: the
    reference has no reduced-precision path)."""
    case = load_case(os.path.join(GOLDEN, "n53_m20_batch.npz"))
    assert case.meta["branches"]["A"] >= 1 and case.meta["branches"]["B"] >= 1 and case.meta["branches"]["C_select"] >= 1
    rows = len(case.meta["bitstrings_sorted"])
    assert rows == 1024 and len(case.slicing_indices) == 40
    leaves = case.fresh_tensors(device=DEV)
    runner = A.SliceRunner(leaves, case.scheme, case.slicing_indices, (rows,), sparse=True, device=DEV)
    got = runner.run([0]).reshape(-1).cpu().numpy().copy()
    want = case.arrays["slice0"].reshape(-1)
    assert amp_rel(got, want) <= 1e-5
    # loose <= 1e-5 and strict within 2 x the reference's own, both against the complex128 truth of this slice
    # (measured: HIP 6.1e-6 / 6.3e-5, the reference's complex64 run 3.8e-6 / 4.2e-5)
    assert_contract(got, want, "n53_m20_batch_slice0")
    with A.precision("bf16"):
        r16 = A.SliceRunner(leaves, case.scheme, case.slicing_indices, (rows,), sparse=True, device=DEV)
        got16 = r16.run([0]).reshape(-1).cpu().numpy().copy()
    x, y = got.astype(np.complex128), got16.astype(np.complex128)
    fidelity = abs(np.vdot(x, y)) ** 2 / (np.vdot(x, x).real * np.vdot(y, y).real)
    assert fidelity > 0.99, fidelity
    assert amp_rel(got16, got) > 1e-4   # and it is not the fp32 path
    # "Parity unpinned" still gets regression pins (measured r03/r04: fidelity 0.99976, rms error 1.5-1.6e-2 of the rms
    # amplitude, the error unbiased): a build that loses a bfloat16 bit, rounds operands twice or drops a chunk of the
    # 2^15-value sums fails these long before it fails the 0.99 of BASELINE configs[4]'s definition
    rms_amp = float(np.sqrt(np.mean(np.abs(x) ** 2)))
    rms_err = float(np.sqrt(np.mean(np.abs(y - x) ** 2))) / rms_amp
    print(f"bf16 big-batch slice 0: fidelity {fidelity:.6f}, rms error {rms_err:.3e} of the rms amplitude, "
          f"mean error {abs(np.mean(y - x)) / rms_amp:.2e}")
    assert fidelity >= 0.9996, fidelity
    assert rms_err <= 2.2e-2, rms_err
    assert abs(np.mean(y - x)) / rms_amp <= 4 * rms_err / np.sqrt(len(x)), (np.mean(y - x), rms_err)   # no systematic offset


def test_n53_m20_big_batch_of_65536_bitstrings_slice0():
    """BASELINE configs[4] at a batch that deserves the name (round 5; tests/golden/n53_m20_bigbatch.npz): 2^16 correlated
    bitstrings -- half of the 2^17 product over 17 open qubits -- on the bundled n53 m20 circuit, planned by the reference
    (41 sliced bonds), compiled by the vectorised sparse compiler (same tuples as the reference's wherever that one
    finishes: tests/golden/check_boundary.py), slice 0 from the REFERENCE's executor (200 s on 8 cores).  complex64 against
    that value; then the bf16-complex MFMA path by state fidelity."""
    case = load_case(os.path.join(GOLDEN, "n53_m20_bigbatch.npz"))
    br = case.meta["branches"]
    assert br["A"] >= 1 and br["B"] >= 1 and br["C_select"] >= 1
    rows = len(case.meta["bitstrings_sorted"])
    assert rows == 65536 and len(case.slicing_indices) == 41
    # the scheme stored in the fixture IS what this package's compiler returns for that batch: index lists of 2^16 rows
    assert max(int(x.numel()) for st in case.scheme if len(st) > 2 for side in st[2] for x in side) >= 32768
    leaves = case.fresh_tensors(device=DEV)
    runner = A.SliceRunner(leaves, case.scheme, case.slicing_indices, (rows,), sparse=True, device=DEV)
    got = runner.run([0]).reshape(-1).cpu().numpy().copy()
    want = case.arrays["slice0"].reshape(-1)
    # round 6: the complex128 truth of this slice (this package's f64 path on the GPU, pinned to an independent torch-CPU
    # complex128 run of the reference's executor loop at 5.7e-15: tests/test_oracle.py) -- the full contract, like every
    # other big fixture: loose <= 1e-5 against the truth, strict within 2 x the reference's own, and no farther from the
    # reference's complex64 value than 1e-5 + the reference's own distance to the truth (5.1e-6)
    assert_contract(got, want, "n53_m20_bigbatch_slice0")
    with A.precision("bf16"):
        r16 = A.SliceRunner(leaves, case.scheme, case.slicing_indices, (rows,), sparse=True, device=DEV)
        got16 = r16.run([0]).reshape(-1).cpu().numpy().copy()
    x, y = got.astype(np.complex128), got16.astype(np.complex128)
    fidelity = abs(np.vdot(x, y)) ** 2 / (np.vdot(x, x).real * np.vdot(y, y).real)
    rms_err = float(np.sqrt(np.mean(np.abs(y - x) ** 2))) / float(np.sqrt(np.mean(np.abs(x) ** 2)))
    print(f"bf16 big-batch (65 536 bitstrings) slice 0: fidelity {fidelity:.6f}, rms error {rms_err:.3e} of the rms amplitude")
    assert fidelity >= 0.999, fidelity
    assert amp_rel(got16, got) > 1e-4   # and it is not the fp32 path


def _bf16_round(x):
    """Round to nearest even onto bfloat16, returned as float32 (complex parts separately)."""
    def r(f):
        u = np.ascontiguousarray(f, dtype=np.float32).view(np.uint32).astype(np.uint64)
        u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
        return u.astype(np.uint32).view(np.float32)
    x = np.asarray(x)
    return (r(x.real) + 1j * r(x.imag)).astype(np.complex64).reshape(x.shape)


def test_bf16_precision_mode(monkeypatch):
    """precision("bf16"): complex64 in memory, MFMA operands rounded to bfloat16, fp32 accumulation.
    The reference has no such mode (parity unpinned): single steps and fused pairs are checked
    against a numpy restatement of exactly that arithmetic, and the n30 amplitudes against the
    complex64 reference by state fidelity."""
    from artensor_amd.contraction import contract2
    monkeypatch.setenv("ARTN_FORCE_BITS", "1")
    rng = np.random.default_rng(21)
    for eq, sa, sb in [("abcdefghijklmnop,pcfx->abdeghijklmnox", (2,) * 16, (2, 2, 2, 2)),
                       ("abcdefghijklmnopq,qhcfaxyz->bdegijklmnopzyx", (2,) * 17, (2,) * 8),
                       ("zabcdefghijklmn,znkcfxy->zabdeghijlmyx", (3,) + (2,) * 14, (3, 2, 2, 2, 2, 2, 2)),
                       # 8 and 7 contracted bits: the one-workgroup-per-CU kernel, and its split over the waves
                       ("abcdefghijklmnopqrstu,ucfhkmoqxyzw->abdegijlnprstwzyx", (2,) * 21, (2,) * 12),
                       ("abcdefghijklmnopqrstu,ucfhkmoxyzwv->abdegijlnpqrstvwzyx", (2,) * 21, (2,) * 12),
                       ("abcdefghijklmnopqrst,tcfhkmoqxy->abdegijlnprsyx", (2,) * 20, (2,) * 10)]:
        a, b = crandn(rng, sa), crandn(rng, sb)
        with A.precision("bf16"):
            got = A.contract(eq, gpu(a), gpu(b)).cpu().numpy()
        want = oracle.einsum_pair(eq, _bf16_round(a).astype(np.complex128), _bf16_round(b).astype(np.complex128))
        assert rel(got, want) < 2e-6, eq
        exact = oracle.einsum_pair(eq, a, b)
        assert 1e-4 < rel(got, exact) < 3e-2   # really reduced precision, and sane
    eq1, eq2 = "abcdefghijklmnopq,qhcfxyzw->abdegijklmnopwzyx", "abdegijklmnopwzyx,xbkdouv->aegijlmnpwzyuv"
    a, b1, b2 = crandn(rng, (2,) * 17), crandn(rng, (2,) * 8), crandn(rng, (2,) * 7)
    with A.precision("bf16"):
        got = contract2(eq1, gpu(a), gpu(b1), eq2, gpu(b2))
    assert got is not None
    mid = oracle.einsum_pair(eq1, _bf16_round(a).astype(np.complex128), _bf16_round(b1).astype(np.complex128))
    want = oracle.einsum_pair(eq2, _bf16_round(mid.astype(np.complex64)).astype(np.complex128),
                              _bf16_round(b2).astype(np.complex128))
    assert rel(got.cpu().numpy(), want) < 5e-5   # (the rounding of the fp32 intermediate can flip on ties)
    monkeypatch.delenv("ARTN_FORCE_BITS")
    # whole circuit: fidelity of the bf16 state against the reference amplitudes at Google's bitstrings
    case = load_case(os.path.join(GOLDEN, "n30_dense.npz"))
    with A.precision("bf16"):
        out = A.tensor_contraction(case.fresh_tensors(device=DEV), case.scheme)
    perm = case.meta["permute_dims"]
    fpos = np.array([int(b, 2) for b in case.meta["google_bitstrings"]], dtype=np.int64)
    rpos = np.zeros_like(fpos)
    for d in range(30):
        rpos |= ((fpos >> (29 - d)) & 1) << (29 - perm[d])
    at = out.reshape(-1)[torch.from_numpy(rpos).to(DEV)].cpu().numpy().astype(np.complex128)
    want = case.arrays["amps_at_google"].astype(np.complex128)
    fidelity = abs(np.vdot(want, at)) ** 2 / (np.vdot(want, want).real * np.vdot(at, at).real)
    assert fidelity > 0.99, fidelity
    assert rel(at, want) > 1e-4   # and it is not the fp32 path


def test_c_abi_demo_without_python(tmp_path):
    """examples/abi_demo.cpp: libartn_hip.so driven from plain C++ (hipMalloc'ed buffers, a
    hipStream_t, label lists) -- no torch, no Python in the process."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = str(tmp_path / "abi_demo")
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_dir = os.path.join(ROOT, "artensor_amd")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "abi_demo.cpp"), "-L" + lib_dir, "-lartn_hip",
                    "-Wl,-rpath," + lib_dir, "-o", exe], check=True, capture_output=True, timeout=300)
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "kernel=1" in run.stdout   # the tiled MFMA kernel, not the strided fallback


def test_label_tuple_schemes_execute():
    """Schemes compiled with labels="tuples" (bond labels instead of einsum letters; no 50-symbol
    limit) through the dense and the sparse executor: the reference's n12 outputs."""
    import json
    sys.path.insert(0, os.path.dirname(__file__))
    from test_scheme_compilers import Tree
    trees = json.load(open(os.path.join(GOLDEN, "trees.json")))
    case = load_case(os.path.join(GOLDEN, "n12_dense.npz"))
    scheme, _ = A.contraction_scheme(Tree(trees["n12_dense"]["tree"]), labels="tuples")
    raw = A.tensor_contraction(case.fresh_tensors(device=DEV), scheme).cpu().numpy()
    assert rel(raw, case.arrays["raw"]) < 1e-5
    sp = load_case(os.path.join(GOLDEN, "n12_sparse5.npz"))
    rec = trees["n12_sparse"]
    scheme, _, order = A.contraction_scheme_sparse(Tree(rec["tree"]), rec["bitstrings"], sc_target=rec["sc_target"],
                                                   labels="tuples")
    out = A.tensor_contraction_sparse(sp.fresh_tensors(device=DEV), scheme).cpu().numpy().reshape(-1)
    # (the leaf tensors of the sparse pattern do not depend on the bitstring set; this tree was
    #  compiled for 40 bitstrings, whose amplitudes the reference's state vector holds)
    want = case.arrays["state_vec"].reshape(-1)[[int(b, 2) for b in order]]
    assert out.shape == want.shape and amp_rel(out, want, rms=2.0 ** -6) < 5e-5


def test_tn_contract_bookkeeping_and_values():
    """tensor_network.py:207-226 on the device: a 4-tensor ring with a dangling vector, contracted
    pairwise; bond bookkeeping as the reference leaves it, values against numpy."""
    import types
    rng = np.random.default_rng(31)
    bonds = {0: ["a", "b", "v"], 1: ["b", "c"], 2: ["c", "d", "e"], 3: ["d", "a"], 4: ["v"]}
    arrs = {k: crandn(rng, (2,) * len(v)) for k, v in bonds.items()}
    tn = types.SimpleNamespace(tensor_bonds={k: list(v) for k, v in bonds.items()},
                               bond_tensors={}, tensors={k: gpu(v) for k, v in arrs.items()})
    for k, v in bonds.items():
        for b in v:
            tn.bond_tensors.setdefault(b, set()).add(k)
    A.tn_contract(tn, 0, 4)      # dangling vector into its neighbour
    assert tn.tensor_bonds[0] == ["a", "b"] and 4 not in tn.tensors and "v" not in tn.bond_tensors
    A.tn_contract(tn, 0, 1)      # matrix
    assert tn.tensor_bonds[0] == ["a", "c"] and tn.bond_tensors["c"] == {0, 2}
    A.tn_contract(tn, 2, 3)
    assert tn.tensor_bonds[2] == ["c", "e", "a"]
    out = A.tn_contract(tn, 2, 0)   # two shared bonds at once
    assert tn.tensor_bonds == {2: ["e"]} and set(tn.bond_tensors) == {"e"}
    want = np.einsum("abv,v,bc,cde,da->e", arrs[0], arrs[4], arrs[1], arrs[2], arrs[3])
    assert rel(out.cpu().numpy(), want) < 1e-5


@pytest.mark.parametrize("m,n,k,seed", [(12, 11, 9, 0), (11, 12, 10, 1), (13, 10, 9, 2)])
def test_packed_gemm_bf16(m, n, k, seed):
    """precision("bf16"), big x big steps with 2^9+ contracted values (BASELINE configs[4]'s step in small): both operands
    are packed to bfloat16 once (artn_k_pack_bf16) into a scratch buffer the HOST allocates (artn_contract_ws), the GEMM
    (artn_k_pgemm: 256 x 128 tiles, LDS-DMA, 32x32x16 bf16 MFMA) reads the packed copies.  Labels are shuffled so that
    every operand bit lands somewhere else in memory.  Against the numpy restatement of that arithmetic: operands
    rounded to bfloat16 (nearest even), products and sums exact (complex128): only the fp32 accumulation differs."""
    rng = np.random.default_rng(100 + seed)
    kl = [chr(65 + x) for x in range(k)]
    ml = [chr(97 + x) for x in range(m)]
    nl = [chr(65 + k + x) for x in range(n)]
    assert k + n <= 25 and m <= 25
    la, lb, lo = kl + ml, kl + nl, ml + nl
    for l in (la, lb, lo):
        rng.shuffle(l)
    eq = "".join(la) + "," + "".join(lb) + "->" + "".join(lo)
    info = A.step_info(eq, (2,) * len(la), (2,) * len(lb))
    # (complex64 arithmetic packs from 2^11 contracted values on: these steps run on artn_k_gemm, without scratch)
    assert info["workspace_bytes"] == 0 and info["kernel"] == N.KERNEL_GEMM_MFMA
    with A.precision("bf16"):
        info = A.step_info(eq, (2,) * len(la), (2,) * len(lb))
        assert info["kernel"] == N.KERNEL_PGEMM and info["workspace_bytes"] == 4 * (2 ** (m + k) + 2 ** (n + k)), info
        a, b = crandn(rng, (2,) * len(la)), crandn(rng, (2,) * len(lb))
        got = A.contract(eq, gpu(a), gpu(b)).cpu().numpy()
    # (numpy in chunks of the m labels would be faster; 2^(m+n+k) = 2^32 complex MACs as one matmul is fine)
    want = oracle.einsum_pair(eq, _bf16_round(a).astype(np.complex128), _bf16_round(b).astype(np.complex128))
    assert rel(got, want) < 2e-6, (eq, rel(got, want))
    exact = oracle.einsum_pair(eq, a, b)
    assert 1e-4 < rel(got, exact) < 3e-2


@pytest.mark.parametrize("m,n,k,seed", [(12, 11, 11, 0), (11, 12, 13, 1)])
def test_packed_gemm_complex64(m, n, k, seed):
    """The packed-operand GEMM in complex64 arithmetic (3M on fp32 MFMA, 2^11+ contracted values): operands copied once
    into [tile][chunk][k][row] order, LDS-DMA fills, 256 x 128 tiles; k = 13 crosses a partial-sum flush (2^12 contracted
    values per fp32 chain).  Against artn_contract WITHOUT scratch (the two-operand LDS GEMM artn_k_gemm) on the same
    operands, and -- where numpy finishes in seconds -- against a complex128 einsum."""
    import ctypes
    from artensor_amd import contraction as C
    rng = np.random.default_rng(200 + seed)
    kl = [chr(65 + x) for x in range(k)]
    ml = [chr(97 + x) for x in range(m)]
    nl = [chr(65 + k + x) for x in range(n)]
    la, lb, lo = kl + ml, kl + nl, ml + nl
    for l in (la, lb, lo):
        rng.shuffle(l)
    eq = "".join(la) + "," + "".join(lb) + "->" + "".join(lo)
    info = A.step_info(eq, (2,) * len(la), (2,) * len(lb))
    assert info["kernel"] == N.KERNEL_PGEMM and info["workspace_bytes"] == 8 * (2 ** (m + k) + 2 ** (n + k)), info
    assert abs(info["mfma_flops"] / info["flops"] - 0.75) < 1e-9
    a, b = gpu(crandn(rng, (2,) * len(la))), gpu(crandn(rng, (2,) * len(lb)))
    got = A.contract(eq, a, b)
    d, out_shape = C._descriptor(tuple(la), tuple(lb), tuple(lo), tuple(a.shape), tuple(a.stride()), tuple(b.shape), tuple(b.stride()), a.dtype)
    plain = torch.empty(out_shape, dtype=a.dtype, device=a.device)
    N.check(N.lib().artn_contract(ctypes.byref(d), a.data_ptr(), b.data_ptr(), plain.data_ptr(), N.current_stream_ptr(a.device)))
    assert (got - plain).abs().max().item() <= 1e-5 * plain.abs().max().item()
    assert not torch.equal(got, plain)   # (another kernel, another order of additions)
    if k <= 11:
        want = oracle.einsum_pair(eq, a.cpu().numpy().astype(np.complex128), b.cpu().numpy().astype(np.complex128))
        assert rel(got.cpu().numpy(), want) < 1e-5


def test_step_with_more_labels_than_the_einsum_alphabet():
    """One step whose label union exceeds the reference's 50-letter alphabet (contraction.py:9-10; the C ABI carries 96
    labels): 62 labels -- 42 of extent 1 (20 only in the first operand, 12 only in the second, 10 contracted) woven
    between 20 real ones -- as label tuples, on the MFMA kernel, against numpy on the squeezed operands (numpy and torch
    stop at 64 dims per tensor: 48 here); and a 97-label step is refused with a message, not truncated."""
    rng = np.random.default_rng(77)
    real_a = [f"a{x}" for x in range(14)] + [f"k{x}" for x in range(4)]      # 2^18 elements
    real_b = [f"k{x}" for x in range(4)] + [f"n{x}" for x in range(2)]
    ua, uk, ub = [f"ua{x}" for x in range(20)], [f"uk{x}" for x in range(10)], [f"ub{x}" for x in range(12)]
    la = list(real_a)
    for q, u in enumerate(ua + uk):
        la.insert((q * 7) % (len(la) + 1), u)
    lb = list(real_b)
    for q, u in enumerate(uk + ub):
        lb.insert((q * 3) % (len(lb) + 1), u)
    lo = [x for x in la if x[0] == "a" or x.startswith("ua")] + [x for x in lb if x[0] == "n" or x.startswith("ub")]
    rng.shuffle(lo)
    assert len(set(la) | set(lb)) == 62 and max(len(la), len(lb), len(lo)) <= 64
    one = lambda x: x.startswith("u")
    a = crandn(rng, tuple(1 if one(x) else 2 for x in la))
    b = crandn(rng, tuple(1 if one(x) else 2 for x in lb))
    assert A.step_info((tuple(la), tuple(lb), tuple(lo)), a.shape, b.shape)["kernel"] == N.KERNEL_BITS_MFMA
    got = A.contract((tuple(la), tuple(lb), tuple(lo)), gpu(a), gpu(b))
    assert tuple(got.shape) == tuple(1 if one(x) else 2 for x in lo)
    sq = lambda arr, labs: arr.reshape([2] * sum(not one(x) for x in labs))
    mp = {x: chr(65 + q) for q, x in enumerate([x for x in dict.fromkeys(la + lb) if not one(x)])}
    eq = "".join(mp[x] for x in la if x in mp) + "," + "".join(mp[x] for x in lb if x in mp) + "->" + "".join(mp[x] for x in lo if x in mp)
    want = np.einsum(eq, sq(a, la).astype(np.complex128), sq(b, lb).astype(np.complex128))
    assert rel(got.reshape(want.shape).cpu().numpy(), want) < STEP_TOL
    many = tuple(f"z{x}" for x in range(97))   # (no tensor library builds a 97-dim tensor: the descriptor builder is asked directly)
    with pytest.raises(RuntimeError, match="at most 96"):
        A.contraction._descriptor(many, (), many, (1,) * 97, (1,) * 97, (), (), torch.complex64)
