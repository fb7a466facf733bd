"""CPU checks of the planner (artn_plan.h): the bit-GEMM plan, replayed thread by thread by
tests/csrc/plan_emulate.cpp, must reproduce the oracle on the very steps the kernel will see."""
import os

import numpy as np
import pytest

from artensor_amd import step_info
from artensor_amd.fixtures import load_case
from oracle import oracle
from helpers import emulate_pgemm, GOLDEN, crandn, dense_scheme_shapes, emulate, emulate128, emulate_gemm, shrink_step

KERNEL_BITS = 1
KERNEL_GEMM = 2
MFMA_KERNELS = (KERNEL_BITS, KERNEL_GEMM)


def check(eq, a_shape, b_shape, seed=0, expect_bits=None):
    rng = np.random.default_rng(seed)
    a, b = crandn(rng, a_shape), crandn(rng, b_shape)
    got, used = emulate(eq, a, b)
    want = oracle.einsum_pair(eq, a, b)
    assert got.shape == want.shape
    err = np.abs(got - want).max() / np.abs(want).max()
    assert err < 1e-5, (eq, err)
    if expect_bits is not None:
        assert (used in MFMA_KERNELS) == expect_bits, (eq, used)
    return used


def test_n30_big_steps_surrogates():
    """Every big step of the n30 m14 scheme, M truncated to 2^16."""
    case = load_case(os.path.join(GOLDEN, "n30_dense.npz"))
    steps = dense_scheme_shapes(case)
    big = [(n, s) for n, s in enumerate(steps) if np.prod(s[1]) >= 2 ** 20]
    assert len(big) == 28
    n_bits = 0
    for n, (eq, sa, sb) in big:
        eq2, sa2, sb2 = shrink_step(eq, sa, sb, max_log2=16)
        used = check(eq2, sa2, sb2, seed=n)
        n_bits += used in MFMA_KERNELS
        # and the full-size step must be planned onto an MFMA kernel (the growth steps whose second
        # operand is itself big go to the two-operand GEMM kernel when they run unfused)
        info = step_info(eq, sa, sb)
        assert info["kernel"] in MFMA_KERNELS, (n, eq, info)
        assert info["lds_bytes"] <= 72 * 1024
    assert n_bits >= 20  # the surrogates themselves mostly take the MFMA plan


@pytest.mark.parametrize("k,n,ra", [(1, 1, 12), (2, 0, 13), (3, 3, 14), (4, 4, 15), (5, 5, 15), (6, 6, 16),
                                    (4, 7, 13), (6, 2, 16), (1, 6, 12), (5, 1, 14), (7, 3, 15), (8, 4, 15), (7, 6, 16), (8, 0, 14),
                                    (6, 7, 16), (6, 8, 15), (5, 6, 15)])   # (growth steps: 7 / 6 result bits in the tile)
def test_random_bit_steps(k, n, ra):
    """Random bit permutations: K and N bits scattered, output order scrambled."""
    rng = np.random.default_rng(100 * k + n)
    for trial in range(3):
        la = [chr(65 + x) for x in range(ra)]
        kl = list(rng.choice(la, size=k, replace=False))
        nl = [chr(97 + x) for x in range(n)]
        lb = kl + nl
        rng.shuffle(lb)
        lo = [x for x in la if x not in kl] + nl
        rng.shuffle(lo)
        eq = "".join(la) + "," + "".join(lb) + "->" + "".join(lo)
        check(eq, (2,) * ra, (2,) * len(lb), seed=trial)


def test_batch_and_generic_dims():
    # shared batch label (sparse path label -3) with a non power-of-two extent
    check("zabcdefghijk,zkcxy->zabdefghijxy", (5,) + (2,) * 11, (5, 2, 2, 2, 2), expect_bits=True)
    # outer product of batch rows (-1, -2) around a contraction
    check("pabcdefghijkl,qlcx->pqabdefghijkx", (3,) + (2,) * 12, (3, 2, 2, 2), expect_bits=True)
    # bond dimension 4 (two bits per label) and 3 (falls back to the strided kernel)
    check("abcdefg,gcx->abdefx", (4,) * 7, (4, 4, 4), expect_bits=True)
    check("abcdef,fcx->abdex", (3,) * 6, (3, 3, 3), expect_bits=False)
    # label summed out of one operand only, scalar result, pure outer product
    check("abc,cd->a", (2, 3, 4), (4, 2), expect_bits=False)
    check("ab,ab->", (4, 4), (4, 4), expect_bits=False)
    check("ab,cd->acbd", (2, 2), (2, 2), expect_bits=False)


def test_whole_n12_scheme_through_emulator():
    case = load_case(os.path.join(GOLDEN, "n12_dense.npz"))
    tensors = {i: t.numpy().copy() for i, t in case.tensors.items()}
    for (i, j), eq in case.scheme:
        tensors[i], _ = emulate(eq, np.ascontiguousarray(tensors[i]), np.ascontiguousarray(tensors[j]))
    raw = tensors[case.scheme[-1][0][0]]
    want = case.arrays["raw"]
    assert np.abs(raw - want).max() / np.abs(want).max() < 5e-6


def test_fusion_schedule_covers_every_step_once():
    from artensor_amd.contraction import fusion_schedule
    for name in ["n12_dense", "n30_dense", "n30_dense_sliced3", "rand_D2_closed"]:
        case = load_case(os.path.join(GOLDEN, name + ".npz"))
        sched = fusion_schedule(case.scheme)
        seen = []
        for e in sched:
            seen += list(e[1:])
        assert sorted(seen) == list(range(len(case.scheme)))
        # dependencies: a step may only run after every earlier step touching its tensors
        when = {}
        for t, e in enumerate(sched):
            for n in e[1:]:
                when[n] = t
        for n, step in enumerate(case.scheme):
            for m in range(n):
                if set(step[0]) & set(case.scheme[m][0]):
                    assert when[m] <= when[n], (name, m, n)


def test_chain_schedule_covers_every_step_once_and_keeps_dependencies():
    """chain_schedule (what triples are cut from): every step exactly once, and a step never runs before an earlier step
    that touches one of its tensors -- steps moved in front of a chain are independent of it; the cut of every chain
    (_cut_chain through triple_schedule) keeps the members in order and is, without a byte-saving triple, the pairing of
    fusion_schedule restricted to pairs the planner accepts."""
    from artensor_amd.contraction import chain_schedule, triple_schedule
    for name in ["n12_dense", "n30_dense", "n30_dense_sliced3", "rand_D2_closed", "rand_D2_nv260_sliced", "rand_D4_nv100", "n53_m14_sliced"]:
        case = load_case(os.path.join(GOLDEN, name + ".npz"))
        scheme = [(s[0], s[1]) for s in case.scheme]
        sched = chain_schedule(scheme)
        seen, when = [], {}
        for t, e in enumerate(sched):
            members = [e[1]] if e[0] == "one" else list(e[1])
            assert members == sorted(members)
            if e[0] == "chain":
                assert len(members) >= 2 and len({scheme[n][0][0] for n in members}) == 1
            for n in members:
                seen.append(n)
                when[n] = (t, n)
        assert sorted(seen) == list(range(len(scheme))), name
        for n, step in enumerate(scheme):
            for m in range(n):
                if set(step[0]) & set(scheme[m][0]):
                    assert when[m] <= when[n], (name, m, n)
    case = load_case(os.path.join(GOLDEN, "n30_dense.npz"))
    cut = triple_schedule(case.scheme, {k: tuple(t.shape) for k, t in case.tensors.items()})
    flat = [n for e in cut for n in e[1:]]
    assert sorted(flat) == list(range(len(case.scheme)))
    assert not [e for e in cut if e[0] == "triple"]              # as shipped: no triple saves bytes on n30
    assert len([e for e in cut if e[0] == "pair"]) == 13


def test_n30_fused_pairs_surrogates():
    """Every fusable pair of big n30 steps, both steps truncated consistently to 2^16."""
    from artensor_amd.contraction import fusion_schedule, pair_info
    from helpers import emulate2, shrink_pair
    case = load_case(os.path.join(GOLDEN, "n30_dense.npz"))
    steps = dense_scheme_shapes(case)
    pairs = [e for e in fusion_schedule(case.scheme) if e[0] == "pair" and np.prod(steps[e[1]][1]) >= 2 ** 22]
    assert len(pairs) == 13
    fused = 0
    for _, n, m in pairs:
        eq1, sa, sb1 = steps[n]
        eq2, _, sb2 = steps[m]
        assert pair_info(eq1, sa, sb1, eq2, sb2) is not None, (n, m)
        e1, a_s, b1_s, e2, b2_s = shrink_pair(eq1, sa, sb1, eq2, sb2, max_log2=16)
        rng = np.random.default_rng(n)
        a, b1, b2 = crandn(rng, a_s), crandn(rng, b1_s), crandn(rng, b2_s)
        want = oracle.einsum_pair(e2, oracle.einsum_pair(e1, a, b1), b2)
        got, info = emulate2(e1, a, b1, e2, b2)
        if got is None:
            continue
        fused += 1
        assert got.shape == want.shape
        assert np.abs(got - want).max() / np.abs(want).max() < 1e-5, (n, m)
    assert fused >= 10


def test_random_fused_pairs():
    from helpers import emulate2
    rng = np.random.default_rng(3)
    done = 0
    for trial in range(40):
        ra = int(rng.integers(13, 17))
        k1, n1, k2, n2 = (int(x) for x in rng.integers(1, 5, size=4))
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
        got, info = emulate2(eq1, a, b1, eq2, b2)
        if got is None:
            continue
        want = oracle.einsum_pair(eq2, oracle.einsum_pair(eq1, a, b1), b2)
        assert np.abs(got - want).max() / np.abs(want).max() < 1e-5, (eq1, eq2)
        done += 1
    assert done >= 15


def test_fused_pairs_with_batch_labels():
    """Fused pairs whose steps carry a batch label (in A, B and C: the shared rows of the sparse
    executor): batch in both steps, in the first only, in the second only; power-of-two and
    ragged batch extents."""
    from helpers import emulate2
    rng = np.random.default_rng(11)
    done = 0
    for trial in range(30):
        ra = int(rng.integers(12, 15))
        k1, n1, k2, n2 = (int(x) for x in rng.integers(1, 4, size=4))
        hext = int(rng.choice([2, 3, 4, 5]))
        mode = trial % 3  # 0: batch in both steps, 1: first only, 2: second only
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
        # the batch label "z" leads every operand that carries it (dim 0, like the reference's rows)
        la_, lo1_, lo2_ = ["z"] + la, ["z"] + lo1, ["z"] + lo2
        lb1_ = (["z"] if mode in (0, 1) else []) + lb1
        lb2_ = (["z"] if mode in (0, 2) else []) + lb2
        shape = lambda labs: tuple(hext if x == "z" else 2 for x in labs)
        eq1 = "".join(la_) + "," + "".join(lb1_) + "->" + "".join(lo1_)
        eq2 = "".join(lo1_) + "," + "".join(lb2_) + "->" + "".join(lo2_)
        a, b1, b2 = crandn(rng, shape(la_)), crandn(rng, shape(lb1_)), crandn(rng, shape(lb2_))
        got, info = emulate2(eq1, a, b1, eq2, b2)
        if got is None:
            continue
        want = oracle.einsum_pair(eq2, oracle.einsum_pair(eq1, a, b1), b2)
        assert got.shape == want.shape
        assert np.abs(got - want).max() / np.abs(want).max() < 1e-5, (eq1, eq2, hext, mode)
        done += 1
    assert done >= 12


def test_fused_row_gather_emulated():
    """artn_contract_gather: the batch rows of both operands are picked through index arrays inside
    the tile-offset computation (sparse executor branches A/B, reference contraction.py:149-156)."""
    import ctypes
    import torch
    from artensor_amd import contraction as C
    from helpers import emulator
    emu = emulator()
    emu.artn_emulate_gather.restype = ctypes.c_int
    rng = np.random.default_rng(17)
    # (the last three: 7-8 contracted bits and 5+ free bits in the second operand -- the GEMM kernel with row gather,
    #  the very last with the operands swapped inside the plan: too few free bits in the first)
    for (na, nb, n, free, kb, nn) in [(7, 5, 6, 11, 3, 2), (16, 16, 9, 12, 4, 3), (3, 9, 20, 10, 2, 4),
                                      (5, 4, 7, 8, 8, 6), (6, 3, 5, 7, 7, 5), (4, 4, 6, 4, 7, 7)]:
        la = ["z"] + [chr(65 + x) for x in range(free + kb)]
        kl = la[1:1 + kb]
        nl = [chr(97 + x) for x in range(nn)]
        lb = ["z"] + kl[::-1] + nl
        lo = ["z"] + [x for x in la[1:] if x not in kl] + nl
        eq = "".join(la) + "," + "".join(lb) + "->" + "".join(lo)
        a = crandn(rng, (na,) + (2,) * (len(la) - 1))
        b = crandn(rng, (nb,) + (2,) * (len(lb) - 1))
        ra = rng.integers(0, na, size=n).astype(np.int64)
        rb = rng.integers(0, nb, size=n).astype(np.int64)
        want = oracle.einsum_pair(eq, a[ra], b[rb])
        ta, tb = torch.from_numpy(a), torch.from_numpy(b)
        d, out_shape = C._descriptor(tuple(la), tuple(lb), tuple(lo), (n,) + tuple(a.shape[1:]), tuple(ta.stride()),
                                     (n,) + tuple(b.shape[1:]), tuple(tb.stride()), torch.complex64)
        out = np.zeros(out_shape, dtype=np.complex64)
        flag = ctypes.c_int32(0)
        rc = emu.artn_emulate_gather(ctypes.byref(d), a.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p),
                                     out.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(0),
                                     ra.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(na),
                                     rb.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(nb), ctypes.byref(flag))
        assert rc == (2 if kb >= 7 else 0), rc     # 2: planned onto the GEMM kernel
        assert flag.value == 0
        assert np.abs(out - want).max() / np.abs(want).max() < 1e-5, eq


def test_row_gather_emulated_complex128():
    """The complex128 chunk loop's fused row gather (artn_k_gemm128<NB, GATHER>, reference contraction.py:149-156 in
    complex128) replayed on the CPU from the plan make_plan builds for artn_contract_gather: 3 to 8 contracted bits, ragged
    row counts, operands exchanged inside the plan; fewer than 3 contracted bits are declined (the caller gathers)."""
    import ctypes
    import torch
    from artensor_amd import contraction as C
    from helpers import emulator
    emu = emulator()
    emu.artn_emulate_gather.restype = ctypes.c_int
    rng = np.random.default_rng(19)
    for (na, nb, n, free, kb, nn) in [(3, 9, 20, 10, 2, 4), (7, 5, 6, 11, 3, 2), (16, 16, 9, 10, 4, 3), (5, 4, 7, 8, 8, 6),
                                      (6, 3, 5, 7, 7, 5), (4, 4, 6, 4, 7, 7)]:
        la = ["z"] + [chr(65 + x) for x in range(free + kb)]
        kl = la[1:1 + kb]
        nl = [chr(97 + x) for x in range(nn)]
        lb = ["z"] + kl[::-1] + nl
        lo = ["z"] + [x for x in la[1:] if x not in kl] + nl
        eq = "".join(la) + "," + "".join(lb) + "->" + "".join(lo)
        a = (rng.standard_normal((na,) + (2,) * (len(la) - 1)) + 1j * rng.standard_normal((na,) + (2,) * (len(la) - 1)))
        b = (rng.standard_normal((nb,) + (2,) * (len(lb) - 1)) + 1j * rng.standard_normal((nb,) + (2,) * (len(lb) - 1)))
        ra = rng.integers(0, na, size=n).astype(np.int64)
        rb = rng.integers(0, nb, size=n).astype(np.int64)
        want = oracle.einsum_pair(eq, a[ra], b[rb])
        ta, tb = torch.from_numpy(a), torch.from_numpy(b)
        d, out_shape = C._descriptor(tuple(la), tuple(lb), tuple(lo), (n,) + tuple(a.shape[1:]), tuple(ta.stride()),
                                     (n,) + tuple(b.shape[1:]), tuple(tb.stride()), torch.complex128)
        out = np.zeros(out_shape, dtype=np.complex128)
        flag = ctypes.c_int32(0)
        rc = emu.artn_emulate_gather(ctypes.byref(d), a.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p),
                                     out.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(0),
                                     ra.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(na),
                                     rb.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(nb), ctypes.byref(flag))
        if kb < 3:
            assert rc < 0, rc    # declined: fewer contracted bits than one chunk of the f64 GEMM kernel
            continue
        assert rc == 2, (eq, rc)
        assert flag.value == 0
        assert np.abs(out - want).max() / np.abs(want).max() < 1e-13, eq
    # an index outside the operand reads row 0 and raises the flag
    bad = ra.copy()
    bad[0] = 99
    emu.artn_emulate_gather(ctypes.byref(d), a.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p),
                            out.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(0), bad.ctypes.data_as(ctypes.c_void_p),
                            ctypes.c_int64(na), rb.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(nb), ctypes.byref(flag))
    assert flag.value == 1


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
    return np.einsum("".join(mp[x] for x in la) + "," + "".join(mp[x] for x in lb) + "->" + "".join(mp[x] for x in lo),
                     a.astype(np.complex128), b.astype(np.complex128))


@pytest.mark.parametrize("m3", [0, 1])
@pytest.mark.parametrize("m,n,k,batch", [(7, 7, 4, 0), (7, 7, 6, 0), (8, 7, 5, 0), (9, 3, 8, 0), (6, 6, 4, 3), (5, 4, 9, 0),
                                         (10, 0, 5, 0), (3, 9, 6, 0), (7, 5, 7, 2), (6, 7, 5, 0), (8, 2, 10, 0), (5, 3, 13, 0),
                                         (5, 5, 8, 0), (5, 5, 13, 2)])   # (one 32 x 32 block: chunks of 2^6 contracted values)
def test_gemm_plan_emulated(monkeypatch, m, n, k, batch, m3):
    """The two-operand GEMM kernel replayed thread by thread from its plan (global -> LDS images, MFMA
    lane maps, Gray-code walk over the looped contracted bits, C-ordered swizzled result image in one
    or two passes, copy-out): random bit permutations, full 128 x 128 tiles (ARTN_EMU_NCU=1 keeps the
    planner from shrinking tiles for want of workgroups), tiles with fewer than 16 columns, operands
    exchanged (m < 5), a ragged batch axis, and more than 2^12 contracted values (partial sums flushed
    into C and added up there); with four real products per complex product and with three (m3: 32-column
    blocks, T1 / T2 / T3 accumulators, combined in the epilogue)."""
    monkeypatch.setenv("ARTN_EMU_NCU", "1")
    rng = np.random.default_rng(100 * m + 10 * n + k)
    eq, sa, sb = _random_gemm_step(rng, m, n, k, batch)
    a, b = crandn(rng, sa), crandn(rng, sb)
    got, info = emulate_gemm(eq, a, b, m3=m3)
    assert got is not None and info["kernel"] == KERNEL_GEMM
    want = _einsum128(eq, a, b)
    assert got.shape == want.shape
    assert np.abs(got - want).max() / np.abs(want).max() < 1e-5, (eq, info)
    if m >= 7 and n >= 7:
        assert info["tile_out_bits"] == (13 if m3 else 14)   # 4M: two epilogue passes covered; 3M: 128 x 64 tiles


@pytest.mark.parametrize("m,n,k", [(7, 7, 5), (7, 6, 6), (8, 3, 7), (5, 4, 9), (3, 9, 6), (5, 3, 11)])
def test_gemm_plan_emulated_bf16(monkeypatch, m, n, k):
    """bf16 operand mode of the GEMM kernel (chunks of 32 contracted values, [kc >> 2][row][kc & 3] images)
    against a complex128 einsum of the bf16-rounded operands: only fp32 accumulation order differs."""
    import torch
    monkeypatch.setenv("ARTN_EMU_NCU", "1")
    rng = np.random.default_rng(7 * m + n + 31 * k)
    eq, sa, sb = _random_gemm_step(rng, m, n, k)
    a, b = crandn(rng, sa), crandn(rng, sb)

    def bf(x):
        r = torch.view_as_real(torch.from_numpy(np.ascontiguousarray(x))).to(torch.bfloat16).to(torch.float32)
        return torch.view_as_complex(r.contiguous()).numpy()

    got, info = emulate_gemm(eq, a, b, bf16=True)
    assert got is not None and info["kernel"] == KERNEL_GEMM and info["n_tile_bits"] <= 6
    want = _einsum128(eq, bf(a), bf(b))
    assert np.abs(got - want).max() / np.abs(want).max() < 2e-6, (eq, info)


def test_planner_routes_big_contractions_to_the_gemm_kernel():
    # 15 contracted bits between two big operands (the n53 m20 big-batch step): one launch, no split-K
    la = tuple(f"m{x}" for x in range(15)) + tuple(f"k{x}" for x in range(15))
    lb = tuple(f"k{x}" for x in range(15)) + tuple(f"n{x}" for x in range(14))
    lo = tuple(f"m{x}" for x in range(15)) + tuple(f"n{x}" for x in range(14))
    # -- the packed-operand GEMM (256 x 128 tiles) where the caller can supply scratch (the query says how much: two
    # packed copies of 8-byte elements), the two-operand LDS GEMM otherwise; either way every contracted bit in the kernel
    info = step_info((la, lb, lo), (2,) * 30, (2,) * 29)
    assert info["kernel"] == 4 and info["k_bits"] == 15 and info["m_tile_bits"] == 8 and info["n_tile_bits"] == 7
    assert info["workspace_bytes"] == 8 * (2 ** 30 + 2 ** 29)
    from artensor_amd.contraction import _big_k_outer
    assert _big_k_outer(la, lb, lo, (2,) * 30, None, (2,) * 29) is None
    # a closing step (two 2^26 tensors down to 2^10 amplitudes): contracted labels are split off only
    # until enough workgroups have work
    la = tuple(f"m{x}" for x in range(5)) + tuple(f"k{x}" for x in range(21))
    lb = tuple(f"k{x}" for x in range(21)) + tuple(f"n{x}" for x in range(5))
    lo = la[:5] + lb[21:]
    outer = _big_k_outer(la, lb, lo, (2,) * 26, None, (2,) * 26)
    assert outer is not None and 1 <= len(outer) <= 10
    # round 6: a step that cannot fill the CUs even once is a chain of chunks per workgroup -- 16 tiles x 2^16 contracted values
    # between operands of 2^26 / 2^24 elements (0.7 GB: 0.13 ms of streaming) is split down to 2^11 values per tile; the same
    # shape between two operands of 2^30 elements streams for longer than its chains take and keeps the 512-tile target
    la = tuple(f"m{x}" for x in range(8)) + tuple(f"h{x}" for x in range(2)) + tuple(f"k{x}" for x in range(16))
    lb = tuple(f"k{x}" for x in range(16)) + tuple(f"h{x}" for x in range(2)) + tuple(f"n{x}" for x in range(6))
    lo = tuple(f"h{x}" for x in range(2)) + la[:8] + lb[18:]
    outer = _big_k_outer(la, lb, lo, (2,) * 26, None, (2,) * 24)
    assert outer is not None and len(outer) == 5, outer
    import artensor_amd.contraction as _C
    few, _C.SPLIT_K_FEW_TILES = _C.SPLIT_K_FEW_TILES, 0   # (the 512-tile target alone: 2^14 values per tile)
    try:
        assert len(_big_k_outer(la, lb, lo, (2,) * 26, None, (2,) * 24)) == 2
    finally:
        _C.SPLIT_K_FEW_TILES = few
    la = tuple(f"m{x}" for x in range(5)) + tuple(f"k{x}" for x in range(25))
    lb = tuple(f"k{x}" for x in range(25)) + tuple(f"n{x}" for x in range(5))
    lo = la[:5] + lb[25:]
    outer = _big_k_outer(la, lb, lo, (2,) * 30, None, (2,) * 30)
    assert outer is not None and len(outer) == 9, outer   # (1 tile -> 512 tiles of 2^16 values)
    # small second operand, 5 contracted bits: stays on the state-streaming kernel
    info = step_info("ABCDEFGHIJKLMNOPQRST,DHKOSwxyz->ABCEFGIJLMNPQRTwxyz", (2,) * 20, (2,) * 9)
    assert info["kernel"] == KERNEL_BITS


@pytest.mark.parametrize("m,n,k,batch", [(7, 5, 3, 0), (6, 6, 4, 0), (5, 7, 5, 0), (7, 3, 6, 0), (8, 2, 3, 0), (9, 0, 4, 0),
                                         (3, 8, 5, 0), (6, 7, 7, 0), (6, 5, 4, 3), (8, 8, 3, 0)])
def test_complex128_gemm_plan_emulated(monkeypatch, m, n, k, batch):
    """complex128 on v_mfma_f64_16x16x4_f64 (artn_k_gemm128): 16-byte elements, one per copy lane, 16 x 8 MFMA
    blocks with the f64 C/D lane map, 2^12-element epilogue passes -- replayed thread by thread in double."""
    monkeypatch.setenv("ARTN_FORCE_BITS", "1")
    rng = np.random.default_rng(50 * m + 5 * n + k)
    eq, sa, sb = _random_gemm_step(rng, m, n, k, batch)
    a = rng.standard_normal(sa) + 1j * rng.standard_normal(sa)
    b = rng.standard_normal(sb) + 1j * rng.standard_normal(sb)
    got, used = emulate128(eq, a, b)
    assert got is not None and used == KERNEL_GEMM
    want = _einsum128(eq, a, b)
    assert np.abs(got - want).max() / np.abs(want).max() < 1e-13, eq


def test_complex128_big_steps_of_n30_plan_onto_the_matrix_cores():
    """Every big step of the n30 scheme is given to the f64 MFMA kernel when the tensors are complex128
    (round 1: the strided one-thread-per-element kernel)."""
    import torch
    case = load_case(os.path.join(GOLDEN, "n30_dense.npz"))
    steps = dense_scheme_shapes(case)
    big = [s for s in steps if np.prod(s[1]) >= 2 ** 20]
    for eq, sa, sb in big:
        info = step_info(eq, sa, sb, dtype=torch.complex128)
        assert info["kernel"] == KERNEL_GEMM, (eq, info)


@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("m,n,k", [(9, 8, 9), (8, 9, 10), (10, 7, 8), (8, 7, 13)])
def test_packed_gemm_plan_emulated(monkeypatch, m, n, k, bf16):
    """The packed-operand GEMM replayed on the CPU from its plan: both packing passes (which source element lands in which
    16-byte unit of which tile / chunk / plane), the tile index -> (m-outer, n-outer) map, the chunk images as the MFMA
    lanes address them, 3M arithmetic with the partial-sum flush past 2^12 contracted values (k = 13), the swizzled
    C-ordered result image in four passes and the copy-out strides -- on random bit permutations, operands exchanged
    (n > m), against a complex128 einsum (of the bfloat16-rounded operands in the reduced-precision form)."""
    monkeypatch.setenv("ARTN_EMU_NCU", "1")   # (a handful of tiles: the planner's fill-the-chip test is not the subject)
    if bf16 and k < 9:
        pytest.skip("the reduced-precision form packs from 2^9 contracted values on")
    rng = np.random.default_rng(1000 * m + 10 * n + k)
    eq, sa, sb = _random_gemm_step(rng, m, n, k, 0)
    a, b = crandn(rng, sa), crandn(rng, sb)
    got, info = emulate_pgemm(eq, a, b, bf16=bf16)
    assert got is not None, "planner declined"
    assert info["kernel"] == 4 and info["workspace_bytes"] == (4 if bf16 else 8) * (2 ** (max(m, n) + k) + 2 ** (min(m, n) + k))
    if bf16:
        import torch

        def bf(x):
            r = torch.view_as_real(torch.from_numpy(np.ascontiguousarray(x))).to(torch.bfloat16).to(torch.float32)
            return torch.view_as_complex(r.contiguous()).numpy()
        want = _einsum128(eq, bf(a), bf(b))
        tol = 2e-6 if k <= 10 else 1e-5   # (one unflushed fp32 chain over 2^13 values: 4e-6)
    else:
        want = _einsum128(eq, a, b)
        tol = 1e-5
    assert got.shape == want.shape
    assert np.abs(got - want).max() / np.abs(want).max() < tol, (eq, info)


# ---- complex128 on the state-streaming kernel (artn_k_bits128): the lane-by-lane replay of the f64 MFMA stages ----------
def _c128(rng, shape):
    return (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(np.complex128)


@pytest.mark.parametrize("k,n,ra", [(1, 1, 12), (2, 0, 13), (3, 3, 14), (4, 4, 14), (5, 5, 14), (6, 5, 15), (4, 6, 13), (6, 2, 15),
                                    (1, 5, 12), (5, 1, 14), (2, 4, 11)])
def test_complex128_random_bit_steps(k, n, ra):
    """Single steps forced onto the complex128 state-streaming plan: K and N bits scattered, output order scrambled."""
    from helpers import emulate_bits
    rng = np.random.default_rng(1000 + 100 * k + n)
    done = 0
    for trial in range(3):
        la = [chr(65 + x) for x in range(ra)]
        kl = list(rng.choice(la, size=k, replace=False))
        nl = [chr(97 + x) for x in range(n)]
        lb = kl + nl
        rng.shuffle(lb)
        lo = [x for x in la if x not in kl] + nl
        rng.shuffle(lo)
        eq = "".join(la) + "," + "".join(lb) + "->" + "".join(lo)
        a, b = _c128(rng, (2,) * ra), _c128(rng, (2,) * len(lb))
        got, info = emulate_bits(eq, a, b)
        if got is None:
            continue
        want = np.einsum(eq, a, b)
        assert np.abs(got - want).max() / np.abs(want).max() < 1e-13, eq
        assert info["lds_bytes"] <= 80 * 1024
        done += 1
    assert done >= 2


def test_complex128_random_fused_pairs():
    from helpers import emulate2
    rng = np.random.default_rng(5)
    done = 0
    for trial in range(40):
        ra = int(rng.integers(12, 16))
        k1, n1, k2, n2 = (int(x) for x in rng.integers(1, 5, size=4))
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
        a, b1, b2 = _c128(rng, (2,) * ra), _c128(rng, (2,) * len(lb1)), _c128(rng, (2,) * len(lb2))
        got, info = emulate2(eq1, a, b1, eq2, b2)
        if got is None:
            continue
        want = np.einsum(eq2, np.einsum(eq1, a, b1), b2)
        assert np.abs(got - want).max() / np.abs(want).max() < 1e-13, (eq1, eq2)
        assert info["arith"] == 3
        done += 1
    assert done >= 15


def test_complex128_n30_fused_pairs_surrogates():
    """The 13 fusable pairs of big n30 steps in complex128: planned as fused pairs at full size (2^11-element tiles: two
    32 KiB regions, two workgroups per CU) and replayed at 2^15."""
    import torch
    from artensor_amd.contraction import fusion_schedule, pair_info
    from helpers import emulate2, shrink_pair
    case = load_case(os.path.join(GOLDEN, "n30_dense.npz"))
    steps = dense_scheme_shapes(case)
    pairs = [e for e in fusion_schedule(case.scheme) if e[0] == "pair" and np.prod(steps[e[1]][1]) >= 2 ** 22]
    assert len(pairs) == 13
    planned = fused = 0
    for _, n, m in pairs:
        eq1, sa, sb1 = steps[n]
        eq2, _, sb2 = steps[m]
        info = pair_info(eq1, sa, sb1, eq2, sb2, dtype=torch.complex128)
        if info is None:
            continue
        planned += 1
        assert info["arith"] == 3 and info["lds_bytes"] <= 80 * 1024, info
        e1, a_s, b1_s, e2, b2_s = shrink_pair(eq1, sa, sb1, eq2, sb2, max_log2=15)
        rng = np.random.default_rng(n)
        a, b1, b2 = _c128(rng, a_s), _c128(rng, b1_s), _c128(rng, b2_s)
        got, _ = emulate2(e1, a, b1, e2, b2)
        if got is None:
            continue
        fused += 1
        want = np.einsum(e2, np.einsum(e1, a, b1), b2)
        assert np.abs(got - want).max() / np.abs(want).max() < 1e-13, (n, m)
    assert planned >= 10 and fused >= 8, (planned, fused)


@pytest.mark.parametrize("k1,k2", [(5, 2), (5, 3), (5, 4), (6, 2), (6, 3), (6, 4), (2, 5), (3, 5), (4, 5), (2, 6), (3, 6), (4, 6)])
def test_fused_pairs_3m_with_a_narrow_stage(k1, k2):
    """3M pairs whose other stage contracts 2-4 bits: that stage runs three products too, on v_mfma_f32_16x16x4_f32 blocks
    (ArtnStage::m3 = 2: lane group l >> 4 carries contracted bits 0, 1 and column bits 2, 3; two 16-column halves per
    sub-tile).  Size-preserving steps, a growing narrow stage and one with fewer than 16 columns."""
    from helpers import emulate2
    rng = np.random.default_rng(100 * k1 + k2)
    done = 0
    for trial in range(4):
        ra = int(rng.integers(15, 17))
        n1 = k1 + (1 if (trial == 1 and k1 <= 4) else 0) - (1 if (trial == 2 and k1 <= 4) else 0)
        n2 = k2 + (1 if (trial == 1 and k2 <= 4) else 0) - (1 if (trial == 2 and k2 <= 4) else 0)
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
        a, b1, b2 = crandn(rng, (2,) * ra), crandn(rng, (2,) * len(lb1)), crandn(rng, (2,) * len(lb2))
        got, info = emulate2(eq1, a, b1, eq2, b2)
        if got is None:
            continue
        want = oracle.einsum_pair(eq2, oracle.einsum_pair(eq1, a, b1), b2)
        assert np.abs(got - want).max() / np.abs(want).max() < 1e-5, (eq1, eq2)
        if info["arith"] == 1:
            done += 1
    assert done >= 2, done   # at least two of the trials really ran the 3M plan


def _random_triple(rng, ra, ks, reuse=0.6):
    """Three consecutive rank-preserving steps on a rank-`ra` tensor; with probability `reuse` a contracted label of a later
    step is one the step before has just produced (consecutive gates on overlapping qubits: what makes triples fit)."""
    labels = [chr(65 + x) for x in range(ra)]
    fresh = iter("abcdefghijklmnopqrstuvwxyz")
    eqs, shapes = [], []
    cur, last_new = list(labels), []
    for k in ks:
        pick = []
        for x in last_new:
            if len(pick) < k and rng.random() < reuse:
                pick.append(x)
        rest = [x for x in cur if x not in pick]
        pick += [str(x) for x in rng.choice(rest, size=k - len(pick), replace=False)]
        new = [next(fresh) for _ in range(k)]
        lb = pick + new
        rng.shuffle(lb)
        lo = [x for x in cur if x not in pick] + new
        rng.shuffle(lo)
        eqs.append("".join(cur) + "," + "".join(lb) + "->" + "".join(lo))
        shapes.append((2,) * len(lb))
        cur, last_new = lo, new
    return eqs, shapes


@pytest.mark.parametrize("ks", [(4, 3, 4), (4, 4, 5), (3, 3, 3), (5, 4, 3), (3, 5, 4), (5, 5, 4), (4, 4, 4), (5, 3, 5)])
def test_fused_triples_emulated(ks):
    """Three steps in one pass (artn_k_bits3: region 0 -> 1 -> 0 -> 1; plan make_bits3), replayed stage by stage on the CPU
    from the same ArtnBitsPlan the kernel gets: random label orders with later steps contracting bits the step before has
    just produced; 3M instantiations (a 5-bit stage: three products there, 16 x 16 x 4 blocks in the narrow stages) and
    four-product ones; against three oracle steps.  Triples that do not fit a 2^12 tile are declined."""
    from helpers import emulate3
    rng = np.random.default_rng(1000 * ks[0] + 100 * ks[1] + 10 * ks[2])
    done = declined = 0
    for trial in range(6):
        ra = int(rng.integers(14, 17))
        eqs, bshapes = _random_triple(rng, ra, ks, reuse=0.5 + 0.1 * trial)
        a = crandn(rng, (2,) * ra)
        bs = [crandn(rng, sh) for sh in bshapes]
        got, info = emulate3(eqs[0], a, bs[0], eqs[1], bs[1], eqs[2], bs[2])
        if got is None:
            declined += 1
            continue
        want = oracle.einsum_pair(eqs[2], oracle.einsum_pair(eqs[1], oracle.einsum_pair(eqs[0], a, bs[0]), bs[1]), bs[2])
        assert got.shape == want.shape
        assert np.abs(got - want).max() / np.abs(want).max() < 1e-5, eqs
        assert (info["k_bits"], info["k2_bits"], info["k3_bits"]) == ks and info["tile_in_bits"] == 12
        assert info["arith"] == (1 if 5 in ks else 0)
        done += 1
    assert done >= 2, (done, declined)


def test_triples_that_do_not_fit_are_declined():
    from helpers import emulate3
    rng = np.random.default_rng(5)
    # 5 + 5 + 5: fragments of 96 registers; 6-bit stages; a step that grows its tensor; twelve old contracted bits
    for ks, grow in [((5, 5, 5), 0), ((6, 3, 3), 0), ((4, 4, 4), 1)]:
        ra = 16
        eqs, bshapes = _random_triple(rng, ra, ks, reuse=1.0)
        if grow:   # the second step brings one more bit than it contracts
            lhs, lo = eqs[1].split("->")
            la, lb = lhs.split(",")
            eqs[1] = la + "," + lb + "Z->" + lo + "Z"
            bshapes[1] = bshapes[1] + (2,)
            lhs3, lo3 = eqs[2].split("->")
            la3, lb3 = lhs3.split(",")
            eqs[2] = la3 + "Z," + lb3 + "->" + lo3 + "Z"
        a = crandn(rng, (2,) * ra)
        bs = [crandn(rng, sh) for sh in bshapes]
        got, info = emulate3(eqs[0], a, bs[0], eqs[1], bs[1], eqs[2], bs[2])
        assert got is None, ks
    eqs, bshapes = _random_triple(rng, 16, (5, 4, 4), reuse=0.0)   # 13 old contracted bits cannot share a 2^12 tile
    got, _ = emulate3(eqs[0], crandn(rng, (2,) * 16), crandn(rng, bshapes[0]), eqs[1], crandn(rng, bshapes[1]), eqs[2], crandn(rng, bshapes[2]))
    assert got is None


def test_n30_triples_planned_and_emulated():
    """The n30 m14 scheme: the two runs of three consecutive rank-30 steps whose contracted old bits and 128-byte runs fit
    one 2^12 tile -- steps (125, 128, 131) and (149, 155, 159) -- planned at full size and replayed on surrogates
    truncated to 2^16 elements; every other run of three is declined with 128-byte runs (the planner's answer: this is why
    three-step fusion does not shorten THIS scheme: 26 chain members = 2 triples + 10 pairs would need the triples to
    sit at even distances; tools/fusion_depth.py)."""
    from helpers import emulate3, shrink_triple
    import torch
    case = load_case(os.path.join(GOLDEN, "n30_dense.npz"))
    steps = dense_scheme_shapes(case)
    big = [n for n, (eq, sa, sb) in enumerate(steps) if len(sa) == 30]
    fit = []
    for p in range(len(big) - 2):
        (e1, sa, sb1), (e2, _, sb2), (e3, _, sb3) = (steps[n] for n in big[p:p + 3])
        # (plan only, at full size, through the CPU emulator's copy of the planner: the product library carries no triples)
        meta = lambda sh: torch.empty(sh, dtype=torch.complex64, device="meta")
        if emulate3(e1, meta(sa), meta(sb1), e2, meta(sb2), e3, meta(sb3), run=False)[0] is not None:
            fit.append(tuple(big[p:p + 3]))
    assert fit == [(125, 128, 131), (149, 155, 159)], fit
    rng = np.random.default_rng(3)
    for n1, n2, n3 in fit:
        (eq1, sa, sb1), (eq2, _, sb2), (eq3, _, sb3) = steps[n1], steps[n2], steps[n3]
        e1, sa_, sb1_, e2, sb2_, e3, sb3_ = shrink_triple(eq1, sa, sb1, eq2, sb2, eq3, sb3, max_log2=16)
        a, b1, b2, b3 = crandn(rng, sa_), crandn(rng, sb1_), crandn(rng, sb2_), crandn(rng, sb3_)
        got, info = emulate3(e1, a, b1, e2, b2, e3, b3)
        assert got is not None, (n1, n2, n3)
        want = oracle.einsum_pair(e3, oracle.einsum_pair(e2, oracle.einsum_pair(e1, a, b1), b2), b3)
        assert np.abs(got - want).max() / np.abs(want).max() < 1e-5, (n1, n2, n3)


def _synthetic_pair(k1, k2, rank=24):
    """a fused pair on a 2^rank state: k1 contracted bits (the low ones bar the 4-bit run), then k2 of the bits just produced"""
    la = [chr(65 + x) for x in range(rank)]
    kl1 = la[4:4 + k1]
    nl1 = [chr(97 + x) for x in range(k1)]
    lo1 = la[:4] + nl1 + la[4 + k1:]
    kl2 = nl1[:k2] if k2 <= k1 else nl1 + la[4 + k1:4 + k1 + (k2 - k1)]
    nl2 = [chr(110 + x) for x in range(k2)]
    lo2 = [x for x in lo1 if x not in kl2]
    lo2 = lo2[:4] + nl2 + lo2[4:]
    e1 = "".join(la) + "," + "".join(kl1 + nl1) + "->" + "".join(lo1)
    e2 = "".join(lo1) + "," + "".join(kl2 + nl2) + "->" + "".join(lo2)
    return e1, (2,) * rank, (2,) * (2 * k1), e2, (2,) * (2 * k2)


def test_wide_kernel_is_planned_for_pairs_with_11_or_more_contracted_bits():
    """ArtnBitsPlan::wide8 (artn_k_wide): the default sends the fused pairs whose fragments artn_k_bits cannot
    hold -- 5+6, 6+5, 6+6 -- to the one-workgroup-per-CU kernel (four 32 KiB regions, one workgroup per CU, three-product
    arithmetic) and nothing else; with ARTN_WIDE=0 a 6+6 pair does not fuse at all (checked in a process of its own: the
    planner reads its tuning once)."""
    from artensor_amd.contraction import pair_info
    is_wide = lambda info: info["lds_bytes"] >= 4 * 32768 and info["grid"] <= 256
    for k1, k2, want in ((5, 6, True), (6, 5, True), (6, 6, True), (5, 5, False), (6, 4, False), (4, 4, False)):
        info = pair_info(*_synthetic_pair(k1, k2))
        assert info is not None and (info["k_bits"], info["k2_bits"]) == (k1, k2), (k1, k2, info)
        assert info["tile_in_bits"] == 12 and info["tile_out_bits"] == 12
        assert is_wide(info) == want, (k1, k2, info)
        if want:
            assert info["arith"] == 1 and abs(info["mfma_flops"] - 0.75 * info["flops"]) < 1.0
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from test_plan_emulation import _synthetic_pair\n"
            "from artensor_amd.contraction import pair_info\n"
            "print('66', pair_info(*_synthetic_pair(6, 6)) is None)\n"
            "i = pair_info(*_synthetic_pair(5, 6)); print('56', i['lds_bytes'] < 4 * 32768)\n") % (ROOT, os.path.join(ROOT, "tests"))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ARTN_WIDE="0"), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "66 True" in out.stdout and "56 True" in out.stdout, out.stdout + out.stderr[-2000:]


@pytest.mark.parametrize("k,nt,seed", [(5, 2, 0), (5, 4, 1), (6, 3, 2), (6, 1, 3), (5, 0, 4)])
def test_shrinking_single_steps_on_narrow_three_product_blocks(k, nt, seed):
    """ArtnBitsPlan::narrow3: a single step with 5 or 6 contracted bits that keeps at most 4 result bits in the tile (the steps
    that shrink their tensor; reference loop contraction.py:66-70) runs three products on 16 x 16 x 4 blocks -- the stage of
    artn_k_wide on four waves -- instead of four on 32 x 32 blocks of which at most 16 rows are results.  Planned that way
    (arithmetic 1 = 3M, state-streaming kernel) and replayed lane by lane against the oracle, scattered bit positions."""
    rng = np.random.default_rng(900 + seed)
    ra = 18
    la = [chr(65 + x) for x in range(ra)]
    kl = list(rng.choice(la, size=k, replace=False))
    nl = [chr(97 + x) for x in range(nt)]
    lb = kl + nl
    rng.shuffle(lb)
    lo = [x for x in la if x not in kl]
    for x in nl:
        lo.insert(int(rng.integers(0, len(lo) + 1)), x)
    eq = "".join(la) + "," + "".join(lb) + "->" + "".join(lo)
    info = step_info(eq, (2,) * ra, (2,) * len(lb))
    a, b = crandn(rng, (2,) * ra), crandn(rng, (2,) * len(lb))
    got, used = emulate(eq, a, b)
    want = oracle.einsum_pair(eq, a, b)
    assert np.abs(got - want).max() <= 2e-5 * np.abs(want).max(), eq
    if info["kernel"] == 1 and used == 1:   # (the state-streaming kernel took it: then as a three-product stage)
        assert info["arith"] == 1 and abs(info["mfma_flops"] - 0.75 * info["flops"]) < 1.0, info
