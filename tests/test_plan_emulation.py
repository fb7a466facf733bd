"""CPU checks of the planner (artn_plan.h): the bit-GEMM plan, replayed thread by thread by
tests/csrc/plan_emulate.cpp, must reproduce the oracle on the very steps the kernel will see."""
import os

import numpy as np
import pytest

from artensor_amd import step_info
from artensor_amd.fixtures import load_case
from oracle import oracle
from helpers import GOLDEN, crandn, dense_scheme_shapes, emulate, shrink_step

KERNEL_BITS = 1


def check(eq, a_shape, b_shape, seed=0, expect_bits=None):
    rng = np.random.default_rng(seed)
    a, b = crandn(rng, a_shape), crandn(rng, b_shape)
    got, used = emulate(eq, a, b)
    want = oracle.einsum_pair(eq, a, b)
    assert got.shape == want.shape
    err = np.abs(got - want).max() / np.abs(want).max()
    assert err < 1e-5, (eq, err)
    if expect_bits is not None:
        assert (used == KERNEL_BITS) == expect_bits, (eq, used)
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
        n_bits += used == KERNEL_BITS
        # and the full-size step must be planned onto the MFMA kernel
        info = step_info(eq, sa, sb)
        assert info["kernel"] == KERNEL_BITS, (n, eq, info)
        assert info["lds_bytes"] <= 72 * 1024
    assert n_bits >= 20  # the surrogates themselves mostly take the MFMA plan


@pytest.mark.parametrize("k,n,ra", [(1, 1, 12), (2, 0, 13), (3, 3, 14), (4, 4, 15), (5, 5, 15), (6, 6, 16),
                                    (4, 7, 13), (6, 2, 16), (1, 6, 12), (5, 1, 14), (7, 3, 15), (8, 4, 15), (7, 6, 16), (8, 0, 14)])
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
    for (na, nb, n, free, kb, nn) in [(7, 5, 6, 11, 3, 2), (16, 16, 9, 12, 4, 3), (3, 9, 20, 10, 2, 4)]:
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
        assert rc == 0, rc
        assert flag.value == 0
        assert np.abs(out - want).max() / np.abs(want).max() < 1e-5, eq
