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
        assert info["lds_bytes"] <= 64 * 1024
    assert n_bits >= 20  # the surrogates themselves mostly take the MFMA plan


@pytest.mark.parametrize("k,n,ra", [(1, 1, 12), (2, 0, 13), (3, 3, 14), (4, 4, 15), (5, 5, 15), (6, 6, 16),
                                    (4, 7, 13), (6, 2, 16), (1, 6, 12), (5, 1, 14)])
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
