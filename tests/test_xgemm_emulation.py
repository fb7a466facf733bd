"""CPU checks of the extent-based GEMM (artn_k_xgemm, artensor_amd/csrc/artn_xgemm_plan.h / artn_xgemm_kernel.h): steps whose
labels have extents that are not powers of two -- the reference contracts any bond_dims
(/root/reference/artensor/tensor_network.py:4-30, the einsum at contraction.py:70) -- planned by make_xgemm and replayed
thread by thread by tests/csrc/plan_emulate.cpp::run_xgemm, against the oracle."""
import numpy as np
import pytest

from artensor_amd import step_info
from oracle import oracle
from helpers import crandn, emulate, emulate_xgemm

KERNEL_XGEMM = 5


def random_step(rng, ext_choices, n_m, n_n, n_k, n_h=0):
    """Random label orders in A, B and C; extents drawn per label."""
    M = [f"m{i}" for i in range(n_m)]
    Nn = [f"n{i}" for i in range(n_n)]
    K = [f"k{i}" for i in range(n_k)]
    H = [f"h{i}" for i in range(n_h)]
    ext = {x: int(rng.choice(ext_choices)) for x in M + Nn + K + H}
    la, lb, lo = M + K + H, Nn + K + H, M + Nn + H
    for lst in (la, lb, lo):
        rng.shuffle(lst)
    return (tuple(la), tuple(lb), tuple(lo)), tuple(ext[x] for x in la), tuple(ext[x] for x in lb)


def einsum_labels(eq, a, b):
    la, lb, lo = eq
    sym = {}
    for x in la + lb + lo:
        sym.setdefault(x, chr(65 + len(sym)) if len(sym) < 26 else chr(97 + len(sym) - 26))
    s = "".join(sym[x] for x in la) + "," + "".join(sym[x] for x in lb) + "->" + "".join(sym[x] for x in lo)
    return np.einsum(s, a.astype(np.complex128), b.astype(np.complex128))


def check(eq, a, b, tol=2e-6, n_cu=None, rows16=False):
    got, info, modes = emulate_xgemm(eq, a, b, n_cu=n_cu, rows16=rows16)
    assert got is not None, "make_xgemm declined"
    want = einsum_labels(eq, a, b)
    assert got.shape == want.shape
    assert not np.isnan(got).any(), "an output element was never stored"
    err = np.abs(got - want).max() / np.abs(want).max()
    assert err < tol, (eq, err, modes)
    assert info["kernel"] == KERNEL_XGEMM
    return info, modes


@pytest.mark.parametrize("seed", range(12))
def test_random_bond_dimension_3_steps(seed):
    rng = np.random.default_rng(seed)
    n_m, n_n, n_k = int(rng.integers(3, 8)), int(rng.integers(1, 5)), int(rng.integers(0, 5))
    eq, sa, sb = random_step(rng, [3], n_m, n_n, n_k)
    check(eq, crandn(rng, sa), crandn(rng, sb))


@pytest.mark.parametrize("seed", range(12))
def test_random_mixed_extents(seed):
    """Extents 2..7 mixed (D = 6 = 2 x 3 networks, D = 5), batch labels, every copy mode and both operand roles."""
    rng = np.random.default_rng(1000 + seed)
    n_m, n_n, n_k, n_h = int(rng.integers(2, 6)), int(rng.integers(1, 4)), int(rng.integers(1, 4)), int(rng.integers(0, 2))
    eq, sa, sb = random_step(rng, [2, 3, 5, 6, 7], n_m, n_n, n_k, n_h)
    check(eq, crandn(rng, sa), crandn(rng, sb))


@pytest.mark.parametrize("seed", range(10))
def test_random_steps_planned_for_a_one_cu_device(seed):
    """The same random steps planned for ONE CU: a test-size step is then a launch of many rounds and the planner may cut the
    last columns off into a second launch (tail_nb); the replay runs both launches."""
    rng = np.random.default_rng(3000 + seed)
    M = {f"m{i}": int(rng.choice([3, 5, 6])) for i in range(5)}           # 243 .. 7 776 rows
    Nn = {f"n{i}": int(rng.choice([5, 6, 7])) for i in range(3)}          # 125 .. 343 columns: 4 .. 11 blocks of 32
    K = {f"k{i}": int(rng.choice([3, 5])) for i in range(int(rng.integers(1, 3)))}
    H = {"h0": 2} if seed % 3 == 0 else {}
    ext = {**M, **Nn, **K, **H}
    la, lb, lo = list(M) + list(K) + list(H), list(Nn) + list(K) + list(H), list(M) + list(Nn) + list(H)
    for lst in (la, lb, lo):
        rng.shuffle(lst)
    info, modes = check((tuple(la), tuple(lb), tuple(lo)), crandn(rng, tuple(ext[x] for x in la)), crandn(rng, tuple(ext[x] for x in lb)), n_cu=1)
    cols = min(int(np.prod(list(M.values()))), int(np.prod(list(Nn.values()))))
    blocks = -(-cols // 32)
    assert modes["nb"] == 3 and modes["tail_nb"] == blocks % 3, (modes, cols)


@pytest.mark.parametrize("seed", range(10))
def test_random_row_streaming_steps(seed):
    """Random steps of the row-streaming form: 2^15+ rows over 4-7 row labels, at most 48 contracted values and 48 columns over
    one or two labels each, labels shuffled in both operands and in the result -- whose fastest label stays a row label."""
    rng = np.random.default_rng(4000 + seed)
    while True:
        M = {f"m{i}": int(rng.choice([2, 3, 4, 5, 7, 9])) for i in range(int(rng.integers(4, 8)))}
        rows = int(np.prod(list(M.values())))
        if 1 << 15 <= rows <= 1 << 17:
            break
    def small(prefix):
        while True:
            d = {f"{prefix}{i}": int(rng.choice([2, 3, 4, 5, 6, 7])) for i in range(int(rng.integers(1, 3)))}
            if int(np.prod(list(d.values()))) <= 48:
                return d
    K, Nn = small("k"), small("n")
    ext = {**M, **K, **Nn}
    la, lb, lo = list(M) + list(K), list(Nn) + list(K), list(M) + list(Nn)
    for lst in (la, lb, lo):
        rng.shuffle(lst)
    last_m = max(i for i, x in enumerate(lo) if x in M)
    lo[last_m], lo[-1] = lo[-1], lo[last_m]
    a, b = crandn(rng, tuple(ext[x] for x in la)), crandn(rng, tuple(ext[x] for x in lb))
    info, modes = check((tuple(la), tuple(lb), tuple(lo)), a, b)
    small = int(np.prod(list(K.values()))) <= 32 and int(np.prod(list(Nn.values()))) <= 32
    assert modes["rowmode"] == (2 if small else 1), (modes, ext)   # a lane per row (artn_k_xrow64) up to 32 x 32, 16-row blocks beyond
    if small:
        info, modes = check((tuple(la), tuple(lb), tuple(lo)), a, b, rows16=True)
        assert modes["rowmode"] == 1, (modes, ext)


def test_all_modes_are_covered():
    """amode / bmode / trans / swapped each take both values over a set of steps (otherwise a copy mode could rot)."""
    seen = {name: set() for name in ("amode", "bmode", "trans", "swapped", "nb")}
    for seed in range(40):
        rng = np.random.default_rng(5000 + seed)
        eq, sa, sb = random_step(rng, [3], int(rng.integers(3, 7)), int(rng.integers(1, 5)), int(rng.integers(1, 4)))
        _, modes = check(eq, crandn(rng, sa), crandn(rng, sb))
        for name in seen:
            seen[name].add(modes[name])
    rng = np.random.default_rng(6000)   # 60 columns: two blocks of 32
    _, modes = check((("m", "k"), ("k", "p", "q"), ("q", "m", "p")), crandn(rng, (200, 11)), crandn(rng, (11, 5, 12)))
    seen["nb"].add(modes["nb"])
    assert seen["amode"] == {0, 1} and seen["bmode"] == {0, 1} and seen["trans"] == {0, 1} and seen["swapped"] == {0, 1}
    assert seen["nb"] == {1, 2, 3}


def test_columns_behind_the_full_tiles_run_as_a_second_launch():
    """Round 6 (ArtnXGemmPlan::tail_nb, artn_xg_tail_plan): 216 columns are two tiles of three 32-column blocks plus ONE block in a
    second launch (7 blocks in 3 tiles; one launch needs 4 tiles of 2 = 8 blocks), 243 columns 2 x 3 + 2; both operand roles
    (lanes of a store along rows / along columns), a batch label, partial sums through C; 192 columns need no tail."""
    rng = np.random.default_rng(77)
    # 6 x 6 x 6 columns, rows fastest in the result
    eq = (("k0", "m1", "m0"), ("n2", "k0", "n1", "n0"), ("n2", "n1", "n0", "m1", "m0"))
    info, modes = check(eq, crandn(rng, (11, 30, 13)), crandn(rng, (6, 11, 6, 6)))
    assert (modes["nb"], modes["tail_nb"]) == (2, 0), modes   # 4 x 4 tiles on 256 CUs: one round -- one launch, as before
    info, modes = check(eq, crandn(rng, (11, 30, 13)), crandn(rng, (6, 11, 6, 6)), n_cu=1)   # (8+ tiles per CU: many rounds)
    assert (modes["nb"], modes["tail_nb"], modes["trans"]) == (3, 1, 0), modes
    assert info["n_tiles"] == 4 * 2 + 4 and info["a_rereads"] == 3, info   # 390 rows: 4 row tiles x (2 full column tiles + the tail's 1)
    # 3^5 columns, columns fastest in the result (the MFMA roles swapped), a batch label
    eq = (("h", "m0", "k0", "k1"), ("k1", "h", "n0", "k0"), ("h", "m0", "n0"))
    info, modes = check(eq, crandn(rng, (2, 300, 5, 7)), crandn(rng, (7, 2, 243, 5)), n_cu=1)
    assert (modes["nb"], modes["tail_nb"], modes["trans"]) == (3, 2, 1), modes
    # more than 4 096 contracted values: the partial sums of the tail launch go through C as well
    eq = (("m0", "k0", "k1"), ("k1", "n0", "k0"), ("n0", "m0"))
    info, modes = check(eq, crandn(rng, (1100, 81, 63)), crandn(rng, (63, 130, 81)), tol=5e-6, n_cu=1)
    assert (modes["nb"], modes["tail_nb"], modes["swapped"]) == (3, 2, 0) and modes["flush_chunks"] > 0, modes   # (130 columns: 3 + 2 blocks; 9 row tiles x 1 full column tile)
    # 192 = 2 x 96 columns: nothing behind the full tiles; 60 columns: one tile of two blocks
    info, modes = check((("m", "k"), ("k", "n"), ("n", "m")), crandn(rng, (1200, 11)), crandn(rng, (11, 192)), n_cu=1)
    assert (modes["nb"], modes["tail_nb"]) == (3, 0), modes
    info, modes = check((("m", "k"), ("k", "n"), ("n", "m")), crandn(rng, (1200, 11)), crandn(rng, (11, 60)), n_cu=1)
    assert (modes["nb"], modes["tail_nb"]) == (2, 0), modes


def test_row_streaming_form_for_small_blocks_on_many_rows():
    """Round 6 (artn_k_xrow, ArtnXGemmPlan::rowmode): at most 48 contracted values into at most 48 columns on 2^15+ rows, lanes
    along rows in the first operand and the result, no batch label -- the small operand in registers, rows straight into the
    MFMA operand registers, 16-row blocks (rowmode 1) or 64-row superblocks with a lane per row and register butterflies (rowmode 2,
    artn_k_xrow64: up to 32 x 32) dealt round-robin to the waves, three levels of row-offset tables, buffer loads and stores.  The 9 x 9 and 27 x 27 blocks of the bond-dimension-3 network (rows past the end, an odd count of contracted
    values), all three table levels in use, a mixed-extent step; declined: a batch label, a result whose fastest label is a
    column, few rows, more than 4 096 values left for the third level."""
    rng = np.random.default_rng(70)
    # 9 contracted values, 9 columns, 3^10 - style rows (label order as the network's: row labels fastest in A and C)
    eq = (("k1", "m2", "k0", "m1", "m0"), ("n0", "k1", "k0", "n1"), ("n1", "n0", "m2", "m1", "m0"))
    a, b = crandn(rng, (3, 150, 3, 27, 9)), crandn(rng, (3, 3, 3, 3))
    info, modes = check(eq, a, b)
    assert modes["rowmode"] == 2 and info["n_tiles"] == -(-150 * 27 * 9 // 64), (modes, info["n_tiles"])   # 64-row superblocks: a lane per row
    info, modes = check(eq, a, b, rows16=True)
    assert modes["rowmode"] == 1 and info["n_tiles"] == -(-150 * 27 * 9 // 16), (modes, info["n_tiles"])   # the 16-row shape of the same step
    # eleven row labels of extent 3 (243 x 243 x 3 rows: every level of the tables carries), 9 -> 9 with the contracted labels inside
    eq = (tuple("abcdKefgLhijk"), ("x", "K", "L", "y"), ("y", "x") + tuple("abcdefghijk"))
    info, modes = check(eq, crandn(rng, (3,) * 13), crandn(rng, (3,) * 4))
    assert modes["rowmode"] == 2 and info["lds_bytes"] == 4096 + 8 * 3, (modes, info)
    info, modes = check(eq, crandn(rng, (3,) * 13), crandn(rng, (3,) * 4), rows16=True)
    assert modes["rowmode"] == 1, modes
    # 27 -> 27: 7 MFMA steps of four contracted values in registers (the last one holds three), two column blocks, 36 000 rows
    eq = (("k0", "m1", "m0"), ("k0", "n0"), ("n0", "m1", "m0"))
    a, b = crandn(rng, (27, 1125, 32)), crandn(rng, (27, 27))
    info, modes = check(eq, a, b)
    assert modes["rowmode"] == 2, modes
    info, modes = check(eq, a, b, rows16=True)
    assert modes["rowmode"] == 1, modes
    # mixed extents, 5 x 6 = 30 contracted values, 7 columns, a row count that is not a multiple of 32
    eq = (("k1", "m1", "k0", "m0"), ("k0", "n0", "k1"), ("n0", "m1", "m0"))
    a, b = crandn(rng, (5, 4001, 6, 9)), crandn(rng, (6, 7, 5))
    info, modes = check(eq, a, b)
    assert modes["rowmode"] == 2, modes
    info, modes = check(eq, a, b, rows16=True)
    assert modes["rowmode"] == 1, modes
    # NOT taken: the result's fastest label is a column (lanes of a store would not run along rows) ...
    eq = (("k0", "m1", "m0"), ("k0", "n0"), ("m1", "m0", "n0"))
    info, modes = check(eq, crandn(rng, (9, 1200, 30)), crandn(rng, (9, 9)))
    assert modes["rowmode"] == 0, modes
    # ... a batch label; too few rows; more than 48 columns
    info, modes = check((("h", "k0", "m1", "m0"), ("h", "k0", "n0"), ("h", "n0", "m1", "m0")), crandn(rng, (2, 9, 1200, 30)), crandn(rng, (2, 9, 9)))
    assert modes["rowmode"] == 0, modes
    info, modes = check((("k0", "m1", "m0"), ("k0", "n0"), ("n0", "m1", "m0")), crandn(rng, (9, 100, 30)), crandn(rng, (9, 9)))
    assert modes["rowmode"] == 0, modes
    info, modes = check((("k0", "m1", "m0"), ("k0", "n0"), ("n0", "m1", "m0")), crandn(rng, (9, 1200, 30)), crandn(rng, (9, 49)))
    assert modes["rowmode"] == 0, modes
    # 36 -> 36 (bond dimension 6: two labels each): nine MFMA steps, three column blocks; 48 contracted values into 40 columns
    eq = (("k1", "m1", "k0", "m0"), ("n1", "k0", "n0", "k1"), ("n1", "n0", "m1", "m0"))
    info, modes = check(eq, crandn(rng, (6, 1100, 6, 31)), crandn(rng, (6, 6, 6, 6)))
    assert modes["rowmode"] == 1, modes
    info, modes = check((("m1", "m0", "k0"), ("k0", "n0"), ("n0", "m1", "m0")), crandn(rng, (1100, 31, 48)), crandn(rng, (48, 40)))
    assert modes["rowmode"] == 1, modes
    # ... a row index that leaves more than 4 096 values to the third level of the offset tables (2 x 5 000 above one level of 7)
    info, modes = check((("k0", "m2", "m1", "m0"), ("k0", "n0"), ("n0", "m2", "m1", "m0")), crandn(rng, (3, 2, 5000, 7)), crandn(rng, (3, 3)))
    assert modes["rowmode"] == 0, modes


def test_long_contraction_flushes_partial_sums():
    """More than 4096 contracted values: the partial sums go through C (read-add-write); one group is not a multiple of 16."""
    rng = np.random.default_rng(7)
    eq = (("m0", "k0", "k1", "k2", "m1"), ("k2", "n0", "k0", "k1"), ("m0", "n0", "m1"))
    a, b = crandn(rng, (5, 3, 81, 21, 7)), crandn(rng, (21, 6, 3, 81))   # K = 5103: levels of 243 x 21 groups
    info, modes = check(eq, a, b, tol=5e-6)
    assert modes["flush_chunks"] == 256


def test_outer_product_and_single_contracted_value():
    rng = np.random.default_rng(8)
    check((("a", "b"), ("c", "d"), ("a", "c", "b", "d")), crandn(rng, (9, 27)), crandn(rng, (5, 7)))       # nothing contracted
    check((("a", "k", "b"), ("k", "c"), ("c", "a", "b")), crandn(rng, (30, 1, 11)), crandn(rng, (1, 13)))  # extent-1 label


def test_strided_views_as_operands():
    """A and B may be arbitrary non-overlapping strided views (include/artn.h): a slice of a bigger tensor."""
    rng = np.random.default_rng(9)
    big_a, big_b = crandn(rng, (7, 9, 5, 6)), crandn(rng, (6, 4, 9))
    a, b = big_a[1:6, :, ::2, :], big_b[:, 1:4, :]
    check((("m", "k", "p", "q"), ("q", "n", "k"), ("p", "n", "m")), a, b)


def test_levels_beyond_the_tables():
    """More rows than two level tables hold (3^12 > 256^2): the outer labels are decoded per row."""
    rng = np.random.default_rng(10)
    la = tuple(f"m{i}" for i in range(12)) + ("k",)
    eq = (la, ("k", "n"), tuple(reversed(la[:12])) + ("n",))
    check(eq, crandn(rng, (3,) * 12 + (2,)), crandn(rng, (2, 3)))


def test_planner_sends_non_power_of_two_steps_to_the_extent_gemm():
    # D = 3: rank 12 x rank 6 over 3 bonds (a benchmark-size step would be rank 17 x rank 9)
    info = step_info("ABCDEFGHIJKL,DHKxyz->ABCEFGIJLxyz", (3,) * 12, (3,) * 6)
    assert info["kernel"] == KERNEL_XGEMM, info
    assert info["arith"] == 1 and abs(info["mfma_flops"] - 0.75 * info["flops"]) < 1
    # D = 6 = 2 x 3
    assert step_info("ABCDEFG,CFxy->ABDEGxy", (6,) * 7, (6,) * 4)["kernel"] == KERNEL_XGEMM
    # powers of two stay where they were; tiny steps stay on the strided kernel
    assert step_info("ABCDEFGHIJKLMNOPQRST,DHKOwxyz->ABCEFGIJLMNPQRSTwxyz", (2,) * 20, (2,) * 8)["kernel"] == 1
    assert step_info("ABC,Cxy->ABxy", (3,) * 3, (3,) * 3)["kernel"] == 0
    # tensors of 2^31 .. 2^32 elements (3^20 = 2^31.7: one SA level above rand_D3_nv112) stay on the matrix cores since round 6
    # -- element offsets are unsigned 32-bit in the kernel; 3^21 elements (2^33.3) are declined with a note
    la = tuple(range(20))
    lo = tuple(20 if x == 3 else (21 if x == 11 else x) for x in la)
    big = step_info((la, (3, 11, 20, 21), lo), (3,) * 20, (3,) * 4)
    # (9 contracted values into 9 columns on 3^18 rows; NOT the row-streaming form, whose buffer offsets are 32-bit BYTES)
    assert big["kernel"] == KERNEL_XGEMM and big["n_tiles"] == -(-3 ** 18 // 128), big
    la = tuple(range(18))
    lo = (18, 19) + tuple(x for x in la if x not in (3, 11))
    rows = step_info((la, (3, 11, 18, 19), lo), (3,) * 18, (3,) * 4)   # the benchmark network's own size: 3.1 GB tensors
    assert rows["kernel"] == KERNEL_XGEMM and rows["n_tiles"] == -(-3 ** 16 // 64) and rows["m_tile_bits"] == 6, rows   # (64-row superblocks)
    la = tuple(range(21))
    lo = tuple(30 if x == 3 else (31 if x == 11 else x) for x in la)
    huge = step_info((la, (3, 11, 30, 31), lo), (3,) * 21, (3,) * 4)
    assert huge["kernel"] == 0 and "2^32 elements" in huge["note"], huge


def test_default_plan_of_the_emulator_entry_point():
    """artn_emulate (the default planner) routes a D = 3 step through run_xgemm."""
    rng = np.random.default_rng(11)
    a, b = crandn(rng, (3,) * 9), crandn(rng, (3,) * 5)
    eq = "ABCDEFGHI,CFHxy->yABDEGIx"
    got, used = emulate(eq, a, b)
    assert used == KERNEL_XGEMM
    want = oracle.einsum_pair(eq, a, b)
    assert np.abs(got - want).max() / np.abs(want).max() < 2e-6


def test_own_layouts_of_intermediates_keep_the_result():
    """_own_layouts reorders the labels of intermediates of a bond-dimension-3 scheme (the free labels of the bigger operand
    fastest, in that operand's order); the rewritten scheme, run by the oracle, gives the scheme's own result, and the
    last step keeps the scheme's label order."""
    import os
    import torch
    from artensor_amd import contraction as C
    from artensor_amd.fixtures import load_case
    from helpers import GOLDEN
    case = load_case(os.path.join(GOLDEN, "rand_D3_open.npz"))
    shapes = {i: tuple(t.shape) for i, t in case.tensors.items()}
    main = list(range(len(case.scheme)))
    new = C._own_layouts(case.scheme, main, shapes, torch.complex64)
    assert new is not case.scheme and len(new) == len(case.scheme)
    assert any(tuple(C._labels(a[1])[2]) != tuple(C._labels(b[1])[2]) for a, b in zip(new, case.scheme))
    assert tuple(C._labels(new[-1][1])[2]) == tuple(C._labels(case.scheme[-1][1])[2])
    want = oracle.tensor_contraction({i: t.numpy().copy() for i, t in case.tensors.items()}, case.scheme)
    got = oracle.tensor_contraction({i: t.numpy().copy() for i, t in case.tensors.items()}, new)
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max()
    # a tail of the scheme only (the steps a small-step program did not take), and power-of-two schemes are left alone
    tail = C._own_layouts(case.scheme, main[3:], shapes_after(case, 3), torch.complex64)
    got = oracle.tensor_contraction(run_prefix(case, 3), tail[3:])
    assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max()
    case2 = load_case(os.path.join(GOLDEN, "rand_D2_closed.npz"))
    assert C._own_layouts(case2.scheme, list(range(len(case2.scheme))), {i: tuple(t.shape) for i, t in case2.tensors.items()},
                          torch.complex64) is case2.scheme


def run_prefix(case, n):
    tensors = {i: t.numpy().copy() for i, t in case.tensors.items()}
    oracle.tensor_contraction(tensors, case.scheme[:n])
    return tensors


def shapes_after(case, n):
    return {i: tuple(t.shape) for i, t in run_prefix(case, n).items()}


def crandn128(rng, shape):
    return rng.standard_normal(shape) + 1j * rng.standard_normal(shape)


@pytest.mark.parametrize("seed", range(10))
def test_complex128_steps_on_the_f64_matrix_cores(seed):
    """artn_k_xgemm128 (round 5): the same plan with 16-byte elements, chunks of 8 and the lane map of v_mfma_f64_16x16x4_f64,
    replayed by run_xgemm128 -- bond dimension 3 and mixed extents, batch labels, both copy modes, partial tiles and chunks --
    against numpy's complex128 einsum to 1e-13."""
    rng = np.random.default_rng(5000 + seed)
    if seed < 5:
        n_m, n_n, n_k = int(rng.integers(3, 8)), int(rng.integers(1, 5)), int(rng.integers(0, 5))
        eq, sa, sb = random_step(rng, [3], n_m, n_n, n_k)
    else:
        n_m, n_n, n_k, n_h = int(rng.integers(2, 6)), int(rng.integers(1, 4)), int(rng.integers(1, 4)), int(rng.integers(0, 2))
        eq, sa, sb = random_step(rng, [2, 3, 5, 6, 7], n_m, n_n, n_k, n_h)
    a, b = crandn128(rng, sa), crandn128(rng, sb)
    info, modes = check(eq, a, b, tol=1e-13)
    assert modes["kc"] == 8 and modes["nb"] == 1 and info["arith"] in (0, 3)   # (arith is filled in by make_plan: 3 there)


def test_complex128_long_contraction_with_partial_sums():
    """5 103 contracted values in ONE pass: f64 accumulators take no periodic read-add-write of C (the flush interval
    bounds fp32 rounding growth; since round 6 complex128 plans carry flush_chunks = 0)."""
    rng = np.random.default_rng(77)
    eq = (("m0", "k0", "k1", "k2"), ("k2", "n0", "k1", "k0"), ("n0", "m0"))
    a, b = crandn128(rng, (5, 3, 243, 7)), crandn128(rng, (7, 4, 243, 3))
    info, modes = check(eq, a, b, tol=1e-13)
    assert modes["flush_chunks"] == 0


def test_planner_gives_complex128_odd_extents_to_the_extent_gemm():
    import torch
    info = step_info("abcdefghijklmn,nmx->abcdefghijklx", (3,) * 14, (3, 3, 3), dtype=torch.complex128)
    assert info["kernel"] == KERNEL_XGEMM and info["arith"] == 3 and info["mfma_flops"] == info["flops"]
