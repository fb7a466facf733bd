"""Host logic of the slice loop that needs no GPU: Gray-ordered shards and the split of a scheme
into slice-reusable small steps and the remaining big steps (artensor_amd/simulation.py)."""
import os

import numpy as np
import pytest
import torch

import artensor_amd as A
from artensor_amd import simulation as S
from artensor_amd.fixtures import load_case
from oracle import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.mark.parametrize("n,world", [(16, 1), (16, 4), (1 << 14, 8), (64, 2)])
def test_gray_shard_is_a_permutation_with_single_bit_steps(n, world):
    for rank in range(world):
        plain = list(A.rank_slices(n, rank, world))
        gray = list(A.rank_slices(n, rank, world, gray=True))
        assert sorted(gray) == plain
        for a, b in zip(gray, gray[1:]):
            assert bin(a ^ b).count("1") == 1


def test_gray_shard_uneven_world():
    # 3 ranks over 16 slices: counts 6, 5, 5; still the same sets
    seen = []
    for rank in range(3):
        gray = list(A.rank_slices(16, rank, 3, gray=True))
        assert sorted(gray) == list(A.rank_slices(16, rank, 3))
        seen += gray
    assert sorted(seen) == list(range(16))


def _sliced_shapes(case, leaves):
    sel = {}
    for x, (bond, lst) in enumerate(case.slicing_indices.items()):
        for tid, dim in lst:
            sel.setdefault(tid, {})[dim] = x
    items = leaves.items() if isinstance(leaves, dict) else enumerate(leaves)
    return {k: tuple(e for d, e in enumerate(t.shape) if d not in sel.get(k, {})) for k, t in items}


@pytest.mark.parametrize("name,limit", [("rand_D2_open_sliced", 8), ("rand_D2_closed_sliced", 4),
                                        ("n12_sparse_sliced", 32), ("n12_sparse_sliced", 1 << 16)])
def test_split_scheme_small_then_main_equals_whole(name, limit, monkeypatch):
    """Evaluating the small steps first (what SliceRunner keeps across slices) and handing the
    rest to the executor gives the executor's result on the whole scheme -- checked with the CPU
    oracle as executor on slice 0 of the sliced fixtures."""
    monkeypatch.setattr(S, "SMALL_NUMEL", limit)
    case = load_case(os.path.join(GOLDEN, name + ".npz"))
    sparse = case.meta.get("pattern") == "sparse"
    n_b = len(case.slicing_indices)
    leaves = A.apply_slice(case.fresh_tensors(device="cpu"), case.slicing_indices, A.slice_assignments(n_b, 0))
    items = leaves.items() if isinstance(leaves, dict) else enumerate(leaves)
    np_leaves = {k: t.numpy() for k, t in items}
    small, main, _ = S.split_scheme(case.scheme, {k: v.shape for k, v in np_leaves.items()})
    assert len(small) + len(main) == len(case.scheme)
    assert [s for s in case.scheme if any(s is m for m in main)] == main  # scheme order kept
    run = oracle.tensor_contraction_sparse if sparse else oracle.tensor_contraction
    whole = run(dict(np_leaves), case.scheme)
    cur = dict(np_leaves)
    for n in small:
        step = case.scheme[n]
        i, j = step[0]
        scratch = {i: cur[i], j: cur[j]}
        run(scratch, [step])
        cur[i] = scratch[i]
    got = run(cur, main) if main else cur[case.scheme[-1][0][0]]
    assert np.abs(np.asarray(got) - np.asarray(whole)).max() <= 1e-6 * max(np.abs(whole).max(), 1e-30)
    if limit == 1 << 16:
        assert not main  # n12: every step is small
    else:
        assert small and main


def test_split_scheme_keeps_a_reused_operand_behind_the_main_step_that_reads_it():
    """a main step (X, Y) followed by a small step that overwrites Y.  The executors'
    semantics are sequential (reference contraction.py:66-70), so the small step may not be hoisted in
    front of the main step that still has to read the old Y."""
    rng = np.random.default_rng(5)

    def t(*shape):
        return (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(np.complex64)

    leaves = {0: t(2, 2, 2, 2), 1: t(2, 2), 2: t(2, 2), 3: t(2, 2)}
    scheme = [((0, 1), "abcd,de->abce"),   # main (operand 0 is "big" under the limit below); reads Y = 1
              ((1, 2), "de,ef->df"),       # small, but overwrites Y after the main step read it
              ((0, 1), "abce,ef->abcf"),   # main again, reads the NEW Y
              ((0, 3), "abcf,fa->bc")]
    import artensor_amd.simulation as S2
    old = S2.SMALL_NUMEL
    S2.SMALL_NUMEL = 8
    try:
        small, main, _ = S.split_scheme(scheme, {k: v.shape for k, v in leaves.items()})
    finally:
        S2.SMALL_NUMEL = old
    assert small == []                      # step 1 stays behind step 0
    assert main == scheme
    whole = oracle.tensor_contraction(dict(leaves), scheme)
    cur = dict(leaves)
    for n in small:
        scratch = {scheme[n][0][0]: cur[scheme[n][0][0]], scheme[n][0][1]: cur[scheme[n][0][1]]}
        oracle.tensor_contraction(scratch, [scheme[n]])
        cur[scheme[n][0][0]] = scratch[scheme[n][0][0]]
    got = oracle.tensor_contraction(cur, main)
    assert np.allclose(got, whole)


@pytest.mark.parametrize("name,limit", [("n12_dense", 1 << 14), ("n12_dense", 64), ("rand_D2_closed", 1 << 14), ("rand_D4_closed", 256)])
def test_small_step_program_plan_keeps_sequential_semantics(name, limit, monkeypatch):
    """contraction._plan_small_program hoists the steps that only combine small leaf-derived tensors into a
    one-launch program (groups = independent components).  Running those steps in program order and then
    the remaining steps in scheme order must equal the scheme run step by step (CPU oracle as executor)."""
    from artensor_amd import contraction as C
    monkeypatch.setattr(C, "PROGRAM_MAX_NUMEL", limit)
    case = load_case(os.path.join(GOLDEN, name + ".npz"))
    if case.slicing_indices:
        pytest.skip("unsliced cases only")
    leaves = {k: t.numpy().copy() for k, t in case.tensors.items()}
    shapes = {k: v.shape for k, v in leaves.items()}
    prog, main = C._plan_small_program(case.scheme, shapes, torch.complex64)
    whole = oracle.tensor_contraction(dict(leaves), case.scheme)
    if prog is None:
        assert main == list(range(len(case.scheme)))
        return
    assert prog.n_steps + len(main) == len(case.scheme)
    assert len(prog.ext_ids) == len(set(prog.ext_ids)) <= 256
    groups = prog.host_groups.tolist()
    assert groups[0] == 0 and groups[-1] == prog.n_steps and sorted(groups) == groups
    # replay: the program's steps are the scheme's small steps sorted by (group, index); recover that order
    small = [n for n in range(len(case.scheme)) if n not in set(main)]
    cur = dict(leaves)
    for n in small:   # scheme order is a valid order inside every group
        oracle.tensor_contraction(cur, [case.scheme[n]])
    for t, (off, shape) in prog.outputs.items():   # what the program leaves for the remaining steps
        assert tuple(cur[t].shape) == tuple(shape) and off % 16 == 0
    got = oracle.tensor_contraction(cur, [case.scheme[n] for n in main]) if main else cur[case.scheme[-1][0][0]]
    assert np.abs(np.asarray(got) - np.asarray(whole)).max() <= 1e-6 * max(np.abs(whole).max(), 1e-30)


@pytest.mark.parametrize("name,limit", [("n12_dense", 1 << 14), ("n12_dense", 256), ("n30_dense", 1 << 14), ("rand_D2_closed", 1 << 14),
                                        ("rand_D4_closed", 256), ("rand_D3_open", 1 << 12),
                                        # sparse-state schemes: the plain (branch D) small steps are hoisted the same way
                                        ("n12_sparse5", 1 << 14), ("n30_sparse100", 1 << 14)])
def test_small_step_program_image_emulated(name, limit, monkeypatch):
    """The compiled image itself -- levels, wave tasks, LDS arena with reuse, preloaded leaves, which results
    go to the workspace -- executed on the CPU the way artn_k_program executes it (tests/helpers.emulate_program)
    must leave in the workspace exactly what the scheme's small steps leave for the remaining ones."""
    from artensor_amd import contraction as C
    from helpers import emulate_program
    monkeypatch.setattr(C, "PROGRAM_MAX_NUMEL", limit)
    case = load_case(os.path.join(GOLDEN, name + ".npz"))
    if case.slicing_indices:
        pytest.skip("unsliced cases only")
    rng = np.random.default_rng(5)
    leaves = {k: (rng.standard_normal(tuple(t.shape)) + 1j * rng.standard_normal(tuple(t.shape))).astype(np.complex64) for k, t in case.tensors.items()}
    shapes = {k: v.shape for k, v in leaves.items()}
    prog, main = C._plan_small_program(case.scheme, shapes, torch.complex64)
    if prog is None:
        pytest.skip("no program for this scheme")
    ws, stats = emulate_program(prog, leaves)
    small = [n for n in range(len(case.scheme)) if n not in set(main)]
    cur = dict(leaves)
    for n in small:
        oracle.tensor_contraction(cur, [case.scheme[n]])
    assert prog.outputs
    for t, (off, shape) in prog.outputs.items():
        want = np.asarray(cur[t]).reshape(-1)
        got = ws[off // 8: off // 8 + want.size]
        assert np.abs(got - want).max() <= 2e-6 * max(np.abs(want).max(), 1e-30), (name, t)
    # the point of the image: far fewer barriers than steps, intermediates in LDS
    assert stats["levels"] < prog.n_steps or prog.n_steps < 4
    if name == "n12_dense" and limit == 1 << 14:
        # 19 levels for 68 steps; the 15 steps of the stem (a 2^12-element tensor absorbing one small tensor each,
        # and a few 2^10-element ones before it) are matrix-core steps; only the final result goes to the workspace
        assert stats["levels"] == 19 and stats["fast"] >= 15 and stats["to_ws"] == 1


@pytest.mark.parametrize("name,limit", [("n12_dense", 1 << 14), ("n12_dense", 256), ("n30_dense", 1 << 14), ("rand_D3_open", 1 << 12),
                                        ("n30_sparse100", 1 << 14)])
def test_small_step_program_image_in_complex128_emulated(name, limit, monkeypatch):
    """The complex128 image (artn_k_program<double>): 16-byte elements in the arena and the workspace, no matrix-core steps;
    replayed on the CPU it must leave what numpy's complex128 einsums of the small steps leave, to 1e-13."""
    from artensor_amd import contraction as C
    from helpers import emulate_program
    monkeypatch.setattr(C, "PROGRAM_MAX_NUMEL", limit)
    case = load_case(os.path.join(GOLDEN, name + ".npz"))
    rng = np.random.default_rng(6)
    leaves = {k: rng.standard_normal(tuple(t.shape)) + 1j * rng.standard_normal(tuple(t.shape)) for k, t in case.tensors.items()}
    shapes = {k: v.shape for k, v in leaves.items()}
    prog, main = C._plan_small_program(case.scheme, shapes, torch.complex128)
    assert prog is not None and prog.dtype == torch.complex128
    ws, stats = emulate_program(prog, leaves)
    assert ws.dtype == np.complex128 and stats["fast"] == 0
    small = [n for n in range(len(case.scheme)) if n not in set(main)]
    cur = dict(leaves)
    for n in small:
        oracle.tensor_contraction(cur, [case.scheme[n]])
    assert prog.outputs
    for t, (off, shape) in prog.outputs.items():
        want = np.asarray(cur[t]).reshape(-1)
        assert off % 16 == 0
        got = ws[off // 16: off // 16 + want.size]
        assert np.abs(got - want).max() <= 1e-13 * max(np.abs(want).max(), 1e-300), (name, t)
    assert stats["levels"] < prog.n_steps


@pytest.mark.parametrize("mb,kb,extra", [(12, 2, 3), (9, 1, 3), (11, 3, 20), (12, 2, 60), (8, 3, 4)])
def test_small_step_program_with_fully_contracted_second_operand(mb, kb, extra):
    """A matrix-core step whose second operand is contracted away entirely ('abcdefghijkl,kl->abcdefghij') is cut
    into out/32 wave tasks -- more than artn_program_image_bytes once allowed for (such schemes raised out of
    tensor_contraction).  The image must build and, run by the CPU emulator, give the scheme's result."""
    from artensor_amd import contraction as C
    from helpers import emulate_program
    rng = np.random.default_rng(mb * 100 + kb)
    c = lambda shape: (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(np.complex64)
    la = list(range(mb))
    leaves = {0: c((2,) * mb), 1: c((2,) * kb)}
    out = la[:mb - kb]
    scheme = [((0, 1), (tuple(la), tuple(la[mb - kb:]), tuple(out)))]
    for q in range(extra):   # tiny steps (a 2 x 2 matrix on one label each) so that the scheme is worth a program
        leaves[2 + q] = c((2, 2))
        old, new = out[q % len(out)], 100 + q
        nxt = [new if x == old else x for x in out]
        scheme.append(((0, 2 + q), (tuple(out), (old, new), tuple(nxt))))
        out = nxt
    shapes = {k: v.shape for k, v in leaves.items()}
    prog, main = C._plan_small_program(scheme, shapes, torch.complex64)
    assert prog is not None and main == []
    ws, stats = emulate_program(prog, leaves)
    cur = {k: v.astype(np.complex128) for k, v in leaves.items()}
    for (i, j), (a, b, o) in scheme:   # (label tuples: numpy's sublist einsum takes labels < 52)
        m = {x: n for n, x in enumerate(dict.fromkeys(a + b))}
        cur[i] = np.einsum(cur[i], [m[x] for x in a], cur[j], [m[x] for x in b], [m[x] for x in o])
    want = cur[0].reshape(-1)
    (off, shape), = [v for t, v in prog.outputs.items() if t == 0]
    got = ws[off // 8: off // 8 + want.size]
    assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max()


def test_gray_ordered_shard_is_lazy():
    """A plan with 41 sliced bonds (tests/golden/n53_m20_bigbatch.npz) has 2^41 slices: the Gray-ordered shard of rank_slices is
    a sequence object that costs what is touched -- a list of a rank's 2^38 slices does not fit any host."""
    g = A.rank_slices(2 ** 41, 3, 8, gray=True)
    assert len(g) == 2 ** 38
    assert g[:5] == [3 + 8 * x for x in (0, 1, 3, 2, 6)] and g[7] == 3 + 8 * 4 and g[2 ** 30] == 3 + 8 * (2 ** 30 ^ 2 ** 29)
    it = iter(g)
    assert [next(it) for _ in range(3)] == [3, 11, 27]
    # counts that are not powers of two: the same set as the plain shard, each slice once, and the list API of before
    h = A.rank_slices(16, 1, 3, gray=True)
    assert sorted(h) == list(A.rank_slices(16, 1, 3)) and len(h) == 5 and h[1:3] == list(h)[1:3] and h[-1] == list(h)[-1]
    assert h == list(h)
    with pytest.raises(IndexError):
        h[5]
    # an integer index or a late start never walks the shard from its first slice: a shard of
    # 2^41 - 12345 slices answers positions 10^11 and len - 1 at once, and agrees with the walk where the walk is cheap
    big = A.rank_slices(8 * (2 ** 38 - 12345) + 3, 3, 8, gray=True)
    assert len(big) == 2 ** 38 - 12345
    t0 = __import__("time").time()
    far = [big[10 ** 11], big[len(big) - 1]] + big[10 ** 11:10 ** 11 + 3]
    assert __import__("time").time() - t0 < 1.0 and far[0] == far[2] and len(set(far)) == 4
    for count in (5, 6, 7, 100, 257, 1000, 4097):
        sh = A.rank_slices(count * 3, 1, 3, gray=True)
        walk = list(sh)
        assert len(walk) == count and [sh[i] for i in range(count)] == walk
        assert sh[count // 2:count // 2 + 9] == walk[count // 2:count // 2 + 9]


@pytest.mark.parametrize("n,world", [(1024, 3), (4096, 8), (100, 7), (8, 8), (5, 8)])
def test_gray_ordered_shards_partition_the_slices(n, world):
    """Every slice belongs to exactly one rank's Gray-ordered shard (what the N-rank slice loop sums, reference
    simulation.py:107-114 sharded), and inside a power-of-two shard consecutive slices differ in ONE sliced bond."""
    shards = [A.rank_slices(n, r, world, gray=True) for r in range(world)]
    flat = [s for sh in shards for s in sh]
    assert sorted(flat) == list(range(n)) and sum(len(sh) for sh in shards) == n
    for r, sh in enumerate(shards):
        count = len(sh)
        if count and count & (count - 1) == 0 and world & (world - 1) == 0:
            vals = list(sh)
            for x, y in zip(vals, vals[1:]):
                assert bin(x ^ y).count("1") == 1, (r, x, y)
