"""The chain planner of the sparse executor (contraction._plan_chain / _cut_sparse_chain, host only): which consecutive
steps on the same big tensor run as one fused pass.  Reference loop: contraction.py:140-191 (one einsum per step)."""
import string

import pytest
import torch

import artensor_amd as A
from artensor_amd import contraction as C

LETTERS = string.ascii_lowercase + string.ascii_uppercase


def _chain(n_state, steps):
    """A scheme of plain steps on tensor 0 (n_state labels of extent 2): steps = [(contracted, new)], each with its own
    second operand.  Returns (scheme, b_shapes)."""
    state = list(LETTERS[:n_state])
    scheme, b_shapes = [], []
    for q, (k, new) in enumerate(steps):
        gone = state[len(state) - k:]              # the fastest labels are contracted
        add = [x for x in LETTERS if x not in state][:new]   # (labels contracted by an earlier step are free again)
        lb = gone + add
        out = state[:len(state) - k] + add
        scheme.append(((0, q + 1), "".join(state) + "," + "".join(lb) + "->" + "".join(out), ([], [])))
        b_shapes.append((2,) * len(lb))
        state = out
    return scheme, b_shapes


def test_plain_form_of_the_four_branches():
    eq = "ab,ac->abc"
    # (D) plain step
    assert C._plain_form(((0, 1), "ab,bc->ac", ([], [])), (4, 2), (2, 8), 8)[:4] == (None, None, None, None)
    # (A) chunk loop: never part of a pair
    assert C._plain_form(((0, 1), eq, ([torch.arange(2), torch.arange(2)], [torch.arange(2), torch.arange(2)]), None, None), (4, 2), (4, 8), 8) is None
    # (B) identity select on both operands: the whole tensors
    ident = ((0, 1), eq, ([torch.arange(4)], [torch.arange(4)]), None, None)
    f = C._plain_form(ident, (4, 2), (4, 8), 8)
    assert f[:2] == (None, None) and f[6] == (4, 2, 8)
    # (B) one row of each: views (rows of 16+ bytes only)
    rows = ((0, 1), eq, ([torch.tensor([3])], [torch.tensor([1])]), None, None)
    f = C._plain_form(rows, (4, 2), (4, 8), 8)
    assert f[:2] == (3, 1) and f[4] == (1, 2) and f[5] == (1, 8) and f[6] == (1, 2, 8)
    assert C._plain_form(rows, (4, 1), (4, 8), 8) is None          # an 8-byte row is not a 16-byte aligned view
    # (B) a real gather
    assert C._plain_form(((0, 1), eq, ([torch.tensor([3, 0])], [torch.tensor([1, 1])]), None, None), (4, 2), (4, 8), 8) is None
    # (C) reshape and row select
    sel = torch.tensor([0, 2, 5])
    f = C._plain_form(((0, 1), "ab,cd->acbd", ([sel], []), (-1, 2, 8), None), (4, 2), (2, 8), 8)
    assert f[2] == (-1, 2, 8) and f[3] is sel and f[6] == (3, 2, 8) and f[7] == 8


def test_a_declined_pair_does_not_cost_the_next_one():
    """(k = 8 GEMM step, k = 3) is declined by the planner; pairs from the left then left BOTH steps single and tried the
    third step with the fourth.  The chain cut pairs the second step with the third."""
    scheme, b_shapes = _chain(26, [(8, 8), (3, 3), (3, 3), (8, 8)])
    assert C.pair_info(scheme[0][1], (2,) * 26, b_shapes[0], scheme[1][1], b_shapes[1]) is None
    assert C.pair_info(scheme[1][1], (2,) * 26, b_shapes[1], scheme[2][1], b_shapes[2]) is not None
    assert C.fusion_schedule(scheme) == [("pair", 0, 1), ("pair", 2, 3)]
    groups = C._cut_sparse_chain(scheme, [0, 1, 2, 3], (2,) * 26, b_shapes, torch.complex64)
    assert groups == [(0,), (1, 2), (3,)]


def test_cut_covers_the_chain_in_order_and_only_forms_pairs_the_planner_accepts():
    steps = [(3, 4), (4, 3), (5, 5), (3, 3), (4, 4), (2, 3), (6, 4), (5, 2), (3, 4)]
    scheme, b_shapes = _chain(26, steps)
    groups = C._cut_sparse_chain(scheme, list(range(len(steps))), (2,) * 26, b_shapes, torch.complex64)
    assert [n for g in groups for n in g] == list(range(len(steps)))
    n_state = 26
    sizes = [n_state]
    for k, new in steps:
        sizes.append(sizes[-1] - k + new)
    for g in groups:
        assert len(g) in (1, 2)
        if len(g) == 2:
            n = g[0]
            assert C.pair_info(scheme[n][1], (2,) * sizes[n], b_shapes[n], scheme[n + 1][1], b_shapes[n + 1]) is not None
    assert any(len(g) == 2 for g in groups)


def test_planning_stops_at_a_step_that_is_no_plain_contraction():
    scheme, b_shapes = _chain(24, [(3, 3), (3, 3), (3, 3)])
    # the second step becomes a chunk loop (two index chunks): it ends the plannable head of the chain
    i2 = ([torch.arange(1), torch.arange(1)], [torch.arange(1), torch.arange(1)])
    scheme[1] = (scheme[1][0], scheme[1][1], i2, None, None)
    groups = C._cut_sparse_chain(scheme, [0, 1, 2], (2,) * 24, b_shapes, torch.complex64)
    assert groups == [(0,)]
    assert C._cut_sparse_chain(scheme, [1, 2], (2,) * 24, b_shapes[1:], torch.complex64) == [(1,)]


def _reads(scheme, order):
    """per step: the versions of its two operands it reads when the steps run in `order` (version = number of writes so far)"""
    version, seen = {}, {}
    for n in order:
        i, j = scheme[n][0]
        seen[n] = (version.get(i, 0), version.get(j, 0))
        version[i] = version.get(i, 0) + 1
        version[j] = version.get(j, 0) + 1     # consumed: `tensors[j] = []` (reference contraction.py:72, :157, :189)
    return seen


@pytest.mark.parametrize("name", ["n53_m14_sliced", "n53_m20_sliced", "n30_sparse10000", "n12_sparse_sliced", "n30_dense"])
def test_chain_schedule_is_a_reordering_that_changes_no_operand(name):
    import os
    from artensor_amd.fixtures import load_case
    case = load_case(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    scheme = case.scheme
    sched = C.chain_schedule(scheme)
    order = [n for e in sched for n in ([e[1]] if e[0] == "one" else e[1])]
    assert sorted(order) == list(range(len(scheme)))
    assert _reads(scheme, order) == _reads(scheme, range(len(scheme)))
    # chains: same first operand, increasing step numbers; the circuit schemes have long ones (the state tensor)
    chains = [e[1] for e in sched if e[0] == "chain"]
    for members in chains:
        assert members == sorted(members) and len({scheme[n][0][0] for n in members}) == 1
    assert max(len(m) for m in chains) >= 12


def test_stage1_reruns_counts_only_the_outer_labels_of_the_second_step():
    """ArtnStepInfo.stage1_reruns (ABI 7): result labels that only the SECOND step brings and that do not fit the tile
    repeat the first stage per value -- the chain planner prices that; outer labels of the FIRST step (a growth step reads its
    small input once per value: a_rereads) repeat nothing."""
    def info(steps):
        scheme, b = _chain(26, steps)
        return C.pair_info(scheme[0][1], (2,) * 26, b[0], scheme[1][1], b[1])
    plain = info([(3, 3), (3, 3)])
    assert plain["stage1_reruns"] == 1 and plain["a_rereads"] == 1
    two = info([(3, 3), (3, 5)])       # five new labels in the second step, four in the tile
    assert two["n2_tile_bits"] == 4 and two["stage1_reruns"] == 2 and two["a_rereads"] == 2
    four = info([(4, 4), (4, 6)])
    assert four["stage1_reruns"] == 4
    grow = info([(3, 7), (3, 3)])      # seven new labels in the FIRST step: re-reads of A, no repeated stage
    assert grow["a_rereads"] == 8 and grow["stage1_reruns"] == 1


def test_identity_memo_holds_its_objects_and_sees_edited_scheme_lists():
    """contraction._IdMemo: the one place where the host-side memos decide whether an entry is still the caller's --
    identity of the key objects (held by the entry, so an id cannot be recycled) and, for scheme lists, the same step
    objects in the same order (the reference re-reads the list on every call, contraction.py:66)."""
    memo = C._IdMemo(8)
    a, b = [1, 2], [1, 2]                       # equal, not identical
    assert memo.find((a,)) is C._MISS
    assert memo.keep((a,), "A") == "A" and memo.find((a,)) == "A" and memo.find((b,)) is C._MISS
    assert memo.find((a,), extra=("cuda:0", 4)) is C._MISS
    memo.keep((a, b), "AB", extra=(7,))
    assert memo.find((a, b), (7,)) == "AB" and memo.find((b, a), (7,)) is C._MISS
    scheme = [((0, 1), "ab,bc->ac"), ((0, 2), "ac,cd->ad")]
    memo.keep((scheme,), "plan", schemes=(scheme,))
    assert memo.find((scheme,), schemes=(scheme,)) == "plan"
    scheme.append(((0, 3), "ad,de->ae"))        # edited in place: same list object, other steps
    assert memo.find((scheme,), schemes=(scheme,)) is C._MISS
    scheme.pop()
    assert memo.find((scheme,), schemes=(scheme,)) == "plan"
    scheme[0] = tuple([(0, 1), "ab,bc->ac"])    # an equal but different step object (built at run time): treated as an edit
    assert memo.find((scheme,), schemes=(scheme,)) is C._MISS
    for q in range(20):                          # bounded: forgets everything rather than grow
        memo.keep(([q],), q)
    assert len(memo) <= 8
