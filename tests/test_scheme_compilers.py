"""Scheme compilers (artensor_amd.contraction_scheme / contraction_scheme_sparse) against the
schemes the reference compiled from the same trees (tests/golden/trees.json, produced by
tests/golden/make_golden.py::case_trees).  The planner is not part of this package, so the
tree is rebuilt here as a plain test double exposing exactly what the compilers consume
(reference contraction_tree.py:10-50, :305-314, :334-357)."""
import json
import os

import numpy as np
import pytest

import artensor_amd as A

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


class Vertex:
    def __init__(self, rec, verts):
        self.contain_tensors = frozenset(rec["contain_tensors"])
        self.contain_bonds = list(rec["contain_bonds"])  # iteration order the reference saw
        self.sc = rec["sc"]
        self.left = Vertex(verts[rec["left"]], verts) if rec["left"] else None
        self.right = Vertex(verts[rec["right"]], verts) if rec["right"] else None
        self.rep_tensor = -1

    def is_leaf(self):
        return not (self.left and self.right)


class TN:
    def __init__(self, rec):
        self.tensor_bonds = {int(k): list(v) for k, v in rec["tensor_bonds"].items()}
        self.final_qubits = list(rec["final_qubits"])


class Tree:
    """Test double of the reference ContractionTree: the attributes and the two traversal
    helpers the scheme compilers call."""

    def __init__(self, rec):
        self.tn = TN(rec)
        root = Vertex(rec["vertices"][rec["root"]], rec["vertices"])
        self.all_tensors = root.contain_tensors
        self.tree = {self.all_tensors: root}

    def _post_order(self):
        out, stack = [], [self.tree[self.all_tensors]]
        while stack:
            v = stack.pop()
            out.append(v)
            if not v.is_leaf():
                stack += [v.left, v.right]
        return reversed(out)

    def mark_rep_tensor(self):  # contraction_tree.py:305-314
        for v in self._post_order():
            if v.is_leaf():
                v.rep_tensor = min(v.contain_tensors)
            else:
                v.rep_tensor = v.left.rep_tensor if v.left.sc > v.right.sc else v.right.rep_tensor

    def tree_order_dfs(self):  # contraction_tree.py:334-357
        self.mark_rep_tensor()
        order, stack = [], [self.tree[self.all_tensors]]
        while stack:
            v = stack.pop()
            if v.is_leaf():
                continue
            if v.rep_tensor == v.left.rep_tensor:
                order.append((v.left.rep_tensor, v.right.rep_tensor))
            else:
                order.append((v.right.rep_tensor, v.left.rep_tensor))
            stack += [v.left, v.right] if v.left.sc > v.right.sc else [v.right, v.left]
        order.reverse()
        return order


def canon(eq):
    """Equations compare up to a renaming of letters (the letter assignment follows set
    iteration order, i.e. PYTHONHASHSEED, in the reference and here alike)."""
    m = {}
    return "".join(c if c in ",->" else m.setdefault(c, chr(65 + len(m))) for c in eq)


@pytest.fixture(scope="module")
def trees():
    with open(os.path.join(GOLDEN, "trees.json")) as f:
        return json.load(f)


def test_dense_scheme_matches_reference(trees):
    rec = trees["n12_dense"]
    scheme, output_bonds = A.contraction_scheme(Tree(rec["tree"]))
    assert [list(e) for e, _ in scheme] == [e for e, _ in rec["scheme"]]
    assert [canon(eq) for _, eq in scheme] == [canon(eq) for _, eq in rec["scheme"]]
    assert list(output_bonds) == rec["output_bonds"]


@pytest.mark.parametrize("name", ["n12_sparse", "n12_sparse_chunked", "n12_sparse_chunked6"])
def test_sparse_scheme_matches_reference(trees, name):
    rec = trees[name]
    scheme, bonds, sorted_bits = A.contraction_scheme_sparse(Tree(rec["tree"]), rec["bitstrings"],
                                                             sc_target=rec["sc_target"])
    assert len(scheme) == len(rec["scheme"])
    assert list(bonds) == rec["bonds"]
    assert list(sorted_bits) == rec["bitstrings_sorted"]
    for got, want in zip(scheme, rec["scheme"]):
        assert list(got[0]) == want["edge"]
        assert canon(got[1]) == canon(want["eq"])
        for side in (0, 1):
            assert len(got[2][side]) == len(want["batch"][side])
            for g, w in zip(got[2][side], want["batch"][side]):
                assert np.asarray(g).tolist() == w
        assert (len(got) == 5) == ("next_shape" in want)
        if len(got) == 5:
            assert (None if got[3] is None else list(got[3])) == want["rshape"]
            assert list(got[4]) == want["next_shape"]


def test_compiled_scheme_runs_through_the_oracle(trees):
    """The scheme this package compiles, executed by the oracle on the fixture's leaf
    tensors, reproduces the reference's amplitudes."""
    from artensor_amd.fixtures import load_case
    from oracle import oracle
    case = load_case(os.path.join(GOLDEN, "n12_dense.npz"))
    scheme, _ = A.contraction_scheme(Tree(trees["n12_dense"]["tree"]))
    raw = oracle.tensor_contraction({i: t.numpy().copy() for i, t in case.tensors.items()}, scheme)
    want = case.arrays["raw"]
    assert np.abs(raw - want).max() / np.abs(want).max() < 5e-6


def test_label_tuple_schemes_equal_the_string_schemes(trees):
    """labels="tuples": the same steps with the bond labels themselves instead of einsum letters (no
    50-symbol limit): turning each tuple equation back into a string gives the string scheme."""
    rec = trees["n12_dense"]
    s_str, out_str = A.contraction_scheme(Tree(rec["tree"]))
    s_tup, out_tup = A.contraction_scheme(Tree(rec["tree"]), labels="tuples")
    assert list(out_str) == list(out_tup) and len(s_str) == len(s_tup)
    for (e1, eq), (e2, tup) in zip(s_str, s_tup):
        assert e1 == e2 and isinstance(tup, tuple) and len(tup) == 3
        assert canon(A.einsum_eq_convert((list(tup[0]), list(tup[1])), list(tup[2]))) == canon(eq)
    for name in ("n12_sparse", "n12_sparse_chunked"):
        rec = trees[name]
        a = A.contraction_scheme_sparse(Tree(rec["tree"]), rec["bitstrings"], sc_target=rec["sc_target"])[0]
        b = A.contraction_scheme_sparse(Tree(rec["tree"]), rec["bitstrings"], sc_target=rec["sc_target"],
                                        labels="tuples")[0]
        assert len(a) == len(b)
        for x, y in zip(a, b):
            assert x[0] == y[0] and len(x) == len(y)
            assert canon(A.einsum_eq_convert((list(y[1][0]), list(y[1][1])), list(y[1][2]))) == canon(x[1])
    with pytest.raises(RuntimeError):
        A.contraction_scheme(Tree(trees["n12_dense"]["tree"]), labels="letters")
