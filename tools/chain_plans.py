#!/usr/bin/env python3
"""The chains of one slice of a sparse-state fixture and how _plan_chain cuts them: python3 tools/chain_plans.py n53_m20_sliced.npz"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import artensor_amd as A
from artensor_amd import contraction as C
from artensor_amd.fixtures import load_case
case = load_case(os.path.join(ROOT, "tests", "golden", sys.argv[1]))
leaves = case.fresh_tensors(device="cuda")
nb = len(case.slicing_indices or {})
orig = C._plan_chain
lg = lambda sh: sum((e - 1).bit_length() for e in sh)
def plan(tensors, scheme, members):
    groups = orig(tensors, scheme, members)
    a = tensors[scheme[members[0]][0][0]]
    if hasattr(a, "numel") and len(members) > 3:
        kinds = []
        for n in members:
            st = scheme[n]
            bi, bj = st[2]
            kinds.append(f"{n}:{'A' if len(bi) > 1 else 'D' if len(st) == 3 else 'B' if len(bi) == 1 and len(bj) == 1 else 'C'}")
        print(f"chain on tensor {scheme[members[0]][0][0]} (2^{lg(a.shape)}): members {' '.join(kinds)}  -> groups {groups}")
    return groups
C._plan_chain = plan
sl = A.apply_slice(leaves, case.slicing_indices, A.slice_assignments(nb, 0)) if nb else dict(leaves)
A.tensor_contraction_sparse(sl, case.scheme)
torch.cuda.synchronize()
