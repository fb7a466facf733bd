#!/usr/bin/env python3
"""Which consecutive big steps of a slice fuse, and what the planner says about the ones that do not (and about the
alternative pairing one step later): python3 tools/pair_notes.py n53_m14_sliced.npz"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import artensor_amd as A
from artensor_amd import contraction as C, _native as N
from artensor_amd.fixtures import load_case
case = load_case(os.path.join(ROOT, "tests", "golden", sys.argv[1]))
sparse = "bitstrings_sorted" in case.meta
leaves = case.fresh_tensors(device="cuda")
nb = len(case.slicing_indices or {})
log = []
orig_c2, orig_c = C.contract2, C.contract
def c2(eq1, a, b1, eq2, b2, mid_view=None):
    r = orig_c2(eq1, a, b1, eq2, b2, mid_view)
    note = N.lib().artn_last_plan_note().decode()
    if a.numel() >= 1 << 22:
        log.append(("pair", eq1, tuple(a.shape), tuple(b1.shape), eq2, tuple(b2.shape), r is not None, note))
    return r
prev = [None]
def c1(eq, a, b, out=None):
    big = hasattr(a, "numel") and a.numel() >= 1 << 22
    alt = ""
    if big and prev[0] is not None and len(C._labels(eq)[0]) == len(prev[0][3]) and tuple(a.shape) == prev[0][3]:
        try:
            pi = C.pair_info(prev[0][0], prev[0][1], prev[0][2], eq, tuple(b.shape))
            alt = f"with the step before: {'kernel %d k=%d+%d tiles=%d T=%d/%d' % (pi['kernel'], pi['k_bits'], pi['k2_bits'], pi['n_tiles'], pi['tile_in_bits'], pi['tile_out_bits']) if pi else 'declined: ' + N.lib().artn_last_plan_note().decode()}"
        except Exception as e:
            alt = f"pair query failed: {e}"
    r = orig_c(eq, a, b, out)
    if big:
        log.append(("one", eq, tuple(a.shape), tuple(b.shape), alt))
        prev[0] = (eq, tuple(a.shape), tuple(b.shape), tuple(r.shape))
    return r
C.contract2, C.contract = c2, c1
sl = A.apply_slice(leaves, case.slicing_indices, A.slice_assignments(nb, 0)) if nb else dict(leaves)
(A.tensor_contraction_sparse if sparse else A.tensor_contraction)(sl, case.scheme)
torch.cuda.synchronize()
lg = lambda sh: sum((e - 1).bit_length() for e in sh)
for e in log:
    if e[0] == "pair":
        print(f"PAIR fused={e[6]} A 2^{lg(e[2])} B1 2^{lg(e[3])} B2 2^{lg(e[5])}  {e[1]} | {e[4]}   note: {e[7]}")
    else:
        la, lb, lo = C._labels(e[1])
        k = len(set(la) & set(lb) - set(lo))
        print(f"  one A 2^{lg(e[2])} B 2^{lg(e[3])} k={k} out 2^{len(lo)}  {str(e[1])[:60]}  {e[4]}")
