#!/usr/bin/env python3
"""Brute force over the cuts of the big chain of a sliced fixture (dense or sparse-state): every way to cut N of its members (from index I0) into single steps
and planner-accepted pairs, timed on the GPU (ms per slice) next to the chain planner's own cut -- calibration of
contraction._cut_sparse_chain's cost model.   python3 tools/cut_search.py rand_D2_nv260_sliced.npz [N=9 [I0]]"""
import itertools, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import artensor_amd as A
from artensor_amd import contraction as C
from artensor_amd.fixtures import load_case
case = load_case(os.path.join(ROOT, "tests", "golden", sys.argv[1]))
N_LAST = int(sys.argv[2]) if len(sys.argv) > 2 else 9
leaves = case.fresh_tensors(device="cuda")
orig = C._cut_sparse_chain
seen = {}
def spy(scheme, members, a_shape, b_shapes, dtype):
    g = orig(scheme, members, a_shape, b_shapes, dtype)
    seen[tuple(members)] = (g, a_shape, b_shapes)
    return g
C._cut_sparse_chain = spy
SPARSE = "bitstrings_sorted" in case.meta
ROWS = len(case.meta["bitstrings_sorted"]) if SPARSE else 1
OUT_SHAPE = tuple(int(x) for x in os.environ["OUT_SHAPE"].split(",")) if os.environ.get("OUT_SHAPE") else (ROWS,)   # (open dense fixtures: OUT_SHAPE=2,2,...)
def time_slices(n=6):
    C._chain_cache.clear()
    r = A.SliceRunner(leaves, case.scheme, case.slicing_indices, OUT_SHAPE, sparse=SPARSE, device="cuda")
    total = 2 ** len(case.slicing_indices)
    order = list(A.rank_slices(total, 0, 8 if total >= 64 else 1, gray=True))
    order = (order * (1 + (n + 2) // len(order)))[:n + 2]   # (a fixture of 8 slices: round again)
    r.run(order[:2]); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r.run(order[2:2 + n]); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
base = time_slices()
# the chain whose planned cut holds the most fused pairs: the state tensor's
big = max(seen, key=lambda k: (sum(len(g) == 2 for g in seen[k][0]), len(k)))
groups0 = seen[big][0]
print(f"planner's cut: {base:.3f} ms per slice; chain of {len(big)} members, cut {groups0[-N_LAST:]}")
members = list(big)
I0 = int(sys.argv[3]) if len(sys.argv) > 3 else max(0, len(members) - N_LAST)
print("planner's whole cut:", groups0)
head_groups, tail_groups, covered = [], [], 0
for g in groups0:
    if covered + len(g) <= I0: head_groups.append(g)
    elif covered >= I0 + N_LAST: tail_groups.append(g)
    covered += len(g)
lo = sum(len(g) for g in head_groups)
hi = len(members) - sum(len(g) for g in tail_groups)
tail = members[lo:hi]
def cuts(seq):
    if not seq: yield []; return
    for rest in cuts(seq[1:]): yield [(seq[0],)] + rest
    if len(seq) >= 2:
        for rest in cuts(seq[2:]): yield [(seq[0], seq[1])] + rest
results = []
for cut in cuts(tail):
    forced = head_groups + cut + tail_groups
    def fake(scheme, mem, a_shape, b_shapes, dtype, forced=forced):
        return forced if tuple(mem) == big else orig(scheme, mem, a_shape, b_shapes, dtype)
    C._cut_sparse_chain = fake
    C._plan_cache.clear()
    try:
        t = time_slices(4)
    except Exception as e:
        print('   failed', cut, str(e)[:80]); continue
    results.append((t, cut))
results.sort(key=lambda x: x[0])
for t, cut in results[:8]:
    print(f"  {t:.3f} ms  {cut}")
print(f"  ... worst {results[-1][0]:.3f} ms  {results[-1][1]}  ({len(results)} cuts)")
