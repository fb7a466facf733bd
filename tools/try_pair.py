#!/usr/bin/env python3
"""Force one more fused pair into the chain planner's cut of a sparse-state slice and time the slice:
python3 tools/try_pair.py n53_m14_sliced.npz            -> the planner's cut (members, single-step prices, candidate pairs)
python3 tools/try_pair.py n53_m14_sliced.npz N          -> the same with steps (N, next member) forced into one pair"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import artensor_amd as A
from artensor_amd import contraction as C
from artensor_amd.fixtures import load_case
case = load_case(os.path.join(ROOT, "tests", "golden", sys.argv[1]))
force = int(sys.argv[2]) if len(sys.argv) > 2 else None
leaves = case.fresh_tensors(device="cuda")
rows_n = len(case.meta["bitstrings_sorted"]) if "bitstrings_sorted" in case.meta else 1
orig = C._cut_sparse_chain
def cut(scheme, members, a_shape, b_shapes, dtype):
    groups = orig(scheme, members, a_shape, b_shapes, dtype)
    if force is not None and force in members:
        flat = [n for g in groups for n in g]
        if force in flat and flat.index(force) + 1 < len(flat):
            nxt = flat[flat.index(force) + 1]
            out, skip = [], set()
            for g in groups:
                g = tuple(n for n in g if n not in skip)
                if not g: continue
                if force in g:
                    pre = tuple(n for n in g if n != force)
                    if pre: out.append(pre)
                    out.append((force, nxt)); skip.add(nxt)
                elif nxt in g:
                    rest = tuple(n for n in g if n != nxt)
                    if rest: out.append(rest)
                else:
                    out.append(g)
            groups = out
    return groups
C._cut_sparse_chain = cut
def tr(d):
    if max(d["single_ms"]) > 0.3:
        print("members", d["members"]); print("single ms", [round(x, 2) for x in d["single_ms"]])
        print("pairs ms", {k: round(v, 2) for k, v in d["pair_ms"].items()}); print("cut", d["groups"])
if force is None: C._chain_trace = tr
r = A.SliceRunner(leaves, case.scheme, case.slicing_indices, (rows_n,), sparse="bitstrings_sorted" in case.meta, device="cuda")
order = A.rank_slices(2 ** len(case.slicing_indices), 0, 8, gray=True)
r.run(order[:2]); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); r.run(order[2:8]); e1.record(); torch.cuda.synchronize()
print(f"forced pair at {force}: {e0.elapsed_time(e1) / 6:.3f} ms per slice")
class Prof:
    def __init__(s): s.rows = []
    def record(s, info, a, b): s.rows.append((info, a, b))
p = Prof(); C.profiler = p; r.run(order[8:9]); torch.cuda.synchronize(); C.profiler = None
for info, a, b in p.rows:
    ms = a.elapsed_time(b)
    if ms > 0.4: print(f"   {ms * 1e3:8.1f} us k={info.get('k_bits')}+{info.get('k2_bits')} tiles={info.get('n_tiles')} GF={info['flops'] / 1e9:.0f} GB={info['bytes'] / 1e9:.2f} rereads {info.get('a_rereads')} reruns {info.get('stage1_reruns')}")
