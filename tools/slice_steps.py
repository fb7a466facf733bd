#!/usr/bin/env python3
"""Every launch of one slice of a sliced fixture in execution order, with its time, planner kernel, note and label extents:
python3 tools/slice_steps.py n53_m14_sliced.npz [min_us]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import artensor_amd as A
from artensor_amd import contraction as C
from artensor_amd.fixtures import load_case
case = load_case(os.path.join(ROOT, "tests", "golden", sys.argv[1]))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
leaves = case.fresh_tensors(device="cuda")
rows_n = len(case.meta["bitstrings_sorted"]) if "bitstrings_sorted" in case.meta else 1
sparse = "bitstrings_sorted" in case.meta
orig_q = C._query
desc_of = {}
def q(d):
    info = orig_q(d)
    rows = []
    for i in range(d.n_labels):
        e, sa, sb, sc = d.extent[i], d.stride_a[i], d.stride_b[i], d.stride_c[i]
        rows.append(("".join(c for c, st in zip("ABC", (sa, sb, sc)) if st >= 0), e, sa, sb, sc))
    desc_of[id(info)] = (rows, info)
    return info
C._query = q
class Prof:
    def __init__(s): s.rows = []
    def record(s, info, e0, e1): s.rows.append((info, e0, e1))
r = A.SliceRunner(leaves, case.scheme, case.slicing_indices, (rows_n,), sparse=sparse, device="cuda")
order = A.rank_slices(2 ** len(case.slicing_indices), 0, 8, gray=True)
r.run(order[:3]); torch.cuda.synchronize()
p = Prof(); C.profiler = p; r.run(order[3:4]); torch.cuda.synchronize(); C.profiler = None
tot = 0.0
for info, e0, e1 in p.rows:
    ms = e0.elapsed_time(e1); tot += ms
    if ms * 1e3 < min_us: continue
    d = desc_of.get(id(info))
    cls = {}
    if d:
        for c, e, sa, sb, sc in d[0]: cls.setdefault(c, []).append(e)
    sizes = {c: int(torch.tensor(v).prod()) for c, v in cls.items()}
    print(f"{ms*1e3:8.1f} us kernel={info['kernel']} k={info.get('k_bits')}+{info.get('k2_bits')} tiles={info.get('n_tiles')} GF={info['flops']/1e9:.2f} GB={info['bytes']/1e9:.3f} "
          f"sizes={sizes} note={info.get('note', '')}")
print(f"total {tot:.2f} ms in {len(p.rows)} launches")
