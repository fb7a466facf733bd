#!/usr/bin/env python3
"""Where does a slice of the n53 m14 plan spend its time?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import artensor_amd as A
from artensor_amd import contraction as C
from artensor_amd.fixtures import load_case
case = load_case(os.path.join(ROOT, "tests", "golden", "n53_m14_sliced.npz"))
leaves = case.fresh_tensors(device="cuda")
nb = len(case.slicing_indices)
def one(s):
    sl = A.apply_slice(leaves, case.slicing_indices, A.slice_assignments(nb, s))
    return A.tensor_contraction_sparse(sl, case.scheme)
one(0); torch.cuda.synchronize()
t0 = time.perf_counter()
for s in range(1, 9): one(s)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8
t0 = time.perf_counter()
for s in range(1, 9): A.apply_slice(leaves, case.slicing_indices, A.slice_assignments(nb, s))
torch.cuda.synchronize(); ds = (time.perf_counter() - t0) / 8
class Prof:
    def __init__(s): s.rows = []
    def record(s, info, e0, e1): s.rows.append((info, e0, e1))
p = Prof(); C.profiler = p; one(3); torch.cuda.synchronize(); C.profiler = None
rows = sorted(((e0.elapsed_time(e1), info) for info, e0, e1 in p.rows), key=lambda r: -r[0])
tot = sum(r[0] for r in rows); big = sum(r[0] for r in rows if r[1]["kernel"] in (1, 2))
print(f"per slice wall {dt*1e3:.1f} ms; apply_slice {ds*1e3:.2f} ms; contract launches {len(rows)}: {tot:.1f} ms in kernels ({big:.1f} ms MFMA kernel)")
for ms, info in rows[:int(os.environ.get("TOP", "10"))]:
    print(f"   {ms:6.2f} ms kernel={info['kernel']} k={info['k_bits']}+{info['k2_bits']} T={info['tile_in_bits']}/{info['tile_out_bits']} tiles={info['n_tiles']} GF={info['flops']/1e9:.0f} -> {info['flops']/ms/1e9:.1f} TF/s  {info.get('note','')}")
