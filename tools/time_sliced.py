#!/usr/bin/env python3
"""Per-launch table of one slice of a sliced fixture through SliceRunner (reuse of small intermediates on)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import artensor_amd as A
from artensor_amd import contraction as C
from artensor_amd.fixtures import load_case
name = sys.argv[1]
case = load_case(os.path.join(ROOT, "tests", "golden", name + ".npz"))
leaves = case.fresh_tensors(device="cuda")
sparse = case.meta.get("pattern") == "sparse"
rows = len(case.meta["bitstrings_sorted"]) if sparse else 1
nb = len(case.slicing_indices or {})
order = A.rank_slices(2 ** nb, 0, 8, gray=True) if nb else [0] * 16
class Prof:
    def __init__(s): s.rows = []
    def record(s, info, e0, e1): s.rows.append((info, e0, e1))
r = A.SliceRunner(leaves, case.scheme, case.slicing_indices, (rows,), sparse=sparse, device="cuda")
r.run(order[:2]); torch.cuda.synchronize()
t0 = time.perf_counter(); r.run(order[2:6]); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4
p = Prof(); C.profiler = p; r.run(order[6:7]); torch.cuda.synchronize(); C.profiler = None
rows_ = sorted(((e0.elapsed_time(e1), info) for info, e0, e1 in p.rows), key=lambda x: -x[0])
tot = sum(x[0] for x in rows_)
fl = 8.0 * 10 ** case.meta["log10_tc"]
print(f"{name}: {dt*1e3:.2f} ms per slice = {fl/dt/1e12:.1f} TF; launches {len(rows_)}, {tot:.1f} ms in contract kernels")
for ms, info in rows_[:int(os.environ.get("TOP", "16"))]:
    print(f"   {ms:6.2f} ms ({100*ms/tot:4.1f}%) kernel={info['kernel']} k={info['k_bits']}+{info['k2_bits']} mt={info['m_tile_bits']} nt={info['n_tile_bits']} T={info['tile_in_bits']}/{info['tile_out_bits']} tiles={info['n_tiles']} rr={info['a_rereads']} GF={info['flops']/1e9:.0f} -> {info['flops']/ms/1e9:.1f} TF/s  {info['bytes']/ms/1e6:.0f} GB/s")
