#!/usr/bin/env python3
"""n53 slices through SliceRunner with/without reuse of small intermediates: wall per slice + launch list."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import artensor_amd as A
from artensor_amd import contraction as C
from artensor_amd.fixtures import load_case
case = load_case(os.path.join(ROOT, "tests", "golden", "n53_m14_sliced.npz"))
leaves = case.fresh_tensors(device="cuda")
order = A.rank_slices(2 ** 14, 0, 8, gray=True)
class Prof:
    def __init__(s): s.rows = []
    def record(s, info, e0, e1): s.rows.append((info, e0, e1))
for reuse in (True,):
    r = A.SliceRunner(leaves, case.scheme, case.slicing_indices, (1,), sparse=True, device="cuda", reuse_small=reuse)
    r.run(order[:2]); torch.cuda.synchronize()
    t0 = time.perf_counter(); r.run(order[2:10]); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8
    p = Prof(); C.profiler = p; r.run(order[10:11]); torch.cuda.synchronize(); C.profiler = None
    rows = sorted(((e0.elapsed_time(e1), info) for info, e0, e1 in p.rows), key=lambda x: -x[0])
    tot = sum(x[0] for x in rows); big = sum(x[0] for x in rows if x[1]["kernel"] in (1, 2))
    print(f"reuse_small={reuse}: {dt*1e3:.2f} ms per slice; launches {len(rows)}, {tot:.1f} ms in contract kernels ({big:.1f} MFMA)")
    for ms, info in rows[:int(os.environ.get("TOP", "14"))]:
        print(f"   {ms:6.2f} ms kernel={info['kernel']} k={info['k_bits']}+{info['k2_bits']} T={info['tile_in_bits']}/{info['tile_out_bits']} tiles={info['n_tiles']} GF={info['flops']/1e9:.0f} -> {info['flops']/ms/1e9:.1f} TF/s  {info['bytes']/1e9:.2f} GB -> {info['bytes']/ms/1e9:.2f} TB/s")
