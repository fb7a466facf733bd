#!/usr/bin/env python3
"""Time the sparse-state executor on the n30 fixtures (BASELINE config 3)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import artensor_amd as A
from artensor_amd import contraction as C
from artensor_amd.fixtures import load_case
class Prof:
    def __init__(s): s.rows = []
    def record(s, info, e0, e1): s.rows.append((info, e0, e1))
for name in ("n30_sparse100", "n30_sparse10000"):
    case = load_case(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    leaves = case.fresh_tensors(device="cuda")
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = A.tensor_contraction_sparse(dict(leaves), case.scheme)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    flops = 8 * 10 ** case.meta["log10_tc"]
    err = np.abs(out.cpu().numpy() - case.arrays["final"]).max() / np.abs(case.arrays["final"]).max()
    print(f"{name}: {dt * 1e3:.1f} ms, {flops / dt / 1e12:.1f} TFLOP/s (tc-based), rel err {err:.1e}, reference CPU {case.meta['reference_cpu_seconds']:.0f} s")
    p = Prof(); C.profiler = p
    A.tensor_contraction_sparse(dict(leaves), case.scheme); torch.cuda.synchronize(); C.profiler = None
    rows = sorted(((e0.elapsed_time(e1), info) for info, e0, e1 in p.rows), key=lambda r: -r[0])
    tot = sum(r[0] for r in rows)
    print(f"   contract launches {len(rows)}, {tot:.1f} ms in contract kernels; top:")
    for ms, info in rows[:8]:
        print(f"     {ms:7.2f} ms kernel={info['kernel']} k={info['k_bits']} T={info['tile_in_bits']}/{info['tile_out_bits']} tiles={info['n_tiles']} rereads={info['a_rereads']} GF={info['flops'] / 1e9:.0f} -> {info['flops'] / ms / 1e9:.1f} TF/s")
