#!/usr/bin/env python3
"""Error figures of slice 0 of the n53 m20 big-batch fixture against the reference's value, for the default
in-kernel accumulation over all 2^15 contracted values and with the contracted index split 2^SPLIT ways
(partial sums in HBM, tree-summed): how much of the distance is accumulation order."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import artensor_amd as A
from artensor_amd import contraction as C
from artensor_amd.fixtures import load_case
case = load_case(os.path.join(ROOT, "tests", "golden", "n53_m20_batch.npz"))
leaves = case.fresh_tensors(device="cuda")
want = case.arrays["slice0"].reshape(-1)
rms = np.sqrt(np.mean(np.abs(want) ** 2))
def figs(got):
    d = np.abs(got - want)
    sel = np.abs(want) >= 1e-3 * rms
    return dict(loose_vs_max=float(d.max() / np.abs(want).max()), per_amp_vs_max_abs_or_rms=float((d / np.maximum(np.abs(want), rms)).max()),
                strict=float((d[sel] / np.abs(want)[sel]).max()), max_over_rms=float(np.abs(want).max() / rms))
for tiles in (512, 1 << 18):   # 2^15 tiles without a split: 2^18 asks for 3 split bits (32 GiB of partial sums)
    C.SPLIT_K_MIN_TILES = tiles
    C._desc_cache.clear(); C._info_cache.clear(); C._plan_cache.clear(); C._schedule_cache.clear()
    r = A.SliceRunner(leaves, case.scheme, case.slicing_indices, (1024,), sparse=True, device="cuda")
    got = r.run([0]).reshape(-1).cpu().numpy()
    print("SPLIT_K_MIN_TILES", tiles, figs(got), flush=True)
