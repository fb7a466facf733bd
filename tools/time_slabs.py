#!/usr/bin/env python3
"""Per-rank time of the re-planned n30 slabs on ONE GPU (tests/golden/n30_dense_part{2,4,8}.npz): what a rank of an N-GPU run of
`bench.py --gpus N` executes; projected strong-scaling speed-up = time of the unsliced contraction / time of one slab."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import artensor_amd as A
from artensor_amd.fixtures import load_case
full = load_case(os.path.join(ROOT, "tests", "golden", "n30_dense.npz"))
leaves = full.fresh_tensors(device="cuda")
def timed(fn, n=5):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n
t1 = timed(lambda: A.tensor_contraction(dict(leaves), full.scheme))
print(f"unsliced n30: {t1*1e3:.2f} ms ({8 * 10 ** full.meta['log10_tc'] / t1 / 1e12:.1f} TFLOP/s)")
for n in (2, 4, 8):
    part = load_case(os.path.join(ROOT, "tests", "golden", f"n30_dense_part{n}.npz"))
    pl = part.fresh_tensors(device="cuda")
    ts = [timed(lambda r=r: A.slab_contraction(pl, part.scheme, part.meta["fixed"], r, device="cuda")) for r in (0, n - 1)]
    t = max(ts)
    f = 8 * 10 ** part.meta["log10_tc"]
    print(f"N={n}: slab {t*1e3:.2f} ms ({f / t / 1e12:.1f} TFLOP/s executed per rank; all ranks execute {part.meta['executed_flop_over_unsliced']:.2f} x the unsliced FLOP): "
          f"projected speed-up {t1 / t:.2f} x, {8 * 10 ** full.meta['log10_tc'] / t / 1e12:.0f} TFLOP/s of the metric")
    with __import__("contextlib").redirect_stdout(None):
        pass
    os.environ["ARTN_BENCH_SAME_TREE"] = "1"
    t_old = timed(lambda: A.partitioned_contraction(leaves, full.scheme, {2: 1, 4: 2, 8: 3}[n], 0, device="cuda")[0])
    print(f"      (the one tree of the full network, round 4: slab {t_old*1e3:.2f} ms, projected {t1 / t_old:.2f} x)")
