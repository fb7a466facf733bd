#!/usr/bin/env python3
"""Where does a slice spend time it should not?  Per launch of one slice of a fixture: measured ms against the launch's own
lower bound max(bytes / 5 TB/s, FLOP / 125 TFLOP/s) (2.0 PFLOP/s in the reduced-precision mode), sorted by the EXCESS.
    [PRECISION=bf16] [TOP=n] python tools/scan_launches.py <fixture.npz> [sparse]"""
import os, sys, time, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import artensor_amd as A
from artensor_amd import contraction as C
from artensor_amd import _native as N
from artensor_amd.fixtures import load_case
case = load_case(os.path.join(ROOT, "tests", "golden", sys.argv[1]))
sparse = len(sys.argv) > 2 and sys.argv[2] == "sparse"
leaves = case.fresh_tensors(device="cuda")
nb = len(case.slicing_indices or {})
bf16 = os.environ.get("PRECISION") == "bf16"
orig_q = C._query
def q(d):
    info = orig_q(d)
    import math, collections
    cnt = collections.Counter()
    for i in range(d.n_labels):
        cls = "".join(c for c, st in zip("ABC", (d.stride_a[i], d.stride_b[i], d.stride_c[i])) if st >= 0)
        cnt[cls] += math.log2(d.extent[i])
    info["shape"] = {k: round(v, 1) for k, v in cnt.items()}
    return info
C._query = q
def ctx():
    return C.precision("bf16") if bf16 else contextlib.nullcontext()
def one(s):
    with ctx():
        sl = A.apply_slice(leaves, case.slicing_indices, A.slice_assignments(nb, s)) if nb else dict(leaves)
        return (A.tensor_contraction_sparse if sparse else A.tensor_contraction)(sl, case.scheme)
one(0); torch.cuda.synchronize()
class Prof:
    def __init__(s): s.rows = []
    def record(s, info, e0, e1): s.rows.append((info, e0, e1))
p = Prof(); C.profiler = p
torch.cuda.synchronize(); t0 = time.perf_counter(); one(1 if nb else 0); torch.cuda.synchronize(); wall = time.perf_counter() - t0
C.profiler = None
BW, FL = 5.0e12, (2.0e15 if bf16 else 125e12)
rows = []
for info, e0, e1 in p.rows:
    ms = e0.elapsed_time(e1)
    lb = max(info.get("bytes", 0.0) / BW, info["flops"] / FL) * 1e3
    rows.append((ms - lb, ms, lb, info))
tot = sum(r[1] for r in rows); totlb = sum(r[2] for r in rows)
print(f"{sys.argv[1]}{' bf16' if bf16 else ''}: wall {wall*1e3:.1f} ms (with per-launch events), {len(rows)} launches, {tot:.1f} ms in kernels, sum of per-launch lower bounds {totlb:.1f} ms")
rows.sort(key=lambda r: -r[0])
for ex, ms, lb, info in rows[:int(os.environ.get("TOP", "16"))]:
    print(f"   excess {ex:6.2f} ms  ({ms:6.2f} measured, {lb:5.2f} bound)  kernel={info['kernel']} k={info['k_bits']}+{info['k2_bits']} tiles={info['n_tiles']} rereads={info['a_rereads']} "
          f"GF={info['flops']/1e9:.1f} GB={info.get('bytes',0)/1e9:.2f} {info.get('shape','')}")
small = [r for r in rows if r[1] < 0.05]
print(f"   launches under 50 us: {len(small)}, {sum(r[1] for r in small):.2f} ms in all")
