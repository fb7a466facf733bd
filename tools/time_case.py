#!/usr/bin/env python3
"""Per-launch breakdown of one slice of a fixture: [DTYPE=c128] [PRECISION=bf16] [TOP=n] python tools/time_case.py <fixture.npz> [sparse]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import artensor_amd as A
from artensor_amd import contraction as C
from artensor_amd import _native as N
from artensor_amd.fixtures import load_case
case = load_case(os.path.join(ROOT, "tests", "golden", sys.argv[1]))
sparse = len(sys.argv) > 2 and sys.argv[2] == "sparse"
leaves = case.fresh_tensors(device="cuda", dtype=torch.complex128 if os.environ.get("DTYPE") == "c128" else torch.complex64)
nb = len(case.slicing_indices or {})
rec = []
orig_q = C._query
def q(d):
    info = orig_q(d)
    info["note"] = N.lib().artn_last_plan_note().decode()
    ext = [d.extent[i] for i in range(d.n_labels)]
    cls = ["".join(c for c, st in zip("ABC", (d.stride_a[i], d.stride_b[i], d.stride_c[i])) if st >= 0) for i in range(d.n_labels)]
    import collections
    cnt = collections.Counter()
    for e, c in zip(ext, cls): cnt[c] += __import__('math').log2(e)
    info["shape"] = {k: round(v, 1) for k, v in cnt.items()}
    return info
C._query = q
import contextlib
def ctx():
    return C.precision(os.environ["PRECISION"]) if os.environ.get("PRECISION") else contextlib.nullcontext()
def one(s):
  with ctx():
    sl = A.apply_slice(leaves, case.slicing_indices, A.slice_assignments(nb, s)) if nb else dict(leaves)
    return (A.tensor_contraction_sparse if sparse else A.tensor_contraction)(sl, case.scheme)
one(0); torch.cuda.synchronize()
t0 = time.perf_counter(); one(1 if nb else 0); torch.cuda.synchronize(); dt = time.perf_counter() - t0
class Prof:
    def __init__(s): s.rows = []
    def record(s, info, e0, e1): s.rows.append((info, e0, e1))
p = Prof(); C.profiler = p; one(2 if nb else 0); torch.cuda.synchronize(); C.profiler = None
rows = sorted(((e0.elapsed_time(e1), info) for info, e0, e1 in p.rows), key=lambda r: -r[0])
tot = sum(r[0] for r in rows)
print(f"{sys.argv[1]}: wall {dt*1e3:.1f} ms; launches {len(rows)}: {tot:.1f} ms in kernels")
for ms, info in rows[:int(os.environ.get("TOP", "12"))]:
    print(f"   {ms:7.2f} ms kernel={info['kernel']} k={info['k_bits']}+{info['k2_bits']} T={info['tile_in_bits']}/{info['tile_out_bits']} runs={info.get('run_in_bits')}/{info.get('run_out_bits')} arith={info.get('arith')} tiles={info['n_tiles']} rereads={info['a_rereads']} GF={info['flops']/1e9:.1f} -> {info['flops']/ms/1e9:.1f} TF/s bits{info.get('shape','')} {info.get('note','')}")
