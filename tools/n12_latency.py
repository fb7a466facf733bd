#!/usr/bin/env python3
"""Where the n12 end-to-end latency goes: host time per call (no sync), kernel time (events), sync'ed wall."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import artensor_amd as A
from artensor_amd import contraction as C
from artensor_amd.fixtures import load_case
case = load_case(os.path.join(ROOT, "tests", "golden", "n12_dense.npz"))
leaves = case.fresh_tensors(device="cuda")
for _ in range(5):
    A.tensor_contraction(dict(leaves), case.scheme)
torch.cuda.synchronize()
N = 200
t0 = time.perf_counter()
for _ in range(N):
    A.tensor_contraction(dict(leaves), case.scheme)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host per call (async): {(t1 - t0) / N * 1e6:.1f} us; drained after {(t2 - t1) * 1e6:.0f} us more -> device-bound rate {(t2 - t0) / N * 1e6:.1f} us per call")
best = 1e9
for _ in range(50):
    t0 = time.perf_counter(); A.tensor_contraction(dict(leaves), case.scheme); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
print(f"sync'ed wall, best of 50: {best * 1e6:.1f} us")
class Prof:
    def __init__(s): s.rows = []
    def record(s, info, e0, e1): s.rows.append((info, e0, e1))
p = Prof(); C.profiler = p
for _ in range(20):
    A.tensor_contraction(dict(leaves), case.scheme)
torch.cuda.synchronize(); C.profiler = None
ks = [e0.elapsed_time(e1) * 1e3 for _, e0, e1 in p.rows]
print(f"program kernel (events): min {min(ks):.1f} us, median {sorted(ks)[len(ks)//2]:.1f} us over {len(ks)} launches")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
x = torch.zeros(16, device="cuda")
e0.record(); x.add_(1); e1.record(); torch.cuda.synchronize()
print(f"(an empty-ish torch kernel between events: {e0.elapsed_time(e1) * 1e3:.1f} us)")
