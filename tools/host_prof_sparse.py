#!/usr/bin/env python3
"""Diagnostic: host-side profile (cProfile) of the sparse executor on an n30 sparse fixture, plus which torch
operations run per call (torch.profiler kernel names with counts)."""
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import artensor_amd as A
from artensor_amd.fixtures import load_case
name = sys.argv[1] if len(sys.argv) > 1 else "n30_sparse100.npz"
case = load_case(os.path.join(ROOT, "tests", "golden", name))
leaves = case.fresh_tensors(device="cuda")
run = lambda: A.tensor_contraction_sparse(dict(leaves), case.scheme)
for _ in range(3): run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): run()
torch.cuda.synchronize()
print(f"wall {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per run")
t0 = time.perf_counter()
for _ in range(5): run()
t_host = (time.perf_counter() - t0) / 5
torch.cuda.synchronize()
print(f"host time to enqueue one run: {t_host * 1e3:.2f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(10): run()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(18)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    run(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
