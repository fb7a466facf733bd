#!/usr/bin/env python3
"""Diagnostic (needs `make phases`): per-phase durations of a tile iteration of one chunked (gathered) step of the
n30 x 10000-bitstring scheme.   ARTN_LIB=tools/libartn_hip_phases.so python tools/chunk_phases.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import artensor_amd as A
from artensor_amd import _native as N
from artensor_amd import contraction as C
from artensor_amd.fixtures import load_case
lib = N.lib()
case = load_case(os.path.join(ROOT, "tests", "golden", "n30_sparse10000.npz"))
rec = []
orig = C.contract_gathered
def hook(eq, a, rows_a, b, rows_b, out=None, label=None, _validate=True):
    if a.numel() >= 1 << 22 and label is None:
        rec.append((eq, a, rows_a, b, rows_b))
    return orig(eq, a, rows_a, b, rows_b, out=out, label=label, _validate=_validate)
C.contract_gathered = hook
A.tensor_contraction_sparse(case.fresh_tensors(device="cuda"), case.scheme)
torch.cuda.synchronize()
C.contract_gathered = orig
eq, a, ra, b, rb = rec[len(rec) // 2]
NAMES = ["stage 1", "barrier (+ stage 2 + barrier)", "x reads (+barrier)", "wait loads + refill", "stores", "issue loads", "end barrier", "top (offsets, W)"]
orig(eq, a, ra, b, rb); torch.cuda.synchronize()
orig(eq, a, ra, b, rb); torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (1024 * 20))()
assert lib.artn_debug_read_phases(buf) == 0
nb = 256
d = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 20)[:nb]
t = d[:, 2:20].astype(np.int64).reshape(nb, 2, 9)[:, :, :8]   # 100 MHz ticks
seg = np.diff(t[:, 0, :], axis=1)
top = t[:, 1, 0] - t[:, 0, 7]
period = (t[:, 1, 0] - t[:, 0, 0]).mean()
print(f"tile period {period * 10:.0f} ns per workgroup")
for i in range(7):
    print(f"   {NAMES[i]:30s} {seg[:, i].mean() * 10:7.0f} ns  {100 * seg[:, i].mean() / period:5.1f} %")
print(f"   {NAMES[7]:30s} {top.mean() * 10:7.0f} ns  {100 * top.mean() / period:5.1f} %")
