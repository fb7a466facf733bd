#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of artn_k_bits on chosen n30 steps (needs `make stamps`).
    ARTN_LIB=tools/libartn_hip_stamps.so python tools/stamps.py
Never quote this build's run time: the stamps forbid overlaps the real kernel has."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import artensor_amd as A
from artensor_amd import _native as N
from artensor_amd.contraction import contract2, fusion_schedule
from artensor_amd.fixtures import load_case
from helpers import dense_scheme_shapes, GOLDEN

NAMES = ["offsets+W", "barrier(pre-fill)", "load-wait+LDS fill", "barrier(fill)", "issue loads", "stage other", "barrier(stage)", "copy-out", "subtile setup", "mfma chain", "scatter", "-"]
lib = N.lib()
lib.artn_debug_read_stamps.restype = ctypes.c_int

def report(tag, ms):
    n = 2048
    buf = (ctypes.c_ulonglong * (12 * n))()
    assert lib.artn_debug_read_stamps(buf, n) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 12).astype(np.float64)
    a = a[a.sum(axis=1) > 0]
    tot = a.sum(axis=1).mean()
    print(f"{tag}: {ms:.2f} ms, {len(a)} waves, mean cycles/wave {tot:.3e}")
    for i, nm in enumerate(NAMES):
        print(f"   {nm:22s} {100 * a[:, i].mean() / tot:5.1f} %")

case = load_case(os.path.join(GOLDEN, "n30_dense.npz"))
steps = dense_scheme_shapes(case)
gen = torch.Generator(device="cuda").manual_seed(0)
def rnd(shape):
    return torch.view_as_complex(torch.randn(tuple(shape) + (2,), device="cuda", generator=gen))
def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); e0.record(); out = fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
for n in (78, 88, 139):
    eq, sa, sb = steps[n]
    a, b = rnd(sa), rnd(sb)
    ms = timed(lambda: A.contract(eq, a, b))
    report(f"single step {n} (k={len(sb) - (len(eq.split('->')[1]) - (len(sa) - 0)) if False else ''})", ms)
    del a, b
for (n, m) in ((101, 104), (93, 97), (108, 112)):
    eq1, sa, sb1 = steps[n]; eq2, _, sb2 = steps[m]
    a, b1, b2 = rnd(sa), rnd(sb1), rnd(sb2)
    ms = timed(lambda: contract2(eq1, a, b1, eq2, b2))
    report(f"fused pair {n}+{m}", ms)
    del a, b1, b2
