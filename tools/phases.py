#!/usr/bin/env python3
"""Diagnostic (needs `make phases`): per-phase durations of one tile iteration and the phase
relation of the two workgroups sharing a CU.   ARTN_LIB=tools/libartn_hip_phases.so python tools/phases.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import artensor_amd as A
from artensor_amd import _native as N
from artensor_amd.contraction import contract2
from artensor_amd.fixtures import load_case
from helpers import dense_scheme_shapes, GOLDEN
lib = N.lib()
case = load_case(os.path.join(GOLDEN, "n30_dense.npz"))
steps = dense_scheme_shapes(case)
gen = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda shape: torch.view_as_complex(torch.randn(tuple(shape) + (2,), device="cuda", generator=gen))
NAMES = ["stage 1", "barrier + stage 2 + barrier", "x reads (+barrier)", "wait loads + refill", "stores", "issue loads", "end barrier", "top (offsets, W)"]
def run(tag, fn):
    fn(); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (1024 * 20))()
    assert lib.artn_debug_read_phases(buf) == 0
    d = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 20)[:512]
    hw, xcc = d[:, 0], d[:, 1]
    t = d[:, 2:20].astype(np.int64).reshape(512, 2, 9)[:, :, :8]   # 100 MHz ticks
    seg = np.diff(t[:, 0, :], axis=1)                               # 7 segments of iteration 20
    top = t[:, 1, 0] - t[:, 0, 7]
    period = (t[:, 1, 0] - t[:, 0, 0]).mean()
    print(f"{tag}: tile period {period * 10:.0f} ns per workgroup")
    for i in range(7):
        print(f"   {NAMES[i]:30s} {seg[:, i].mean() * 10:7.0f} ns  {100 * seg[:, i].mean() / period:5.1f} %")
    print(f"   {NAMES[7]:30s} {top.mean() * 10:7.0f} ns  {100 * top.mean() / period:5.1f} %")
    key = {}
    for b in range(512):
        h = int(hw[b]); key.setdefault((int(xcc[b]) & 0xf, (h >> 13) & 7, (h >> 12) & 1, (h >> 8) & 0xf), []).append(b)
    offs = []
    for k, v in key.items():
        if len(v) == 2:
            dlt = (t[v[0], 0, 0] - t[v[1], 0, 0]) % period
            offs.append(min(dlt, period - dlt) / period)
    print("   phase offset of the two workgroups of a CU: mean %.2f (0 lockstep, 0.5 alternating)" % np.mean(offs))
for (n, m) in ((101, 104), (93, 97), (108, 112), (139, 144)):
    eq1, sa, sb1 = steps[n]; eq2, _, sb2 = steps[m]
    a, b1, b2 = rnd(sa), rnd(sb1), rnd(sb2)
    run(f"fused pair {n}+{m}", lambda: contract2(eq1, a, b1, eq2, b2))
    del a, b1, b2
