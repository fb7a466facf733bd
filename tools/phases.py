#!/usr/bin/env python3
"""Diagnostic: are the two workgroups of a CU in lockstep?  (needs `make stamps`)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from artensor_amd import _native as N
from artensor_amd.contraction import contract2
from artensor_amd.fixtures import load_case
from helpers import dense_scheme_shapes, GOLDEN
lib = N.lib()
case = load_case(os.path.join(GOLDEN, "n30_dense.npz"))
steps = dense_scheme_shapes(case)
gen = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda shape: torch.view_as_complex(torch.randn(tuple(shape) + (2,), device="cuda", generator=gen))
n, m = 93, 97
eq1, sa, sb1 = steps[n]; eq2, _, sb2 = steps[m]
a, b1, b2 = rnd(sa), rnd(sb1), rnd(sb2)
contract2(eq1, a, b1, eq2, b2); torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (1024 * 16))()
assert lib.artn_debug_read_phases(buf) == 0
d = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 16)[:512]
hw, xcc = d[:, 0], d[:, 1]
print("HW_ID sample:", [hex(int(x)) for x in hw[:8]], "XCC:", [int(x) & 0xf for x in xcc[:8]])
key = {}
for b in range(512):
    h = int(hw[b]); cu = (h >> 8) & 0xf; sh = (h >> 12) & 1; se = (h >> 13) & 0x7
    key.setdefault((int(xcc[b]) & 0xf, se, sh, cu), []).append(b)
sizes = [len(v) for v in key.values()]
print("distinct CUs:", len(key), "WGs per CU histogram:", np.bincount(sizes))
t = d[:, 2:16].astype(np.int64)  # start,end of stage phase for iterations 20..26 (100 MHz ticks)
period = np.diff(t[:, 0::2], axis=1).mean()
stage = (t[:, 1::2] - t[:, 0::2]).mean()
print(f"mean tile period {period * 10:.0f} ns, mean stage phase {stage * 10:.0f} ns")
offs = []
for k, v in key.items():
    if len(v) == 2:
        x, y = v
        dlt = (t[x, 0] - t[y, 0]) % period
        offs.append(min(dlt, period - dlt) / period)
        if len(offs) <= 6:
            print("CU", k, "WGs", v, "wave slots", hex(int(hw[x]) & 0xf), hex(int(hw[y]) & 0xf), f"phase offset {offs[-1]:.2f} of a period")
print("phase offset between the two WGs of a CU (0 = lockstep, 0.5 = alternating): mean %.3f, hist %s" %
      (np.mean(offs), np.histogram(offs, bins=5, range=(0, 0.5))[0]))
