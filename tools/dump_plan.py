#!/usr/bin/env python3
"""Diagnostic (host only; g++ -shared tools/csrc/dump_plan.cpp): tile layouts of the fused pairs of the n30 fixture."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from artensor_amd import contraction as C
from artensor_amd.contraction import fusion_schedule
from artensor_amd.fixtures import load_case
from helpers import dense_scheme_shapes, GOLDEN
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "libdump_plan.so"))
case = load_case(os.path.join(GOLDEN, sys.argv[1] if len(sys.argv) > 1 else "n30_dense.npz"))
steps = dense_scheme_shapes(case)
for e in fusion_schedule(case.scheme):
    if e[0] != "pair" or np.prod(steps[e[1]][1]) < 2 ** 22:
        continue
    _, n, m = e
    eq1, sa, sb1 = steps[n]; eq2, _, sb2 = steps[m]
    mk = lambda s: torch.empty(tuple(s), dtype=torch.complex64, device="meta")
    d1, d2, _ = C._pair_descriptors(eq1, mk(sa), mk(sb1), eq2, mk(sb2))
    print(f"pair {n}+{m}: {eq1} ; {eq2}"); sys.stdout.flush()
    lib.artn_dump_plan(ctypes.byref(d1), ctypes.byref(d2))
