#!/usr/bin/env python3
"""Run fused pairs of the n30 scheme on random operands (diagnostics: rocprofv3 --pmc / --kernel-trace around it).
    python3 tools/run_pair.py 172+179 139+144 [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import artensor_amd as A
from artensor_amd.contraction import contract2
from artensor_amd.fixtures import load_case
from helpers import dense_scheme_shapes, GOLDEN
case = load_case(os.path.join(GOLDEN, "n30_dense.npz"))
steps = dense_scheme_shapes(case)
gen = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda shape: torch.view_as_complex(torch.randn(tuple(shape) + (2,), device="cuda", generator=gen))
pairs = [tuple(int(x) for x in a.split("+")) for a in sys.argv[1:] if "+" in a]
reps = [int(a) for a in sys.argv[1:] if "+" not in a]
reps = reps[0] if reps else 3
for (n, m) in pairs:
    eq1, sa, sb1 = steps[n]; eq2, _, sb2 = steps[m]
    a, b1, b2 = rnd(sa), rnd(sb1), rnd(sb2)
    contract2(eq1, a, b1, eq2, b2); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = contract2(eq1, a, b1, eq2, b2)
    e1.record(); torch.cuda.synchronize()
    print(f"pair {n}+{m}: {e0.elapsed_time(e1) / reps:.3f} ms per launch", flush=True)
    del a, b1, b2, out
