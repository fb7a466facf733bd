#!/usr/bin/env python3
"""Timing of the n30 pairs whose first stage has 4 contracted bits (development builds with ARTN_DEV_FEW=4)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from artensor_amd.contraction import contract2
from artensor_amd.fixtures import load_case
from helpers import dense_scheme_shapes, GOLDEN
case = load_case(os.path.join(GOLDEN, "n30_dense.npz"))
steps = dense_scheme_shapes(case)
gen = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda shape: torch.view_as_complex(torch.randn(tuple(shape) + (2,), device="cuda", generator=gen))
def timed(fn, reps=4):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best
out = []
for (n, m) in ((101, 104), (93, 97), (116, 118), (149, 155)):
    eq1, sa, sb1 = steps[n]; eq2, _, sb2 = steps[m]
    a, b1, b2 = rnd(sa), rnd(sb1), rnd(sb2)
    out.append(f"p{n}+{m}:{timed(lambda: contract2(eq1, a, b1, eq2, b2)):.2f}")
    del a, b1, b2
print(f"{os.path.basename(os.environ.get('ARTN_LIB', 'product')):34s}", " ".join(out))
