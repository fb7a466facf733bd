#!/usr/bin/env python3
"""Diagnostic (wide TU built with -DARTN_WIDE_MARKS, linked as tools/libartn_wide_marks.so): shader-clock marks of
artn_k_wide's tile loop.   ARTN_WIDE=1 ARTN_LIB=tools/libartn_wide_marks.so python tools/wide_marks.py [pairs...]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import artensor_amd as A
from artensor_amd import _native as N
from artensor_amd.contraction import contract2
from artensor_amd.fixtures import load_case
from helpers import dense_scheme_shapes, GOLDEN
lib = ctypes.CDLL(os.environ["ARTN_LIB"])
case = load_case(os.path.join(GOLDEN, "n30_dense.npz"))
steps = dense_scheme_shapes(case)
gen = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda shape: torch.view_as_complex(torch.randn(tuple(shape) + (2,), device="cuda", generator=gen))
NAMES = {0: "loop top", 1: "s1: first sub-tile done", 2: "s1: done (scatter issued)", 3: "barrier A passed", 4: "s2: first sub-tile done",
         5: "s2: done (scatter issued)", 6: "DMA of next tile landed", 7: "barrier B passed"}
def run(tag, fn):
    fn(); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (64 * 2 * 4 * 24))()
    assert lib.artn_debug_read_wide_marks(buf) == 0
    d = np.frombuffer(buf, dtype=np.uint64).reshape(64, 2, 4, 24).astype(np.int64)
    period = (d[:, :, 1:, 0] - d[:, :, :-1, 0]).mean()
    print(f"{tag}: tile period {period:.0f} shader-clock ticks (wave 0 and wave 4 of 64 workgroups, iterations 20..23)")
    for w in (0, 1):
        print(f"  wave {4 * w}:")
        prev = d[:, w, 1, 0]
        for k in range(1, 8):
            seg = d[:, w, 1, k] - prev
            print(f"    {NAMES[k]:36s} +{seg.mean():7.0f}  (at {(d[:, w, 1, k] - d[:, w, 1, 0]).mean():7.0f})")
            prev = d[:, w, 1, k]
        print(f"    {'next loop top':36s} +{(d[:, w, 2, 0] - prev).mean():7.0f}")
        un = [(d[:, w, 1, 8 + u] - d[:, w, 1, 0]).mean() for u in range(8)]
        print("    stage-1 units of the first sub-tile queued at", " ".join(f"{x:.0f}" for x in un if x > 0))
    print(f"  wave 4 - wave 0 at loop top: {(d[:, 1, 1, 0] - d[:, 0, 1, 0]).mean():.0f}")
pairs = [(101, 104), (139, 144), (93, 97)]
for (n, m) in pairs:
    eq1, sa, sb1 = steps[n]; eq2, _, sb2 = steps[m]
    a, b1, b2 = rnd(sa), rnd(sb1), rnd(sb2)
    info = A.contraction.pair_info(eq1, sa, sb1, eq2, sb2) if hasattr(A.contraction, "pair_info") else None
    run(f"fused pair {n}+{m} {info and (info.get('k_bits'), info.get('k2_bits'))}", lambda: contract2(eq1, a, b1, eq2, b2))
    del a, b1, b2
