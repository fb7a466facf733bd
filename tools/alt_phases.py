#!/usr/bin/env python3
"""Diagnostic (dev build with -DARTN_PHASES): where a half period of artn_k_alt goes, per role.
   ARTN_LIB=tools/libartn_hip_dev4.so python tools/alt_phases.py"""
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
for (n, m) in ((101, 104), (93, 97), (139, 144)):
    eq1, sa, sb1 = steps[n]; eq2, _, sb2 = steps[m]
    if len([c for c in eq1.split(",")[0] if c in eq1.split(",")[1].split("->")[0]]) != 4 and os.environ.get("DEV4"):
        pass
    a, b1, b2 = rnd(sa), rnd(sb1), rnd(sb2)
    try:
        contract2(eq1, a, b1, eq2, b2); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); contract2(eq1, a, b1, eq2, b2); e1.record(); torch.cuda.synchronize()
    except RuntimeError as e:
        print(n, m, "skipped:", str(e)[:80]); continue
    buf = (ctypes.c_ulonglong * (1024 * 20))()
    assert lib.artn_debug_read_phases(buf) == 0
    raw = np.frombuffer(buf, dtype=np.uint64).astype(np.int64)
    d = raw[:64 * 2 * 4 * 8].reshape(64, 2, 4, 8)
    sm = raw[4096:4096 + 64 * 2 * 4 * 2 * 8].reshape(64, 2, 4, 2, 8)
    print(f"pair {n}+{m}: {e0.elapsed_time(e1):.2f} ms")
    # role of group g in half h (h = 40..43): compute when (h ^ g) even
    for g in range(2):
        for hh in range(4):
            role = "compute" if ((40 + hh) ^ g) % 2 == 0 else "copy"
            t = d[:, g, hh, :]
            if role == "compute":
                names = [("stage 1", 0, 1), ("wait barrier 1", 1, 2), ("stage 2", 2, 3), ("wait barrier 2", 3, 4)]
            else:
                names = [("x reads", 0, 1), ("wait barrier 1", 1, 2), ("wait loads + refill", 2, 5), ("stores", 5, 6), ("issue loads", 6, 3), ("wait barrier 2", 3, 4)]
            parts = ", ".join(f"{nm} {np.median(t[:, b] - t[:, a]):.0f}" for nm, a, b in names)
            print(f"   group {g} half {40 + hh} {role:8s} total {np.median(t[:, 4] - t[:, 0]):6.0f} cycles: {parts}")
            if role == "compute":
                for st in range(2):
                    q = sm[:, g, hh, st, :]
                    seg = lambda a, b: np.median(q[:, b] - q[:, a])
                    last = 3 if np.median(q[:, 3]) > np.median(q[:, 2]) else 2
                    print(f"        stage {st + 1}: first operands {seg(0, 1):.0f}, chain(s) to last sub-tile {np.median(q[:, last] - q[:, 1]):.0f}, final scatter {np.median(q[:, 4] - q[:, last]):.0f}; stage start after half start {np.median(q[:, 0] - t[:, 0]):.0f}")
    del a, b1, b2
