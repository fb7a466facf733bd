#!/usr/bin/env python3
"""Per-phase durations of one tile iteration for the 7-8 contracted-bit steps of n53 (needs `make phases`):
   ARTN_LIB=tools/libartn_hip_phases.so python tools/phases_big.py"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import artensor_amd as A
from artensor_amd import _native as N
lib = N.lib()
steps = json.load(open(os.path.join(ROOT, "tools", "heavy_steps.json")))
gen = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda shape: torch.view_as_complex(torch.randn(tuple(shape) + (2,), device="cuda", generator=gen))
NAMES = ["stage 1", "barrier (+stage 2)", "x reads (+barrier)", "wait loads + refill", "stores", "issue loads", "end barrier", "top (offsets, W)"]
for st in steps:
    k = len([x for x in st["la"] if x in st["lb"] and x not in st["lo"]])
    if st["case"] != "n53" or k < 7 or len(st["la"]) < 28:
        continue
    a, b = rnd(st["a_shape"]), rnd(st["b_shape"])
    eq = (tuple(st["la"]), tuple(st["lb"]), tuple(st["lo"]))
    A.contract(eq, a, b); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (1024 * 20))()
    assert lib.artn_debug_read_phases(buf) == 0
    d = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 20)[:256]
    t = d[:, 2:20].astype(np.int64).reshape(256, 2, 9)[:, :, :8]
    seg = np.diff(t[:, 0, :], axis=1)
    top = t[:, 1, 0] - t[:, 0, 7]
    period = (t[:, 1, 0] - t[:, 0, 0]).mean()
    print(f"k={k} rankA={len(st['la'])}: tile period {period * 10:.0f} ns per workgroup")
    for i in range(7):
        print(f"   {NAMES[i]:24s} {seg[:, i].mean() * 10:7.0f} ns  {100 * seg[:, i].mean() / period:5.1f} %")
    print(f"   {NAMES[7]:24s} {top.mean() * 10:7.0f} ns  {100 * top.mean() / period:5.1f} %")
    del a, b
