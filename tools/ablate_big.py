#!/usr/bin/env python3
"""Timing-only ablation of the 7-8 contracted-bit steps captured in gpurun_out/heavy_steps.json
(tools/capture_steps.py); needs `make ablate`.  Results are wrong by design."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import artensor_amd as A
steps = json.load(open(os.path.join(ROOT, "tools", "heavy_steps.json")))
gen = torch.Generator(device="cuda").manual_seed(0)
def rnd(shape):
    return torch.view_as_complex(torch.randn(tuple(shape) + (2,), device="cuda", generator=gen))
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best
tag = os.path.basename(os.environ.get("ARTN_LIB", "product"))
out = []
for st in steps:
    k = len([x for x in st["la"] if x in st["lb"] and x not in st["lo"]])
    if st["case"] != "n53" or k < 6:
        continue
    a, b = rnd(st["a_shape"]), rnd(st["b_shape"])
    eq = (tuple(st["la"]), tuple(st["lb"]), tuple(st["lo"]))
    out.append(f"k{k}r{len(st['la'])}:{timed(lambda: A.contract(eq, a, b)):.2f}")
    del a, b
print(f"{tag:34s}", " ".join(out))
