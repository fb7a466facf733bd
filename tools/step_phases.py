#!/usr/bin/env python3
"""Diagnostic (needs `make phases`): per-phase durations of a tile iteration of ONE single-stage launch of a fixture's
first slice, picked by (contracted bits, tile_in_bits, tile_out_bits):
  ARTN_LIB=tools/libartn_hip_phases.so python tools/step_phases.py n53_m14_sliced.npz 6 12 10"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import artensor_amd as A
from artensor_amd import _native as N
from artensor_amd import contraction as C
from artensor_amd.fixtures import load_case
lib = N.lib()
case = load_case(os.path.join(ROOT, "tests", "golden", sys.argv[1]))
want = tuple(int(x) for x in sys.argv[2:5])
leaves = case.fresh_tensors(device="cuda")
nb = len(case.slicing_indices or {})
rec = []
orig = C._launch_step
def hook(d, a, b, out, stream):
    info = C._step_info_cached(d)
    if (info["k_bits"], info["tile_in_bits"], info["tile_out_bits"]) == want and info["k2_bits"] == 0 and a.numel() >= 1 << 24 and not rec:
        rec.append((d, a.clone(), b.clone(), out, info))
    return orig(d, a, b, out, stream)
C._launch_step = hook
sl = A.apply_slice(leaves, case.slicing_indices, A.slice_assignments(nb, 0)) if nb else dict(leaves)
A.tensor_contraction_sparse(sl, case.scheme)
torch.cuda.synchronize()
C._launch_step = orig
d, a, b, out, info = rec[0]
print({k: info[k] for k in ("kernel", "k_bits", "tile_in_bits", "tile_out_bits", "n_tiles", "grid", "lds_bytes", "a_rereads", "run_in_bits", "run_out_bits")})
out = torch.empty_like(out)
st = N.current_stream_ptr(a.device)
def timeit(n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): N.check(orig(d, a, b, out, st))
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
N.check(orig(d, a, b, out, st)); torch.cuda.synchronize()
print(f"launch: {timeit():.3f} ms")
buf = (ctypes.c_ulonglong * (1024 * 20))()
if lib.artn_debug_read_phases(buf) == 0:
    nbk = min(512, info["grid"])
    dd = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 20)[:nbk]
    t = dd[:, 2:20].astype(np.int64).reshape(nbk, 2, 9)[:, :, :8]
    seg = np.diff(t[:, 0, :], axis=1)
    top = t[:, 1, 0] - t[:, 0, 7]
    period = (t[:, 1, 0] - t[:, 0, 0]).mean()
    NAMES = ["stage 1", "barrier (+ stage 2 + barrier)", "x reads (+barrier)", "wait loads + refill", "stores", "issue loads", "end barrier", "top (offsets, W)"]
    print(f"tile period {period * 10:.0f} ns per workgroup")
    for i in range(7):
        print(f"   {NAMES[i]:30s} {seg[:, i].mean() * 10:7.0f} ns  {100 * seg[:, i].mean() / period:5.1f} %")
    print(f"   {NAMES[7]:30s} {top.mean() * 10:7.0f} ns  {100 * top.mean() / period:5.1f} %")
