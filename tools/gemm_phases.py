#!/usr/bin/env python3
"""Diagnostic (needs `make phases`): where a chunk period of artn_k_gemm_deep goes, on one GEMM-planned launch of a fixture's
first slice picked by (contracted bits, tile_in_bits, tile_out_bits):
  ARTN_LIB=tools/libartn_hip_phases.so python tools/gemm_phases.py n53_m14_sliced.npz 7 11 12"""
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
    if (info["k_bits"], info["tile_in_bits"], info["tile_out_bits"]) == want and info["kernel"] == 2 and a.numel() >= 1 << 24 and not rec:
        rec.append((d, a.clone(), b.clone(), out, info))
    return orig(d, a, b, out, stream)
C._launch_step = hook
sl = A.apply_slice(leaves, case.slicing_indices, A.slice_assignments(nb, 0)) if nb else dict(leaves)
(A.tensor_contraction_sparse if len(sys.argv) <= 5 else A.tensor_contraction)(sl, case.scheme)
torch.cuda.synchronize()
C._launch_step = orig
d, a, b, out, info = rec[0]
out = torch.empty_like(out)
st = N.current_stream_ptr(a.device)
for _ in range(3):
    N.check(orig(d, a, b, out, st))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); N.check(orig(d, a, b, out, st)); e1.record(); torch.cuda.synchronize()
print({k: info[k] for k in ("kernel", "k_bits", "tile_in_bits", "tile_out_bits", "n_tiles", "grid", "lds_bytes")}, f"launch {e0.elapsed_time(e1):.3f} ms")
buf = (ctypes.c_ulonglong * (1024 * 20))()
assert lib.artn_debug_read_phases(buf) == 0
t = np.frombuffer(buf, dtype=np.uint64)[:64 * 16].astype(np.int64).reshape(64, 2, 8)[:, :, :6]   # shader-clock ticks (100 MHz? s_memtime)
seg = np.diff(t, axis=2).reshape(-1, 5)
names = ["offsets + load issue", "MFMA section (incl. LDS operand reads)", "wait for the loads of chunk c + 1", "LDS fill", "barrier"]
nxt = (t[:, 1, 0] - t[:, 0, 0]).mean()
print(f"chunk period {nxt:.0f} ticks")
for n, v in zip(names, seg.mean(axis=0)):
    print(f"   {n:45s} {v:8.0f} ticks  {100 * v / nxt:5.1f} %")
