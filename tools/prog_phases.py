import ctypes, os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import artensor_amd as A
from artensor_amd import _native as N
from artensor_amd.fixtures import load_case
case = load_case("/root/repo/tests/golden/n12_dense.npz")
leaves = case.fresh_tensors(device="cuda")
for _ in range(3): A.tensor_contraction(dict(leaves), case.scheme)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (1024 * 20))()
assert N.lib().artn_debug_read_phases(buf) == 0
t = np.frombuffer(buf, dtype=np.uint64)[:42].astype(np.int64)
print("phase0 (tables, preloads):", (t[1]-t[0])*10, "ns")
prev = t[1]
for L in range(19):
    print(f"level {L+1}: work {(t[2+2*L]-prev)*10} ns, barrier {(t[3+2*L]-t[2+2*L])*10} ns")
    prev = t[3+2*L]
print("total", (t[3+2*18]-t[0])*10, "ns")
