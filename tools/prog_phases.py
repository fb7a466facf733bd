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
raw = np.frombuffer(buf, dtype=np.uint64).astype(np.int64)
t = raw[:42]
c = raw[512:512 + 42]
print(f"shader clock during the kernel: {(c[39] - c[0]) / ((t[39] - t[0]) * 10e-9) / 1e9:.2f} GHz (s_memtime ticks per s_memrealtime second)")
print("phase0 (tables, preloads):", (t[1]-t[0])*10, "ns")
prev = t[1]
for L in range(19):
    print(f"level {L+1}: work {(t[2+2*L]-prev)*10} ns, barrier {(t[3+2*L]-t[2+2*L])*10} ns")
    prev = t[3+2*L]
print("total", (t[3+2*18]-t[0])*10, "ns")

f = raw[100:100 + 8 * 19].reshape(19, 8)
print("wave 0, first task of a level (ns): level entry -> level table | task entry | record fields | task done")
for L in range(19):
    base = t[1] if L == 0 else t[3 + 2 * (L - 1)]
    if f[L, 3] == 0:
        print(f" level {L+1}: (generic task)  table {(f[L,0]-base)*10}  task {(f[L,1]-f[L,0])*10}  record {(f[L,2]-f[L,1])*10}")
    else:
        print(f" level {L+1}: table {(f[L,0]-base)*10}  task {(f[L,1]-f[L,0])*10}  record {(f[L,2]-f[L,1])*10}  mfma task {(f[L,3]-f[L,2])*10} = offsets {(f[L,4]-f[L,2])*10} + chain {(f[L,5]-f[L,4])*10} + scatter {(f[L,3]-f[L,5])*10}")
