#!/usr/bin/env python3
"""Diagnostic: run the n12 small-step program on the GPU with every result kept in the workspace and compare each
step's result with the CPU emulation of the same image (ARTN_PROG_KEEP_ALL=1)."""
import os, sys, ctypes
os.environ["ARTN_PROG_KEEP_ALL"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from artensor_amd import contraction as C, _native as N
from artensor_amd.fixtures import load_case
from helpers import emulate_program, parse_program_image
case = load_case(os.path.join(ROOT, "tests", "golden", (sys.argv[1] if len(sys.argv) > 1 else "n12_dense") + ".npz"))
leaves = {k: t.numpy() for k, t in case.tensors.items()}
shapes = {k: tuple(v.shape) for k, v in leaves.items()}
prog, main = C._plan_small_program(case.scheme, shapes, torch.complex64)
ws_cpu, stats = emulate_program(prog, leaves)
dev = {k: torch.from_numpy(v).cuda() for k, v in leaves.items()}
image = prog.device_copy(torch.device("cuda:0"))
ws = torch.zeros(prog.ws_bytes, dtype=torch.uint8, device="cuda")
ext = (ctypes.c_void_p * len(prog.ext_ids))(*[dev[t].data_ptr() for t in prog.ext_ids])
N.check(N.lib().artn_program_run(image.data_ptr(), prog.n_groups, ext, len(prog.ext_ids), ws.data_ptr(), N.ARTN_C64, None))
torch.cuda.synchronize()
ws_gpu = ws.cpu().numpy().view(np.complex64)
groups, levels, wtasks, recs = parse_program_image(prog.host_image.numpy())
bad = 0
for i, R in enumerate(recs):
    o, n = int(R["loc_c"]) // 8, int(R["out_numel"])
    a, b = ws_gpu[o:o + n], ws_cpu[o:o + n]
    err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
    if not (err < 1e-5):
        bad += 1
        print(f"record {i} level {R['level']} fast {R['fast']} mbits {R['n_mbits']} nbits {R['n_nbits']} red {R['red_numel']} out {n} lds_a {R['lds_a']} lds_b {R['lds_b']} lds_c {R['lds_c']}: err {err:.2e}, wrong elements {int((np.abs(a-b) > 1e-5*np.abs(b).max()).sum())}")
print("bad records:", bad, "of", len(recs))
