# first GPU check of artn_k_xgemm: random D=3 / mixed steps against torch einsum (c128), and a timing of benchmark-size steps
import sys, time
import numpy as np, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import artensor_amd as A
from test_xgemm_emulation import random_step, einsum_labels
from helpers import crandn
bad = 0
for seed in range(40):
    rng = np.random.default_rng(seed)
    exts = [3] if seed % 2 == 0 else [2, 3, 5, 6, 7]
    eq, sa, sb = random_step(rng, exts, int(rng.integers(3, 9)), int(rng.integers(1, 5)), int(rng.integers(0, 5)), int(rng.integers(0, 2)) if seed % 2 else 0)
    a, b = crandn(rng, sa), crandn(rng, sb)
    info = A.step_info(eq, sa, sb)
    got = A.contract(eq, torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()).cpu().numpy()
    want = einsum_labels(eq, a, b)
    err = np.abs(got - want).max() / np.abs(want).max()
    ok = err < 3e-6
    bad += not ok
    print(seed, "kernel", info["kernel"], "tiles", info["n_tiles"], "err %.2e" % err, "" if ok else "FAIL", flush=True)
print("bad", bad)
# benchmark-size steps of the D=3 112-vertex network
steps = [
 ("KLOPMCSDTEFRUNVHIW,ABPSTEFGQRUNVHJW->AKLOBMCDGQIJ", 18, 16),
 ("IJQNOKPASRBCDMFGH,NKLBTEUVF->IJQVOLPASRCTDEMUGH", 17, 9),
 ("HIPUMJNARQCSDEKTFG,PLNBOG->HILUMJAQRBCSDEOKTF", 18, 6),
 ("HIFRJOATKUBCDQ,MNOPKTSLEQG->HIRMNJPAUSBCDLEFG", 14, 11),
 ("HILSMJAOPCDQEFNKRG,SBTE->HILMJAOBPCDQFNKRGT", 18, 4),
]
for eq, ra, rb in steps:
    a = torch.randn((3,) * ra, dtype=torch.complex64, device="cuda")
    b = torch.randn((3,) * rb, dtype=torch.complex64, device="cuda")
    info = A.step_info(eq, (3,) * ra, (3,) * rb)
    out = A.contract(eq, a, b)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3):
        out = A.contract(eq, a, b)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 3
    print(eq, "kernel", info["kernel"], "tiles", info["n_tiles"], "%.3f ms  %.1f TFLOP/s  %.2f TB/s" % (dt * 1e3, info["flops"] / dt / 1e12, info["bytes"] / dt / 1e12), flush=True)
    # spot check against a c128 einsum on a slice of the output (first label of the output fixed to 1)
