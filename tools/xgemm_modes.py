#!/usr/bin/env python3
"""A/B of the extent GEMM's copy modes on one step whose label layout is given low -> high stride per tensor:
   python tools/xgemm_modes.py KMMMMMMMMKMMMKMMK KNNKNKKNN [D]   (C = M labels lowest, then N)"""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 4 and sys.argv[4] == "child":
    import torch
    import artensor_amd as A
    a_lo, b_lo, D = sys.argv[1], sys.argv[2], int(sys.argv[3])
    def names(s):
        cnt = {}
        out = []
        for c in s:
            out.append(f"{c}{cnt.get(c, 0)}")
            cnt[c] = cnt.get(c, 0) + 1
        return out
    la, lb = names(a_lo), names(b_lo)
    lo = [x for x in la if x[0] == "M"] + [x for x in lb if x[0] == "N"]
    eq = (tuple(reversed(la)), tuple(reversed(lb)), tuple(reversed(lo)))
    a = torch.randn((D,) * len(la), dtype=torch.complex64, device="cuda")
    b = torch.randn((D,) * len(lb), dtype=torch.complex64, device="cuda")
    info = A.step_info(eq, a.shape, b.shape)
    out = A.contract(eq, a, b)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(5):
        out = A.contract(eq, a, b)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 5
    print(f"AMODE={os.environ.get('ARTN_XG_AMODE', '-')} BMODE={os.environ.get('ARTN_XG_BMODE', '-')} kernel {info['kernel']} tiles {info['n_tiles']}: {dt*1e3:.3f} ms  {info['flops']/dt/1e12:.1f} TFLOP/s  {info['bytes']/dt/1e12:.2f} TB/s", flush=True)
    sys.exit(0)
D = sys.argv[3] if len(sys.argv) > 3 else "3"
for am, bm in ((None, None), ("0", "0"), ("1", "0"), ("0", "1"), ("1", "1")):
    env = dict(os.environ)
    if am is not None:
        env["ARTN_XG_AMODE"], env["ARTN_XG_BMODE"] = am, bm
    subprocess.call([sys.executable, __file__, sys.argv[1], sys.argv[2], D, "child"], env=env)
