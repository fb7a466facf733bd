#!/usr/bin/env python3
"""Randomised stress of the planner + kernels on the GPU: single steps (contract), fused pairs (contract2) and gathered
steps (contract_gathered) with random label orders, contracted / free bit counts and batch extents, complex64 and
complex128, against torch.einsum in complex128 on the host.   python tools/stress_random.py [cases] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import artensor_amd as A
from artensor_amd import contraction as C

def rnd(gen, shape, dtype):
    x = torch.randn(tuple(shape) + (2,), device="cuda", generator=gen, dtype=torch.float64)
    return torch.view_as_complex(x).to(dtype).contiguous()

def main(cases=200, seed=0):
    rng = np.random.default_rng(seed)
    gen = torch.Generator(device="cuda").manual_seed(seed)
    worst = {torch.complex64: 0.0, torch.complex128: 0.0}
    kinds = {"single": 0, "pair": 0, "pair_fused": 0, "gather": 0}
    t0 = time.time()
    for case in range(cases):
        dtype = torch.complex128 if rng.random() < 0.3 else torch.complex64
        tol = 1e-11 if dtype == torch.complex128 else 2e-5
        kind = rng.choice(["single", "pair", "gather"], p=[0.45, 0.35, 0.2])
        ra = int(rng.integers(10, 21))
        k1 = int(rng.integers(1, 9))
        n1 = int(rng.integers(0, 8))
        if k1 > ra - 5:
            k1 = ra - 5
        la = [chr(65 + x) for x in range(ra)]
        kl1 = list(rng.choice(la, size=k1, replace=False))
        nl1 = [chr(97 + x) for x in range(n1)]
        lb1 = kl1 + nl1
        rng.shuffle(lb1)
        lo1 = [x for x in la if x not in kl1] + nl1
        rng.shuffle(lo1)
        batch = int(rng.choice([0, 0, 3, 4, 7])) if kind != "gather" else int(rng.integers(2, 40))
        if batch:
            la_, lb1_, lo1_ = ["z"] + la, ["z"] + lb1, ["z"] + lo1
        else:
            la_, lb1_, lo1_ = la, lb1, lo1
        ext = lambda labs: tuple(batch if x == "z" else 2 for x in labs)
        eq1 = "".join(la_) + "," + "".join(lb1_) + "->" + "".join(lo1_)
        if kind == "gather":
            na, nb = int(rng.integers(1, 12)), int(rng.integers(1, 12))
            a = rnd(gen, (na,) + ext(la_)[1:], dtype)
            b = rnd(gen, (nb,) + ext(lb1_)[1:], dtype)
            ia, ib = torch.from_numpy(rng.integers(0, na, size=batch)), torch.from_numpy(rng.integers(0, nb, size=batch))
            got = C.contract_gathered(eq1, a, ia, b, ib)
            want = torch.einsum(eq1, a.cpu()[ia].to(torch.complex128), b.cpu()[ib].to(torch.complex128))
            if got is None:
                got = A.contract(eq1, a[ia.cuda()].contiguous(), b[ib.cuda()].contiguous())
            kinds["gather"] += 1
        else:
            a, b1 = rnd(gen, ext(la_), dtype), rnd(gen, ext(lb1_), dtype)
            if kind == "single":
                got = A.contract(eq1, a, b1)
                want = torch.einsum(eq1, a.cpu().to(torch.complex128), b1.cpu().to(torch.complex128))
                kinds["single"] += 1
            else:
                k2 = int(rng.integers(1, 7))
                n2 = int(rng.integers(0, 7))
                cand = [x for x in lo1 if True]
                if k2 > len(cand) - 5:
                    k2 = max(1, len(cand) - 5)
                kl2 = list(rng.choice(cand, size=k2, replace=False))
                nl2 = [chr(110 + x) for x in range(n2)]
                lb2 = kl2 + nl2
                rng.shuffle(lb2)
                lo2 = [x for x in lo1 if x not in kl2] + nl2
                rng.shuffle(lo2)
                lb2_, lo2_ = (["z"] + lb2, ["z"] + lo2) if batch and rng.random() < 0.5 else (lb2, (["z"] if batch else []) + lo2)
                eq2 = "".join(lo1_) + "," + "".join(lb2_) + "->" + "".join(lo2_)
                b2 = rnd(gen, ext(lb2_), dtype)
                got = C.contract2(eq1, a, b1, eq2, b2)
                kinds["pair"] += 1
                if got is None:
                    got = A.contract(eq2, A.contract(eq1, a, b1), b2)
                else:
                    kinds["pair_fused"] += 1
                want = torch.einsum(eq2, torch.einsum(eq1, a.cpu().to(torch.complex128), b1.cpu().to(torch.complex128)), b2.cpu().to(torch.complex128))
        err = float((got.cpu().to(torch.complex128) - want).abs().max() / want.abs().max())
        worst[dtype] = max(worst[dtype], err)
        if not err <= tol:
            print("FAIL", kind, dtype, eq1, locals().get("eq2"), "batch", batch, "err", err)
            return 1
    torch.cuda.synchronize()
    print(f"{cases} cases ok in {time.time() - t0:.1f} s: {kinds}; worst rel err c64 {worst[torch.complex64]:.2e}, c128 {worst[torch.complex128]:.2e}")
    return 0

if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 0))
