#!/usr/bin/env python3
"""Development check of the two-operand GEMM kernel on the GPU.
    python tools/gemm_check.py correct     random steps forced onto the GEMM plan (ARTN_GEMM=2) vs a complex128 einsum
    python tools/gemm_check.py time        the heavy captured steps (tools/heavy_steps.json) + synthetic big x big steps,
                                           timed under whatever ARTN_GEMM says (run twice: ARTN_GEMM=0 and =1)
"""
import json
import os
import string
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import artensor_amd as A
from artensor_amd import contraction as C

mode = sys.argv[1] if len(sys.argv) > 1 else "correct"
dev = "cuda"
BF16 = os.environ.get("PREC") == "bf16"
if BF16:
    A.precision("bf16").__enter__()


def bf(x):
    t = torch.from_numpy(np.ascontiguousarray(x))
    r = torch.view_as_real(t).to(torch.bfloat16).to(torch.float32)
    return torch.view_as_complex(r.contiguous()).numpy()


def crandn(rng, shape):
    return (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(np.complex64)


def einsum128(la, lb, lo, a, b):
    allab = list(dict.fromkeys(list(la) + list(lb)))
    mp = {x: string.ascii_letters[i] for i, x in enumerate(allab)}
    eq = "".join(mp[x] for x in la) + "," + "".join(mp[x] for x in lb) + "->" + "".join(mp[x] for x in lo)
    return np.einsum(eq, a.astype(np.complex128), b.astype(np.complex128))


if mode == "correct":
    assert os.environ.get("ARTN_GEMM") == "2", "run with ARTN_GEMM=2"
    bad = 0
    for (m, n, k, batch) in [(7, 7, 4, 0), (11, 11, 5, 0), (12, 11, 4, 0), (11, 12, 6, 0), (9, 8, 6, 0), (12, 3, 8, 0), (8, 8, 9, 0), (6, 10, 5, 0), (14, 0, 7, 0), (3, 12, 6, 0),
                             (10, 6, 7, 3), (9, 9, 4, 5), (11, 2, 10, 0), (5, 5, 12, 0), (13, 1, 4, 0), (8, 4, 11, 7)]:
        for seed in range(3):
            rng = np.random.default_rng(1000 * m + 10 * n + k + seed)
            ml = [f"m{x}" for x in range(m)]
            kl = [f"k{x}" for x in range(k)]
            nl = [f"n{x}" for x in range(n)]
            la, lb, lo = ml + kl, kl + nl, ml + nl
            rng.shuffle(la); rng.shuffle(lb); rng.shuffle(lo)
            sa, sb = [2] * len(la), [2] * len(lb)
            if batch:
                la, lb, lo = ["z"] + la, ["z"] + lb, ["z"] + lo
                sa, sb = [batch] + sa, [batch] + sb
            a, b = crandn(rng, sa), crandn(rng, sb)
            info = A.step_info((tuple(la), tuple(lb), tuple(lo)), a.shape, b.shape)
            got = A.contract((tuple(la), tuple(lb), tuple(lo)), torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)).cpu().numpy()
            want = einsum128(la, lb, lo, bf(a), bf(b)) if BF16 else einsum128(la, lb, lo, a, b)
            err = np.abs(got - want).max() / np.abs(want).max()
            ok = err < 1e-5 and info["kernel"] == 2
            bad += not ok
            print(f"m={m} n={n} k={k} batch={batch} seed={seed}: kernel={info['kernel']} mt={info['m_tile_bits']} nt={info['n_tile_bits']} "
                  f"tiles={info['n_tiles']} err={err:.2e} {'ok' if ok else 'FAIL ' + info.get('note', '')}", flush=True)
    print("FAILURES:", bad)
    sys.exit(1 if bad else 0)

# ---- timing
gen = torch.Generator(device=dev).manual_seed(0)


def rnd(shape, stride=None):
    n = 1
    for e in shape:
        n *= e
    t = torch.view_as_complex(torch.randn((n, 2), device=dev, generator=gen)).reshape(shape)
    return t


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


steps = json.load(open(os.path.join(ROOT, "tools", "heavy_steps.json")))
seen = set()
tot = {}
for st in steps:
    la, lb, lo = tuple(st["la"]), tuple(st["lb"]), tuple(st["lo"])
    if st["a_stride"] != list(C._dense_strides(tuple(st["a_shape"]))) or st["b_stride"] != list(C._dense_strides(tuple(st["b_shape"]))):
        continue
    key = (st["case"], la, lb, lo)
    if key in seen:
        continue
    seen.add(key)
    info = A.step_info((la, lb, lo), st["a_shape"], st["b_shape"])
    kb = sum(1 for x in la if x in lb and x not in lo)
    if kb < int(os.environ.get("MINK", "7")) and info["kernel"] != 2:
        continue
    a, b = rnd(st["a_shape"]), rnd(st["b_shape"])
    ms = timed(lambda: A.contract((la, lb, lo), a, b))
    fl = info["flops"]
    tot[st["case"]] = tot.get(st["case"], 0.0) + ms
    print(f"{st['case']:16s} A 2^{np.log2(a.numel()):.0f} B 2^{np.log2(b.numel()):.0f} K {kb:2d} kernel={info['kernel']} mt={info['m_tile_bits']} nt={info['n_tile_bits']} "
          f"tiles={info['n_tiles']:7d} {ms:8.3f} ms {fl / ms / 1e9:7.1f} TF/s", flush=True)
    del a, b
print("total ms per case:", {k: round(v, 2) for k, v in tot.items()})

for (m, n, k) in [(14, 14, 8), (15, 14, 10), (12, 12, 12), (13, 13, 13)] + ([(15, 14, 15)] if os.environ.get("BIG") else []):
    rng = np.random.default_rng(m * 100 + n)
    ml = [f"m{x}" for x in range(m)]
    kl = [f"k{x}" for x in range(k)]
    nl = [f"n{x}" for x in range(n)]
    la, lb, lo = ml + kl, kl + nl, ml + nl
    rng.shuffle(la); rng.shuffle(lb); rng.shuffle(lo)
    a, b = rnd([2] * len(la)), rnd([2] * len(lb))
    info = A.step_info((tuple(la), tuple(lb), tuple(lo)), a.shape, b.shape)
    t0 = time.perf_counter()
    ms = timed(lambda: A.contract((tuple(la), tuple(lb), tuple(lo)), a, b), reps=1 if k >= 13 else 2)
    fl = 8.0 * 2.0 ** (m + n + k)
    print(f"synthetic m={m} n={n} k={k}: kernel={info['kernel']} mt={info['m_tile_bits']} nt={info['n_tile_bits']} tiles={info['n_tiles']} "
          f"{ms:9.2f} ms {fl / ms / 1e9:7.1f} TF/s", flush=True)
    del a, b
