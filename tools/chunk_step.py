#!/usr/bin/env python3
"""Diagnostic: the chunked (gathered) steps of the n30 x 10000-bitstring scheme, one of them replayed alone."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import artensor_amd as A
from artensor_amd import contraction as C
from artensor_amd.fixtures import load_case
case = load_case(os.path.join(ROOT, "tests", "golden", "n30_sparse10000.npz"))
rec = []
orig = C.contract_gathered
def hook(eq, a, rows_a, b, rows_b, out=None, label=None, _validate=True):
    if a.numel() >= 1 << 22 and label is None:
        rec.append((eq, a, rows_a, b, rows_b))
    return orig(eq, a, rows_a, b, rows_b, out=out, label=label, _validate=_validate)
C.contract_gathered = hook
A.tensor_contraction_sparse(case.fresh_tensors(device="cuda"), case.scheme)
torch.cuda.synchronize()
C.contract_gathered = orig
print("gathered steps recorded:", len(rec))
eq, a, ra, b, rb = rec[len(rec) // 2]
la, lb, lo = C._parse(eq) if isinstance(eq, str) else eq
print("la", la, "\nlb", lb, "\nlo", lo)
print("a", tuple(a.shape), a.stride(), "rows", None if ra is None else len(ra), "distinct", None if ra is None else len(set(ra.tolist())))
print("b", tuple(b.shape), b.stride(), "rows", None if rb is None else len(rb), "distinct", None if rb is None else len(set(rb.tolist())))
def timeit(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
t = timeit(lambda: orig(eq, a, ra, b, rb))
info = C.step_info if False else None
print(f"contract_gathered: {t:.3f} ms")
ia = torch.as_tensor(ra, device="cuda") if ra is not None else None
ib = torch.as_tensor(rb, device="cuda") if rb is not None else None
ga = a[ia] if ia is not None else a
gb = b[ib] if ib is not None else b
print(f"gather A alone (torch index): {timeit(lambda: a[ia]):.3f} ms for {ga.numel() * 8 / 2**20:.0f} MiB")
print(f"plain contract on gathered operands: {timeit(lambda: C.contract(eq, ga, gb)):.3f} ms")
print(f"copy of gathered A (clone): {timeit(lambda: ga.clone()):.3f} ms")
si = C.step_info(eq, tuple(ga.shape), tuple(gb.shape))
print({k: si[k] for k in ("kernel", "k_bits", "tile_in_bits", "tile_out_bits", "n_tiles", "run_in_bits", "run_out_bits", "lds_bytes", "grid")})
