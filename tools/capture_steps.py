#!/usr/bin/env python3
"""Record the heavy pairwise steps (labels, shapes, strides) of one n53 slice / the n30 sparse runs
to gpurun_out/heavy_steps.json, so single steps can be replayed under the diagnostic builds."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import artensor_amd as A
from artensor_amd import contraction as C
from artensor_amd.fixtures import load_case
rec = []
orig = C.contract
def wrapped(eq, a, b, out=None):
    la, lb, lo = C._parse(eq) if isinstance(eq, str) else eq
    if a.numel() >= 1 << 24:
        rec.append(dict(case=tag, la=[str(x) for x in la], lb=[str(x) for x in lb], lo=[str(x) for x in lo],
                        a_shape=list(a.shape), a_stride=list(a.stride()), b_shape=list(b.shape), b_stride=list(b.stride())))
    return orig(eq, a, b, out)
C.contract = wrapped
os.environ["ARTN_NO_FUSE"] = "1"
tag = "n53"
case = load_case(os.path.join(ROOT, "tests", "golden", "n53_m14_sliced.npz"))
leaves = case.fresh_tensors(device="cuda")
sl = A.apply_slice(leaves, case.slicing_indices, A.slice_assignments(len(case.slicing_indices), 1))
A.tensor_contraction_sparse(sl, case.scheme)
for tag in ("n30_sparse10000", "n30_sparse100"):
    case = load_case(os.path.join(ROOT, "tests", "golden", tag + ".npz"))
    A.tensor_contraction_sparse(case.fresh_tensors(device="cuda"), case.scheme)
torch.cuda.synchronize()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rec, open(os.path.join(ROOT, "gpurun_out", "heavy_steps.json"), "w"))
print(len(rec), "steps")
