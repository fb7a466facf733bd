#!/usr/bin/env python3
"""Run the n30 dense fixture in complex128 a few times (for rocprofv3 --kernel-trace --stats / --pmc)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import artensor_amd as A
from artensor_amd.fixtures import load_case
case = load_case(os.path.join(ROOT, "tests", "golden", "n30_dense.npz"))
leaves = case.fresh_tensors(dtype=torch.complex128, device="cuda")
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    out = A.tensor_contraction(dict(leaves), case.scheme)
    del out
torch.cuda.synchronize()
