#!/usr/bin/env python3
"""Run the n30 sparse fixtures a few times (for rocprofv3 --kernel-trace --stats)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import artensor_amd as A
from artensor_amd.fixtures import load_case
tag = sys.argv[1] if len(sys.argv) > 1 else "n30_sparse10000"
case = load_case(os.path.join(ROOT, "tests", "golden", tag + ".npz"))
for _ in range(3):
    A.tensor_contraction_sparse(case.fresh_tensors(device="cuda"), case.scheme)
torch.cuda.synchronize()
