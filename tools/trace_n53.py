#!/usr/bin/env python3
"""Run a few n53 slices (for rocprofv3 --kernel-trace): python3 tools/trace_n53.py [graph|eager]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import artensor_amd as A
from artensor_amd.fixtures import load_case
case = load_case(os.path.join(ROOT, "tests", "golden", "n53_m14_sliced.npz"))
leaves = case.fresh_tensors(device="cuda")
mode = sys.argv[1] if len(sys.argv) > 1 else "eager"
r = A.SliceRunner(leaves, case.scheme, case.slicing_indices, (1,), sparse=True, device="cuda", graph=(mode == "graph"))
r.run(range(0, 6))
torch.cuda.synchronize()
