#!/usr/bin/env python3
"""One workload leg of bench.py, UNITS times, with nothing else in the process (for rocprofv3 --pmc passes: the HBM
traffic of every kernel family per unit of work -- tools/profile_traffic.sh).

    python3 tools/trace_leg.py LEG UNITS      LEG: n30 | n30_sparse10000 | n30_c128 | n30_sliced3 | n53 | n53m20 | n53m20b |
                                                   n53m20b_bf16 | rand2 | rand4 | rand3 | rand6 | n53m20bb | n53m20bb_bf16
A unit is one whole contraction (n30*, rand4) or one slice (the sliced fixtures, Gray order from slice 0).  Prints
"units N" on the last line."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import artensor_amd as A  # noqa: E402
from artensor_amd.fixtures import load_case  # noqa: E402

SLICED = {"n53": ("n53_m14_sliced.npz", True), "n53m20": ("n53_m20_sliced.npz", True), "n53m20b": ("n53_m20_batch.npz", True),
          "n53m20b_bf16": ("n53_m20_batch.npz", True), "rand2": ("rand_D2_nv260_sliced.npz", False), "rand4": ("rand_D4_nv100.npz", False),
          "rand3": ("rand_D3_nv112.npz", False), "rand6": ("rand_D6_nv64.npz", False),
          "n53m20bb": ("n53_m20_bigbatch.npz", True), "n53m20bb_bf16": ("n53_m20_bigbatch.npz", True)}
leg, units = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 2
G = os.path.join(ROOT, "tests", "golden")
dev = "cuda:0"
with A.precision("bf16" if leg.endswith("_bf16") else "fp32"):
    if leg in ("n30", "n30_c128"):
        case = load_case(os.path.join(G, "n30_dense.npz"))
        leaves = case.fresh_tensors(dtype=torch.complex128 if leg == "n30_c128" else torch.complex64, device=dev)
        for _ in range(units):
            out = A.tensor_contraction(dict(leaves), case.scheme)
            del out
    elif leg == "n30_sparse10000":
        case = load_case(os.path.join(G, "n30_sparse10000.npz"))
        leaves = case.fresh_tensors(device=dev)
        for _ in range(units):
            A.tensor_contraction_sparse(dict(leaves), case.scheme)
    elif leg == "n30_sliced3":
        case = load_case(os.path.join(G, "n30_dense_sliced3.npz"))
        runner = A.SliceRunner(case.fresh_tensors(device=dev), case.scheme, case.slicing_indices, (2,) * 30, device=dev)
        runner.run(range(units))
    else:
        fixture, sparse = SLICED[leg]
        case = load_case(os.path.join(G, fixture))
        rows = len(case.meta["bitstrings_sorted"]) if sparse else 1
        n_b = len(case.slicing_indices or {})
        runner = A.SliceRunner(case.fresh_tensors(device=dev), case.scheme, case.slicing_indices, (rows,), sparse=sparse, device=dev)
        mine = [(q ^ (q >> 1)) % (2 ** n_b) if n_b else 0 for q in range(units)]
        A.sliced_contraction(None, case.scheme, case.slicing_indices, (rows,), sparse=sparse, device=dev, slices=mine, reduce=None,
                             runner=runner)
torch.cuda.synchronize()
print("units", units)
