#!/usr/bin/env python3
"""An INDEPENDENT complex128 truth for the big cases: the reference's executor loop (torch.einsum step by step,
oracle.tensor_contraction[_sparse]_torch_cpu -- /root/reference/artensor/contraction.py:62-76, :132-205) run in complex128
by torch on the HOST cores of the GPU box, on the same complex64 leaves (widened exactly) and schemes as the fixtures.
No planner, descriptor or kernel of this package is involved.  The build container cannot hold these runs (2^30-element
complex128 intermediates, 16 GiB each, three alive per step); the GPU box's host can.

    python tests/golden/make_c128_truth_cpu.py [key ...]   ->  gpurun_out/truth_cpu/c128_truth_torch_cpu.npz + report.json

The committed copy lives in tests/golden/c128_truth_torch_cpu.npz; tests/test_oracle.py::test_gpu_truth_equals_the_independent_cpu_truth
compares tests/golden/c128_truth_gpu.npz (this package's f64-MFMA path) with it key by key, on the CPU, every round.
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from artensor_amd.fixtures import load_case  # noqa: E402  (fixture loader only: pure numpy / torch-CPU)
from artensor_amd.simulation import apply_slice, slice_assignments  # noqa: E402  (host-side indexing, no kernels)
from oracle import oracle  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
OUT = os.path.join(ROOT, "gpurun_out", "truth_cpu")
CASES = {   # key in c128_truth_gpu.npz -> (fixture, sparse executor, sliced)
    "rand_D4_nv100_slice0": ("rand_D4_nv100.npz", False, True),
    "n30_sparse100_final": ("n30_sparse100.npz", True, False),
    "rand_D2_nv260_sliced_slice0": ("rand_D2_nv260_sliced.npz", False, True),
    "n53_m14_sliced_slice0": ("n53_m14_sliced.npz", True, True),
    "n53_m20_sliced_slice0": ("n53_m20_sliced.npz", True, True),
    "n30_sparse10000_final": ("n30_sparse10000.npz", True, False),
    "n30_dense_at_google": ("n30_dense.npz", False, False),
    "n53_m20_batch_slice0": ("n53_m20_batch.npz", True, True),
    "n53_m20_bigbatch_slice0": ("n53_m20_bigbatch.npz", True, True),   # round 6: 65 536 amplitudes
}


def main(keys):
    os.makedirs(OUT, exist_ok=True)
    mem_gib = os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES") / 2 ** 30
    out, report = {}, {"threads": torch.get_num_threads(), "host_memory_GiB": mem_gib, "cases": {}}
    path = os.path.join(OUT, "c128_truth_torch_cpu.npz")
    if os.path.exists(path):
        out.update({k: v for k, v in np.load(path).items()})
    for key in keys:
        fixture, sparse, sliced = CASES[key]
        case = load_case(os.path.join(GOLDEN, fixture))
        if key == "n30_dense_at_google" and mem_gib < 100:
            report["cases"][key] = {"skipped": f"host memory {mem_gib:.0f} GiB < 100 GiB"}
            continue
        leaves = case.fresh_tensors(dtype=torch.complex128)
        if sliced and case.slicing_indices:
            leaves = apply_slice(leaves, case.slicing_indices, slice_assignments(len(case.slicing_indices), 0))
        t0 = time.time()
        fn = oracle.tensor_contraction_sparse_torch_cpu if sparse else oracle.tensor_contraction_torch_cpu
        res = fn(leaves, case.scheme)["result"]
        assert res.dtype == torch.complex128
        if key == "n30_dense_at_google":
            perm = case.meta["permute_dims"]
            fpos = np.array([int(b, 2) for b in case.meta["google_bitstrings"]], dtype=np.int64)
            rpos = np.zeros_like(fpos)
            for d in range(30):
                rpos |= ((fpos >> (29 - d)) & 1) << (29 - perm[d])
            val = res.reshape(-1)[torch.from_numpy(rpos)].numpy().copy()
        else:
            val = res.reshape(-1).numpy().copy()
        del res, leaves
        out[key] = val
        report["cases"][key] = {"seconds": time.time() - t0, "n": int(val.size)}
        print(key, val.shape, f"{time.time() - t0:.1f} s", flush=True)
        np.savez_compressed(path, **out)
        json.dump(report, open(os.path.join(OUT, "report.json"), "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1:] or list(CASES))
