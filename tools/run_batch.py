#!/usr/bin/env python3
"""Slice 0 of a sliced sparse fixture (default: the n53 m20 big-batch case) on the GPU in fp32 and
under precision("bf16"): wall per slice, fidelity of bf16 against fp32, and -- when the fixture
carries the reference's slice-0 output -- both error figures against it.  Saves the outputs under
gpurun_out/ so a run made before the reference value existed can be compared later."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import artensor_amd as A
from artensor_amd import contraction as C
from artensor_amd.fixtures import load_case

name = sys.argv[1] if len(sys.argv) > 1 else "n53_m20_batch"
case = load_case(os.path.join(ROOT, "tests", "golden", name + ".npz"))
leaves = case.fresh_tensors(device="cuda")
rows = len(case.meta["bitstrings_sorted"])
flops = 8.0 * 10 ** case.meta["log10_tc"]
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)


class Prof:
    def __init__(s): s.rows = []
    def record(s, info, e0, e1): s.rows.append((info, e0, e1))


def run(mode, reps=2):
    with A.precision(mode):
        r = A.SliceRunner(leaves, case.scheme, case.slicing_indices, (rows,), sparse=True, device="cuda")
        out = r.run([0]).clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(reps):
            r.collect.zero_()
            r.run([0])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        p = Prof()
        C.profiler = p
        r.collect.zero_()
        r.run([0])
        torch.cuda.synchronize()
        C.profiler = None
    rws = sorted(((e0.elapsed_time(e1), info) for info, e0, e1 in p.rows), key=lambda x: -x[0])
    tot = sum(x[0] for x in rws)
    print(f"{name} [{mode}]: {dt * 1e3:.1f} ms per slice = {flops / dt / 1e12:.1f} TFLOP/s; {len(rws)} contract launches, "
          f"{tot:.1f} ms in them", flush=True)
    for ms, info in rws[:int(os.environ.get("TOP", "12"))]:
        print(f"   {ms:7.2f} ms kernel={info['kernel']} k={info['k_bits']}+{info['k2_bits']} T={info['tile_in_bits']}/"
              f"{info['tile_out_bits']} tiles={info['n_tiles']} GF={info['flops'] / 1e9:.0f} -> {info['flops'] / ms / 1e9:.1f} TF/s")
    return out.cpu().numpy()


f32 = run("fp32")
np.save(os.path.join(ROOT, "gpurun_out", name + "_slice0_fp32.npy"), f32)
b16 = run("bf16")
np.save(os.path.join(ROOT, "gpurun_out", name + "_slice0_bf16.npy"), b16)
a, b = f32.astype(np.complex128), b16.astype(np.complex128)
print("bf16 vs fp32 fidelity", abs(np.vdot(a, b)) ** 2 / (np.vdot(a, a).real * np.vdot(b, b).real))
if "slice0" in case.arrays:
    want = case.arrays["slice0"].reshape(-1)
    rms = np.sqrt(np.mean(np.abs(want) ** 2))
    loose = np.abs(f32 - want).max() / max(np.abs(want).max(), rms)
    sel = np.abs(want) >= 1e-3 * rms
    strict = (np.abs(f32 - want)[sel] / np.abs(want)[sel]).max()
    print(f"fp32 vs reference slice 0: loose {loose:.3e}, strict (|amp| >= 1e-3 rms, {sel.sum()} of {len(want)}) {strict:.3e}")
