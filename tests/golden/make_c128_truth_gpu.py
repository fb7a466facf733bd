#!/usr/bin/env python3
"""complex128 truth for the big cases, computed ON THE GPU BOX by this package's own complex128 path
(artn_k_gemm128 on v_mfma_f64_16x16x4_f64 + the strided kernel; 1e-12 against a complex128 einsum,
tests/test_gpu_parity.py::test_complex128_on_the_matrix_cores), on the same complex64 leaves (widened
exactly) and the same schemes as the fixtures.  The reference's own complex128 run of these cases does not
fit the build container (n30 dense alone is 2 x 16 GiB); 288 GB of HBM holds all of them.

    python tests/golden/make_c128_truth_gpu.py truth          -> gpurun_out/truth/c128_truth_gpu.npz
    python tests/golden/make_c128_truth_gpu.py c64 <tag>      -> gpurun_out/truth/c64_<tag>.npz   (HIP complex64, env as set)
    python tests/golden/make_c128_truth_gpu.py report         -> gpurun_out/truth/report.md + .json

`tools/truth_round.sh` runs the three in order (c64 twice: 3M arithmetic on = default, and
ARTN_BITS_3M=0 ARTN_GEMM_3M=0).  The committed copy of the truth lives in tests/golden/c128_truth_gpu.npz;
what the tests read from it are the amplitudes they check (Google's 10 000 positions, the strided probe, block
sums, slice values) -- never a full state.
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
OUT = os.path.join(ROOT, "gpurun_out", "truth")
GOLDEN = os.path.join(ROOT, "tests", "golden")
DEV = "cuda:0"

import artensor_amd as A  # noqa: E402
from artensor_amd.fixtures import load_case  # noqa: E402


def raw_index(final_pos, perm, n=30):
    out = np.zeros_like(final_pos)
    for d in range(n):
        out |= ((final_pos >> (n - 1 - d)) & 1) << (n - 1 - perm[d])
    return out


def block_sums(raw, lead):
    n = raw.dim()
    flat = raw.reshape(-1)
    out = torch.zeros(2 ** len(lead), dtype=torch.complex128, device=raw.device)
    chunk = 2 ** 26
    for s in range(0, flat.numel(), chunk):
        idx = torch.arange(s, min(s + chunk, flat.numel()), device=raw.device, dtype=torch.int64)
        blk = torch.zeros_like(idx)
        for r, d in enumerate(lead):
            blk |= ((idx >> (n - 1 - d)) & 1) << (len(lead) - 1 - r)
        out.index_add_(0, blk, flat[s:s + chunk].to(torch.complex128))
    return out.cpu().numpy()


def n30_dense(dtype):
    case = load_case(os.path.join(GOLDEN, "n30_dense.npz"))
    raw = A.tensor_contraction(case.fresh_tensors(dtype=dtype, device=DEV), case.scheme)
    perm = case.meta["permute_dims"]
    flat = raw.reshape(-1)
    fpos = np.array([int(b, 2) for b in case.meta["google_bitstrings"]], dtype=np.int64)
    at = flat[torch.from_numpy(raw_index(fpos, perm)).to(DEV)].cpu().numpy()
    spos = np.arange(len(case.arrays["strided"]), dtype=np.int64) * (2 ** 14 + 1)
    strided = flat[torch.from_numpy(raw_index(spos, perm)).to(DEV)].cpu().numpy()
    blocks = block_sums(raw, perm[:10])
    norm2 = float((flat.real.double() ** 2 + flat.imag.double() ** 2).sum())
    return {"n30_dense_at_google": at, "n30_dense_strided": strided, "n30_dense_block_sums": blocks,
            "n30_dense_norm2": np.array(norm2)}


def sparse_whole(name, dtype):
    case = load_case(os.path.join(GOLDEN, name + ".npz"))
    out = A.tensor_contraction_sparse(case.fresh_tensors(dtype=dtype, device=DEV), case.scheme)
    return {name + "_final": out.reshape(-1).cpu().numpy()}


def slice0(name, sparse, dtype):
    case = load_case(os.path.join(GOLDEN, name + ".npz"))
    leaves = case.fresh_tensors(dtype=dtype, device=DEV)
    n_b = len(case.slicing_indices or {})
    sliced = A.apply_slice(leaves, case.slicing_indices, A.slice_assignments(n_b, 0)) if n_b else leaves
    ex = A.tensor_contraction_sparse if sparse else A.tensor_contraction
    return {name + "_slice0": ex(sliced, case.scheme).reshape(-1).cpu().numpy()}


CASES = [
    ("n30_dense", lambda dt: n30_dense(dt)),
    ("n30_sparse10000", lambda dt: sparse_whole("n30_sparse10000", dt)),
    ("n30_sparse100", lambda dt: sparse_whole("n30_sparse100", dt)),
    ("n53_m14_sliced", lambda dt: slice0("n53_m14_sliced", True, dt)),
    ("n53_m20_sliced", lambda dt: slice0("n53_m20_sliced", True, dt)),
    ("n53_m20_batch", lambda dt: slice0("n53_m20_batch", True, dt)),
    ("n53_m20_bigbatch", lambda dt: slice0("n53_m20_bigbatch", True, dt)),   # round 6: 65 536 amplitudes
    ("rand_D2_nv260_sliced", lambda dt: slice0("rand_D2_nv260_sliced", False, dt)),
    ("rand_D4_nv100", lambda dt: slice0("rand_D4_nv100", False, dt)),
]

# the reference's complex64 value of the same quantity: (fixture, array) per key of the truth file
REFERENCE_C64 = {
    "n30_dense_at_google": ("n30_dense", "amps_at_google"), "n30_dense_strided": ("n30_dense", "strided"),
    "n30_dense_block_sums": ("n30_dense", "block_sums"),
    "n30_sparse10000_final": ("n30_sparse10000", "final"), "n30_sparse100_final": ("n30_sparse100", "final"),
    "n53_m14_sliced_slice0": ("n53_m14_sliced", "slice0"), "n53_m20_sliced_slice0": ("n53_m20_sliced", "slice0"),
    "n53_m20_batch_slice0": ("n53_m20_batch", "slice0"), "n53_m20_bigbatch_slice0": ("n53_m20_bigbatch", "slice0"),
    "rand_D2_nv260_sliced_slice0": ("rand_D2_nv260_sliced", "slice0"),
    "rand_D4_nv100_slice0": ("rand_D4_nv100", "slice0"),
}


def run_all(dtype, path, only=None):
    os.makedirs(OUT, exist_ok=True)
    res, secs = {}, {}
    for name, fn in CASES:
        if only and name not in only:
            continue
        torch.cuda.synchronize()
        t0 = time.time()
        try:
            res.update(fn(dtype))
        except Exception:   # keep what the other cases give; the report lists what is there
            import traceback
            traceback.print_exc()
            print(f"{name}: FAILED", flush=True)
            continue
        torch.cuda.synchronize()
        secs[name] = time.time() - t0
        print(f"{name}: {secs[name]:.1f} s", flush=True)
        torch.cuda.empty_cache()
    res["meta"] = np.frombuffer(json.dumps({
        "dtype": str(dtype), "seconds": secs, "env": {k: v for k, v in os.environ.items() if k.startswith("ARTN_")},
        "made_by": "tests/golden/make_c128_truth_gpu.py on one MI355X (this package's complex128 / complex64 path)"}).encode(),
        dtype=np.uint8)
    np.savez_compressed(path, **res)


def figures(got, want, rms=None):
    """loose: max |got-want| / max(|want|, rms); strict: max relative error over |want| >= 1e-3 rms."""
    got, want = np.asarray(got, dtype=np.complex128).reshape(-1), np.asarray(want, dtype=np.complex128).reshape(-1)
    if rms is None:
        rms = float(np.sqrt(np.mean(np.abs(want) ** 2)))
    d = np.abs(got - want)
    loose = float((d / np.maximum(np.abs(want), rms)).max())
    sel = np.abs(want) >= 1e-3 * rms
    strict = float((d[sel] / np.abs(want)[sel]).max()) if sel.any() else 0.0
    return loose, strict


def report():
    truth = np.load(os.path.join(OUT, "c128_truth_gpu.npz"))
    runs = {}
    for f in sorted(os.listdir(OUT)):
        if f.startswith("c64_") and f.endswith(".npz"):
            runs[f[4:-4]] = np.load(os.path.join(OUT, f))
    rows, table = [], {}
    for key, (fixture, arr) in REFERENCE_C64.items():
        if key not in truth.files:
            continue
        want = truth[key]
        rms = 2.0 ** -15 if key.startswith("n30_dense_at") or key.startswith("n30_dense_str") else None
        ref = load_case(os.path.join(GOLDEN, fixture + ".npz")).arrays.get(arr)
        rec = {}
        if ref is not None:
            rec["reference_c64"] = figures(ref, want, rms)
        for tag, z in runs.items():
            if key in z.files:
                rec["hip_c64_" + tag] = figures(z[key], want, rms)
                if ref is not None:
                    rec["hip_c64_" + tag + "_vs_reference_c64"] = figures(z[key], ref, rms)
        table[key] = {k: {"loose": v[0], "strict": v[1]} for k, v in rec.items()}
        rows.append((key, len(np.asarray(want).reshape(-1)), rec))
    with open(os.path.join(OUT, "report.json"), "w") as f:
        json.dump(table, f, indent=1)
    with open(os.path.join(OUT, "report.md"), "w") as f:
        f.write("| quantity | n | who | loose vs c128 truth | strict vs c128 truth |\n|---|---|---|---|---|\n")
        for key, n, rec in rows:
            for who, (lo, st) in rec.items():
                f.write(f"| {key} | {n} | {who} | {lo:.2e} | {st:.2e} |\n")
    print(open(os.path.join(OUT, "report.md")).read())


if __name__ == "__main__":
    mode = sys.argv[1]
    if mode == "truth":
        run_all(torch.complex128, os.path.join(OUT, "c128_truth_gpu.npz"), sys.argv[2:] or None)
    elif mode == "c64":
        run_all(torch.complex64, os.path.join(OUT, f"c64_{sys.argv[2]}.npz"), sys.argv[3:] or None)
    elif mode == "report":
        report()
    else:
        sys.exit(__doc__)
