#!/usr/bin/env python3
"""bench.py -- contracted TFLOP/s of the hot path on MI355X.

A "step" is one pass of the hot path over one batch of synthetic input: one full
`tensor_contraction` of the Sycamore n30 m14 full-amplitude scheme (BASELINE.json configs[1]:
180 pairwise steps, 5.3705e12 real FLOP, all 2^30 complex64 amplitudes, no slicing) with the
181 leaf tensors already resident in HBM.  Leaf tensors and scheme come from the committed
fixture tests/golden/n30_dense.npz (circuit gate tensors are tiny and fixed: "synthetic"
here means no dataset is read; amplitudes are checked against the fixture's statistics).

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1 (launched by torch.distributed.run, one rank per GPU over RCCL): the unsliced n30
contraction cannot be sharded the reference's way (slicing inner bonds would end in an 8 GiB
all-reduce), so it is partitioned over OUTPUT qubits instead -- a build-side
extension (artensor_amd.partitioned_contraction): rank r fixes log2(N) output labels to the bits of
r at the leaves and contracts its own 2^30 / N slab; slabs are disjoint, there is no exchange.
One step is still ONE full n30 contraction (all 2^30 amplitudes, 5.3705e12 nominal FLOP) done by
the N ranks together: strong scaling.  (Fixing an output label only thins the steps that already
carry it, so the slabs cost more than 1/N of the whole: the line reports that overhead.)  The same
line carries, under "sliced", the slice-sharded workload north_star names -- Sycamore n53 m14,
slices dealt round-robin to the ranks, ONE all-reduce over RCCL closing the timed region.

`python bench.py --gpus N` with N > 1 as a PLAIN command (no WORLD_SIZE in the environment) starts its own N ranks
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` as a CHILD process, before
this process has touched the GPU) and exits with their status; launched by torch.distributed.run it is a rank.

Output (rank 0): one line per secondary workload leg as it finishes (`{"leg": ...}`), the full record in
`bench_detail.json` next to this script (and under gpurun_out/ when that directory exists), and as the LAST stdout line
ONE compact JSON object (< 1.5 KB: metric, value, roofline, cpu_baseline, one [value, frac, check] triple per leg).
Exit code 1 when a result check fails (the line is still printed, with "check": "FAILED").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 matrix peak (spec)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 MFMA, dense (spec, ~2.5 PF)
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E peak (spec)
KERNEL_NAMES = {0: "artn_k_generic", 1: "artn_k_bits", 2: "artn_k_gemm", 3: "artn_k_program", 4: "artn_k_pgemm / artn_k_pgemm3m (+ packing passes)",
                5: "artn_k_xgemm"}
LOOSE_TOL = 1e-5              # |got - want| <= tol * max(|want|, rms(want)) for every amplitude
STRICT_FACTOR = 2.0           # relative error per amplitude over |truth| >= 1e-3 rms, against the complex128 truth
#                               of the same leaves and scheme (tests/golden/c128_truth_gpu.npz, pinned to an independent torch-CPU
#                               complex128 run): allowed up to this many times the distance of the reference's OWN complex64
#                               run from that truth (n30 at Google's 10 000 bitstrings: reference 5.4e-5, this package 4.7-5.2e-5)
BF16_MIN_FIDELITY = 0.99


def kernel_source_sha16():
    """sha256 (first 16 hex digits) of the kernel sources: profiles/rNN_traffic.json records the sources its
    counters were collected on (tools/summarize_profile.py)."""
    import hashlib
    h = hashlib.sha256()
    for name in ("artn_kernels.hip", "artn_gemm_kernel.h", "artn_gemm128_kernel.h", "artn_bits128_kernel.h", "artn_bits3_kernel.h",
                 "artn_wide_kernel.h", "artn_pgemm_kernel.h", "artn_plan.h", "artn_xgemm_kernel.h", "artn_xgemm128_kernel.h", "artn_xgemm_plan.h", "artn_xrow_kernel.h"):
        with open(os.path.join(ROOT, "artensor_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


class KernelTimes:
    """Per-launch HIP-event timing of artn_contract calls inside the timed region."""

    def __init__(self):
        self.rows = []

    def record(self, info, e0, e1):
        self.rows.append((info, e0, e1))

    def summarize(self):
        out = {}
        for info, e0, e1 in self.rows:
            d = out.setdefault(info["kernel"], dict(launches=0, ms=0.0, flops=0.0, bytes=0.0, mfma_flops=0.0, bf16_flops=0.0))
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += info["flops"]
            d["bytes"] += info["bytes"]
            d["mfma_flops"] += info.get("mfma_flops", 0.0)
            if info.get("arith") == 2:
                d["bf16_flops"] += info["flops"]
            if info.get("arith") == 3:
                d["f64_flops"] = d.get("f64_flops", 0.0) + info["flops"]
        return out


_probe = {}


def mfma_rate_probe(kind, dev):
    """TFLOP/s of back-to-back MFMAs with operands in registers, measured on this device (artn_probe_mfma_rate):
    kind 0 fp32 32x32x2, 1 bf16 32x32x16, 2 f64 16x16x4."""
    if kind not in _probe:
        import ctypes
        from artensor_amd import _native as N
        scratch = torch.zeros(4, dtype=torch.uint8, device=dev)
        out = ctypes.c_double(0.0)
        torch.cuda.synchronize()
        N.check(N.lib().artn_probe_mfma_rate(kind, scratch.data_ptr(), ctypes.byref(out)))
        _probe[kind] = float(out.value)
    return _probe[kind]


_traffic = []


def traffic_record():
    """profiles/rNN_traffic.json of the newest round -- HBM bytes from rocprofv3 PMC passes (FETCH_SIZE x 2 on gfx950,
    WRITE_SIZE; tools/profile_traffic.sh) -- if it was collected on THESE kernel sources; else (None, why)."""
    if not _traffic:
        import glob
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
        if not files:
            _traffic.append((None, "no profiles/r*_traffic.json"))
        else:
            with open(files[-1]) as f:
                tj = json.load(f)
            name = os.path.basename(files[-1])
            if tj.get("kernel_source_sha16") == kernel_source_sha16():
                _traffic.append((tj, f"{name} (same kernel sources)"))
            else:   # counters belong to the sources they were collected on: a stale file is not quoted
                _traffic.append((None, f"{name} was collected on other kernel sources "
                                       f"({tj.get('kernel_source_sha16')} != {kernel_source_sha16()}): not quoted"))
    return _traffic[0]


FAMILY_KEYS = {0: "generic", 1: "bits", 2: "gemm", 3: "program", 4: "pgemm", 5: "xgemm"}   # keys of rNN_traffic.json's per-leg tables


def roofline_of(ks, gpu_ms, launches_div=1, f64_peak=None, leg=None, units=1):
    """roofline block of a workload leg from per-launch HIP-event timings (KernelTimes.summarize()): the kernel
    family that takes most of the GPU time; `bound` is whichever roof the family's arithmetic intensity puts it
    under.  HBM-bound: algorithmic bytes / time against 8 TB/s.  MFMA-bound: `achieved` is the NOMINAL rate (8 FLOP
    per complex multiply-add, the metric's count) and `peak` the nominal rate at which the matrix pipe would be
    saturated by this family's mix of arithmetic -- spec peak / (executed FLOP / nominal FLOP): 3M stages execute 6 of
    the 8 counted FLOP, so a pure-3M kernel can reach 4/3 of the spec peak in nominal terms -- so that `frac` is the
    executed-FLOP fraction of the matrix pipe and can never exceed 1; the nominal fraction of the SPEC peak is beside
    it (`mfma_frac_nominal`).  `traffic`: HBM bytes per launch from the PMC counters of profiles/rNN_traffic.json."""
    if not ks:
        return None
    kid = max(ks, key=lambda k: ks[k]["ms"])
    d = ks[kid]
    sec = d["ms"] * 1e-3
    bf16 = d["bf16_flops"] > 0.5 * d["flops"]
    f64 = d.get("f64_flops", 0.0) > 0.5 * d["flops"]
    peak = MFMA_BF16_PEAK_TFLOPS if bf16 else (f64_peak if f64 and f64_peak else MFMA_F32_PEAK_TFLOPS)
    tf = d["flops"] / sec / 1e12 if sec else 0.0
    gbs = d["bytes"] / sec / 1e9 if sec else 0.0
    ai = d["flops"] / d["bytes"] if d["bytes"] else float("inf")
    ridge = peak * 1e12 / (HBM_PEAK_GBS * 1e9)
    mfma_bound = ai >= ridge
    kname = KERNEL_NAMES.get(kid, str(kid))
    if f64:   # (complex128 runs on its own kernels behind the same planner ids)
        kname = {1: "artn_k_bits128", 2: "artn_k_gemm128"}.get(kid, kname)
    exec_frac = d["mfma_flops"] / d["flops"] if d["flops"] and d["mfma_flops"] else 1.0
    eff_peak = peak / exec_frac
    alg_per_launch = d["bytes"] / max(d["launches"], 1)
    traffic, ratio, tnote = None, None, None
    tj, tnote = traffic_record()
    if tj is not None and leg is not None:
        legs = tj.get("workloads", {}).get(leg) or {}
        fam = legs.get(FAMILY_KEYS.get(kid, str(kid)))
        if fam and kid == 5 and legs.get("xrow"):   # (planner id 5 covers artn_k_xgemm and its row-streaming form artn_k_xrow)
            fam = dict(fam, hbm_bytes_per_unit=fam["hbm_bytes_per_unit"] + legs["xrow"]["hbm_bytes_per_unit"])
        if fam:   # bytes per contraction (slice) of this family / its contract calls per contraction
            traffic = fam["hbm_bytes_per_unit"] / (d["launches"] / launches_div / units)
            ratio = traffic / alg_per_launch if alg_per_launch else None
        else:
            tnote += f": no entry for {leg}/{FAMILY_KEYS.get(kid, kid)}"
    r = {"bound": "mfma" if mfma_bound else "hbm", "kernel": kname,
         "achieved": tf if mfma_bound else gbs, "peak": eff_peak if mfma_bound else HBM_PEAK_GBS,
         "unit": "TFLOP/s" if mfma_bound else "GB/s", "frac": (tf / eff_peak) if mfma_bound else gbs / HBM_PEAK_GBS,
         "traffic": traffic, "traffic_over_algorithmic": ratio, "traffic_source": tnote,
         "algorithmic_bytes_per_launch": alg_per_launch,
         "peak_is": ((f"{peak:.1f} TFLOP/s matrix peak / {exec_frac:.3f} (share of the counted FLOP this family executes on the "
                      "matrix pipe: 3M stages run 6 of 8)") if mfma_bound else "HBM3E spec"),
         "arithmetic": "bf16 operands, fp32 accumulate" if bf16 else ("f64 MFMA (peak = measured back-to-back v_mfma_f64_16x16x4_f64 rate)" if f64 else "fp32 MFMA"),
         "nominal_TFLOPs": tf, "mfma_peak_TFLOPs": peak, "mfma_frac_nominal": tf / peak,
         "executed_mfma_flop_frac": exec_frac,
         "mfma_frac_executed": tf * exec_frac / peak,
         "hbm_GBs": gbs, "hbm_frac": gbs / HBM_PEAK_GBS, "flop_per_byte": ai, "ridge_flop_per_byte": ridge,
         "hbm_frac_is": "algorithmic bytes (cold-cache compulsory bytes per launch) / time / 8 TB/s",
         "launches": d["launches"] / launches_div, "avg_launch_ms": d["ms"] / max(d["launches"], 1),
         "kernel_ms": d["ms"] / launches_div, "share_of_gpu_time": d["ms"] / gpu_ms if gpu_ms else None,
         "other_kernels_ms": {KERNEL_NAMES.get(k, str(k)): v["ms"] / launches_div for k, v in ks.items() if k != kid}}
    # The per-launch byte model counts every operand once, from a cold cache.  Where the counters show LESS HBM traffic than
    # that (a producer's output still in the 256 MB MALL / the L2s when its consumer starts; short-lived chunk results
    # overwritten before they are evicted), the model is not a lower bound and the HBM fraction is priced on what really
    # moved (review r04: n30 x 10 000 bitstrings, 0.87)
    if ratio is not None and ratio < 1.0 and sec:
        moved = traffic * max(d["launches"], 1)
        r["hbm_GBs_algorithmic"], r["hbm_frac_algorithmic"] = r["hbm_GBs"], r["hbm_frac"]
        r["hbm_GBs"] = moved / sec / 1e9
        r["hbm_frac"] = r["hbm_GBs"] / HBM_PEAK_GBS
        r["hbm_frac_is"] = ("MEASURED HBM bytes (PMC counters) / time / 8 TB/s: fewer than the cold-cache byte model counts -- "
                            "consecutive launches meet in the MALL / L2; the model's figure is kept as hbm_frac_algorithmic")
        if not mfma_bound:
            r["achieved"], r["frac"] = r["hbm_GBs"], r["hbm_frac"]
    return r


def error_figures(got, want):
    """(loose, strict, n_strict): loose = max |got-want| / max(|want|, rms); strict = max relative error
    over the amplitudes with |want| >= 1e-3 rms (SURVEY 8c's contract)."""
    got, want = np.asarray(got).reshape(-1), np.asarray(want).reshape(-1)
    rms = float(np.sqrt(np.mean(np.abs(want) ** 2)))
    loose = float((np.abs(got - want) / np.maximum(np.abs(want), rms)).max())
    sel = np.abs(want) >= 1e-3 * rms
    strict = float((np.abs(got - want)[sel] / np.abs(want)[sel]).max()) if sel.any() else 0.0
    return loose, strict, int(sel.sum())


def fidelity_of(got, want):
    w, a = np.asarray(want, dtype=np.complex128).reshape(-1), np.asarray(got, dtype=np.complex128).reshape(-1)
    return float(abs(np.vdot(w, a)) ** 2 / (np.vdot(w, w).real * np.vdot(a, a).real))


_truth = []


def truth_value(key):
    """complex128 value of a checked quantity, computed on the GPU by this package's complex128 path and committed
    (tests/golden/make_c128_truth_gpu.py); None when the file or the key is absent."""
    if not _truth:
        path = os.path.join(ROOT, "tests", "golden", "c128_truth_gpu.npz")
        _truth.append(np.load(path) if os.path.exists(path) else None)
    z = _truth[0]
    return z[key].reshape(-1) if z is not None and key in z.files else None


def cpu_leg(case, sparse, sliced, budget_s, what):
    """cpu_baseline of one secondary workload: the reference's executor as it runs on a CPU (torch-CPU einsum loop,
    oracle.tensor_contraction[_sparse]_torch_cpu) on this box's host cores, the scheme's steps in order (slice 0 of
    a sliced workload) until `budget_s` seconds have gone."""
    from oracle import oracle
    import artensor_amd as A
    leaves = {k: t.clone() for k, t in case.tensors.items()}
    if sliced and case.slicing_indices:
        leaves = A.apply_slice(leaves, case.slicing_indices, A.slice_assignments(len(case.slicing_indices), 0))
    fn = oracle.tensor_contraction_sparse_torch_cpu if sparse else oracle.tensor_contraction_torch_cpu
    r = fn(leaves, case.scheme, budget_s=budget_s)
    whole = r["steps_done"] == len(case.scheme)
    return {"value": r["flops_done"] / r["seconds"] / 1e12, "unit": "TFLOP/s", "cores": int(r["threads"]), "kind": "port",
            "covers": "whole scheme" if whole else "PREFIX of the scheme (its first, small steps: not the rate of the leg)",
            "sample": (f"torch-CPU einsum loop over {what}, steps 0..{r['steps_done'] - 1} of {len(case.scheme)} in order"
                       f"{' (the whole scheme)' if r['steps_done'] == len(case.scheme) else ''}: {r['flops_done']:.3e} FLOP "
                       f"(8 x product of the extents of every label, per einsum) in {r['seconds']:.1f} s, "
                       f"torch.set_num_threads({r['threads']}); budget {budget_s:.0f} s"),
            "steps_done": r["steps_done"], "seconds": r["seconds"]}


def cpu_baseline(case_n30, case_n12, budget_s=25.0):
    """SURVEY 8d / BASELINE.md 3: the reference's executor as it runs on a CPU -- a torch-CPU einsum
    loop over the scheme (oracle.tensor_contraction_torch_cpu restates contraction.py:62-76) -- on this
    box's host cores: the n12 m14 scheme whole (min of 3), and the SAME n30 m14 scheme step by step in
    order until `budget_s` seconds have gone (the full scheme takes minutes: 215 s on 8 cores); the
    fraction of the scheme's FLOPs reached is in the record.  Reported, not optimised."""
    from oracle import oracle
    threads = torch.get_num_threads()
    best = None
    for _ in range(3):
        r = oracle.tensor_contraction_torch_cpu({k: t.clone() for k, t in case_n12.tensors.items()}, case_n12.scheme)
        best = r if best is None or r["seconds"] < best["seconds"] else best
    r30 = oracle.tensor_contraction_torch_cpu({k: t.clone() for k, t in case_n30.tensors.items()}, case_n30.scheme,
                                              budget_s=budget_s)
    frac = r30["flops_done"] / r30["flops_total"]
    return {
        "value": r30["flops_done"] / r30["seconds"] / 1e12, "unit": "TFLOP/s", "cores": int(threads), "kind": "port",
        "sample": (f"torch-CPU einsum loop over the n30 m14 scheme, steps 0..{r30['steps_done'] - 1} of {len(case_n30.scheme)} in order "
                   f"({100 * frac:.1f} % of the scheme's FLOPs, {r30['flops_done']:.3e} FLOP in {r30['seconds']:.1f} s, "
                   f"torch.set_num_threads({threads})); budget {budget_s:.0f} s"),
        "n30_fraction_of_flops": frac, "n30_seconds": r30["seconds"], "n30_steps_done": r30["steps_done"],
        "n12_ms": best["seconds"] * 1e3, "n12_gflops": best["flops_done"] / best["seconds"] / 1e9,
    }


SLICED_WORKLOADS = {
    # name: (fixture, sparse executor?, description)
    "n53": ("n53_m14_sliced.npz", True,
            "Sycamore n53 m14 (first 14 cycles of the bundled m20 circuit), 1 bitstring, 14 sliced bonds"),
    "n53m20": ("n53_m20_sliced.npz", True,
               "Sycamore n53 m20 (bundled circuit_n53_m20_s0_e0_pABCDCDAB), 1 bitstring, 29 sliced bonds"),
    "n53m20b": ("n53_m20_batch.npz", True,
                "Sycamore n53 m20 big-batch sampling (BASELINE configs[4]): 1 024 correlated bitstrings over 16 open "
                "qubits, 40 sliced bonds"),
    "n53m20bb": ("n53_m20_bigbatch.npz", True,
                 "Sycamore n53 m20 big-batch sampling at 2^16 bitstrings (half of the 2^17 product over 17 open qubits), "
                 "41 sliced bonds; scheme from the vectorised sparse compiler"),
    "rand2": ("rand_D2_nv260_sliced.npz", False,
              "random 3-regular tensor network, bond dimension 2, 260 tensors, closed, 12 sliced bonds (sc 30)"),
    "rand4": ("rand_D4_nv100.npz", False,
              "random 3-regular tensor network, bond dimension 4, 100 tensors, closed, no slicing (sc 28)"),
    "rand3": ("rand_D3_nv112.npz", False,
              "random 3-regular tensor network, bond dimension 3, 112 tensors, closed, no slicing (largest intermediate 3^18 elements)"),
    "rand6": ("rand_D6_nv64.npz", False,
              "random 3-regular tensor network, bond dimension 6 = 2 x 3, 64 tensors, closed, no slicing (largest intermediate 6^11 elements)"),
}


def run_sliced(A, name, dev, world, rank, dist, steps, warmup, per_step, precision, profile=False):
    """Slice-sharded workloads: every step each rank contracts `per_step` slices of its round-robin
    shard, in Gray-code order, and accumulates; ONE all-reduce of the accumulator over RCCL closes the
    timed region.  Weak scaling (slices per rank fixed).  Returns the result dict (on every rank)."""
    from artensor_amd.fixtures import load_case
    fixture, sparse, what = SLICED_WORKLOADS[name]
    case = load_case(os.path.join(ROOT, "tests", "golden", fixture))
    leaves = case.fresh_tensors(device=dev)
    n_b = len(case.slicing_indices or {})
    rows = len(case.meta["bitstrings_sorted"]) if sparse else 1
    shape = (rows,)
    flops_slice = 8.0 * 10 ** case.meta["log10_tc"]
    runner = A.SliceRunner(leaves, case.scheme, case.slicing_indices, shape, sparse=sparse, device=dev)
    # slice 0 against the reference's value (outside the timed region); under bf16 by fidelity against it
    want = case.arrays["slice0"].reshape(-1)
    got = A.sliced_contraction(None, case.scheme, case.slicing_indices, shape, sparse=sparse, device=dev,
                               slices=[0], reduce=None, runner=runner).reshape(-1).cpu().numpy()
    loose, strict, n_strict = error_figures(got, want)
    fid = fidelity_of(got, want) if rows > 1 else None
    if precision == "bf16":
        ok = (fid is None and loose < 3e-2) or (fid is not None and fid >= BF16_MIN_FIDELITY)
    else:
        ok = loose <= 2 * LOOSE_TOL   # (against the reference's complex64 value; the contract is checked against the truth below)

    def run(first, count):
        if n_b == 0:
            mine = [0] * count
        else:  # Gray-code order over this rank's shard: consecutive slices differ in one sliced bond
            mine = [(((first + q) ^ ((first + q) >> 1)) * world + rank) % (2 ** n_b) for q in range(count)]
        return A.sliced_contraction(None, case.scheme, case.slicing_indices, shape, sparse=sparse, device=dev,
                                    slices=mine, reduce=None, runner=runner)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for w in range(warmup):
        run(w * per_step, per_step)
    barrier()
    t0 = time.perf_counter()
    acc = torch.zeros(shape, dtype=torch.complex64, device=dev)
    base = warmup * per_step
    for k in range(steps):
        A.accumulate(acc, run(base + k * per_step, per_step))
    if world > 1:
        dist.all_reduce(torch.view_as_real(acc))   # THE collective of the path: one sum of the accumulators
    barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    n_slices = world * steps * per_step
    value = n_slices * flops_slice / dt / 1e12
    kernel_times = None
    if profile:   # one more step with per-launch HIP events on the launch stream (outside the timed region)
        from artensor_amd import contraction as C
        prof = KernelTimes()
        C.profiler = prof
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run(base + steps * per_step, per_step)
        e1.record()
        torch.cuda.synchronize()
        C.profiler = None
        kernel_times = (prof.summarize(), e0.elapsed_time(e1))
    truth = truth_value(fixture[:-4] + "_slice0")
    truth_is = "complex128 on the GPU (tests/golden/c128_truth_gpu.npz)"
    if truth is None and "exact128" in case.arrays:   # (round-5 fixtures carry the REFERENCE executor's own complex128 value)
        truth, truth_is = case.arrays["exact128"].reshape(-1), f"the reference's executor in complex128 on the CPU (tests/golden/{fixture}: exact128)"
    vs_truth = None
    if truth is not None:
        tl, ts, _ = error_figures(got, truth)
        rl, rs, _ = error_figures(want, truth)
        vs_truth = {"hip_loose": tl, "hip_strict": ts, "reference_c64_loose": rl, "reference_c64_strict": rs,
                    "truth": truth_is}
        if precision != "bf16":   # the contract: 1e-5 of the truth, or no farther from it than 2 x the reference's own complex64 run
            ok = ok and (tl <= LOOSE_TOL or tl <= 2 * rl)
    return {
        "kernel_times_": kernel_times, "vs_c128_truth": vs_truth, "case_": case,
        "workload": f"{what}, tests/golden/{fixture}", "value": value, "unit": "TFLOP/s",
        "ms_per_step": dt / steps * 1e3, "ms_per_slice_per_rank": dt / (steps * per_step) * 1e3,
        "slices_per_rank_per_step": per_step, "slices_timed": n_slices, "flops_per_slice": flops_slice,
        "parallelism": (f"slices sharded round-robin over {world} rank(s), one all-reduce ({dist.get_backend() if world > 1 else 'none'})"
                        if n_b else ("replicas" if world > 1 else "single")),
        "ranks_in_collective": world, "frac_mfma_peak": value / world / MFMA_F32_PEAK_TFLOPS,
        "slices_executed_in_process": 1 + (warmup + steps + (1 if profile else 0)) * per_step,
        "slice0_err_rel_to_max_abs_or_rms": loose, "slice0_rel_err_strict_over_1e-3rms": strict,
        "slice0_fidelity_vs_reference": fid, "check": "ok" if ok else "FAILED",
        "partial_sum_abs": float(acc.abs().sum().item()),
    }


def strip_private(d):
    return {k: v for k, v in d.items() if not k.endswith("_")}


def leg_sparse_whole(A, dev, steps, precision):
    """BASELINE configs[2]: Sycamore n30 m14 sparse-state, Google's 10 000 bitstrings, one MI355X: one step = one
    tensor_contraction_sparse of the whole 180-step scheme (151 plain, 28 outer-product and 1 chunked step)."""
    from artensor_amd import contraction as C
    from artensor_amd.fixtures import load_case
    fixture = "n30_sparse10000.npz"
    case = load_case(os.path.join(ROOT, "tests", "golden", fixture))
    leaves = case.fresh_tensors(device=dev)
    flops = 8.0 * 10 ** case.meta["log10_tc"]
    run = lambda: A.tensor_contraction_sparse(dict(leaves), case.scheme)
    got = run().reshape(-1).cpu().numpy()
    want = case.arrays["final"].reshape(-1)
    loose, strict, _ = error_figures(got, want)
    ok = loose <= LOOSE_TOL if precision != "bf16" else fidelity_of(got, want) >= BF16_MIN_FIDELITY
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = KernelTimes()
    C.profiler = prof
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    C.profiler = None
    truth = truth_value("n30_sparse10000_final")
    vs_truth = None
    if truth is not None:
        tl, ts, _ = error_figures(got, truth)
        rl, rs, _ = error_figures(want, truth)
        vs_truth = {"hip_loose": tl, "hip_strict": ts, "reference_c64_loose": rl, "reference_c64_strict": rs,
                    "truth": "complex128 on the GPU (tests/golden/c128_truth_gpu.npz)"}
        if precision != "bf16":
            ok = ok and (tl <= LOOSE_TOL or tl <= 2 * rl)
    return {"workload": f"Sycamore n30 m14 sparse-state, Google's 10 000 bitstrings, tests/golden/{fixture}",
            "value": steps * flops / dt / 1e12, "unit": "TFLOP/s", "ms_per_step": dt / steps * 1e3, "flops_per_step": flops,
            "err_rel_to_max_abs_or_rms": loose, "rel_err_strict_over_1e-3rms": strict,
            "fidelity_vs_reference": fidelity_of(got, want), "vs_c128_truth": vs_truth, "check": "ok" if ok else "FAILED",
            "kernel_times_": (prof.summarize(), e0.elapsed_time(e1)), "case_": case}


def leg_n30_c128(A, dev):
    """The headline scheme in complex128 (the reference takes any dtype, simulation.py:90): the 13 fused pairs run on
    artn_k_bits128 (state-streaming, v_mfma_f64_16x16x4_f64 stages, the intermediate stays in LDS), the growth steps on
    artn_k_gemm128, the small ones on the strided kernel.  Priced against the f64 MFMA rate measured on this device."""
    from artensor_amd import contraction as C
    from artensor_amd.fixtures import load_case
    case = load_case(os.path.join(ROOT, "tests", "golden", "n30_dense.npz"))
    leaves = case.fresh_tensors(dtype=torch.complex128, device=dev)
    flops = 8.0 * 10 ** case.meta["log10_tc"]
    run = lambda: A.tensor_contraction(dict(leaves), case.scheme)
    out = run()
    perm = case.meta["permute_dims"]
    fpos = np.array([int(b, 2) for b in case.meta["google_bitstrings"]], dtype=np.int64)
    rpos = np.zeros_like(fpos)
    for d in range(30):
        rpos |= ((fpos >> (29 - d)) & 1) << (29 - perm[d])
    at = out.reshape(-1)[torch.from_numpy(rpos).to(dev)].cpu().numpy()
    del out
    truth = truth_value("n30_dense_at_google")
    want = truth if truth is not None else case.arrays["amps_at_google"]
    rms = 2.0 ** -15
    err = float((np.abs(at - want) / np.maximum(np.abs(want), rms)).max())
    ok = err <= (1e-10 if truth is not None else LOOSE_TOL)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 2
    for _ in range(steps):
        o = run()
        del o
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = KernelTimes()
    C.profiler = prof
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    o = run()
    e1.record()
    torch.cuda.synchronize()
    C.profiler = None
    del o
    return {"workload": "Sycamore n30 m14 full-amplitude in complex128, tests/golden/n30_dense.npz", "value": steps * flops / dt / 1e12,
            "unit": "TFLOP/s", "ms_per_step": dt / steps * 1e3, "flops_per_step": flops,
            "err_rel_to_max_abs_or_rms_vs_c128_truth": err, "check": "ok" if ok else "FAILED",
            "kernel_times_": (prof.summarize(), e0.elapsed_time(e1)), "case_": case}


def leg_n30_sliced3(A, dev):
    """The alternative N > 1 split of the unsliced n30 contraction, timed on ONE GPU: the reference's own mechanism --
    inner-bond slicing (simulation.py:107-114) -- with k = 3 sliced bonds (8 slices, one per rank of an 8-GPU node;
    plan by the reference's order finder, tests/golden/n30_dense_sliced3.npz).  Every slice is a full 2^30-amplitude
    tensor; an 8-rank run would close with a reduce-scatter of 8.6 GB per rank (DESIGN section 6 prices it).  All 8
    slices are run and summed here and the sum is checked against the complex128 truth."""
    from artensor_amd.fixtures import load_case
    case = load_case(os.path.join(ROOT, "tests", "golden", "n30_dense_sliced3.npz"))
    dense = load_case(os.path.join(ROOT, "tests", "golden", "n30_dense.npz"))
    leaves = case.fresh_tensors(device=dev)
    n_b = len(case.slicing_indices)
    flops_slice = 8.0 * 10 ** case.meta["log10_tc_per_slice"]
    runner = A.SliceRunner(leaves, case.scheme, case.slicing_indices, (2,) * 30, device=dev)
    total = runner.run(range(2 ** n_b))
    perm = case.meta["permute_dims"]
    fpos = np.array([int(b, 2) for b in dense.meta["google_bitstrings"]], dtype=np.int64)
    rpos = np.zeros_like(fpos)
    for d in range(30):
        rpos |= ((fpos >> (29 - d)) & 1) << (29 - perm[d])
    at = total.reshape(-1)[torch.from_numpy(rpos).to(dev)].cpu().numpy()
    truth = truth_value("n30_dense_at_google")
    want = truth if truth is not None else dense.arrays["amps_at_google"]
    rms = 2.0 ** -15
    err = float((np.abs(at - want) / np.maximum(np.abs(want), rms)).max())
    ok = err <= LOOSE_TOL
    runner.collect.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    runner.run(range(2 ** n_b))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = 2 ** n_b
    # the 8 slices once more with per-launch HIP events (outside the timed region): the leg's roofline block
    from artensor_amd import contraction as C
    prof = KernelTimes()
    C.profiler = prof
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    runner.run(range(2 ** n_b))
    e1.record()
    torch.cuda.synchronize()
    C.profiler = None
    return {"workload": "Sycamore n30 m14 full-amplitude, 3 inner bonds sliced (8 slices of 2^30 amplitudes), tests/golden/n30_dense_sliced3.npz",
            "value": n * flops_slice / dt / 1e12, "unit": "TFLOP/s", "ms_per_step": dt * 1e3, "ms_per_slice_per_rank": dt / n * 1e3,
            "flops_per_slice": flops_slice, "slices": n, "executed_flop_over_unsliced": n * flops_slice / (8.0 * 10 ** dense.meta["log10_tc"]),
            "sum_of_slices_err_rel_to_max_abs_or_rms_vs_truth": err, "check": "ok" if ok else "FAILED",
            "kernel_times_": (prof.summarize(), e0.elapsed_time(e1)), "profiled_units_": n, "case_": case}


# secondary workloads of the default run: key -> (BASELINE config, runner, precision, slices per step, sparse, sliced)
LEGS = [
    ("n30_sparse10000", "configs[2]", "sparse", "fp32", 1),
    ("n53", "configs[3] (one rank's share: slices are independent)", "n53", "fp32", 4),
    ("n53m20b", "configs[4], complex64 arithmetic", "n53m20b", "fp32", 1),
    ("n53m20b_bf16", "configs[4] as written: bf16-complex MFMA path", "n53m20b", "bf16", 1),
    ("n53m20bb", "configs[4] at 65 536 bitstrings, complex64 arithmetic", "n53m20bb", "fp32", 1),
    ("n53m20bb_bf16", "configs[4] at 65 536 bitstrings: bf16-complex MFMA path", "n53m20bb", "bf16", 1),
    ("n53m20", "the bundled n53 m20 circuit, one bitstring", "n53m20", "fp32", 2),
    ("rand2", "north_star: random tensor network, bond dimension 2", "rand2", "fp32", 4),
    ("rand4", "north_star: random tensor network, bond dimension 4", "rand4", "fp32", 4),
    ("rand3", "north_star: random tensor network, bond dimension 3 (extents that are not powers of two)", "rand3", "fp32", 2),
    ("rand6", "north_star: random tensor network, bond dimension 6 = 2 x 3", "rand6", "fp32", 2),
    ("n30_c128", "configs[1] in complex128 (reference simulation.py:90: any dtype)", "c128", "fp32", 1),
    ("n30_sliced3", "configs[1] split over 3 inner bonds: the N > 1 alternative to output partitioning, on one GPU", "n30s3", "fp32", 1),
]


def run_workloads(A, dev, cpu_budget, only=None):
    """Short legs of every other BASELINE config on the same GPU, after (and outside) the headline's timed region:
    each with its own value, roofline (per-launch HIP events of one extra step) and cpu_baseline."""
    out = {}
    for key, config, kind, precision, per_step in LEGS:
        if only and key not in only:
            continue
        t_leg = time.perf_counter()
        try:
            with A.precision(precision):
                if kind == "sparse":
                    res = leg_sparse_whole(A, dev, 3, precision)
                    sparse, sliced = True, False
                elif kind == "c128":
                    res = leg_n30_c128(A, dev)
                    sparse, sliced = False, False
                elif kind == "n30s3":
                    res = leg_n30_sliced3(A, dev)
                    sparse, sliced = False, True
                else:
                    res = run_sliced(A, kind, dev, 1, 0, None, 2, 1, per_step, precision, profile=True)
                    sparse, sliced = SLICED_WORKLOADS[kind][1], True
            units = per_step if kind not in ("sparse", "c128", "n30s3") else 1
            f64_peak = mfma_rate_probe(2, dev) if kind == "c128" else None
            if res["kernel_times_"] is not None:
                ks, gpu_ms = res["kernel_times_"]
                roof = roofline_of(ks, gpu_ms, 1, f64_peak, leg=key, units=res.get("profiled_units_", units))
            else:
                ks, gpu_ms, roof = None, 0.0, None
            peak = MFMA_BF16_PEAK_TFLOPS if precision == "bf16" else (f64_peak if kind == "c128" else MFMA_F32_PEAK_TFLOPS)
            entry = {"config": config, "value": res["value"], "unit": "TFLOP/s",
                     "ms": res.get("ms_per_slice_per_rank", res["ms_per_step"]), "ms_is_per": "slice" if sliced else "contraction",
                     "frac_of_peak": res["value"] / peak, "peak_TFLOPs": peak,
                     "dtype": ("c64 in memory, bf16 MFMA operands, fp32 accumulate" if precision == "bf16"
                               else ("c128 (f64 MFMA)" if kind == "c128" else "c64 (fp32 MFMA)")),
                     "roofline": roof, "profiled_step_gpu_ms": gpu_ms / res.get("profiled_units_", units),
                     "check": {k: v for k, v in strip_private(res).items() if k not in ("value", "unit", "ms_per_step", "workload")},
                     "workload": res["workload"]}
            if kind == "c128":
                entry["peak_is"] = "v_mfma_f64_16x16x4_f64 back to back, measured on this device (artn_probe_mfma_rate)"
                entry["cpu_baseline"] = {"same_as": "the headline cpu_baseline", "note": "the same scheme on the same host cores "
                                         "(complex64 there; a complex128 torch-CPU run needs 2 x 16 GiB per step)"}
            elif kind == "n30s3":
                entry["cpu_baseline"] = {"same_as": "the headline cpu_baseline", "note": "one slice is the same kind of work at 0.27 x the FLOP"}
            elif cpu_budget > 0 and precision != "bf16":
                entry["cpu_baseline"] = cpu_leg(res["case_"], sparse, sliced, cpu_budget,
                                                ("slice 0 of " if sliced and res["case_"].slicing_indices else "") + key)
            elif precision == "bf16":
                entry["cpu_baseline"] = {"same_as": key.replace("_bf16", ""), "note": "the reference has no reduced-precision path: "
                                         "its CPU executor runs this workload in complex64 (see that entry)"}
            entry["leg_seconds"] = time.perf_counter() - t_leg
            out[key] = entry
            print(json.dumps({"leg": key, **entry}), flush=True)   # (the LAST line of the run is the compact headline)
        except Exception as e:   # a leg that cannot run must not take the headline with it; it is reported as failed
            import traceback
            out[key] = {"config": config, "check": {"check": "FAILED"}, "error": f"{type(e).__name__}: {e}",
                        "traceback": traceback.format_exc()[-800:]}
            print(json.dumps({"leg": key, **out[key]}), flush=True)
        torch.cuda.empty_cache()
    return out


def bench_sliced(args, A, dev, world, rank, dist):
    res = run_sliced(A, args.workload, dev, world, rank, dist, args.steps, args.warmup, args.slices, args.precision,
                     profile=world == 1)
    if rank == 0:
        kt = res.pop("kernel_times_", None)
        res = strip_private(res)
        leg = args.workload + ("_bf16" if args.precision == "bf16" else "")
        roof = roofline_of(*kt, leg=leg, units=args.slices) if kt else None
        line = {
            "metric": f"contracted TFLOP/s, {args.workload} sliced contraction (8 real FLOP per complex MAC)"
                      + (", bf16 operands" if args.precision == "bf16" else ""),
            "value": res["value"], "unit": "TFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("c64 in memory, bf16 MFMA operands, fp32 accumulate" if args.precision == "bf16" else "c64 (fp32 MFMA)"),
            "data": "circuit fixture", "config": {k: v for k, v in res.items() if k not in ("value", "unit", "ms_per_step")},
            "roofline": roof,
        }
        if not args.no_cpu_baseline and world == 1 and args.precision != "bf16":
            from artensor_amd.fixtures import load_case
            fixture, sparse, _ = SLICED_WORKLOADS[args.workload]
            line["cpu_baseline"] = cpu_leg(load_case(os.path.join(ROOT, "tests", "golden", fixture)), sparse, True,
                                           args.cpu_budget_workloads, "slice 0 of " + args.workload)
        write_detail(line)
        # the LAST line stays under COMPACT_LIMIT: rounded numbers, the roofline block's essentials
        short = dict(line)
        short["value"], short["ms_per_step"] = sig(line["value"], 6), sig(line["ms_per_step"], 6)
        cfg = line["config"]
        short["config"] = {k: sig(cfg[k], 5) for k in ("workload", "ms_per_slice_per_rank", "slices_per_rank_per_step", "slices_timed",
                                                       "slices_executed_in_process", "flops_per_slice", "parallelism", "ranks_in_collective",
                                                       "frac_mfma_peak", "slice0_err_rel_to_max_abs_or_rms",
                                                       "slice0_rel_err_strict_over_1e-3rms", "slice0_fidelity_vs_reference", "check") if k in cfg}
        if roof:
            short["roofline"] = {k: sig(roof[k], 5) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic",
                                                              "traffic_over_algorithmic", "mfma_frac_nominal", "hbm_frac",
                                                              "algorithmic_bytes_per_launch", "launches", "kernel_ms", "share_of_gpu_time")}
        if "cpu_baseline" in line:
            cb = line["cpu_baseline"]
            short["cpu_baseline"] = {"value": sig(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                                     "sample": f"torch-CPU einsum loop, slice 0, steps 0..{cb['steps_done'] - 1} in {cb['seconds']:.0f} s"}
        print(json.dumps(short), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0 if res["check"] == "ok" else 1


def n12_latency(A, dev, reps=20):
    """End-to-end latency of BASELINE configs[0] (Sycamore n12 m14 full amplitude, 68 tiny steps) on the GPU:
    a launch-latency workload (the reference's CPU executor: 2.4-4.5 ms)."""
    from artensor_amd.fixtures import load_case
    case = load_case(os.path.join(ROOT, "tests", "golden", "n12_dense.npz"))
    leaves = case.fresh_tensors(device=dev)
    for _ in range(3):
        A.tensor_contraction(dict(leaves), case.scheme)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        out = A.tensor_contraction(dict(leaves), case.scheme)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    err = float(np.abs(out.cpu().numpy() - case.arrays["raw"]).max() / np.abs(case.arrays["raw"]).max())
    return {"n12_gpu_us": best * 1e6, "n12_err": err}, case


def sig(x, n=4):
    """x rounded to n significant digits (the compact line carries no 17-digit floats)."""
    if isinstance(x, bool) or not isinstance(x, (int, float)):
        return x
    if x == 0 or x != x or x in (float("inf"), float("-inf")):
        return x
    if isinstance(x, int) or float(x).is_integer() and abs(x) < 1e15:
        return int(x)
    return float(f"{x:.{n}g}")


COMPACT_LIMIT = 1700   # bytes: the driver keeps a 2 000-character tail of stdout


def compact_line(full):
    """The LAST stdout line: what the driver parses (it keeps a 2 000-character tail).  Everything else of `full` goes to
    bench_detail.json and the per-leg lines.  Optional keys are dropped from the least important end until the line fits
    COMPACT_LIMIT; metric, value, roofline, cpu_baseline and the per-leg triples are never dropped."""
    cfg, roof = full.get("config", {}), full.get("roofline") or {}
    vt = cfg.get("vs_c128_truth") or {}
    line = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                 "scaling", "vs_baseline", "dtype", "data") if k in full}
    line["value"], line["ms_per_step"] = sig(full["value"], 6), sig(full["ms_per_step"], 6)
    par = cfg.get("parallelism")
    c = {"workload": "Sycamore n30 m14 full-amplitude, complex64, no slicing (tests/golden/n30_dense.npz)",
         "flops_per_step": sig(cfg.get("flops_per_step"), 6), "parallelism": par if par in (None, "single") else par.split(":")[0],
         "check": cfg.get("check"), "frac_mfma_peak": sig(cfg.get("frac_mfma_peak")),
         "err_loose": sig(vt.get("hip_loose", cfg.get("err_rel_to_max_abs_or_rms")), 3),
         "err_strict": sig(vt.get("hip_strict", cfg.get("rel_err_strict_over_1e-3rms")), 3),
         "ref_c64_loose": sig(vt.get("reference_c64_loose"), 3), "ref_c64_strict": sig(vt.get("reference_c64_strict"), 3),
         # the interpretation of north_star's "<= 1e-5 relative" this line's `check` applies: loose =
         # |d| <= tol_loose * max(|amp|, rms) per amplitude; strict = max relative error over |amp| >= 1e-3 rms within
         # tol_strict_x times the reference's own complex64-vs-complex128 figure (ref_c64_strict)
         "tol_loose": vt.get("tol_loose", LOOSE_TOL), "tol_strict_x_ref": STRICT_FACTOR if vt else None,
         "err_vs": "c128 truth" if vt else "reference c64", "n12_gpu_us": sig(cfg.get("n12_gpu_us")),
         "ms_unprofiled": sig(cfg.get("ms_per_step_unprofiled"), 6)}
    if cfg.get("failed_workloads"):
        c["failed_workloads"] = cfg["failed_workloads"]
    line["config"] = {k: v for k, v in c.items() if v is not None}
    rk = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "mfma_frac_executed",
          "hbm_frac", "algorithmic_bytes_per_launch", "launches_per_step", "kernel_ms_per_step")
    line["roofline"] = {k: sig(roof.get(k), 5) for k in rk if k in roof and (roof.get(k) is not None or k == "traffic")}
    cb = full.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {"value": sig(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                                "sample": (f"torch-CPU einsum loop, n30 scheme steps 0..{cb.get('n30_steps_done', 0) - 1}"
                                           f" ({100 * cb.get('n30_fraction_of_flops', 0):.0f}% of its FLOP), {cb.get('n30_seconds', 0):.0f} s"),
                                "n12_ms": sig(cb.get("n12_ms"))}
    sl = full.get("sliced")
    if sl:
        line["sliced"] = {"workload": "n53 m14, slices sharded, one all-reduce", "value": sig(sl["value"], 6),
                          "ms_per_slice": sig(sl.get("ms_per_slice_per_rank"), 5), "slices": sl.get("slices_timed"),
                          "ranks_in_collective": sl.get("ranks_in_collective"), "backend": sl.get("backend"), "check": sl.get("check")}
        r2 = sl.get("rand2")
        if r2:   # [TFLOP/s, ms per slice per rank, fraction of N x the fp32 MFMA peak, check]: random network D = 2, same sharding
            line["sliced"]["rand_D2"] = [sig(r2["value"], 5), sig(r2.get("ms_per_slice_per_rank"), 4), sig(r2.get("frac_mfma_peak"), 3), r2.get("check")]
    wl = full.get("workloads")
    if wl:   # [TFLOP/s, roofline fraction of the dominant kernel (executed FLOP or bytes: never above 1), check]
        line["workloads"] = {k: [sig(v.get("value"), 4), sig((v.get("roofline") or {}).get("frac"), 3), (v.get("check") or {}).get("check")]
                             for k, v in wl.items()}
    line["detail"] = "bench_detail.json + one {\"leg\":..} line per workload"
    for drop in (("detail",), ("config", "n12_gpu_us"), ("cpu_baseline", "n12_ms"), ("roofline", "launches_per_step"),
                 ("sliced", "workload"), ("config", "parallelism"), ("config", "err_vs"), ("roofline", "algorithmic_bytes_per_launch"),
                 ("config", "ref_c64_loose"), ("config", "flops_per_step"), ("sliced", "slices"), ("config", "failed_workloads")):
        if len(json.dumps(line)) <= COMPACT_LIMIT:
            break
        d = line
        for k in drop[:-1]:
            d = d.get(k, {})
        d.pop(drop[-1], None)
    if len(json.dumps(line)) > COMPACT_LIMIT and "workloads" in line:   # (many legs: value and check only)
        line["workloads"] = {k: [v[0], v[2]] for k, v in line["workloads"].items()}
    return line


def write_detail(full):
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, "bench_detail.json"), "w") as f:
                    json.dump(full, f, indent=1)
            except OSError:
                pass


def launch_ranks(n):
    """`python bench.py --gpus N` as a plain command: start the N ranks as a CHILD process (never exec, and before this
    process has initialised the GPU: torch.cuda.device_count() does not), relay their output, exit with their status."""
    import socket
    import subprocess
    if "ARTN_BENCH_DEVICE" not in os.environ:   # (the one-GPU self-test knob runs every rank on that device)
        have = torch.cuda.device_count()
        if have < n:
            print(json.dumps({"error": f"--gpus {n}: this node has {have} visible GPU(s)", "n_gpus": n}), flush=True)
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=420.0,
                    help="seconds the torch-CPU einsum loop may spend on the n30 scheme (it stops after the first step that "
                         "ends past the budget; the default lets the whole scheme finish on the GPU box's host: SURVEY 8d)")
    ap.add_argument("--cpu-budget-workloads", type=float, default=10.0,
                    help="the same, per secondary workload")
    ap.add_argument("--no-workloads", action="store_true", help="N = 1: skip the legs of the other BASELINE configs")
    ap.add_argument("--only-workloads", default=None, help="comma-separated subset of the legs (diagnostics)")
    ap.add_argument("--no-sliced", action="store_true", help="N > 1: skip the embedded slice-sharded n53 run")
    ap.add_argument("--detail", default=None, help="write a per-launch table of the MFMA kernels to this file")
    ap.add_argument("--workload", default="n30", choices=["n30"] + sorted(SLICED_WORKLOADS),
                    help="n30: BASELINE configs[1] (default, the metric's config); n53: configs[3], the "
                         "slice-sharded Sycamore n53 m14 contraction with one RCCL reduce at the end; "
                         "n53m20: the bundled n53 m20 circuit, per-slice throughput; n53m20b: configs[4], n53 m20 "
                         "big-batch sampling (1 024 bitstrings; use with --precision bf16); "
                         "rand2 / rand4: random 3-regular tensor networks of bond dimension 2 (sliced) / 4")
    ap.add_argument("--slices", type=int, default=4, help="sliced workloads: slices per rank per step")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16"],
                    help="fp32 (default; the metric's arithmetic) or bf16: complex64 in memory, MFMA operands "
                         "rounded to bfloat16, fp32 accumulation (BASELINE configs[4]'s reduced-precision mode; "
                         "reported as its own metric, checked by state fidelity)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus)   # (nothing above this line has touched the GPU)

    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Self-test knobs (never set by the driver): ARTN_BENCH_BACKEND=gloo with ARTN_BENCH_DEVICE=0 runs
    # the N > 1 code path with every rank on one GPU, where RCCL cannot
    backend = os.environ.get("ARTN_BENCH_BACKEND", "nccl")
    if "ARTN_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["ARTN_BENCH_DEVICE"])
    if world > 1:
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    import artensor_amd as A
    from artensor_amd import contraction as C
    from artensor_amd.fixtures import load_case

    bf16 = args.precision == "bf16"
    if bf16:
        A.precision("bf16").__enter__()   # for the whole run
    if args.workload in SLICED_WORKLOADS:
        return bench_sliced(args, A, dev, world, rank, dist)
    case = load_case(os.path.join(ROOT, "tests", "golden", "n30_dense.npz"))
    leaves = case.fresh_tensors(device=dev)  # resident in HBM before the timed region
    flops_per_step = 8.0 * 10 ** case.meta["log10_tc"]
    # final = raw.permute(permute_dims) is a view (reference simulation.py:115-116): map the
    # positions of Google's 10 000 bitstrings in `final` to positions in the raw result
    perm = case.meta["permute_dims"]
    fpos = np.array([int(b, 2) for b in case.meta["google_bitstrings"]], dtype=np.int64)
    rpos = np.zeros_like(fpos)
    for d in range(30):
        rpos |= ((fpos >> (29 - d)) & 1) << (29 - perm[d])
    want_all = case.arrays["amps_at_google"]

    n_fix = 0
    parts = None   # the committed per-slab re-plan of this N (tests/golden/n30_dense_part{N}.npz), when there is one
    if world > 1:
        n_fix = int(np.log2(world))
        if (1 << n_fix) != world:
            sys.exit("--gpus N > 1: N must be a power of two (output-qubit partitioning fixes log2 N output labels)")
        ppath = os.path.join(ROOT, "tests", "golden", f"n30_dense_part{world}.npz")
        if os.path.exists(ppath) and not os.environ.get("ARTN_BENCH_SAME_TREE"):
            parts = load_case(ppath)
            parts_leaves = parts.fresh_tensors(device=dev)

    def one_step():
        if world == 1:
            return A.tensor_contraction(dict(leaves), case.scheme)
        if parts is not None:
            return A.slab_contraction(parts_leaves, parts.scheme, parts.meta["fixed"], rank, device=dev)
        return A.partitioned_contraction(leaves, case.scheme, n_fix, rank, device=dev)[0]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    out = None
    for _ in range(args.warmup):
        out = one_step()
    if out is None:   # (--warmup 0: the check still needs a result; it is outside the timed region either way)
        out = one_step()
    # result check outside the timed region: amplitudes at Google's 10 000 bitstrings (this rank's share of them)
    if world == 1:
        sel = np.ones(len(rpos), dtype=bool)
        local = rpos
        overhead = 1.0
    elif parts is not None:
        # slab `rank` of the re-planned reduced network: fixed qubit j of meta["fixed"] holds bit j of the rank; raw result dim x
        # holds final qubit out_qubits[x] (qubit 0 = most significant bit of a position in `final`)
        sel = np.ones(len(fpos), dtype=bool)
        for jq, (_leaf, _dim, q) in enumerate(parts.meta["fixed"]):
            sel &= ((fpos >> (29 - q)) & 1) == ((rank >> jq) & 1)
        oq = parts.meta["out_qubits"]
        local = np.zeros(int(sel.sum()), dtype=np.int64)
        for x, q in enumerate(oq):
            local |= ((fpos[sel] >> (29 - q)) & 1) << (len(oq) - 1 - x)
        overhead = float(parts.meta["executed_flop_over_unsliced"])
    else:
        _, fixed_dims, values = A.partitioned_contraction(leaves, case.scheme, n_fix, rank, device=dev)
        sel = np.ones(len(rpos), dtype=bool)
        for d, v in zip(fixed_dims, values):
            sel &= ((rpos >> (29 - d)) & 1) == v
        keep = [d for d in range(30) if d not in fixed_dims]
        local = np.zeros(int(sel.sum()), dtype=np.int64)
        for q, d in enumerate(keep):
            local |= ((rpos[sel] >> (29 - d)) & 1) << (len(keep) - 1 - q)
        from artensor_amd.contraction import _labels
        from artensor_amd import simulation as S
        new_scheme = next(v for k, v in S._partition_cache.items() if k[0] == id(case.scheme) and k[1] == n_fix)[1]
        overhead = world * sum(2.0 ** len(set(_labels(e)[0]) | set(_labels(e)[1])) for _, e in new_scheme) / \
            sum(2.0 ** len(set(_labels(e)[0]) | set(_labels(e)[1])) for _, e in case.scheme)
    at = out.reshape(-1)[torch.from_numpy(local).to(dev)].cpu().numpy()
    # error figures relative to the rms of the WHOLE 10 000-amplitude sample, whatever share this rank holds
    want = want_all[sel]
    rms_all = float(np.sqrt(np.mean(np.abs(want_all) ** 2)))
    diff = np.abs(at - want)
    loose = float((diff / np.maximum(np.abs(want), rms_all)).max()) if len(want) else 0.0
    big = np.abs(want) >= 1e-3 * rms_all
    strict = float((diff[big] / np.abs(want)[big]).max()) if big.any() else 0.0
    if world > 1:
        e = torch.tensor([loose, strict], dtype=torch.float64, device=dev)
        dist.all_reduce(e, op=dist.ReduceOp.MAX)
        loose, strict = float(e[0].item()), float(e[1].item())
        fidelity = None
    else:
        fidelity = fidelity_of(at, want)
    # the contract proper: against the complex128 truth of the same leaves and scheme, the reference's own
    # complex64 distance from it measured beside it
    truth_all = truth_value("n30_dense_at_google")
    vs_truth = None
    if truth_all is not None:
        tw = truth_all[sel]
        d_hip, d_ref = np.abs(at - tw), np.abs(want - tw)
        den = np.maximum(np.abs(tw), rms_all)
        bigt = np.abs(tw) >= 1e-3 * rms_all
        e = [float((d_hip / den).max()) if len(tw) else 0.0, float((d_hip[bigt] / np.abs(tw)[bigt]).max()) if bigt.any() else 0.0,
             float((d_ref / den).max()) if len(tw) else 0.0, float((d_ref[bigt] / np.abs(tw)[bigt]).max()) if bigt.any() else 0.0]
        if world > 1:
            et = torch.tensor(e, dtype=torch.float64, device=dev)
            dist.all_reduce(et, op=dist.ReduceOp.MAX)
            e = [float(x) for x in et.tolist()]
        vs_truth = {"hip_loose": e[0], "hip_strict": e[1], "reference_c64_loose": e[2], "reference_c64_strict": e[3],
                    "tol_loose": LOOSE_TOL, "tol_strict": f"{STRICT_FACTOR} x reference_c64_strict",
                    "truth": "complex128 on the GPU, this package's f64-MFMA path (tests/golden/c128_truth_gpu.npz)"}
    if bf16:
        ok = fidelity is None or fidelity >= BF16_MIN_FIDELITY
    elif vs_truth is not None:
        ok = (loose <= LOOSE_TOL + vs_truth["reference_c64_loose"] and vs_truth["hip_loose"] <= LOOSE_TOL
              and vs_truth["hip_strict"] <= STRICT_FACTOR * vs_truth["reference_c64_strict"])
    else:
        ok = loose <= LOOSE_TOL
    del out

    prof = KernelTimes()
    C.profiler = prof
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one_step()
    barrier()
    dt = time.perf_counter() - t0
    C.profiler = None
    del out

    # the same steps once more WITHOUT the per-launch events (what the 2 events per launch inside the timed region cost)
    n_plain = max(1, min(args.steps, 5))
    barrier()
    t1 = time.perf_counter()
    for _ in range(n_plain):
        out = one_step()
    barrier()
    dt_plain = time.perf_counter() - t1
    del out

    t = torch.tensor([dt, dt_plain], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt, dt_plain = float(t[0].item()), float(t[1].item())

    # the slice-sharded workload north_star's ">= 6x at 8 GPUs" refers to, at EVERY N with the same parameters (weak
    # scaling: slices per rank fixed; at N = 1 no collective): sliced.value(N) / sliced.value(1) is that speed-up
    sliced = None
    if not args.no_sliced and not (world == 1 and args.no_workloads):
        del leaves
        leaves = None
        torch.cuda.empty_cache()
        sliced = strip_private(run_sliced(A, "n53", dev, world, rank, dist, 3, 1, args.slices, args.precision))
        sliced["series"] = "n53 m14 slice-sharded, one all-reduce (the workload north_star's >= 6x at 8 GPUs refers to)"
        sliced["backend"] = dist.get_backend() if world > 1 else None
        ok = ok and sliced["check"] == "ok"
        # north_star: "throughput on random tensor networks of stated bond dimension ... at 1/2/4/8 GPUs": the sliced random
        # 3-regular network of bond dimension 2 (260 tensors, 12 sliced bonds) the same way -- slices sharded, one all-reduce
        # (the bond-dimension-4 network is unsliced: it does not shard and is a leg of the N = 1 run only)
        torch.cuda.empty_cache()
        rand2 = strip_private(run_sliced(A, "rand2", dev, world, rank, dist, 3, 1, args.slices, args.precision))
        rand2["series"] = "random 3-regular network, bond dimension 2, slice-sharded, one all-reduce"
        rand2["backend"] = sliced["backend"]
        ok = ok and rand2["check"] == "ok"
        sliced["rand2"] = rand2

    if rank == 0 and args.detail:
        per = {}
        order = []
        n_per_step = len(prof.rows) // max(args.steps, 1)
        for n, (info, e0, e1) in enumerate(prof.rows):
            key = n % n_per_step
            if key not in per:
                per[key] = [info, 0.0]
                order.append(key)
            per[key][1] += e0.elapsed_time(e1) / args.steps
        with open(args.detail, "w") as f:
            f.write("launch kernel k k2 mt nt Tin Tout tiles rereads ms GB/s TFLOP/s\n")
            for key in order:
                info, ms = per[key]
                if info["kernel"] == 0 and ms < 0.05:
                    continue
                f.write(f"{key} {info['kernel']} {info['k_bits']} {info['k2_bits']} {info['m_tile_bits']} {info['n_tile_bits']} "
                        f"{info['tile_in_bits']} {info['tile_out_bits']} {info['n_tiles']} {info['a_rereads']} "
                        f"{ms:.3f} {info['bytes'] / ms / 1e6:.0f} {info['flops'] / ms / 1e9:.1f}\n")
    if rank == 0:
        ks = prof.summarize()
        bits = ks.get(1, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0, mfma_flops=0.0))
        ms_per_step = dt / args.steps * 1e3
        value = args.steps * flops_per_step / dt / 1e12   # one full n30 contraction per step, whatever N
        achieved = bits["flops"] / (bits["ms"] * 1e-3) / 1e12 if bits["ms"] else 0.0
        hbm_gbs = bits["bytes"] / (bits["ms"] * 1e-3) / 1e9 if bits["ms"] else 0.0
        # HBM bytes per launch of the dominant kernel come from separate rocprofv3 PMC passes
        # (tools/profile_round.sh -> profiles/rNN_traffic.json); bench.py cannot run them itself
        traffic, traffic_note = None, "N > 1: not collected"
        if world == 1:
            tj, traffic_note = traffic_record()
            if tj is not None:
                traffic = tj.get("hbm_bytes_per_launch")
        line = {
            "metric": "contracted TFLOP/s, Sycamore n30 m14 full-amplitude (8 real FLOP per complex MAC)"
                      + (", bf16 operands" if bf16 else ""),
            "value": value, "unit": "TFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if world > 1 else "weak",
            "vs_baseline": None,
            "dtype": "c64 in memory, bf16 MFMA operands, fp32 accumulate" if bf16 else "c64 (fp32 MFMA)",
            "data": "circuit fixture",
            "config": {"workload": "Sycamore n30 m14 full-amplitude, complex64, no slicing, 180-step scheme "
                                   "(tests/golden/n30_dense.npz)",
                       "flops_per_step": flops_per_step,
                       "parallelism": (f"output-qubit partitioning: {n_fix} output qubit(s) fixed per rank at the leaves, "
                                       f"{world} disjoint slabs of 2^{30 - n_fix} amplitudes, no collective on the data path; "
                                       + ("the reduced network RE-PLANNED by the reference's order finder "
                                          f"(tests/golden/n30_dense_part{world}.npz)" if parts is not None
                                          else "the one tree of the full network")
                                       + f" (build-side extension; executed FLOP of all ranks = {overhead:.2f} x the unsliced plan's)") if world > 1 else "single",
                       "series": ("n30 m14 full amplitude, output-partitioned over the ranks (strong scaling; value = the unsliced plan's "
                                  "FLOP / time, whatever the ranks execute)"
                                  if world > 1 else "n30 m14 full amplitude, one GPU"),
                       "ranks_in_collective": 0 if world > 1 else None,
                       "frac_mfma_peak": value / world / MFMA_F32_PEAK_TFLOPS,
                       "ms_per_step_unprofiled": dt_plain / n_plain * 1e3,
                       "ms_per_step_unprofiled_is": f"{n_plain} more steps with the per-launch HIP events of the roofline block switched off",
                       "check": "ok" if ok else "FAILED",
                       "checked_amplitudes": "Google's 10 000 bitstrings (examples/amplitudes_n30_m14...txt positions) vs the "
                                             "reference's complex64 CPU run",
                       "err_rel_to_max_abs_or_rms": loose, "tol_rel_to_max_abs_or_rms": LOOSE_TOL,
                       "rel_err_strict_over_1e-3rms": strict, "vs_c128_truth": vs_truth,
                       "fidelity_vs_reference": fidelity},
            "roofline": {
                "bound": "mfma", "achieved": achieved, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / MFMA_F32_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_note,
                "traffic_over_algorithmic": (traffic / (bits["bytes"] / max(bits["launches"], 1))) if traffic and bits["bytes"] else None,
                "frac_is": "nominal FLOP (8 per complex multiply-add) / kernel time / fp32 MFMA spec peak",
                "executed_mfma_flop_frac": bits["mfma_flops"] / bits["flops"] if bits["flops"] else 0.0,
                "mfma_frac_executed": (bits["mfma_flops"] / (bits["ms"] * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS) if bits["ms"] else 0.0,
                "algorithmic_bytes_per_launch": bits["bytes"] / max(bits["launches"], 1),
                "kernel": "artn_k_bits", "launches_per_step": bits["launches"] / max(args.steps, 1),
                "avg_launch_ms": bits["ms"] / max(bits["launches"], 1),
                "hbm_achieved_GBs": hbm_gbs, "hbm_frac": hbm_gbs / HBM_PEAK_GBS,
                "kernel_ms_per_step": bits["ms"] / max(args.steps, 1),
                "other_kernels_ms_per_step": sum(v["ms"] for k, v in ks.items() if k != 1) / max(args.steps, 1),
            },
        }
        if bf16:  # with bf16 operands every big launch is bound by its one pass over HBM
            line["roofline"].update({"bound": "hbm", "achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": hbm_gbs / HBM_PEAK_GBS, "traffic": None})
        if sliced is not None:
            line["sliced"] = sliced
        if world == 1:
            lat, case12 = n12_latency(A, dev)
            line["config"].update(lat)
            if not args.no_workloads and not bf16:
                leaves = None
                torch.cuda.empty_cache()
                only = set(args.only_workloads.split(",")) if args.only_workloads else None
                line["workloads"] = run_workloads(A, dev, 0.0 if args.no_cpu_baseline else args.cpu_budget_workloads, only)
                bad = [k for k, v in line["workloads"].items() if v["check"].get("check") != "ok"]
                if bad:
                    ok = False
                    line["config"]["check"] = "FAILED"
                    line["config"]["failed_workloads"] = bad
            if not args.no_cpu_baseline:
                line["cpu_baseline"] = cpu_baseline(case, case12, args.cpu_budget)
        write_detail(line)
        print(json.dumps(compact_line(line)), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
