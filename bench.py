#!/usr/bin/env python3
"""bench.py -- contracted TFLOP/s of the hot path on MI355X.

A "step" is one pass of the hot path over one batch of synthetic input: one full
`tensor_contraction` of the Sycamore n30 m14 full-amplitude scheme (BASELINE.json configs[1]:
180 pairwise steps, 5.3705e12 real FLOP, all 2^30 complex64 amplitudes, no slicing) with the
181 leaf tensors already resident in HBM.  Leaf tensors and scheme come from the committed
fixture tests/golden/n30_dense.npz (circuit gate tensors are tiny and fixed: "synthetic"
here means no dataset is read; amplitudes are checked against the fixture's statistics).

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1 (launched by torch.distributed.run, one rank per GPU): the unsliced n30 contraction
does not shard (its 8 GiB dense output would need an 8 GiB all-reduce), so
every rank contracts its own replica -- weak scaling, no data-path collective; the
barrier + max-over-ranks timing of the contract stays.  The slice-sharded path with its
single RCCL reduce is `artensor_amd.sliced_contraction` (tests/test_distributed.py).

Prints ONE JSON line on rank 0.
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
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E peak (spec)


class KernelTimes:
    """Per-launch HIP-event timing of artn_contract calls inside the timed region."""

    def __init__(self):
        self.rows = []

    def record(self, info, e0, e1):
        self.rows.append((info, e0, e1))

    def summarize(self):
        out = {}
        for info, e0, e1 in self.rows:
            d = out.setdefault(info["kernel"], dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += info["flops"]
            d["bytes"] += info["bytes"]
        return out


def cpu_baseline(case, budget_log2=25):
    """The oracle (numpy port of the reference executor) timed on this box's host cores on a
    bounded sample: the 28 big steps of the same n30 scheme with the state operand truncated
    to 2^budget_log2 elements (surrogates keep each step's contracted/free bit pattern)."""
    from oracle import oracle
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import dense_scheme_shapes, shrink_step, crandn
    steps = dense_scheme_shapes(case)
    big = [s for s in steps if np.prod(s[1]) >= 2 ** 20]
    rng = np.random.default_rng(0)
    work = []
    flops = 0.0
    for eq, sa, sb in big:
        eq2, sa2, sb2 = shrink_step(eq, sa, sb, max_log2=budget_log2)
        a, b = crandn(rng, sa2), crandn(rng, sb2)
        lhs, _ = eq2.split("->")
        la, lb = lhs.split(",")
        flops += 8.0 * float(2 ** len(set(la) | set(lb)))
        work.append((eq2, a, b))
    t0 = time.perf_counter()
    for eq2, a, b in work:
        oracle.einsum_pair(eq2, a, b)
    dt = time.perf_counter() - t0
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [os.cpu_count()])
    except Exception:
        threads = os.cpu_count()
    return {
        "value": flops / dt / 1e12, "unit": "TFLOP/s", "cores": int(threads), "kind": "port",
        "sample": f"28 big steps of the n30 m14 scheme, state operand truncated to 2^{budget_log2} "
                  f"elements ({flops:.3e} FLOP, {dt:.1f} s, numpy oracle)",
    }


SLICED_WORKLOADS = {
    # name: (fixture, sparse executor?, description)
    "n53": ("n53_m14_sliced.npz", True,
            "Sycamore n53 m14 (first 14 cycles of the bundled m20 circuit), 1 bitstring, 14 sliced bonds"),
    "n53m20": ("n53_m20_sliced.npz", True,
               "Sycamore n53 m20 (bundled circuit_n53_m20_s0_e0_pABCDCDAB), 1 bitstring, 29 sliced bonds"),
    "rand2": ("rand_D2_nv260_sliced.npz", False,
              "random 3-regular tensor network, bond dimension 2, 260 tensors, closed, 12 sliced bonds (sc 30)"),
    "rand4": ("rand_D4_nv100.npz", False,
              "random 3-regular tensor network, bond dimension 4, 100 tensors, closed, no slicing (sc 28)"),
}


def bench_sliced(args, A, dev, world, rank, dist):
    """Slice-sharded workloads (BASELINE configs[3] and the random networks of north_star): every
    step each rank contracts `--slices` slices of its round-robin shard, in Gray-code order, and
    accumulates; ONE reduce of the accumulator over RCCL closes the timed region.  Weak scaling.
    A network without sliced bonds (rand4) is contracted whole, `--slices` times per step."""
    from artensor_amd.fixtures import load_case
    fixture, sparse, what = SLICED_WORKLOADS[args.workload]
    case = load_case(os.path.join(ROOT, "tests", "golden", fixture))
    leaves = case.fresh_tensors(device=dev)
    n_b = len(case.slicing_indices)
    flops_slice = 8.0 * 10 ** case.meta["log10_tc"]
    per_step = args.slices
    # one runner for the whole job: small intermediates are kept across slices
    runner = A.SliceRunner(leaves, case.scheme, case.slicing_indices, (1,), sparse=sparse, device=dev)
    # slice 0 against the reference's value (outside the timed region)
    want = case.arrays["slice0"].reshape(-1)
    got = A.sliced_contraction(None, case.scheme, case.slicing_indices, (1,), sparse=sparse, device=dev,
                               slices=[0], reduce=None, runner=runner).reshape(-1).cpu().numpy()
    rel_err = float(np.abs(got - want).max() / np.abs(want).max())

    def run(first, count):
        if n_b == 0:
            mine = [0] * count
        else:  # Gray-code order over this rank's shard: consecutive slices differ in one sliced bond
            mine = [(((first + q) ^ ((first + q) >> 1)) * world + rank) % (2 ** n_b) for q in range(count)]
        return A.sliced_contraction(None, case.scheme, case.slicing_indices, (1,), sparse=sparse, device=dev,
                                    slices=mine, reduce=None, runner=runner)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for w in range(args.warmup):
        run(w * per_step, per_step)
    barrier()
    t0 = time.perf_counter()
    acc = torch.zeros(1, dtype=torch.complex64, device=dev)
    base = args.warmup * per_step
    for k in range(args.steps):
        A.accumulate(acc, run(base + k * per_step, per_step))
    if world > 1:
        dist.all_reduce(torch.view_as_real(acc))
    barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    if rank == 0:
        n_slices = world * args.steps * per_step
        value = n_slices * flops_slice / dt / 1e12
        print(json.dumps({
            "metric": f"contracted TFLOP/s, {args.workload} sliced contraction (8 real FLOP per complex MAC)",
            "value": value, "unit": "TFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("c64 in memory, bf16 MFMA operands, fp32 accumulate" if args.precision == "bf16"
                      else "c64 (fp32 MFMA)"), "data": "synthetic",
            "config": {"workload": f"{what}, tests/golden/{fixture}",
                       "slices_per_rank_per_step": per_step, "slices_timed": n_slices,
                       "flops_per_slice": flops_slice,
                       "parallelism": (f"slices sharded over {world} rank(s), one reduce" if n_b else
                                       ("replicas" if world > 1 else "single")),
                       "frac_mfma_peak": value / world / MFMA_F32_PEAK_TFLOPS,
                       "slice0_rel_err_vs_reference": rel_err,
                       "partial_sum": [float(acc.real.item()), float(acc.imag.item())]},
        }), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--detail", default=None, help="write a per-launch table of the MFMA kernel to this file")
    ap.add_argument("--workload", default="n30", choices=["n30"] + sorted(SLICED_WORKLOADS),
                    help="n30: BASELINE configs[1] (default, the metric's config); n53: configs[3], the "
                         "slice-sharded Sycamore n53 m14 contraction with one RCCL reduce at the end; "
                         "n53m20: the bundled n53 m20 circuit, per-slice throughput; "
                         "rand2 / rand4: random 3-regular tensor networks of bond dimension 2 (sliced) / 4")
    ap.add_argument("--slices", type=int, default=4, help="sliced workloads: slices per rank per step")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16"],
                    help="fp32 (default; the metric's arithmetic) or bf16: complex64 in memory, MFMA operands "
                         "rounded to bfloat16, fp32 accumulation (BASELINE configs[4]'s reduced-precision mode; "
                         "reported as its own metric, checked by state fidelity)")
    args = ap.parse_args()

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
    elif args.gpus > 1:
        sys.exit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
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
    pos = torch.from_numpy(rpos).to(dev)

    def one_step():
        return A.tensor_contraction(dict(leaves), case.scheme)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = one_step()
    # result check outside the timed region (amplitudes at Google's 10 000 bitstrings)
    at = out.reshape(-1)[pos].cpu().numpy()
    want = case.arrays["amps_at_google"]
    rel_err = float((np.abs(at - want) / np.maximum(np.abs(want), 2.0 ** -15)).max())
    w128, a128 = want.astype(np.complex128), at.astype(np.complex128)
    fidelity = float(abs(np.vdot(w128, a128)) ** 2 / (np.vdot(w128, w128).real * np.vdot(a128, a128).real))
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

    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

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
            f.write("launch kernel k mt nt Tin Tout tiles rereads ms GB/s TFLOP/s\n")
            for key in order:
                info, ms = per[key]
                if info["kernel"] != 1 and ms < 0.05:
                    continue
                f.write(f"{key} {info['kernel']} {info['k_bits']} {info['m_tile_bits']} {info['n_tile_bits']} "
                        f"{info['tile_in_bits']} {info['tile_out_bits']} {info['n_tiles']} {info['a_rereads']} "
                        f"{ms:.3f} {info['bytes'] / ms / 1e6:.0f} {info['flops'] / ms / 1e9:.1f}\n")
    if rank == 0:
        ks = prof.summarize()
        bits = ks.get(1, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
        ms_per_step = dt / args.steps * 1e3
        value = world * args.steps * flops_per_step / dt / 1e12
        achieved = bits["flops"] / (bits["ms"] * 1e-3) / 1e12 if bits["ms"] else 0.0
        hbm_gbs = bits["bytes"] / (bits["ms"] * 1e-3) / 1e9 if bits["ms"] else 0.0
        # HBM bytes per launch of the dominant kernel come from separate rocprofv3 PMC passes
        # (tools/profile_round.sh -> profiles/rNN_traffic.json); bench.py cannot run them itself
        traffic = None
        import glob
        tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
        if tfiles:
            with open(tfiles[-1]) as f:
                traffic = json.load(f).get("hbm_bytes_per_launch")
        line = {
            "metric": "contracted TFLOP/s, Sycamore n30 m14 full-amplitude (8 real FLOP per complex MAC)"
                      + (", bf16 operands" if bf16 else ""),
            "value": value, "unit": "TFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "c64 in memory, bf16 MFMA operands, fp32 accumulate" if bf16 else "c64 (fp32 MFMA)",
            "data": "synthetic",
            "config": {"workload": "Sycamore n30 m14 full-amplitude, complex64, no slicing, 180-step scheme "
                                   "(tests/golden/n30_dense.npz)",
                       "flops_per_step": flops_per_step, "parallelism": "replicas" if world > 1 else "single",
                       "frac_mfma_peak": value / world / MFMA_F32_PEAK_TFLOPS,
                       "max_rel_err_vs_reference": rel_err, "fidelity_vs_reference": fidelity},
            "roofline": {
                "bound": "mfma", "achieved": achieved, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / MFMA_F32_PEAK_TFLOPS, "traffic": traffic,
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
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(case)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
