#!/usr/bin/env python3
"""gpurun_out/traffic_<round>/ (tools/profile_traffic.sh) -> profiles/<round>_traffic.json + profiles/<round>_traffic.md:
HBM bytes per unit of work (one contraction / one slice) of every kernel family of every bench leg.  FETCH_SIZE and
WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts a 128-byte request as 64 bytes (MI355X_MICROARCH.md, HBM/rocprofv3
section), so it is doubled."""
import collections
import csv
import glob
import json
import os
import sys

R = sys.argv[1] if len(sys.argv) > 1 else "r04"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", f"traffic_{R}")
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def family(name):
    if "artn_k_bits" in name or "artn_k_alt" in name or "artn_k_wide" in name:
        return "bits"          # (artn_k_bits128 too: the complex128 leg's planner id 1)
    if "artn_k_pgemm" in name or "artn_k_pack" in name:
        return "pgemm"         # packing passes + the packed GEMM: one artn_contract_ws call
    if "artn_k_xgemm" in name:
        return "xgemm"         # the extent-based GEMM (non power-of-two extents)
    if "artn_k_xrow" in name:
        return "xrow"          # ... its row-streaming form (round 6)
    if "artn_k_gemm" in name:
        return "gemm"          # artn_k_gemm, artn_k_gemm_deep, artn_k_gemm128
    if "artn_k_program" in name:
        return "program"
    if "artn_k_generic" in name:
        return "generic"
    if "copyBuffer" in name or "fillBuffer" in name:
        return "runtime_copies"
    return "other"             # gathers, axpy, column sums, normalisation, torch's own kernels


def table(path, counter):
    per = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        d = per.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"], "t0": int(r["Start_Timestamp"]), "v": 0.0})
        d["v"] += float(r["Counter_Value"])
    return list(per.values())


out = {"round": R, "kernel_source_sha16": bench.kernel_source_sha16(),
       "method": "rocprofv3 --pmc FETCH_SIZE (KiB, x2 on gfx950) and --pmc WRITE_SIZE (KiB) in separate passes per leg, "
                 "tools/profile_traffic.sh; bytes per unit = one contraction / one slice", "workloads": {}}
md = [f"# HBM traffic per kernel family and bench leg, round {R}", "",
      "`tools/profile_traffic.sh` on one MI355X: per leg two rocprofv3 passes (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`) around "
      "`tools/trace_leg.py LEG UNITS`; FETCH_SIZE x 2 (gfx950 tallies 128-byte requests at 64 bytes), KiB -> bytes.", "",
      "| leg | family | kernel dispatches / unit | fetch GB / unit | write GB / unit | HBM GB / unit |", "|---|---|---|---|---|---|"]
copies = {}
for leg_dir in sorted(glob.glob(os.path.join(SRC, "*"))):
    leg = os.path.basename(leg_dir)
    if not os.path.isdir(leg_dir):
        continue
    units = int(open(os.path.join(leg_dir, "units")).read())
    # (gpurun MERGES a call's files into gpurun_out/: an earlier run of the same tag leaves its files behind -- newest first)
    f = sorted(glob.glob(os.path.join(leg_dir, "fetch", "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime, reverse=True)
    w = sorted(glob.glob(os.path.join(leg_dir, "write", "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime, reverse=True)
    if not f or not w:
        print("missing counters for", leg)
        continue
    fe, wr = table(f[0], "FETCH_SIZE"), table(w[0], "WRITE_SIZE")
    fams = {}
    for rows, key, scale in ((fe, "fetch", 2048.0), (wr, "write", 1024.0)):
        for d in rows:
            e = fams.setdefault(family(d["name"]), {"fetch": 0.0, "write": 0.0, "dispatches_fetch": 0, "dispatches_write": 0})
            e[key] += d["v"] * scale
            e["dispatches_" + key] += 1
    # runtime copies: how many before the first artn kernel (leaf / index uploads at start-up) and how many after
    first = min((d["t0"] for d in fe if "artn_k" in d["name"]), default=None)
    cp = [d for d in fe if "copyBuffer" in d["name"]]
    copies[leg] = {"total": len(cp), "before_first_artn_kernel": sum(1 for d in cp if first is not None and d["t0"] < first),
                   "per_unit_after": (sum(1 for d in cp if first is not None and d["t0"] >= first)) / units}
    out["workloads"][leg] = {}
    for fam, e in sorted(fams.items()):
        rec = {"fetch_bytes_per_unit": e["fetch"] / units, "write_bytes_per_unit": e["write"] / units,
               "hbm_bytes_per_unit": (e["fetch"] + e["write"]) / units, "dispatches_per_unit": e["dispatches_fetch"] / units}
        out["workloads"][leg][fam] = rec
        md.append(f"| {leg} | {fam} | {rec['dispatches_per_unit']:.1f} | {rec['fetch_bytes_per_unit'] / 1e9:.3f} | "
                  f"{rec['write_bytes_per_unit'] / 1e9:.3f} | {rec['hbm_bytes_per_unit'] / 1e9:.3f} |")
# the headline's figure in the form bench.py's roofline block quotes: bytes per artn_k_bits launch (20 per contraction)
n30 = out["workloads"].get("n30", {}).get("bits")
if n30:
    out["kernel"] = "artn_k_bits"
    out["hbm_bytes_per_launch"] = n30["hbm_bytes_per_unit"] / n30["dispatches_per_unit"]
out["runtime_copies"] = copies
md += ["", "## `__amd_rocclr_copyBuffer` dispatches (the runtime's blit kernel: host-to-device uploads and device copies)", "",
       "| leg | total | before the first artn kernel (leaf / index uploads) | per unit afterwards |", "|---|---|---|---|"]
for leg, c in copies.items():
    md.append(f"| {leg} | {c['total']} | {c['before_first_artn_kernel']} | {c['per_unit_after']:.1f} |")
json.dump(out, open(os.path.join(ROOT, "profiles", f"{R}_traffic.json"), "w"), indent=1)
open(os.path.join(ROOT, "profiles", f"{R}_traffic.md"), "w").write("\n".join(md) + "\n")
print("\n".join(md))
