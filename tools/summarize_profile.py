#!/usr/bin/env python3
"""Condense gpurun_out/prof_<round>/ (rocprofv3 CSVs) into profiles/<round>_*.{csv,md,json}."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

R = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", f"prof_{R}")
DST = os.path.join(ROOT, "profiles")
os.makedirs(DST, exist_ok=True)


def one(pattern):
    hits = glob.glob(os.path.join(SRC, pattern), recursive=True)
    return hits[0] if hits else None


stats = one("kt/**/*_kernel_stats.csv")
shutil.copy(stats, os.path.join(DST, f"{R}_kernel_stats.csv"))
rows = list(csv.DictReader(open(stats)))


def counters(sub):
    f = one(f"{sub}/**/*_counter_collection.csv")
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if not any(k in r["Kernel_Name"] for k in ("artn_k_bits", "artn_k_alt", "artn_k_wide")):
            continue
        d = per.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"].split("(")[0].replace("void ", ""),
                                              "t0": int(r["Start_Timestamp"]), "t1": int(r["End_Timestamp"])})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return per


sq, fe, wr = counters("pmc_sq"), counters("pmc_fetch"), counters("pmc_write")
big = lambda per: [d for d in per.values() if d["t1"] - d["t0"] > 1e6]  # launches longer than 1 ms
md = [f"# rocprofv3 summary, round {R}", "",
      "Command: `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-workloads` (Sycamore n30 m14 full amplitude, 1 x MI355X).",
      "Kernel trace and each PMC group were collected in separate rocprofv3 runs.", "",
      "## Kernel time (rocprofv3 --kernel-trace --stats)", "",
      "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
for r in rows[:12]:
    name = r["Name"].split("(")[0].replace("void ", "")[:60]
    md.append(f"| `{name}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")
bits = [r for r in rows if any(k in r["Name"] for k in ("artn_k_bits", "artn_k_alt", "artn_k_wide"))]
tot_ns = sum(float(r["TotalDurationNs"]) for r in bits)
tot_calls = sum(int(r["Calls"]) for r in bits)
md += ["", f"All `artn_k_bits<KB1,KB2>` instantiations together: {tot_calls} launches, {tot_ns / 1e6:.2f} ms, "
           f"average {tot_ns / tot_calls / 1e3:.1f} us per launch ({tot_calls // 20} contractions: warm-up, check, 3 timed with per-launch events, 3 without).", ""]

fb, wb = big(fe), big(wr)
fetch = sum(d.get("FETCH_SIZE", 0) for d in fb) * 1024 * 2  # KB -> B; gfx950 halves wide reads (guide)
write = sum(d.get("WRITE_SIZE", 0) for d in wb) * 1024
n = max(len(fb), 1)
md += ["## HBM traffic of the MFMA kernel (launches > 1 ms)", "",
       f"* FETCH_SIZE x 2 (gfx950 counts 128-B requests at 64 B): {fetch / n / 2**30:.3f} GiB per launch",
       f"* WRITE_SIZE: {write / max(len(wb), 1) / 2**30:.3f} GiB per launch",
       f"* launches counted: {len(fb)} (fetch pass), {len(wb)} (write pass)", ""]
sb = big(sq)
if sb:
    mf = sum(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) for d in sb)
    dur = sum(d["t1"] - d["t0"] for d in sb) * 1e-9
    ga = sum(d.get("GRBM_GUI_ACTIVE", 0) for d in fb)
    clk = ga / 8 / (sum(d["t1"] - d["t0"] for d in fb) * 1e-9) if fb else 0
    conf = sum(d.get("SQ_LDS_BANK_CONFLICT", 0) for d in sb) / max(sum(d.get("SQ_LDS_IDX_ACTIVE", 0) for d in sb), 1)
    md += ["## Matrix-core and LDS counters (launches > 1 ms)", "",
           f"* SQ_VALU_MFMA_BUSY_CYCLES / (duration x 1024 SIMDs x clock): "
           f"{mf / (dur * 1024 * (clk or 2.2e9)):.3f}  (clock from GRBM_GUI_ACTIVE / 8 / time = {clk / 1e9:.2f} GHz)",
           f"* SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE: {conf:.3f}",
           f"* SQ_WAIT_ANY / SQ_WAVE_CYCLES: {sum(d.get('SQ_WAIT_ANY', 0) for d in sb) / sum(d.get('SQ_WAVE_CYCLES', 1) for d in sb):.3f}",
           f"* SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES: {sum(d.get('SQ_WAIT_INST_ANY', 0) for d in sb) / sum(d.get('SQ_WAVE_CYCLES', 1) for d in sb):.3f}", ""]
# per instantiation: launches, average duration, matrix-pipe busy fraction, LDS bank-conflict ratio
if sb:
    clk_eff = clk or 2.1e9
    per_name = collections.OrderedDict()
    for d in sb:
        e = per_name.setdefault(d["name"], {"n": 0, "t": 0.0, "mf": 0.0, "conf": 0.0, "act": 0.0})
        e["n"] += 1
        e["t"] += (d["t1"] - d["t0"]) * 1e-9
        e["mf"] += d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
        e["conf"] += d.get("SQ_LDS_BANK_CONFLICT", 0)
        e["act"] += d.get("SQ_LDS_IDX_ACTIVE", 0)
    md += ["## Per instantiation (launches > 1 ms; template arguments KB1, KB2, BIGK, NP, GATHER, NT, M3, FULL -- artn_k_alt: KB1, KB2, NT, M3; artn_k_wide: KB1, KB2)", "",
           "| kernel | launches | avg ms | SQ_VALU_MFMA_BUSY | LDS conflict / active |", "|---|---|---|---|---|"]
    for name, e in per_name.items():
        md.append(f"| `{name[:70]}` | {e['n']} | {e['t'] / e['n'] * 1e3:.2f} | {e['mf'] / (e['t'] * 1024 * clk_eff):.3f} | {e['conf'] / max(e['act'], 1):.3f} |")
    md.append("")
bj = os.path.join(SRC, "bench_under_rocprof.json")
if os.path.exists(bj) and os.path.getsize(bj):
    md += ["## bench.py line of the kernel-trace run", "", "```", open(bj).read().strip(), "```", ""]
open(os.path.join(DST, f"{R}_summary.md"), "w").write("\n".join(md))
fa, wa = list(fe.values()), list(wr.values())
fetch_all = sum(d.get("FETCH_SIZE", 0) for d in fa) * 1024 * 2 / max(len(fa), 1)
write_all = sum(d.get("WRITE_SIZE", 0) for d in wa) * 1024 / max(len(wa), 1)
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (kernel_source_sha16: the sources these counters belong to)
json.dump({"round": R, "kernel": "artn_k_bits", "kernel_source_sha16": bench.kernel_source_sha16(),
           # averaged over EVERY artn_k_bits launch, like bench.py's roofline.achieved and
           # algorithmic_bytes_per_launch (20 launches per contraction: 13 fused 8-GiB passes, 7 growth steps)
           "hbm_bytes_per_launch": fetch_all + write_all, "launches": len(fa),
           "launches_over_1ms": len(fb),
           "hbm_bytes_per_launch_over_1ms": (fetch / n + write / max(len(wb), 1)),
           "fetch_bytes_per_launch_over_1ms": fetch / n, "write_bytes_per_launch_over_1ms": write / max(len(wb), 1),
           "method": "rocprofv3 --pmc FETCH_SIZE (x2, gfx950) and --pmc WRITE_SIZE in separate passes"},
          open(os.path.join(DST, f"{R}_traffic_headline.json"), "w"), indent=1)   # (bench.py quotes {R}_traffic.json: every leg,
#                                                                     tools/profile_traffic.sh + tools/summarize_traffic.py)
print("\n".join(md))
