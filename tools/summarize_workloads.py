#!/usr/bin/env python3
"""Condense gpurun_out/profw_<round>/ (tools/profile_workloads.sh) into profiles/<round>_workloads.md."""
import csv, glob, json, os, sys
R = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
root = os.path.join(ROOT, "gpurun_out", f"profw_{R}")
md = [f"# rocprofv3 kernel statistics of the secondary workloads, round {R}", "",
      f"`tools/profile_workloads.sh {R}` on one MI355X: `rocprofv3 --kernel-trace --stats` around",
      "`bench.py --workload W --slices 4 --steps 2 --warmup 1` (13 slices incl. warm-up and check) and around",
      "`tools/trace_sparse.py` (3 runs of the n30 sparse fixtures).", ""]
for w in ("n53", "n53m20", "rand2", "rand4", "n30_sparse10000", "n30_sparse100"):
    fs = sorted(glob.glob(f"{root}/{w}/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
    if not fs:
        continue
    rows = list(csv.DictReader(open(fs[-1])))
    md += [f"## {w}", ""]
    j = f"{root}/{w}.json"
    if os.path.exists(j) and os.path.getsize(j):
        d = json.load(open(j))
        md += [f"bench line under rocprof: {d['value']:.1f} TFLOP/s, {d['ms_per_step'] / 4:.2f} ms per slice "
               f"(slice 0 vs reference: {d['config']['slice0_rel_err_vs_reference']:.1e})", ""]
    md += ["| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    for r in rows[:8]:
        name = r["Name"].split("(")[0].replace("void ", "")[:64]
        md.append(f"| `{name}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")
    md.append("")
st = f"{root}/sparse_times.txt"
if os.path.exists(st):
    md += ["## n30 sparse wall times (tools/time_sparse.py)", "", "```", open(st).read().strip(), "```", ""]
open(os.path.join(ROOT, "profiles", f"{R}_workloads.md"), "w").write("\n".join(md))
print("\n".join(md))
