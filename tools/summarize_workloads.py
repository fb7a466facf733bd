#!/usr/bin/env python3
"""Condense gpurun_out/profw_<round>/ (tools/profile_workloads.sh) into profiles/<round>_workloads.md."""
import csv, glob, json, os, sys
R = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
root = os.path.join(ROOT, "gpurun_out", f"profw_{R}")
md = [f"# rocprofv3 kernel statistics of the secondary workloads, round {R}", "",
      f"`tools/profile_workloads.sh {R}` on one MI355X: `rocprofv3 --kernel-trace --stats` around",
      "`bench.py --workload W --slices 4 --steps 2 --warmup 1` (13 slices incl. warm-up and check) and around",
      "`tools/trace_sparse.py` (3 runs of the n30 sparse fixtures) and `tools/trace_c128.py` (3 runs of the n30 dense fixture in complex128).", ""]
for w in ("n53", "n53m20", "n53m20b", "n53m20b_bf16", "n53m20bb", "n53m20bb_bf16", "rand2", "rand4", "rand3", "rand6", "n30_sparse10000", "n30_sparse100", "n30_c128"):
    fs = sorted(glob.glob(f"{root}/{w}/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
    if not fs:
        continue
    rows = list(csv.DictReader(open(fs[-1])))
    md += [f"## {w}", ""]
    j = f"{root}/{w}.json"
    if os.path.exists(j) and os.path.getsize(j):
        d = json.load(open(j))
        c = d["config"]
        md += [f"bench line under rocprof: {d['value']:.1f} TFLOP/s, {c['ms_per_slice_per_rank']:.2f} ms per slice "
               f"(slice 0 vs reference: {c['slice0_err_rel_to_max_abs_or_rms']:.1e} of max(|amp|, rms); strict {c['slice0_rel_err_strict_over_1e-3rms']:.1e}"
               + (f"; fidelity {c['slice0_fidelity_vs_reference']:.5f}" if c.get("slice0_fidelity_vs_reference") else "") + ")", ""]
    md += ["| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    for r in rows[:8]:
        name = r["Name"].split("(")[0].replace("void ", "")[:64]
        md.append(f"| `{name}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")
    md.append("")
# matrix-core counters of the GEMM kernel on the big-batch slice
import collections
pm = glob.glob(f"{root}/n53m20b_pmc/**/*_counter_collection.csv", recursive=True)
ck = glob.glob(f"{root}/n53m20b_clk/**/*_counter_collection.csv", recursive=True)
if pm:
    def load(f):
        per = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            if "artn_k_gemm" not in r["Kernel_Name"]:
                continue
            d = per.setdefault(r["Dispatch_Id"], {"t0": int(r["Start_Timestamp"]), "t1": int(r["End_Timestamp"])})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        return [d for d in per.values() if d["t1"] - d["t0"] > 1e8]   # the 2^15-contracted-values launch(es)
    sb = load(pm[0])
    clk = 0.0
    if ck:
        cb = load(ck[0])
        if cb:
            clk = sum(d.get("GRBM_GUI_ACTIVE", 0) for d in cb) / 8 / (sum(d["t1"] - d["t0"] for d in cb) * 1e-9)
    if sb:
        dur = sum(d["t1"] - d["t0"] for d in sb) * 1e-9
        mf = sum(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) for d in sb)
        md += ["## artn_k_gemm on the n53 m20 big-batch slice (15 contracted bits, fp32, 3M arithmetic): PMC counters", "",
               f"* launches counted: {len(sb)}, {dur / len(sb) * 1e3:.1f} ms each",
               f"* SQ_VALU_MFMA_BUSY_CYCLES / (duration x 1024 SIMDs x clock): {mf / (dur * 1024 * (clk or 2.1e9)):.3f} (clock {clk / 1e9:.2f} GHz from GRBM_GUI_ACTIVE)",
               f"* SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE: {sum(d.get('SQ_LDS_BANK_CONFLICT', 0) for d in sb) / max(sum(d.get('SQ_LDS_IDX_ACTIVE', 0) for d in sb), 1):.3f}",
               f"* SQ_WAIT_ANY / SQ_WAVE_CYCLES: {sum(d.get('SQ_WAIT_ANY', 0) for d in sb) / sum(d.get('SQ_WAVE_CYCLES', 1) for d in sb):.3f}",
               f"* SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES: {sum(d.get('SQ_WAIT_INST_ANY', 0) for d in sb) / sum(d.get('SQ_WAVE_CYCLES', 1) for d in sb):.3f}", ""]
# the same for the packed-operand GEMM of the reduced-precision mode (1 024 bitstrings; round 6: and 65 536)
for leg, flop, what in (("n53m20b_bf16", 1.407e14, "n53 m20 big-batch slice (15 contracted bits"),
                        ("n53m20bb_bf16", 3.518e13, "n53 m20 big-batch slice of 65 536 bitstrings (13 contracted bits")):
    pm = glob.glob(f"{root}/{leg}_pmc/**/*_counter_collection.csv", recursive=True)
    ck = glob.glob(f"{root}/{leg}_clk/**/*_counter_collection.csv", recursive=True)
    if not pm:
        continue
    def load2(f):
        per = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            if "artn_k_pgemm" not in r["Kernel_Name"]:
                continue
            d = per.setdefault(r["Dispatch_Id"], {"t0": int(r["Start_Timestamp"]), "t1": int(r["End_Timestamp"])})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        return [d for d in per.values() if d["t1"] - d["t0"] > 2e7]
    sb = load2(pm[0])
    clk = 0.0
    if ck:
        cb = load2(ck[0])
        if cb:
            clk = sum(d.get("GRBM_GUI_ACTIVE", 0) for d in cb) / 8 / (sum(d["t1"] - d["t0"] for d in cb) * 1e-9)
    if sb:
        dur = sum(d["t1"] - d["t0"] for d in sb) * 1e-9
        mf = sum(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) for d in sb)
        md += [f"## artn_k_pgemm on the {what}, bf16 operands packed once, LDS-DMA): PMC counters", "",
               f"* launches counted: {len(sb)}, {dur / len(sb) * 1e3:.1f} ms each ({flop:.3e} real FLOP: {flop / (dur / len(sb)) / 1e12:.0f} TFLOP/s)",
               f"* SQ_VALU_MFMA_BUSY_CYCLES / (duration x 1024 SIMDs x clock): {mf / (dur * 1024 * (clk or 2.1e9)):.3f} (clock {clk / 1e9:.2f} GHz from GRBM_GUI_ACTIVE)",
               f"* SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE: {sum(d.get('SQ_LDS_BANK_CONFLICT', 0) for d in sb) / max(sum(d.get('SQ_LDS_IDX_ACTIVE', 0) for d in sb), 1):.3f}",
               f"* SQ_WAIT_ANY / SQ_WAVE_CYCLES: {sum(d.get('SQ_WAIT_ANY', 0) for d in sb) / sum(d.get('SQ_WAVE_CYCLES', 1) for d in sb):.3f}",
               f"* SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES: {sum(d.get('SQ_WAIT_INST_ANY', 0) for d in sb) / sum(d.get('SQ_WAVE_CYCLES', 1) for d in sb):.3f}", ""]
# the complex128 state-streaming kernel: matrix-core counters of its big launches
pm = glob.glob(f"{root}/n30_c128_pmc/**/*_counter_collection.csv", recursive=True)
ck = glob.glob(f"{root}/n30_c128_clk/**/*_counter_collection.csv", recursive=True)
if pm:
    def load3(f):
        per = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            if "artn_k_bits128" not in r["Kernel_Name"]:
                continue
            d = per.setdefault(r["Dispatch_Id"], {"t0": int(r["Start_Timestamp"]), "t1": int(r["End_Timestamp"])})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        return [d for d in per.values() if d["t1"] - d["t0"] > 2e6]
    sb = load3(pm[0])
    clk = 0.0
    if ck:
        cb = load3(ck[0])
        if cb:
            clk = sum(d.get("GRBM_GUI_ACTIVE", 0) for d in cb) / 8 / (sum(d["t1"] - d["t0"] for d in cb) * 1e-9)
    if sb:
        dur = sum(d["t1"] - d["t0"] for d in sb) * 1e-9
        mf = sum(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) for d in sb)
        md += ["## artn_k_bits128 on the n30 contraction in complex128 (fused pairs, v_mfma_f64_16x16x4_f64): PMC counters", "",
               f"* launches counted: {len(sb)} (longer than 2 ms), {dur * 1e3:.1f} ms together",
               f"* SQ_VALU_MFMA_BUSY_CYCLES / (duration x 1024 SIMDs x clock): {mf / (dur * 1024 * (clk or 2.1e9)):.3f} (clock {clk / 1e9:.2f} GHz from GRBM_GUI_ACTIVE)",
               f"* SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE: {sum(d.get('SQ_LDS_BANK_CONFLICT', 0) for d in sb) / max(sum(d.get('SQ_LDS_IDX_ACTIVE', 0) for d in sb), 1):.3f}",
               f"* SQ_WAIT_ANY / SQ_WAVE_CYCLES: {sum(d.get('SQ_WAIT_ANY', 0) for d in sb) / sum(d.get('SQ_WAVE_CYCLES', 1) for d in sb):.3f}",
               f"* SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES: {sum(d.get('SQ_WAIT_INST_ANY', 0) for d in sb) / sum(d.get('SQ_WAVE_CYCLES', 1) for d in sb):.3f}", ""]
st = f"{root}/sparse_times.txt"
if os.path.exists(st):
    md += ["## n30 sparse wall times (tools/time_sparse.py)", "", "```", open(st).read().strip(), "```", ""]
open(os.path.join(ROOT, "profiles", f"{R}_workloads.md"), "w").write("\n".join(md))
print("\n".join(md))
