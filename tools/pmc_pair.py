#!/usr/bin/env python3
"""Summary of tools/pmc_pair.sh: per kernel instantiation (launches longer than 1 ms), counters averaged per launch."""
import collections, csv, glob, os, sys
out = sys.argv[1]
for who in ("real", "probe"):
    agg = collections.OrderedDict()
    for sub in ("a", "b", "c"):
        files = glob.glob(os.path.join(out, f"{who}_{sub}", "**", "*_counter_collection.csv"), recursive=True)
        if not files:
            print(who, sub, "no counters:", open(os.path.join(out, f"{who}_{sub}.log")).read()[-400:])
            continue
        per = collections.OrderedDict()
        for r in csv.DictReader(open(files[0])):
            d = per.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"].replace("void ", "")[:90], "t": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        seen = collections.Counter()
        for d in per.values():
            if d["t"] < 1e6: continue
            seen[d["name"]] += 1
            key = (d["name"], seen[d["name"]] if who == "probe" else 0)   # probe: same instantiation launched with different occupancy
            a = agg.setdefault(key, collections.Counter())
            a["n_" + sub] += 1
            for k, v in d.items():
                if k == "t": a["t_" + sub] += v
                elif k != "name": a[k] += v; a["cnt_" + k] += 1
    for (name, which), a in agg.items():
        ms = a["t_a"] / max(1, a["n_a"]) / 1e6
        vals = {k: a[k] / a["cnt_" + k] for k in a if not k.startswith(("n_", "t_", "cnt_"))}
        print(f"{who} {name} #{which} ms={ms:.3f}")
        print("    " + " ".join(f"{k}={v:.4g}" for k, v in sorted(vals.items())))
