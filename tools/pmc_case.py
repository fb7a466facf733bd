#!/usr/bin/env python3
"""Per-kernel PMC summary of tools/pmc_case.sh (launches longer than 1 ms, averaged per instantiation)."""
import collections, csv, glob, os, sys
out = sys.argv[1]
for sub in ("a", "b", "c"):
    files = glob.glob(os.path.join(out, sub, "**", "*_counter_collection.csv"), recursive=True)
    if not files:
        log = os.path.join(out, sub + ".log")
        if os.path.exists(log):   # (pass "c" is optional)
            print(sub, "no counters:", open(log).read()[-600:])
        continue
    per = collections.OrderedDict()
    for r in csv.DictReader(open(files[0])):
        d = per.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"].split("(")[0].replace("void ", ""), "t": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    agg = collections.OrderedDict()
    for d in per.values():
        if d["t"] < 1e6: continue
        a = agg.setdefault(d["name"], collections.Counter())
        a["n"] += 1
        for k, v in d.items():
            if k != "name": a[k] += v
    if os.environ.get("PER_DISPATCH"):   # every launch longer than 1 ms on its own line (same instantiation, different plans)
        for d in sorted(per.values(), key=lambda d: -d["t"]):
            if d["t"] >= 1e6:
                print(f"  {d['name'][:44]:44s} ms={d['t']/1e6:.2f} " + " ".join(f"{k}={v:.3g}" for k, v in d.items() if k not in ("name", "t")))
        continue
    for name, a in agg.items():
        n = a["n"]
        print(f"{name[:60]:60s} n={n} ms={a['t']/n/1e6:.2f} " + " ".join(f"{k}={v/n:.3g}" for k, v in a.items() if k not in ("n", "t")))
