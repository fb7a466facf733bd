#!/usr/bin/env python3
"""Busy/idle analysis of a rocprofv3 kernel trace CSV: python3 tools/timeline.py trace.csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40], r.get("Queue_Id", "")) for r in rows)
# take the last third of the trace (steady state)
t_lo = ev[len(ev) * 2 // 3][0]
ev = [e for e in ev if e[0] >= t_lo]
span = ev[-1][1] - ev[0][0]
busy, cur_s, cur_e = 0, None, None
for s, e, _, _ in ev:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, _, _ in ev)
big = sum(e - s for s, e, n, _ in ev if "artn_k_bits" in n)
print(f"kernels {len(ev)}  span {span/1e6:.2f} ms  union-busy {busy/1e6:.2f} ms  idle {(span-busy)/1e6:.2f} ms  sum-of-durations {tot/1e6:.2f} ms  (artn_k_bits {big/1e6:.2f} ms)")
queues = {}
for s, e, n, q in ev: queues.setdefault(q, 0); queues[q] += 1
print("launches per queue:", queues)
# gap histogram between consecutive kernel starts on the union timeline
gaps = []
pe = None
for s, e, n, _ in ev:
    if pe is not None and s > pe: gaps.append(s - pe)
    pe = max(pe, e) if pe else e
gaps.sort()
if gaps: print(f"gaps: n={len(gaps)} median {gaps[len(gaps)//2]/1e3:.1f} us  mean {sum(gaps)/len(gaps)/1e3:.1f} us  max {gaps[-1]/1e3:.1f} us total {sum(gaps)/1e6:.2f} ms")
