#!/usr/bin/env python3
"""Active launches of the conditional re-sort kernels in a rocprofv3 kernel trace (the idle ones return at once and
drown the averages of kernel_stats.csv).   usage: scripts/resort_times.py <..._kernel_trace.csv> [min_us]"""
import collections, csv, sys
rows = collections.defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "k_rb_" in n:
            rows[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
tot = 0.0
for n in sorted(rows):
    act = [d for d in rows[n] if d > thr]
    idle = [d for d in rows[n] if d <= thr]
    a = sum(act) / max(len(act), 1)
    tot += a
    print(f"{n:22s} active {len(act):4d} avg {a:7.2f} us   idle {len(idle):4d} avg {sum(idle) / max(len(idle), 1):5.2f} us")
print(f"{'sum of active averages':22s} {tot:7.2f} us")
