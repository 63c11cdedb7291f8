#!/usr/bin/env python3
"""kernel_trace.csv of scratch/team_world_prof.py -> launches by kernel: the contact solve's kernels, the team transport's,
and every blit (copyBuffer / fillBuffer) the run made.  usage: team_trace_summary.py <kernel_trace.csv> [<stdout of the run>]"""
import csv
import sys
from collections import Counter

rows = list(csv.DictReader(open(sys.argv[1])))
name_col = next(c for c in rows[0] if c.lower() in ("kernel_name", "name"))
cnt, dur = Counter(), Counter()
for r in rows:
    n = r[name_col].split("(")[0].replace("void ", "").strip()
    cnt[n] += 1
    try:
        dur[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    except Exception:  # noqa: BLE001
        pass
if len(sys.argv) > 2:
    print(open(sys.argv[2]).read().strip())
decides = sum(v for k, v in cnt.items() if "k_ct_decide" in k)
blits = sum(v for k, v in cnt.items() if "copyBuffer" in k or "fillBuffer" in k or "Memcpy" in k)
print(f"kernel launches in the whole process: {sum(cnt.values())}; k_ct_decide launches (two per rank and Newton iteration): {decides}; "
      f"blit kernels (copyBuffer / fillBuffer: set-up, migrations, the final downloads): {blits}")
print(f"{'launches':>9} {'total us':>10} {'avg us':>8}  kernel")
for n, c in sorted(cnt.items(), key=lambda kv: -dur[kv[0]]):
    print(f"{c:9d} {dur[n] / 1e3:10.1f} {dur[n] / 1e3 / c:8.2f}  {n}")
