#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv compactly: name, calls, average us, total ms, max us."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print(f'{r["Name"].split("(")[0][:44]:44s} calls {int(r["Calls"]):6d}  avg {float(r["AverageNs"]) / 1e3:8.2f} us  '
          f'total {float(r["TotalDurationNs"]) / 1e6:8.3f} ms  max {float(r["MaxNs"]) / 1e3:8.2f} us')
