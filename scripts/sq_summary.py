#!/usr/bin/env python3
"""Per-kernel SQ counters of one rocprofv3 --pmc pass -> profiles/<tag>_sq_counters.json.

  python scripts/sq_summary.py <tag> <counter_collection.csv>

Averages over the launches of each engine kernel (idle launches of the conditional re-sort kernels
left out).  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles
(/opt/skills/guides/MI355X_MICROARCH.md, cycle-constants table)."""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag, path = sys.argv[1:3]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if k.startswith("mpm::"):
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = []
    for k in sorted(acc):
        c = acc[k]
        if "k_rb_" in k and "SQ_INSTS_VALU" in c:     # keep the launches that did the work
            lim = 0.05 * max(c["SQ_INSTS_VALU"])
            keep = [i for i, v in enumerate(c["SQ_INSTS_VALU"]) if v > lim]
            c = {n: [v[i] for i in keep if i < len(v)] for n, v in c.items()}
        avg = {n: sum(v) / len(v) for n, v in c.items() if v}
        rec = dict(kernel=k, launches=len(next(iter(c.values()))), counters=avg)
        if avg.get("SQ_WAVE_CYCLES"):
            wc = avg["SQ_WAVE_CYCLES"]
            rec["fractions_of_wave_cycles"] = {n: avg[n] / wc for n in ("SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY")
                                               if n in avg}
        out.append(rec)
    with open(os.path.join(ROOT, "profiles", f"{tag}_sq_counters.json"), "w") as f:
        json.dump(out, f, indent=1)
    # what bench.py replays into roofline.valu_issue_ms
    config = sys.argv[3] if len(sys.argv) > 3 else "cloth_1m"
    with open(os.path.join(ROOT, "profiles", "sq_counters.json"), "w") as f:
        json.dump(dict(config=config, source=f"profiles/{tag}_sq_counters.json", head=os.environ.get("MPM_PROFILE_HEAD"), kernels=out), f, indent=1)
    for r in out:
        print(r["kernel"], {k: round(v) for k, v in r["counters"].items()})


if __name__ == "__main__":
    main()
