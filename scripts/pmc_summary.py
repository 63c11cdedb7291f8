#!/usr/bin/env python3
"""Summarise rocprofv3 outputs into profiles/.

  python scripts/pmc_summary.py <tag> <kernel_stats.csv> <fetch counter_collection.csv> <write counter_collection.csv> [config]

writes profiles/<tag>_kernel_stats.csv (copy), profiles/<tag>_pmc_traffic.csv and
profiles/pmc_traffic.json (read by bench.py for roofline.traffic).

Counter handling follows /opt/skills/guides/MI355X_MICROARCH.md, section HBM: FETCH_SIZE and
WRITE_SIZE come from separate passes, both are in KiB, and on gfx950 FETCH_SIZE tallies 128-byte
read requests at 64 bytes, so it is doubled.  Calibration inside the same runs: k_init_vertex_adjacency
writes exactly 32 B per vertex (WRITE_SIZE matches to the byte) and reads 4 B per vertex + 4 B per
face corner (2 x FETCH_SIZE matches to 4%).  Idle launches of the conditional rebuild kernels are
excluded from their averages (they move no data).
"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(path):
    d = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            d[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return d


def main():
    tag, stats, fetch, write = sys.argv[1:5]
    config = sys.argv[5] if len(sys.argv) > 5 else "cloth_1m"
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    shutil.copy(stats, os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))
    durations = {}
    with open(stats) as f:
        for r in csv.DictReader(f):
            durations[r["Name"].split("(")[0]] = float(r["AverageNs"])
    F, W = per_kernel(fetch), per_kernel(write)
    out = {}
    rows = []
    for k in sorted(set(F) | set(W)):
        if not k.startswith("mpm::") and not k.startswith("void mpm::"):
            continue
        name = k.replace("void ", "")
        fv, wv = F.get(k, []), W.get(k, [])
        if "k_rb_" in name:  # conditional kernels: keep the launches that did the work
            fv = [x for x in fv if x > 100.0]
            wv = [x for x in wv if x > 100.0]
        fk = sum(fv) / len(fv) if fv else 0.0
        wk = sum(wv) / len(wv) if wv else 0.0
        rd, wr = 2.0 * fk * 1024.0, wk * 1024.0
        out[name] = dict(fetch_size_kib=fk, write_size_kib=wk, read_bytes=rd, write_bytes=wr,
                         hbm_bytes_per_launch=rd + wr, launches=max(len(fv), len(wv)),
                         avg_duration_ns=durations.get(k))
        rows.append((name, max(len(fv), len(wv)), fk, wk, rd + wr, durations.get(k)))
    with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.csv"), "w") as f:
        f.write("kernel,launches,FETCH_SIZE_KiB_avg,WRITE_SIZE_KiB_avg,hbm_bytes_per_launch(2*fetch+write),avg_duration_ns(kernel-trace run)\n")
        for r in rows:
            f.write(",".join(str(x) for x in r) + "\n")
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w") as f:
        json.dump(dict(config=config, source=f"profiles/{tag}_pmc_traffic.csv", correction="hbm = 2*FETCH_SIZE + WRITE_SIZE (KiB)",
                       head=os.environ.get("MPM_PROFILE_HEAD"), kernels=out), f, indent=1)
    for r in rows:
        print(r)


if __name__ == "__main__":
    main()
