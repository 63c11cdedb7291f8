#!/usr/bin/env python3
"""Register / LDS / scratch usage of every kernel of the engine, from hipcc's own resource remarks
(no GPU needed).  usage: scripts/kres.py [pattern] [-- extra hipcc flags]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from drake_amd import _build  # noqa: E402

args = sys.argv[1:]
extra = []
if "--" in args:
    k = args.index("--")
    args, extra = args[:k], args[k + 1:]
pat = re.compile(args[0]) if args else None
cmd = ["/opt/rocm/bin/hipcc", *_build.FLAGS, *extra, "-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/kres.so",
       os.path.join(_build.CSRC, "mpm_engine.hip")]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for ln in out.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+: (.*) \[-Rpass", ln) or re.search(r"remark: (.*) \[-Rpass", ln)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:") or t.startswith("Name:"):
        cur = subprocess.run(["c++filt", t.split(":", 1)[1].strip()], capture_output=True,
                             text=True).stdout.strip().split("(")[0]
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
print(f"{'kernel':40s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'LDS':>7s} {'occ':>4s}")
for k, r in rows.items():
    if pat and not pat.search(k):
        continue
    print(f"{k:40s} {r.get('VGPRs', '?'):>5s} {r.get('AGPRs', '?'):>5s} {r.get('TotalSGPRs', '?'):>5s} "
          f"{r.get('ScratchSize [bytes/lane]', '?'):>8s} {r.get('LDS Size [bytes/block]', '?'):>7s} "
          f"{r.get('Occupancy [waves/SIMD]', '?'):>4s}")
