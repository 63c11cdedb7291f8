#!/usr/bin/env python3
"""Device assembly of one kernel of the engine and its instruction mix per loop nest (no GPU needed).
usage: scripts/isa.py <kernel substring> [--src DIR] [--out FILE] [-- extra hipcc flags]"""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = []
if "--" in args:
    k = args.index("--"); args, extra = args[:k], args[k + 1:]
src = os.path.join(ROOT, "drake_amd", "csrc")
out = None
if "--src" in args:
    k = args.index("--src"); src = args[k + 1]; del args[k:k + 2]
if "--out" in args:
    k = args.index("--out"); out = args[k + 1]; del args[k:k + 2]
name = args[0]
asm = "/tmp/isa_%d.s" % os.getpid()
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-slp-vectorize", "-munsafe-fp-atomics", "-fno-gpu-rdc",
                "--cuda-device-only", "-S", *extra, "-o", asm, os.path.join(src, "mpm_engine.hip")], check=True,
               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
lines = open(asm).read().splitlines()
os.remove(asm)
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % name, l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end + 1]
if out:
    open(out, "w").write("\n".join(body))
depth = 0
mix = collections.defaultdict(collections.Counter)
i = 0
while i < len(body):
    l = body[i]
    if re.match(r"^\.LBB", l) or l.startswith("; %bb."):
        # a block header: its loop depth is in its comment, possibly continued on comment-only lines
        # ("Parent Loop ... Depth=1" ... "=> This Inner Loop Header: Depth=4": the last one counts)
        ds = re.findall(r"Depth[= ](\d+)", l)
        j = i + 1
        while j < len(body) and re.match(r"^\s+;", body[j]):
            ds += re.findall(r"Depth[= ](\d+)", body[j])
            j += 1
        depth = int(ds[-1]) if ds else 0
        i = j
        continue
    i += 1
    t = l.strip().split()
    if not t or t[0].startswith(";") or t[0].startswith("."):
        continue
    op = t[0]
    cls = ("MFMA" if op.startswith("v_mfma") else "VALU" if op.startswith("v_") else "SALU" if op.startswith("s_") else
           "LDS" if op.startswith("ds_") else "VMEM" if re.match(r"(global|buffer|scratch|flat)_", op) else "other")
    mix[depth][cls] += 1
print("static instruction counts by loop depth:")
for d in sorted(mix):
    print("  depth", d, dict(mix[d]))
