"""Builds A/B variants of the engine for same-box comparisons (scratch/ab_run.py):
    python scratch/ab_build.py name1="-DFOO=1 -DBAR=0" name2="" ...
Each variant becomes drake_amd/variants/libmpm_hip_<name>.so (git-ignored, travels with gpurun)."""
import os, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from drake_amd import _build
os.makedirs(os.path.join(ROOT, "drake_amd", "variants"), exist_ok=True)
os.environ.pop("MPM_HIP_LIBRARY", None)
def one(arg):
    name, _, flags = arg.partition("=")
    out = os.path.join(ROOT, "drake_amd", "variants", f"libmpm_hip_{name}.so")
    _build.build(force=True, extra=tuple(flags.split()), out=out)
    return out
with ThreadPoolExecutor(4) as ex:
    for o in ex.map(one, sys.argv[1:]):
        print(o)
