#!/bin/bash
# Builds the engine at milestone commits into drake_amd/variants/libmpm_hip_<tag>.so (for a same-box comparison with
# scratch/ab_run.py).  usage: scripts/build_history.sh tag=commit ...
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
for a in "$@"; do
  tag=${a%%=*}; c=${a#*=}
  d=$(mktemp -d)
  git -C "$R" archive "$c" drake_amd/csrc include | tar -x -C "$d"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -shared -fvisibility=hidden --offload-arch=gfx950 -fno-slp-vectorize -munsafe-fp-atomics \
    -fno-gpu-rdc -w -o "$R/drake_amd/variants/libmpm_hip_$tag.so" "$d/drake_amd/csrc/mpm_engine.hip"
  rm -rf "$d"; echo "$tag <- $c"
done
