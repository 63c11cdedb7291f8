#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3i
R=$PWD
cd /tmp && export TMPDIR=/tmp
for v in cur emu64 emu320 emu1088; do
  MPM_HIP_LIBRARY=$R/drake_amd/variants/libmpm_hip_$v.so timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3i/$v -- python3 $R/scripts/bench_contact.py --survey-config3 --device-pairs --steps 20 > $R/gpurun_out/r3i/$v.json 2> $R/gpurun_out/r3i/$v.log
  echo "== $v"; cat $R/gpurun_out/r3i/$v.json | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_substep'], d['newton_iterations_mean'], d['contacts_mean'], d['solve_us_per_iteration'])"
  grep -E "k_ct_tile|k_ct_ls|k_ct_node_dir|k_ct_decide" $R/gpurun_out/r3i/$v/*/*kernel_stats.csv | cut -d, -f1-4
done
