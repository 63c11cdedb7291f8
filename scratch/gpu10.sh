#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3j
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r3j/bench.json 2> gpurun_out/r3j/bench.err; echo "bench rc=$?"
tail -3 gpurun_out/r3j/bench.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r3j/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','steps','warmup')})
print('roofline', {k:(round(v,4) if isinstance(v,float) else v) for k,v in d['roofline'].items() if k not in ('traffic_source','phase_note')})
print('steady', d.get('steady_state'))
c=d['contact']; print('contact', {k:c[k] for k in ('ms_per_substep','contacts','newton_iterations')}, c['settled'], c['roofline'])
print('cpu', {k:v for k,v in d['cpu_baseline'].items() if k not in ('host','sample')})
P
