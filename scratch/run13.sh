cd $GRAFT_REPO_ROOT
timeout -k 10 1150 python -m pytest tests -m gpu -q -p no:faulthandler > gpurun_out/r4n_suite.log 2>&1; echo "suite rc $?" >> gpurun_out/r4n_suite.log
tail -4 gpurun_out/r4n_suite.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r4n_bench_line.json 2> gpurun_out/r4n_bench.err; echo "bench rc $?"
python3 -c "
import json
d=json.load(open('gpurun_out/r4n_bench_line.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_net'], d['roofline']['substep_frac'])
print(d['steady_state']); print(d['reference_call_pattern']['ms_per_step']); print(d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
c=d['contact']; print(c['ms_per_substep'], c['newton_iterations'], c['settled']['ms_per_substep'], c['roofline']['frac'])"
