cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_contact_gpu.py tests/test_contact_pairs_gpu.py -q -x -p no:faulthandler 2>&1 | tail -3
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4i_bench.json 2> gpurun_out/r4i_bench.err
python3 -c "
import json
d=json.load(open('gpurun_out/r4i_bench.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], d['roofline']['frac_net'], 'substep', d['roofline']['substep_frac'])
print('steady', d['steady_state']); print('ref', d['reference_call_pattern']['ms_per_step'])
c=d['contact']; print('contact', c['ms_per_substep'], c['newton_iterations'], c['settled'], c['roofline']['kernel_ms'], c['roofline']['frac'])"
