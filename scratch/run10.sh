cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_contact_gpu.py tests/test_contact_pairs_gpu.py tests/test_chain_native_gpu.py -q -x -p no:faulthandler 2>&1 | tail -2
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4l_bench.json 2> gpurun_out/r4l_bench.err
python3 -c "
import json
d=json.load(open('gpurun_out/r4l_bench.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'steady', d['steady_state']['ms_per_step'])
c=d['contact']; print('contact', c['ms_per_substep'], c['newton_iterations'], c['settled'], c['roofline']['kernel_ms'], c['roofline']['frac'])"
