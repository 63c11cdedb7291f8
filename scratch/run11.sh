cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
timeout -k 10 1150 python -m pytest tests/test_parity_gpu.py tests/test_run_substeps_gpu.py tests/test_capacity_gpu.py tests/test_configs_gpu.py tests/test_world_gpu.py -q -x -p no:faulthandler > gpurun_out/r4k_suite.log 2>&1; echo "suite rc $?" >> gpurun_out/r4k_suite.log
tail -3 gpurun_out/r4k_suite.log
mkdir -p gpurun_out/r4k_trace
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4k_trace -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-contact-leg > $R/gpurun_out/r4k_bench_trace.json 2> $R/gpurun_out/r4k_trace.log
cd $R
python3 - <<'PY'
import csv,glob,collections
f=glob.glob("gpurun_out/r4k_trace/**/*kernel_trace.csv", recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"].split("(")[0]
    if "k_rb_" in n: d[n].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for n,v in sorted(d.items()):
    act=[x for x in v if x>8]; idle=[x for x in v if x<=8]
    print(n, "active",len(act), "avg %.1f"%(sum(act)/max(len(act),1)), "idle avg %.2f"%(sum(idle)/max(len(idle),1)))
PY
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-contact-leg | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['steady_state'], d['reference_call_pattern']['ms_per_step'])"
