#!/bin/bash
# Runs the GPU suite N times, one pytest process after the other, to find tests whose outcome depends on the run
# (non-deterministic summation orders, chaotic iterations).  Stops at once after a run that was killed or timed out
# (a GPU problem, not a test outcome); an ordinary test failure does not stop the series.
n=${1:-3}
mkdir -p gpurun_out
for i in $(seq 1 "$n"); do
  timeout -k 10 900 python -m pytest tests -q -m gpu -p no:cacheprovider > "gpurun_out/suite_repeat_$i.log" 2>&1
  rc=$?
  echo "run $i rc $rc: $(tail -1 gpurun_out/suite_repeat_$i.log)"
  if [ $rc -ge 124 ]; then echo "stopping: run $i was killed or timed out"; exit $rc; fi
done
