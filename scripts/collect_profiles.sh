#!/bin/bash
# Runs on the GPU box (inside gpurun): kernel timing and the two PMC passes of the bench command.
# usage: scripts/collect_profiles.sh <tag>
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r01}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_trace.json 2> $OUT/trace.log
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline > /dev/null 2> $OUT/fetch.log
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline > /dev/null 2> $OUT/write.log
cd $R
python3 scripts/pmc_summary.py $TAG $OUT/trace/*/*kernel_stats.csv $OUT/fetch/*/*counter_collection.csv $OUT/write/*/*counter_collection.csv
mkdir -p $R/gpurun_out/profiles_$TAG && cp $R/profiles/${TAG}_* $R/profiles/pmc_traffic.json $R/gpurun_out/profiles_$TAG/
