#!/bin/bash
# Runs on the GPU box (inside gpurun): kernel timing, the two PMC passes and the SQ pass of the bench command, and the
# kernel trace of the contact leg.  usage: scripts/collect_profiles.sh <tag> <git HEAD the tree was taken from>
# (the box has no .git: the caller passes `git rev-parse HEAD`, and refuses to run on a dirty tree -- VERDICT r4: profiles
# are taken at the commit that ships; every summary records it)
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r01}
export MPM_PROFILE_HEAD=${2:-unknown}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-contact-leg"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 200 --warmup 20 $B > $OUT/bench_trace.json 2> $OUT/trace.log
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $R/bench.py --steps 40 --warmup 10 $B > /dev/null 2> $OUT/fetch.log
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $R/bench.py --steps 40 --warmup 10 $B > /dev/null 2> $OUT/write.log
# instruction counts and wait states (one pass, 8 SQ slots)
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/sq -- python3 $R/bench.py --steps 40 --warmup 10 $B > /dev/null 2> $OUT/sq.log
# the contact substep (config 3): kernel summary of the bench's own contact leg (both call patterns)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/contact -- python3 $R/bench.py --contact-only > $OUT/contact_bench.json 2> $OUT/contact.log
cd $R
python3 scripts/pmc_summary.py $TAG $OUT/trace/*/*kernel_stats.csv $OUT/fetch/*/*counter_collection.csv $OUT/write/*/*counter_collection.csv
python3 scripts/sq_summary.py $TAG $OUT/sq/*/*counter_collection.csv
cp $OUT/contact/*/*kernel_stats.csv $R/profiles/${TAG}_contact_config3_kernel_stats.csv
cp $OUT/contact_bench.json $R/profiles/${TAG}_contact_config3_bench.json
echo "$MPM_PROFILE_HEAD" > $R/profiles/${TAG}_HEAD.txt
mkdir -p $R/gpurun_out/profiles_$TAG && cp $R/profiles/${TAG}_* $R/profiles/pmc_traffic.json $R/profiles/sq_counters.json $R/gpurun_out/profiles_$TAG/
